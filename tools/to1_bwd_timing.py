#!/usr/bin/env python
"""Backward of the prediction heads' Conv3d(32 -> 1) at config K (1 x 32 x 72 x 80 x 320 bf16 NDHWC): ms per backward
call of MfmaConv3dTo1 -- csrc/conv3d_to1_bwd.hip against the former route (DFM_TO1_PADDED_BWD=1).  GPU box."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cv = importlib.import_module('depth-from-motion_amd.conv3d')
dev = torch.device('cuda:0')


def main():
    torch.manual_seed(0)
    m = cv.MfmaConv3dTo1(32, 1, 3, 1, 1, bias=False).to(dev).to(torch.bfloat16)
    x = torch.randn(1, 32, 72, 80, 320, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    y = m(x)
    gy = torch.randn_like(y)
    for mode in ('0', '1', '0', '1'):
        os.environ['DFM_TO1_PADDED_BWD'] = mode
        for _ in range(3):
            torch.autograd.grad(y, [x, m.weight], gy, retain_graph=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.autograd.grad(y, [x, m.weight], gy, retain_graph=True)
        e1.record()
        torch.cuda.synchronize()
        print(f'{"padded to 32 channels (former)" if mode == "1" else "matrix products over the taps":32s} {e0.elapsed_time(e1) / 10:7.3f} ms per backward', flush=True)


if __name__ == '__main__':
    main()
