#!/usr/bin/env python
"""OutdoorImVoxelNeck forward + backward in training mode (batch statistics), config W volume,
bf16 NDHWC: MFMA forward / backward-data / weight gradient, BatchNorm3d through torch.  GPU box."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mods = importlib.import_module('depth-from-motion_amd.modules')
dev = torch.device('cuda:0')


def main():
    iters = int(os.environ.get('DFM_ITERS', '3'))
    torch.manual_seed(0)
    m = mods.OutdoorImVoxelNeck(in_channels=64, out_channels=256).to(dev).to(torch.bfloat16).train()
    x = torch.randn(1, 64, 220, 300, 12, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
    x.requires_grad_(True)

    def step():
        m.zero_grad(set_to_none=True)
        m(x)[0].float().square().mean().backward()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    print(f'OutdoorImVoxelNeck config W bf16 NDHWC forward + backward (training mode): '
          f'{(time.perf_counter() - t) * 1e3 / iters:8.2f} ms', flush=True)


if __name__ == '__main__':
    main()
