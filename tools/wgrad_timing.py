#!/usr/bin/env python
"""MFMA weight gradient (csrc/conv3d_wgrad.hip) on the layer shapes of the path.  GPU box."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cv = importlib.import_module('depth-from-motion_amd.conv3d')
dev = torch.device('cuda:0')
CASES = [  # name, a (g channels), b (x channels), x size, stride, padding
    ('dres1 32->32', 32, 32, (72, 80, 320), 1, 1),
    ('hg.conv1 32->64 s2', 64, 32, (72, 80, 320), 2, 1),
    ('hg.conv2 64->64', 64, 64, (36, 40, 160), 1, 1),
    ('neck 64->64', 64, 64, (220, 300, 12), 1, 1),
    ('neck 64->128 s(1,1,2)', 128, 64, (220, 300, 12), (1, 1, 2), 1),
    ('neck 128->128', 128, 128, (220, 300, 6), 1, 1),
    ('neck 256->256', 256, 256, (220, 300, 3), 1, 1),
]


def main():
    for name, a, b, size, stride, padding in CASES:
        st, pd = cv._triple(stride), cv._triple(padding)
        osz = cv.conv3d_g_out_size(size, st, pd, (False,) * 3)
        x = torch.randn(1, b, *size, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
        g = torch.randn(1, a, *osz, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
        for _ in range(3):
            cv.conv3d_weight_grad(x, g, st, pd)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            cv.conv3d_weight_grad(x, g, st, pd)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10
        flops = 2 * 27 * a * b * osz[0] * osz[1] * osz[2]
        print(f'{name:24s} {str(size):16s} {t:7.3f} ms {flops / t / 1e9:7.1f} TFLOP/s ({flops / t / 1e9 / 25:4.1f} %)', flush=True)


if __name__ == '__main__':
    main()
