#!/usr/bin/env python
"""MFMA Conv3d (csrc/conv3d.hip) vs torch/MIOpen at the config-K cost-volume shapes (GPU box).
usage: python tools/conv_timing.py [--chunks 0,8,12,24]"""
import argparse
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cv = importlib.import_module('depth-from-motion_amd.conv3d')
torch.backends.cudnn.benchmark = True
dev = torch.device('cuda:0')


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--chunks', default='0,6,9,12,18,24,36,72')
    args = ap.parse_args()
    for N, (D, H, W) in ((1, (72, 80, 320)), (8, (72, 80, 320)), (1, (20, 304, 288))):
        V = N * D * H * W
        flops = 2 * 27 * 32 * 32 * V
        x = torch.randn(N, 32, D, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
        conv = torch.nn.Conv3d(32, 32, 3, padding=1, bias=False).to(dev).bfloat16().to(memory_format=torch.channels_last_3d)
        packed = cv.pack_conv3d_weights(conv.weight)
        with torch.no_grad():
            t_mi = timeit(lambda: conv(x))
            print(f'N={N} {D}x{H}x{W} 32->32  MIOpen bf16 ndhwc : {t_mi:7.3f} ms  {flops / t_mi / 1e9:7.1f} TFLOP/s', flush=True)
            for c in (int(v) for v in args.chunks.split(',')):
                t = timeit(lambda: cv.conv3d_k3_c32(x, packed, depth_chunk=c))
                print(f'N={N} {D}x{H}x{W} 32->32  MFMA kernel chunk={c:3d}: {t:7.3f} ms  {flops / t / 1e9:7.1f} TFLOP/s'
                      f'  ({flops / t / 1e9 / 2500 * 100:4.1f} % of 2.5 PF)', flush=True)
            t = timeit(lambda: cv.conv3d_k3_c32(x, packed, stats=True))
            print(f'N={N} {D}x{H}x{W} 32->32  MFMA kernel + GroupNorm statistics epilogue: {t:7.3f} ms', flush=True)
            t = timeit(lambda: cv.conv3d_k3_c32(x, packed, out_f32=True))
            print(f'N={N} {D}x{H}x{W} 32->32  MFMA kernel fp32-partial out: {t:7.3f} ms', flush=True)
        del x


if __name__ == '__main__':
    main()
