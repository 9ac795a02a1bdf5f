#!/usr/bin/env python
"""Time of the MFMA weight-gradient launch (csrc/conv3d_wgrad.hip) on the convolution shapes of config K's backbone.
usage: [DFM_WGRAD_COL=0|1] [DFM_WGRAD_CHUNK=n] python tools/wgrad_timing.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    cv = importlib.import_module('depth-from-motion_amd.conv3d')
    dev = torch.device('cuda:0')
    cl = torch.channels_last_3d
    for (a, b, size, stride) in ((32, 32, (72, 80, 320), 1), (64, 32, (72, 80, 320), 2), (64, 64, (36, 40, 160), 1)):
        x = torch.randn(1, b, *size, device=dev).bfloat16().contiguous(memory_format=cl)
        osz = tuple((s - 1) // stride + 1 for s in size)
        gy = torch.randn(1, a, *osz, device=dev).bfloat16().contiguous(memory_format=cl)
        for _ in range(3):
            cv.conv3d_weight_grad(x, gy, stride, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            cv.conv3d_weight_grad(x, gy, stride, 1)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        flops = 2.0 * 27 * a * b * osz[0] * osz[1] * osz[2]
        print('wgrad %d<-%d %s stride %d: %.1f us  %.0f TFLOP/s (kernel + reduce)  column mode=%s chunk=%s' % (
            a, b, size, stride, us, flops / us / 1e6, os.environ.get('DFM_WGRAD_COL', 'default'), os.environ.get('DFM_WGRAD_CHUNK', '-')))


if __name__ == '__main__':
    main()
