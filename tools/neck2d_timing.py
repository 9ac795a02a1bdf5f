#!/usr/bin/env python
"""Where SPPUNetNeck / BEVHourglass spend their time at config K (bf16 channels_last, GPU box):
wall clock per sub-module and the number of MFMA 2-D convolution launches.
usage: python tools/neck2d_timing.py"""
import importlib
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import path_timing as pt  # noqa: E402

pkg, dev = pt.pkg, pt.dev
cv = importlib.import_module('depth-from-motion_amd.conv3d')
mods = importlib.import_module('depth-from-motion_amd.modules')


def run(fn, iters=10):
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3 / iters


def main():
    model = dict(pt.cfg('dfm_r34_1x8_kitti-3d-3class.py'))
    torch.manual_seed(0)
    path = pkg.DfMStereoPath(model).to(dev).eval().to(torch.bfloat16)
    neck = path.neck
    H, W = 320, 1280
    gen = torch.Generator().manual_seed(1)
    feats = [torch.randn(1, c, H // s, W // s, generator=gen).to(dev).bfloat16().contiguous(
        memory_format=torch.channels_last) for c, s in ((3, 1), (64, 2), (128, 4), (128, 4), (128, 4))]
    calls = {'n': 0}
    real = cv.conv3d_g

    def counted(*a, **k):
        calls['n'] += 1
        return real(*a, **k)
    cv.conv3d_g = counted
    with torch.no_grad():
        neck(feats)
    print('MFMA conv launches per SPPUNetNeck.forward:', calls['n'])
    cv.conv3d_g = real
    print(f'SPPUNetNeck.forward                 : {run(lambda: neck(feats)):7.3f} ms')
    shape = tuple(feats[2].shape[2:])
    with torch.no_grad():
        spp = [F.interpolate(b(feats[-1]), shape, mode='bilinear', align_corners=True) for b in neck.spp_branches]
        cat = torch.cat((*feats[2:], *spp), 1)
        st = neck.upconv_module([cat, feats[1], feats[0]])
    print(f'  4 SPP branches (mean, 1x1 conv, GN, upsample): '
          f'{run(lambda: [F.interpolate(b(feats[-1]), shape, mode="bilinear", align_corners=True) for b in neck.spp_branches]):7.3f} ms')
    print(f'  concat                              : {run(lambda: torch.cat((*feats[2:], *spp), 1)):7.3f} ms')
    print(f'  upconv_module                       : {run(lambda: neck.upconv_module([cat, feats[1], feats[0]])):7.3f} ms')
    up = neck.upconv_module
    with torch.no_grad():
        c0 = mods._conv_norm_2d(up.conv[0], cat)
        u0 = up.up(c0)
    print(f'    conv[0] 512->64 (+BN)             : {run(lambda: mods._conv_norm_2d(up.conv[0], cat)):7.3f} ms')
    print(f'    upsample x2 of it                 : {run(lambda: up.up(c0)):7.3f} ms')
    print(f'    redir[0] 64->64 (+BN)             : {run(lambda: mods._conv_norm_2d(up.redir[0], feats[1])):7.3f} ms')
    print(f'    redir[1] 3->32 (+BN, torch)       : {run(lambda: mods._conv_norm_2d(up.redir[1], feats[0])):7.3f} ms')
    print(f'  lastconv (3x3 + GN + ReLU, 1x1)     : {run(lambda: neck.lastconv(st)):7.3f} ms')
    print(f'  rpnconv (2 x 3x3 + GN + ReLU)       : {run(lambda: neck.rpnconv(cat)):7.3f} ms')
    bev = path.backbone_3d
    x = torch.randn(1, 160, 304, 288, generator=gen).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    print(f'BEVHourglass.forward                : {run(lambda: bev(x)):7.3f} ms')
    print(f'  compress_conv                       : {run(lambda: bev.compress_conv(x)):7.3f} ms')


if __name__ == '__main__':
    main()
