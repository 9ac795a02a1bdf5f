#!/usr/bin/env python
"""Host time per launch of the path's Python entry points (GPU box): how long the CPU takes to ENQUEUE one call
(perf_counter over many calls, no synchronisation in between) next to the device time per call (HIP events) -- a
launch whose host time exceeds its device time is host-bound when issued back to back.
usage: python tools/host_overhead_probe.py"""
import ctypes
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cv = importlib.import_module('depth-from-motion_amd.conv3d')
gn = importlib.import_module('depth-from-motion_amd.group_norm')
mods = importlib.import_module('depth-from-motion_amd.modules')
capi = importlib.import_module('depth-from-motion_amd._capi')
dev = torch.device('cuda:0')
CL = torch.channels_last_3d


def probe(name, fn, n=300):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    host = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    print(f'{name:46s} host {host:6.1f} us / call   device (events, back to back) {e0.elapsed_time(e1) / n * 1e3:6.1f} us / call')


def main():
    with torch.no_grad():
        x = torch.zeros(1, device=dev)
        probe('torch add_ on one element', lambda: x.add_(1))
        lib = capi.lib()
        probe('ctypes call: dfm_version()', lambda: lib.dfm_version())
        for name, cin, cout, size, stride, tr in (('hg.conv4 64->64 (18,20,80)', 64, 64, (18, 20, 80), 1, False),
                                                 ('hg.conv2 64->64 (36,40,160)', 64, 64, (36, 40, 160), 1, False),
                                                 ('hg.conv6 T 64->32 (36,40,160)', 64, 32, (36, 40, 160), 2, True)):
            xx = torch.randn(1, cin, *size, device=dev).bfloat16().contiguous(memory_format=CL)
            ref = (torch.nn.ConvTranspose3d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=False) if tr else
                   torch.nn.Conv3d(cin, cout, 3, stride=stride, padding=1, bias=False)).to(dev).bfloat16()
            pk = cv.pack_conv3d_g_weights(ref.weight, cin, cout, swap=tr)
            st = (1, 1, 1) if tr else cv._triple(stride)
            probe('conv3d_g ' + name, lambda: cv.conv3d_g(xx, pk, cout, st, (1, 1, 1), tr))
            probe('  ... conv3d_g_plan alone', lambda: cv.conv3d_g_plan(1, cin, cout, size, st, (1, 1, 1), tr), n=100)
        y = torch.randn(1, 64, 18, 20, 80, device=dev).bfloat16().contiguous(memory_format=CL)
        g = gn.HipGroupNorm(32, 64).to(dev)
        probe('HipGroupNorm(32, 64) on (18,20,80), relu', lambda: g(y, relu=True))
        c32 = mods.MfmaConv3d(32, 32, 3, padding=1, bias=False).to(dev).bfloat16()
        z = torch.randn(1, 32, 18, 20, 80, device=dev).bfloat16().contiguous(memory_format=CL)
        probe('MfmaConv3d 32->32 (18,20,80) + stats', lambda: c32.forward_with_stats(z))
        bb = mods.DfMBackbone(in_channels=32).to(dev).to(torch.bfloat16).eval()
        bb.volume_memory_format = CL
        hgm = bb.hg_stereo[0]
        zz = torch.randn(1, 32, 72, 80, 320, device=dev).bfloat16().contiguous(memory_format=CL)
        probe('hourglass.forward_add at config K (24 launches)', lambda: hgm.forward_add(zz, None, None, zz), n=50)


if __name__ == '__main__':
    main()
