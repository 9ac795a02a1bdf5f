#!/usr/bin/env python
"""dfm_store_probe over run lengths: what does this part sustain for the tile kernel's store pattern
(N* volume: 8 x 512 planes of 6.5 MB) when a workgroup writes 4 / 8 / 16 / 32 / 64 KiB contiguous per plane?"""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('depth-from-motion_amd')
lib = pkg._capi.lib()
dev = torch.device('cuda:0')
B, C2, D, H, W = 8, 512, 112, 94, 311
out = torch.empty((B, C2, D, H, W), dtype=torch.bfloat16, device=dev)
plane_bytes = D * H * W * 2
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for run, group in ((4096, 0), (16384, 0), (65536, 0), (4096, 256), (4096, 64), (4096, 16), (4096, 8), (4096, 1), (16384, 8),
                   (16384, 1)):
    fn = lambda: pkg._capi.check(lib.dfm_store_probe(ctypes.c_void_p(out.data_ptr()), B, C2, plane_bytes, run, group, st))
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f'run {run // 1024:3d} KiB per plane, {group or C2:3d} planes per workgroup (run index fastest): {5 * out.numel() * 2 / (e0.elapsed_time(e1) * 1e-3) / 1e9:8.1f} GB/s')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out.zero_(); e0.record()
for _ in range(5):
    out.zero_()
e1.record(); torch.cuda.synchronize()
print(f'linear fill: {5 * out.numel() * 2 / (e0.elapsed_time(e1) * 1e-3) / 1e9:8.1f} GB/s')
