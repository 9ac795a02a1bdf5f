#!/usr/bin/env python
"""s_memtime stamps of one workgroup of the fused sweep + dres0 kernel (debug build:
DFM_HIP_LIB=.../libdfm_hip_dbg.so built with build_hip(debug_hooks=True)).  Prints, per wave, the
median cycles between consecutive stamps over the traced planes.
S waves: 0 start, 1 after phase0, 2 after X, 3 after phase1, 4 after Y, 5 after phase2, 6 after epilogue
M waves: 0 start, 1 after issue(pair1), 2 after phase0, 3 after X, 4 after blend1+issue2, 5 after phase1,
         6 after blend2+issue3, 7 after Y, 8 after phase2, 9 after blend3, 10 after epilogue"""
import ctypes, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module('depth-from-motion_amd')
sc = importlib.import_module('depth-from-motion_amd.sweep_conv')
lib = pkg._capi.lib()
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
cur = torch.randn(1, 32, 320, 1280, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
prev = torch.randn(1, 32, 320, 1280, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
depths = pkg.prepare_depth(dict(num_bins=288, depth_min=2, depth_max=59.6, downsample_factor=4))[0].to(dev)
P = torch.from_numpy(bench.KITTI_P2)[None].to(dev)
T = torch.from_numpy(bench.poses(1, 2)).to(dev)
ws = (torch.randn(32, 64, 3, 3, 3, generator=g) * 0.03).to(dev).bfloat16()
wm = (torch.randn(32, 32, 3, 3, 3, generator=g) * 0.04).to(dev).bfloat16()
packed = sc.pack_sweep_conv_weights(ws, wm)
run = lambda: sc.sweep_dres0(cur, prev, depths, 1, 4, P, T, (375, 1242), packed, img_crop_offset=(0, 55))
for _ in range(3):
    run()
torch.cuda.synchronize()
buf = torch.zeros(4 * 16 * 16, dtype=torch.int64, device=dev)
lib.dfm_debug_set_sc_trace.argtypes = [ctypes.c_void_p]
lib.dfm_debug_set_sc_trace(ctypes.c_void_p(buf.data_ptr()))
run()
torch.cuda.synchronize()
lib.dfm_debug_set_sc_trace(None)
t = buf.cpu().numpy().reshape(4, 16, 16).astype(np.float64)
for w in range(4):
    ns = 7 if w < 2 else 11
    planes = [p for p in range(1, 13) if t[w, p, 0] > 0 and t[w, p + 1, 0] > 0]
    seg = np.array([[t[w, p, i + 1] - t[w, p, i] for i in range(ns - 1)] + [t[w, p + 1, 0] - t[w, p, ns - 1]] for p in planes])
    per = np.array([t[w, p + 1, 0] - t[w, p, 0] for p in planes])
    print(f'wave {w}: plane period median {np.median(per):8.0f} ticks; segments', ' '.join(f'{v:7.0f}' for v in np.median(seg, axis=0)))
    print(f'         start offsets vs wave 0 (plane 1..4):', ' '.join(f'{t[w, p, 0] - t[0, p, 0]:8.0f}' for p in range(1, 5)))
