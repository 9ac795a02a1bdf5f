#!/usr/bin/env python
"""Per-phase s_memtime trace of the tile kernel (debug hook dfm_debug_set_trace).
usage: python tools/trace_phases.py [lanes lds_kib planes]   (N* workload, B=8)
Needs a debug build:  python -c "import importlib; b=importlib.import_module('depth-from-motion_amd.build'); b.build_hip(force=True, debug_hooks=True, out='/tmp/libdfm_dbg.so')"
and DFM_HIP_LIB=/tmp/libdfm_dbg.so."""
import ctypes, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
lanes, kib, dbl = (int(v) for v in (sys.argv[1:4] + ['256', '52', '0'])[:3])
pkg = importlib.import_module('depth-from-motion_amd')
sweep = importlib.import_module('depth-from-motion_amd.plane_sweep')
lib = pkg._capi.lib()
pkg._capi.check(lib.dfm_plane_sweep_tune(lanes, kib, 1000, dbl))
w = bench.WORKLOADS['nstar']; dev = torch.device('cuda:0'); B = w['B']
cur = torch.randn(B, w['C'], w['H'], w['W']).to(dev).bfloat16(); prev = torch.randn_like(cur)
depths = torch.from_numpy(bench.depth_planes(w['D'], 2.0, 59.6)).to(dev)
desc = sweep._make_desc(cur, w['D'], w['fsf'], w['csf'], (375, 1242), False, (0, 0), 1.0)
P, Pinv, T = sweep.camera_matrices(torch.from_numpy(np.stack([bench.KITTI_P2] * B)), torch.from_numpy(bench.poses(B, 2)), B, dev)
out = torch.empty((B, 2 * w['C'], w['D'], desc.h_out, desc.w_out), dtype=torch.bfloat16, device=dev)
for _ in range(2): sweep.plane_sweep_forward(desc, cur, prev, depths, P, Pinv, T, out=out)
torch.cuda.synchronize()
nrec = 4096
buf = torch.zeros(nrec * 64, dtype=torch.int64, device=dev)
lib.dfm_debug_set_trace.argtypes = [ctypes.c_void_p]
lib.dfm_debug_set_trace(ctypes.c_void_p(buf.data_ptr()))
sweep.plane_sweep_forward(desc, cur, prev, depths, P, Pinv, T, out=out)
torch.cuda.synchronize()
lib.dfm_debug_set_trace(None)
t = buf.cpu().numpy().reshape(nrec, 64)
t = t[t[:, 0] != 0]
print('traced workgroups', len(t), 'lanes', lanes, 'kib', kib, 'dbl', dbl)
per = 5
d = np.diff(t[:, :2 + per * 11].astype(np.float64), axis=1)
# stamps per iteration -- single: S0 after dma issue, S1 after wait+barrier, S2 after blend,
# S3 after store issue, S4 after barrier2 ; double: T0 top, T1 after vmcnt wait, T2 after
# barrier, T3 after dma issue+blend, T4 after store issue
names = ['wait+barrier', 'reads+blend', 'store_issue', 'barrier2', 'dma_issue(next)']
print('first stamp -> first loop stamp (footprints etc):', np.median(d[:, 0]))
for i in range(per):
    cols = d[:, 1 + i::per][:, 1:9]
    print(f'{names[i]:28s} median {np.median(cols):9.0f}  p90 {np.percentile(cols, 90):9.0f} cycles')
tot = (t[:, 1 + per * 9] - t[:, 1 + per * 1]) / 8.0
print('cycles per block iteration: median', np.median(tot))
