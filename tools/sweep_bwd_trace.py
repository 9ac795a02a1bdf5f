#!/usr/bin/env python
"""Cycles per phase of workgroup 0 of the matrix-product plane-sweep backward (debug build:
DFM_HIP_LIB=.../libdfm_hip_dbg.so).  N* shape, one sample.  Phases per step: 0 loop overhead, 8 footprints
of step t + 2 (the wave whose turn it is), 1 fragment image of step t + 1 (another wave), 2 wait for this
step's gradient words + issue the next step's, 3 meta, 4 flush, 5 fragments + MFMA, 6 barrier, 7 final
flush.  (The debug build runs ~2.5x slower than the release build: relative numbers only.)"""
import ctypes, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module('depth-from-motion_amd')
lib = pkg._capi.lib()
dev = torch.device('cuda:0')
w = bench.WORKLOADS['nstar']
B, C, H, W, D = 1, int(os.environ.get('DFM_C', 256)), w['H'], w['W'], w['D']
g = torch.Generator().manual_seed(1)
cur = torch.randn(B, C, H, W, generator=g).to(dev).bfloat16().requires_grad_(True)
prev = torch.randn(B, C, H, W, generator=g).to(dev).bfloat16().requires_grad_(True)
depths = torch.from_numpy(bench.depth_planes(D, w['dmin'], w['dmax'])).to(dev)
P = torch.from_numpy(bench.KITTI_P2)[None].repeat(B, 1, 1)
T = torch.from_numpy(bench.poses(B, 2))
gout = torch.randn(B, 2 * C, D, H, W, generator=g).to(dev).bfloat16()
def run():
    cur.grad = prev.grad = None
    out = pkg.build_dfm_cost(cur, prev, depths, w['fsf'], w['csf'], P, T, (376, 1244))
    out.backward(gout)
    torch.cuda.synchronize()
run()
buf = torch.zeros(128, dtype=torch.int64, device=dev)
lib.dfm_debug_set_bm_trace.argtypes = [ctypes.c_void_p]
lib.dfm_debug_set_bm_trace(ctypes.c_void_p(buf.data_ptr()))
run()
lib.dfm_debug_set_bm_trace(None)
t = buf.cpu().numpy()[:128].reshape(2, 64)[:, :48].reshape(2, 4, 12).astype(np.float64)
names = ['loop', 'image', 'grad wait+issue', 'meta', 'flush', 'frag+mfma', 'barrier', 'final flush', 'footprints', '-', '-', '-']
for half, nm in ((0, 'cur'), (1, 'prev')):
    print(f'{nm} map, workgroup 0 (cycles summed over its planes; per plane of {D} in brackets)')
    for wv in range(4):
        tot = t[half, wv].sum()
        print(f'  wave {wv}: total {tot:9.0f}  ' + '  '.join(f'{n} {v:8.0f} [{v / D:6.0f}]' for n, v in zip(names, t[half, wv])))
