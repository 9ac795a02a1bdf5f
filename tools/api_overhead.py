#!/usr/bin/env python
"""Where does the public build_dfm_cost() spend its time beyond the raw launch? (N*, B=8, bf16)"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module('depth-from-motion_amd')
sweep = importlib.import_module('depth-from-motion_amd.plane_sweep')
w = bench.WORKLOADS['nstar']; dev = torch.device('cuda:0'); B = w['B']
cur = torch.randn(B, w['C'], w['H'], w['W']).to(dev).bfloat16(); prev = torch.randn_like(cur)
depths = torch.from_numpy(bench.depth_planes(w['D'], 2.0, 59.6)).to(dev)
desc = sweep._make_desc(cur, w['D'], w['fsf'], w['csf'], (375, 1242), False, (0, 0), 1.0)
K = torch.from_numpy(np.stack([bench.KITTI_P2] * B)); T = torch.from_numpy(bench.poses(B, 2))
P, Pinv, Tm = sweep.camera_matrices(K, T, B, dev)
out = torch.empty((B, 2 * w['C'], w['D'], desc.h_out, desc.w_out), dtype=torch.bfloat16, device=dev)
out2 = torch.empty_like(out)
Kd, Td = K.to(dev), T.to(dev)

def timeit(name, fn, n=15):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); print(f'{name:58s} {(time.perf_counter() - t) * 1e3 / n:8.3f} ms', flush=True)

timeit('raw launch, same preallocated volume', lambda: sweep.plane_sweep_forward(desc, cur, prev, depths, P, Pinv, Tm, out=out))
flip = [0]
def alt():
    flip[0] ^= 1
    return sweep.plane_sweep_forward(desc, cur, prev, depths, P, Pinv, Tm, out=out2 if flip[0] else out)
timeit('raw launch, two preallocated volumes alternating', alt)
del out2
timeit('raw launch, volume allocated per call (torch.empty)', lambda: sweep.plane_sweep_forward(desc, cur, prev, depths, P, Pinv, Tm))
keep = [None]
def held():
    keep[0] = sweep.plane_sweep_forward(desc, cur, prev, depths, P, Pinv, Tm)
timeit('  ... previous result still referenced (2 blocks live)', held)
keep[0] = None
timeit('camera_matrices (device inputs)', lambda: sweep.camera_matrices(Kd, Td, B, dev), 50)
timeit('build_dfm_cost (device intrinsics / poses)', lambda: pkg.build_dfm_cost(cur, prev, depths, w['fsf'], w['csf'], Kd, Td, (375, 1242)))
timeit('build_dfm_cost (host intrinsics / poses)', lambda: pkg.build_dfm_cost(cur, prev, depths, w['fsf'], w['csf'], K, T, (375, 1242)))
