#!/usr/bin/env python
"""How many tiles of the N* workload the LDS tile kernel hands to the direct-tap pass."""
import ctypes, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module('depth-from-motion_amd')
sweep = importlib.import_module('depth-from-motion_amd.plane_sweep')
lib = pkg._capi.lib()
w = bench.WORKLOADS['nstar']; dev = torch.device('cuda:0'); B = w['B']
cur = torch.randn(B, w['C'], w['H'], w['W']).to(dev).bfloat16(); prev = torch.randn_like(cur)
depths = torch.from_numpy(bench.depth_planes(w['D'], 2.0, 59.6)).to(dev)
desc = sweep._make_desc(cur, w['D'], w['fsf'], w['csf'], (375, 1242), False, (0, 0), 1.0)
P, Pinv, T = sweep.camera_matrices(torch.from_numpy(np.stack([bench.KITTI_P2] * B)), torch.from_numpy(bench.poses(B, 2)), B, dev)
out = torch.empty((B, 2 * w['C'], w['D'], desc.h_out, desc.w_out), dtype=torch.bfloat16, device=dev)
for kib in (36, 52, 64, 78, 100, 150):
    pkg._capi.check(lib.dfm_plane_sweep_tune(256, kib, 1 << 20, 2))
    sweep.plane_sweep_forward(desc, cur, prev, depths, P, Pinv, T, out=out)
    torch.cuda.synchronize()
    nbytes = lib.dfm_plane_sweep_workspace_bytes(ctypes.byref(desc))
    ws = sweep._Workspace.get(dev, nbytes)
    blocked = ((B * 32 * w['H'] * w['W'] * 16 + 255) // 256) * 256
    count = int(ws[2 * blocked: 2 * blocked + 4].view(torch.int32).item())
    print(f'lds {kib:3d} KiB: {count} tiles queued for the direct-tap pass')
