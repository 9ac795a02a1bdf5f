#!/usr/bin/env python
"""Backward passes of the secondary rows at their config sizes (one sample each):
DepthHead (72x80x320 -> 288x320x1280), FrustumToVoxel sampling (config K), multi-view lifting
(config W).  Times forward+backward and forward alone with CUDA events; backward = difference."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module('depth-from-motion_amd')
dev = torch.device('cuda:0')


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


g = torch.Generator().manual_seed(0)
# DepthHead
x = (torch.randn(1, 1, 72, 80, 320, generator=g) * 4).to(dev).requires_grad_(True)
ds = torch.tensor([(k + 0.5) * (57.6 / 288) + 2 for k in range(288)])
gv = torch.randn(1, 1, 288, 320, 1280, device=dev)
gp = torch.randn(1, 1, 320, 1280, device=dev)
def dh_fwd():
    with torch.no_grad():
        pkg.depth_head_forward(x, ds)
def dh_fb():
    x.grad = None
    v, s, p = pkg.depth_head_forward(x, ds)
    torch.autograd.backward([v, s, p], [gv, gv, gp])
f, fb = timed(dh_fwd), timed(dh_fb)
print(f'depth_head   fwd {f:8.3f} ms   fwd+bwd {fb:8.3f} ms   bwd ~{fb - f:8.3f} ms', flush=True)
del gv, gp
# FrustumToVoxel
C, D, H, W = 32, 72, 80, 320
st = torch.randn(1, C, D, H, W, generator=g).to(dev).requires_grad_(True)
soft = torch.softmax(torch.randn(1, 1, 4 * D, 4 * H, 4 * W, device=dev), dim=2)
sem = torch.randn(1, C, H, W, generator=g).to(dev).requires_grad_(True)
zz, yy, xx = torch.meshgrid(torch.linspace(-2.9, 0.9, 20), torch.linspace(-30.3, 30.3, 304),
                            torch.linspace(2.1, 59.5, 288), indexing='ij')
coords = torch.stack([xx, yy, zz], -1).to(dev)
K = bench.KITTI_P2.copy(); K[1, 2] -= 55.0
metas = [{'cam2img': K.tolist(), 'pad_shape': (320, 1280, 3)}]
cfg = dict(depth_min=2, depth_max=59.6)
go = torch.randn(1, 2 * C, 20, 304, 288, device=dev)
def f2v_fwd():
    with torch.no_grad():
        pkg.frustum_to_voxel_sample(st, soft, metas, sem, coords, cfg)
def f2v_fb():
    st.grad = sem.grad = None
    pkg.frustum_to_voxel_sample(st, soft, metas, sem, coords, cfg).backward(go)
f, fb = timed(f2v_fwd), timed(f2v_fb)
print(f'f2v          fwd {f:8.3f} ms   fwd+bwd {fb:8.3f} ms   bwd ~{fb - f:8.3f} ms', flush=True)
del go, soft
# multi-view lifting
from tests.golden.make_golden import waymo_like_cameras
nv, nf, C, hf, wf, nvox = 5, 2, 64, 208, 312, (220, 300, 12)
feats = torch.randn(1, nv * nf, C, hf, wf, generator=g).to(dev).requires_grad_(True)
cams = waymo_like_cameras(nv, nf, 5); cams[:, 0, :] *= 1248 / 156.0; cams[:, 1, :] *= 832 / 104.0
meta = {'ori_lidar2img': [m for m in cams], 'input_shape': (832, 1248), 'img_shape': [(832, 1248, 3)] * (nv * nf)}
pts = pkg.voxel_centers([-35.0, -75.0, -2.0, 75.0, 75.0, 4.0], nvox).to(dev)
go = torch.randn(1, C * nf, *nvox, device=dev)
def mv_fwd():
    with torch.no_grad():
        pkg.mv_feature_transformation(feats, [meta], nv, nf, None, nvox, 'concat', points=pts)
def mv_fb():
    feats.grad = None
    pkg.mv_feature_transformation(feats, [meta], nv, nf, None, nvox, 'concat', points=pts).backward(go)
f, fb = timed(mv_fwd), timed(mv_fb)
print(f'mv lifting   fwd {f:8.3f} ms   fwd+bwd {fb:8.3f} ms   bwd ~{fb - f:8.3f} ms', flush=True)
