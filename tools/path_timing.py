#!/usr/bin/env python
"""Whole-path inference timing at the config sizes (GPU box):
  stereo : DfMStereoPath of configs/dfm/dfm_r34_1x8_kitti-3d-3class.py (SPPUNetNeck -> DfMBackbone ->
           DepthHead (fused) -> FrustumToVoxel -> voxel_convs -> BEVHourglass), 320x1280 crop, bf16 NDHWC
  mv     : MultiViewVoxelPath of the Waymo config (5 views x 2 frames lifting -> DfMNeck), 220x300x12
usage: python tools/path_timing.py [stereo|mv] [--iters N]   (DFM_MIOPEN_FIND=1: MIOpen autotuning for
the 2-D convolutions of the producers / consumers)"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
pkg = importlib.import_module('depth-from-motion_amd')
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = os.environ.get('DFM_MIOPEN_FIND') == '1'


def cfg(name):
    with open(os.path.join(ROOT, 'tests', 'golden', 'configs_dfm.json')) as f:
        return json.load(f)[name]['model']


def run(fn, iters):
    with torch.no_grad():
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3 / iters


def stereo(iters):
    model = dict(cfg('dfm_r34_1x8_kitti-3d-3class.py'))
    torch.manual_seed(0)
    path = pkg.DfMStereoPath(model).to(dev).eval().to(torch.bfloat16)
    path.backbone_stereo.volume_memory_format = torch.channels_last_3d
    H, W = 320, 1280
    gen = torch.Generator().manual_seed(1)

    integ = importlib.import_module('depth-from-motion_amd.integration')

    def pyramid():
        # what a channels_last bf16 image backbone (LIGAResNet under torch / MIOpen) hands over
        return [torch.randn(1, c, H // s, W // s, generator=gen).to(dev).bfloat16().contiguous(
            memory_format=torch.channels_last) for c, s in ((3, 1), (64, 2), (128, 4), (128, 4), (128, 4))]
    cur, prev = pyramid(), pyramid()
    K = bench.KITTI_P2.copy()
    K2 = K.copy()
    K2[1, 2] -= 55.0

    def meta():
        return dict(ori_cam2img=K, cam2img=K2.tolist(), cur2prevs=torch.from_numpy(bench.poses(1, 2)),
                    ori_shape=(375, 1242, 3), pad_shape=(H, W, 3), crop_offset=[0, 55], flip=False,
                    scale_factor=[1.0])
    ms = run(lambda: path(cur, prev, [meta()]), iters)
    print(f'DfMStereoPath inference (config K, 320x1280, bf16 NDHWC, depth head fused): {ms:8.2f} ms / sample', flush=True)
    path.hip_graphs = True
    ms = run(lambda: path(cur, prev, [meta()]), iters)
    print(f'  ... with the 2-D necks replayed as hipGraphs (path.hip_graphs = True)      : {ms:8.2f} ms / sample', flush=True)
    path.hip_graphs = False
    parts = dict(
        neck=lambda: (path.neck(cur), path.neck(prev)),
        backbone_stereo=None)
    cs, csem = path.neck(cur)
    ps, _ = path.neck(prev)
    m = meta()
    m['cur2prevs'] = m['cur2prevs'].to(dev)
    with torch.no_grad():
        costs, sf, mf = path.backbone_stereo(cs, ps, [m])
        _, preds, soft = path.depth_head(costs, lazy=True)
        vol = path.feature_transformation(sf, soft, [m], csem)
    print(f'  SPPUNetNeck x 2 (3x3 convs: MFMA kernel) : {run(parts["neck"], iters):8.2f} ms')
    print(f'  DfMBackbone                              : {run(lambda: path.backbone_stereo(cs, ps, [m]), iters):8.2f} ms')
    print(f'  DepthHead statistics (+ depth_preds)     : {run(lambda: path.depth_head(costs, lazy=True), iters):8.2f} ms')
    print(f'  FrustumToVoxel (fused head + conv + pool): {run(lambda: path.feature_transformation(sf, soft, [m], csem), iters):8.2f} ms')
    _, cv, nz, ny, nx = vol.shape
    print(f'  BEVHourglass (3x3 convs: MFMA kernel)    : {run(lambda: path.backbone_3d(integ.bev_view(vol)), iters):8.2f} ms', flush=True)


def mv(iters):
    from tests.golden.make_golden import waymo_like_cameras
    with open(os.path.join(ROOT, 'tests', 'golden', 'configs_dfm.json')) as f:
        names = sorted(k for k in json.load(f) if 'multiview' in k)
    for name in names:
        model = dict(cfg(name))
        torch.manual_seed(0)
        path = pkg.MultiViewVoxelPath(model).to(dev).eval().to(torch.bfloat16)
        nv, nf = 5, (2 if path.temporal_aggregate == 'concat' else 1)
        C, hf, wf = 64, 208, 312
        gen = torch.Generator().manual_seed(2)
        feats = torch.randn(1, nv * nf, C, hf, wf, generator=gen).to(dev).bfloat16()
        cams = waymo_like_cameras(nv, nf, 5)
        cams[:, 0, :] *= 1248 / 156.0
        cams[:, 1, :] *= 832 / 104.0
        meta = {'ori_lidar2img': [m for m in cams], 'input_shape': (832, 1248),
                'img_shape': [(832, 1248, 3)] * (nv * nf)}
        ms = run(lambda: path(feats, [meta], nv, nf), iters)
        print(f'MultiViewVoxelPath inference ({name}: {nv} views x {nf} frames -> {path.n_voxels} -> '
              f'{type(path.neck_3d).__name__}, bf16 NDHWC): {ms:8.2f} ms / sample', flush=True)

        def lift():
            return pkg.mv_feature_transformation(feats, [meta], nv, nf, path.voxel_range, path.n_voxels,
                                                 path.temporal_aggregate, memory_format=torch.channels_last_3d)
        vol = lift()
        print(f'  lifting (channels-last volume): {run(lift, iters):8.2f} ms')
        print(f'  neck_3d                       : {run(lambda: path.neck_3d(vol), iters):8.2f} ms', flush=True)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('which', nargs='?', default='both')
    ap.add_argument('--iters', type=int, default=5)
    a = ap.parse_args()
    if a.which in ('stereo', 'both'):
        stereo(a.iters)
    if a.which in ('mv', 'both'):
        mv(a.iters)
