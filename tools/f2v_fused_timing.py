#!/usr/bin/env python
"""DepthHead -> FrustumToVoxel at config K (cost 72x80x320 -> distribution 288x320x1280, voxels
20x304x288): materialised pipeline (dfm_depth_head_fwd + dfm_frustum_to_voxel_fwd) vs the fused one
(dfm_depth_head_stats_fwd + dfm_frustum_to_voxel_fused_fwd).  GPU box."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
pkg = importlib.import_module('depth-from-motion_amd')
dev = torch.device('cuda:0')


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    for dtype in (torch.float32, torch.bfloat16):
        B, C, D, H, W = 1, 32, 72, 80, 320
        gen = torch.Generator().manual_seed(0)
        stereo = torch.randn(B, C, D, H, W, generator=gen).to(dev).to(dtype)
        cost = (torch.randn(B, 1, D, H, W, generator=gen) * 4).to(dev).to(dtype)
        sem = torch.randn(B, C, H, W, generator=gen).to(dev).to(dtype)
        ds = torch.tensor([(k + 0.5) * (57.6 / 288) + 2 for k in range(288)])
        zz, yy, xx = torch.meshgrid(torch.linspace(-2.9, 0.9, 20), torch.linspace(-30.3, 30.3, 304),
                                    torch.linspace(2.1, 59.5, 288), indexing='ij')
        coords = torch.stack([xx, yy, zz], -1).to(dev)
        K = bench.KITTI_P2.copy()
        K[1, 2] -= 55.0
        metas = [{'cam2img': K.tolist(), 'pad_shape': (320, 1280, 3)}] * B
        cfg = dict(depth_min=2, depth_max=59.6)
        with torch.no_grad():
            def unfused():
                _, soft, _ = pkg.depth_head_forward(cost, ds, 4)
                return pkg.frustum_to_voxel_sample(stereo, soft, metas, sem, coords, cfg)

            def fused():
                lazy, _ = pkg.depth_head_statistics(cost, ds, 4)
                return pkg.frustum_to_voxel_sample(stereo, lazy, metas, sem, coords, cfg)
            a, b = unfused(), fused()
            same = bool(torch.equal(a, b))
            t_head = timeit(lambda: pkg.depth_head_forward(cost, ds, 4))
            t_stats = timeit(lambda: pkg.depth_head_statistics(cost, ds, 4))
            t_stats0 = timeit(lambda: pkg.depth_head_statistics(cost, ds, 4, need_preds=False))
            t_un, t_fu = timeit(unfused), timeit(fused)
        print(f'{str(dtype)[6:]:9s} materialised: depth head {t_head:6.3f} ms, head + FrustumToVoxel {t_un:6.3f} ms | '
              f'fused: statistics {t_stats:6.3f} ms ({t_stats0:6.3f} without depth_preds), statistics + FrustumToVoxel {t_fu:6.3f} ms | '
              f'bit-identical: {same}', flush=True)


if __name__ == '__main__':
    main()
