#!/usr/bin/env python
"""FrustumToVoxel backward at config K (cost volume 32 x 72 x 80 x 320 bf16 channels-last + a 32-channel semantic
map, voxels 20 x 304 x 288, fused depth head: what DfMStereoPath trains through): ms per backward call (the
pre-pass, the gather kernel, the semantic map's zero fill), device time by graph replay.  GPU box.
DFM_HIP_LIB selects a variant library (build.build_variant)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
pkg = importlib.import_module('depth-from-motion_amd')
dev = torch.device('cuda:0')


def main():
    dtype = torch.bfloat16
    B, C, D, H, W = 1, 32, 72, 80, 320
    gen = torch.Generator().manual_seed(0)
    stereo = torch.randn(B, C, D, H, W, generator=gen).to(dev).to(dtype).contiguous(
        memory_format=torch.channels_last_3d).requires_grad_(True)
    cost = (torch.randn(B, 1, D, H, W, generator=gen) * 4).to(dev).to(dtype)
    sem = torch.randn(B, C, H, W, generator=gen).to(dev).to(dtype).requires_grad_(True)
    ds = torch.tensor([(k + 0.5) * (57.6 / 288) + 2 for k in range(288)])
    zz, yy, xx = torch.meshgrid(torch.linspace(-2.9, 0.9, 20), torch.linspace(-30.3, 30.3, 304),
                                torch.linspace(2.1, 59.5, 288), indexing='ij')
    coords = torch.stack([xx, yy, zz], -1).to(dev)
    K = bench.KITTI_P2.copy()
    K[1, 2] -= 55.0
    metas = [{'cam2img': K.tolist(), 'pad_shape': (320, 1280, 3)}] * B
    cfg = dict(depth_min=2, depth_max=59.6)
    lazy, _ = pkg.depth_head_statistics(cost, ds, 4)
    out = pkg.frustum_to_voxel_sample(stereo, lazy, metas, sem, coords, cfg, channels_last=True) \
        if 'channels_last' in pkg.frustum_to_voxel_sample.__code__.co_varnames else \
        pkg.frustum_to_voxel_sample(stereo, lazy, metas, sem, coords, cfg)
    go = torch.randn(out.shape, generator=gen).to(dev).to(dtype)
    if not out.is_contiguous():
        go = go.contiguous(memory_format=torch.channels_last_3d)

    def bwd():
        return torch.autograd.grad(out, [stereo, sem], go, retain_graph=True)
    for _ in range(3):
        g = bwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 10
    e0.record()
    for _ in range(iters):
        g = bwd()
    e1.record()
    torch.cuda.synchronize()
    print(f'lib {os.environ.get("DFM_HIP_LIB", "release")}: FrustumToVoxel backward {e0.elapsed_time(e1) / iters:7.3f} ms per call; '
          f'checksum stereo {float(g[0].float().abs().sum()):.6e} sem {float(g[1].float().abs().sum()):.6e}', flush=True)


if __name__ == '__main__':
    main()
