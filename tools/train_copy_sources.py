#!/usr/bin/env python
"""Where do the ATen layout copies of a DfMStereoPath bf16 training step come from?  One profiled step
(torch.profiler, with_stack), `aten::copy_` / `aten::contiguous` / `aten::clone` device time grouped by the innermost
frame inside this repository.  GPU box.  usage: python tools/train_copy_sources.py [--all]"""
import collections
import importlib
import json
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
pkg = importlib.import_module('depth-from-motion_amd')
dev = torch.device('cuda:0')


def main():
    with open(os.path.join(ROOT, 'tests', 'golden', 'configs_dfm.json')) as f:
        model = dict(json.load(f)['dfm_r34_1x8_kitti-3d-3class.py']['model'])
    H, W = 320, 1280
    K = bench.KITTI_P2.copy()
    K2 = K.copy()
    K2[1, 2] -= 55.0
    torch.manual_seed(0)
    path = pkg.DfMStereoPath(model).to(dev).train()
    pkg.enable_fast_path(path)
    path.fuse_depth_head = True
    gen = torch.Generator().manual_seed(1)

    def pyramid():
        return [torch.randn(1, c, H // s, W // s, generator=gen).to(dev).bfloat16()
                for c, s in ((3, 1), (64, 2), (128, 4), (128, 4), (128, 4))]
    cur, prev = pyramid(), pyramid()
    depth_img = (torch.rand(1, 1, H, W, generator=gen) * 60).to(dev)
    depth_img[torch.rand(1, 1, H, W, generator=gen).to(dev) < 0.93] = 0
    fg = (torch.rand(1, 1, H, W, generator=gen) < 0.3).float().to(dev)
    meta = dict(ori_cam2img=K, cam2img=K2.tolist(), cur2prevs=torch.from_numpy(bench.poses(1, 2)),
                ori_shape=(375, 1242, 3), pad_shape=(H, W, 3), crop_offset=[0, 55], flip=False, scale_factor=[1.0])

    def step():
        path.zero_grad(set_to_none=True)
        out = path(cur, prev, [meta])
        loss = path.loss_dense_depth(out, depth_img, fg) + out['bev_feat'].float().square().mean()
        loss.backward()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    by_site = collections.defaultdict(lambda: [0.0, 0])
    total = 0.0
    for ev in prof.events():
        if '--all' in sys.argv:  # every ATen operator with device time of its own (what is left on torch)
            if not ev.name.startswith('aten::'):
                continue
        elif ev.name not in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::_to_copy', 'aten::add', 'aten::add_',
                             'aten::fill_', 'aten::zero_'):
            continue
        t = getattr(ev, 'self_device_time_total', None)
        if t is None:
            t = getattr(ev, 'self_cuda_time_total', 0.0)
        if not t:
            continue
        site = None
        for fr in ev.stack or []:
            if ROOT in fr or 'depth-from-motion_amd' in fr:
                site = fr.replace(ROOT + '/', '')
                break
        if site is None:
            # no Python frame (this build's profiler hands the backward thread's operators over without one): the chain
            # of enclosing operators instead -- the autograd node or the ATen operator the copy was issued for
            chain, p_ = [], ev.cpu_parent
            while p_ is not None and len(chain) < 4:
                chain.append(p_.name.replace('autograd::engine::evaluate_function: ', ''))
                p_ = p_.cpu_parent
            site = ' < '.join(chain) if chain else 'top level (autograd engine: gradient accumulation / layout contract)'
        key = (ev.name, site, str(ev.input_shapes)[:90] if ev.input_shapes else '')
        by_site[key][0] += t
        by_site[key][1] += 1
        total += t
    print(f'# one bf16 training step of DfMStereoPath: device time of copy-like ATen ops by call site, total {total / 1e3:.2f} ms')
    for (name, site, shp), (t, n) in sorted(by_site.items(), key=lambda kv: -kv[1][0])[:60]:
        print(f'{t / 1e3:8.3f} ms {n:4d}x  {name:14s} {shp:70s} {site[:110]}')


if __name__ == '__main__':
    main()
