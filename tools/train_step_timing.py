#!/usr/bin/env python
"""DfMBackbone forward + backward (training step without the optimizer) at config K, bf16 NDHWC:
MFMA forward / backward-data, MIOpen backward-weight, HIP GroupNorm / plane-sweep backward.  GPU box.
DFM_MIOPEN_FIND=1 turns MIOpen's autotuning on (minutes)."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
pkg = importlib.import_module('depth-from-motion_amd')
mods = importlib.import_module('depth-from-motion_amd.modules')
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = os.environ.get('DFM_MIOPEN_FIND') == '1'


def main():
    iters = int(os.environ.get('DFM_ITERS', '3'))
    torch.manual_seed(0)
    m = mods.DfMBackbone(in_channels=32).to(dev).to(torch.bfloat16).train()
    m.downsampled_depth = pkg.prepare_depth(dict(num_bins=288, depth_min=2, depth_max=59.6, downsample_factor=4))[0]
    m.volume_memory_format = torch.channels_last_3d
    meta = dict(ori_cam2img=bench.KITTI_P2, cur2prevs=torch.from_numpy(bench.poses(1, 2)), ori_shape=(375, 1242, 3),
                pad_shape=(320, 1280, 3), crop_offset=[0, 55], flip=False, scale_factor=[1.0])
    g = torch.Generator().manual_seed(1)
    cur = torch.randn(1, 32, 320, 1280, generator=g).to(dev).bfloat16().requires_grad_(True)
    prev = torch.randn(1, 32, 320, 1280, generator=g).to(dev).bfloat16().requires_grad_(True)

    def step():
        m.zero_grad(set_to_none=True)
        cost, sf, mf = m(cur, prev, [meta])
        (cost.float().mean() + sf.float().mean() + mf.float().mean()).backward()

    def fwd():
        with torch.no_grad():
            m(cur, prev, [meta])
    for name, fn in (('forward (no grad)', fwd), ('forward + backward', step)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        print(f'DfMBackbone config K bf16 NDHWC {name:20s}: {(time.perf_counter() - t) * 1e3 / iters:8.2f} ms', flush=True)


if __name__ == '__main__':
    main()
