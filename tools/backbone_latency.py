#!/usr/bin/env python
"""Latency of ONE DfMBackbone.forward at config K (bf16 NDHWC, synchronised after every forward -- bench.py times
back-to-back forwards, where the host runs ahead of the device): the two stacks issued alternately, the whole mono
stack first (round 5), one stream.  GPU box.  usage: python tools/backbone_latency.py"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
pkg = importlib.import_module('depth-from-motion_amd')
mods = importlib.import_module('depth-from-motion_amd.modules')
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = mods.DfMBackbone(in_channels=32).to(dev).to(torch.bfloat16).eval()
m.downsampled_depth = pkg.prepare_depth(dict(num_bins=288, depth_min=2, depth_max=59.6, downsample_factor=4))[0]
m.volume_memory_format = torch.channels_last_3d
meta = dict(ori_cam2img=bench.KITTI_P2, cur2prevs=torch.from_numpy(bench.poses(1, 2)), ori_shape=(375, 1242, 3),
            pad_shape=(320, 1280, 3), crop_offset=[0, 55], flip=False, scale_factor=[1.0])
cur = torch.randn(1, 32, 320, 1280).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
prev = torch.randn(1, 32, 320, 1280).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
def lat(n=40):
    with torch.no_grad():
        for _ in range(5): m(cur, prev, [meta])
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); m(cur, prev, [meta]); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3
for mode in (True, False, True, False):
    m.interleaved_issue = mode
    print('interleaved' if mode else 'sequential ', 'single-forward latency (sync per forward): %.3f ms' % lat())
m.two_streams = False
print('one stream  single-forward latency: %.3f ms' % lat())
