#!/usr/bin/env python
"""Which elementwise adds / copies of the DfMBackbone training step are slow, and where do they come
from?  torch.profiler with shapes + python stacks; prints the device-time-sorted aten::add / add_ /
copy_ / cat events of one forward + backward (config K, bf16 NDHWC)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module('depth-from-motion_amd')
mods = importlib.import_module('depth-from-motion_amd.modules')
dev = torch.device('cuda:0')
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


for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.name in ('aten::add', 'aten::add_', 'aten::copy_', 'aten::cat', 'aten::contiguous',
                                               'aten::clone', 'aten::zeros', 'aten::fill_', 'aten::mean', 'aten::to')]
evs.sort(key=lambda e: -e.device_time_total)
for e in evs[:14]:
    stack = [s for s in (e.stack or []) if 'depth-from-motion_amd' in s or 'tools/' in s][:3]
    print(f'{e.device_time_total:9.1f} us  {e.name:16s} {e.input_shapes}  <- {" | ".join(s.strip()[-90:] for s in stack)}')
