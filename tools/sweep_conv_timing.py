#!/usr/bin/env python
"""The fused plane sweep + dres0 / dres0_mono kernel (csrc/sweep_conv.hip) against the unfused
sequence it replaces, config K: 32-channel 320x1280 bf16 maps (NHWC, as SPPUNetNeck emits them),
csf 4, D = 72 -> 72x80x320 volume.  Times are per forward, CUDA events over DFM_ITERS launches.
DFM_DEPTH_CHUNK=n overrides the fused kernel's depth chunk."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module('depth-from-motion_amd')
sc = importlib.import_module('depth-from-motion_amd.sweep_conv')
cv = importlib.import_module('depth-from-motion_amd.conv3d')
dev = torch.device('cuda:0')
iters = int(os.environ.get('DFM_ITERS', '20'))
B = int(os.environ.get('DFM_BATCH', '1'))
g = torch.Generator().manual_seed(1)
cur = torch.randn(B, 32, 320, 1280, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
prev = torch.randn(B, 32, 320, 1280, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
depths = pkg.prepare_depth(dict(num_bins=288, depth_min=2, depth_max=59.6, downsample_factor=4))[0].to(dev)
P = torch.from_numpy(bench.KITTI_P2)[None].repeat(B, 1, 1).to(dev)
T = torch.from_numpy(bench.poses(B, 2)).to(dev)
ws = (torch.randn(32, 64, 3, 3, 3, generator=g) * 0.03).to(dev).bfloat16()
wm = (torch.randn(32, 32, 3, 3, 3, generator=g) * 0.04).to(dev).bfloat16()
args = (depths, 1, 4, P, T, (375, 1242))
kw = dict(img_crop_offset=(0, 55))


def timed(fn):
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


packed = sc.pack_sweep_conv_weights(ws, wm)
pa, pb, pm_ = cv.pack_conv3d_weights(ws, 0), cv.pack_conv3d_weights(ws, 32), cv.pack_conv3d_weights(wm, 0)
dchunk = int(os.environ.get('DFM_DEPTH_CHUNK', '0'))


def fused():
    return sc.sweep_dres0(cur, prev, *args, packed, depth_chunk=dchunk, **kw)


def unfused():
    vol = pkg.build_dfm_cost(cur, prev, *args, memory_format=torch.channels_last_3d, **kw)
    part = cv.conv3d_k3_c32(vol[:, :32], pa, out_f32=True)
    ys = cv.conv3d_k3_c32(vol[:, 32:], pb, acc_in=part, stats=True)
    ym = cv.conv3d_k3_c32(vol[:, :32], pm_, stats=True)
    return ys, ym


with torch.no_grad():
    tu = timed(unfused)
    tf = timed(fused)
    (ys, ps), (ym, pm) = unfused()
    fs, _, fm, _ = fused()
flops = 2 * 27 * (64 + 32) * 32 * 72 * 80 * 320 * B
print(f'config K, batch {B}: unfused (sweep_cl + dres0 two halves + dres0_mono) {tu:.3f} ms, '
      f'fused {tf:.3f} ms = {flops / tf / 1e9:.0f} TFLOP/s ({flops / tf / 1e9 / 2500 * 100:.1f} % of 2.5 PFLOP/s)')
print(f'max |fused - unfused|: stereo {float((fs.float() - ys.float()).abs().max()):.4f} of {float(ys.float().abs().max()):.3f}, '
      f'mono {float((fm.float() - ym.float()).abs().max()):.4f} of {float(ym.float().abs().max()):.3f}')
