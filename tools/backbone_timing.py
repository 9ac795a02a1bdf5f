#!/usr/bin/env python
"""DfMBackbone forward at the config-K size (1 sample, 32-ch 320x1280 feats, D=72):
plane sweep + 3-D aggregation; reference memory layout vs channels_last_3d end to end
(cost volume written (B,D,H,W,2C), NDHWC Conv3d, channels-last fused GroupNorm).
MIOpen autotuning (cudnn.benchmark) is OFF unless DFM_MIOPEN_FIND=1: it costs minutes of GPU
time per process.  In bf16 + channels_last_3d the 32-channel 3x3x3 convolutions run in the
hand-written MFMA kernel (DFM_NO_MFMA_CONV=1 sends them to MIOpen for an A/B)."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module('depth-from-motion_amd')
mods = importlib.import_module('depth-from-motion_amd.modules')
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = os.environ.get('DFM_MIOPEN_FIND') == '1'
if os.environ.get('DFM_NO_MFMA_CONV') == '1':
    importlib.import_module('depth-from-motion_amd.conv3d').MfmaConv3d.eligible = lambda self, x: False
if os.environ.get('DFM_NO_MFMA_CONV_G') == '1':  # hourglass convolutions back to MIOpen
    _cv = importlib.import_module('depth-from-motion_amd.conv3d')
    _cv.MfmaConv3dG.eligible = lambda self, x: False
    _cv.MfmaConvTranspose3d.eligible = lambda self, x: False
outs = {}
only = os.environ.get('DFM_ONLY', '')
for dtype in ((torch.bfloat16,) if only == 'bf16' else (torch.float32, torch.bfloat16)):
    fmts = (torch.contiguous_format, torch.channels_last_3d)
    if os.environ.get('DFM_ONLY_FMT') == 'cl':
        fmts = (torch.channels_last_3d,)
    for fmt in fmts:
        torch.manual_seed(0)
        m = mods.DfMBackbone(in_channels=32).to(dev).to(dtype).eval()
        m.downsampled_depth = pkg.prepare_depth(dict(num_bins=288, depth_min=2, depth_max=59.6, downsample_factor=4))[0]
        if fmt == torch.channels_last_3d:
            for sub in m.modules():  # the 1x1 Conv2d gate stays as it is
                if isinstance(sub, (torch.nn.Conv3d, torch.nn.ConvTranspose3d)):
                    sub.to(memory_format=torch.channels_last_3d)
            m.volume_memory_format = torch.channels_last_3d
        meta = dict(ori_cam2img=bench.KITTI_P2, cur2prevs=torch.from_numpy(bench.poses(1, 2)), ori_shape=(375, 1242, 3),
                    pad_shape=(320, 1280, 3), crop_offset=[0, 55], flip=False, scale_factor=[1.0])
        g = torch.Generator().manual_seed(1)
        cur = torch.randn(1, 32, 320, 1280, generator=g).to(dev).to(dtype)
        prev = torch.randn(1, 32, 320, 1280, generator=g).to(dev).to(dtype)
        with torch.no_grad():
            for _ in range(2): out = m(cur, prev, [meta])
            torch.cuda.synchronize(); t = time.perf_counter()
            iters = int(os.environ.get('DFM_ITERS', '5'))
            for _ in range(iters): out = m(cur, prev, [meta])
            torch.cuda.synchronize(); ms = (time.perf_counter() - t) * 1e3 / iters
        outs[(dtype, fmt)] = out[0].float()
        print(f'DfMBackbone.forward {str(dtype)[6:]:9s} {str(fmt)[6:]:18s}: {ms:8.2f} ms', flush=True)
    if len(fmts) < 2:
        continue
    a, b = outs[(dtype, torch.contiguous_format)], outs[(dtype, torch.channels_last_3d)]
    print(f'  max |cost(channels_last) - cost(contiguous)| = {float((a - b).abs().max()):.3e} (max |cost| {float(a.abs().max()):.3e})')
