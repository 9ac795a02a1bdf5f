#!/usr/bin/env python
"""OutdoorImVoxelNeck / DfMNeck forward (eval) at the config-W voxel volume (220 x 300 x 12):
bf16 channels_last_3d with the MFMA convolutions (BatchNorm folded into the epilogue) vs the same
module with torch's convolutions (MIOpen) + BatchNorm3d + ReLU.  GPU box.
DFM_MIOPEN_FIND=1 turns MIOpen's autotuning on for the torch path."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mods = importlib.import_module('depth-from-motion_amd.modules')
cv = importlib.import_module('depth-from-motion_amd.conv3d')
dev = torch.device('cuda:0')
torch.backends.cudnn.benchmark = os.environ.get('DFM_MIOPEN_FIND') == '1'
FLOPS = {'OutdoorImVoxelNeck': 3.21e12, 'DfMNeck': 7.65e12}   # SURVEY 8a a8 / a9, per sample


def run(m, x, iters):
    with torch.no_grad():
        for _ in range(2):
            y = m(x)[0]
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(iters):
            y = m(x)[0]
        torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3 / iters, y


def main():
    iters = int(os.environ.get('DFM_ITERS', '5'))
    for name, kw, cin in (('OutdoorImVoxelNeck', dict(in_channels=64, out_channels=256), 64),
                          ('DfMNeck', dict(in_channels=64, out_channels=256, num_frames=2), 128)):
        torch.manual_seed(0)
        m = getattr(mods, name)(**kw).to(dev).eval().to(torch.bfloat16)
        x = torch.randn(1, cin, 220, 300, 12, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
        ms, y = run(m, x, iters)
        print(f'{name:20s} bf16 NDHWC  MFMA (BN folded): {ms:8.2f} ms  {FLOPS[name] / ms / 1e9:7.1f} TFLOP/s '
              f'({FLOPS[name] / ms / 1e9 / 25:4.1f} % of 2.5 PF)', flush=True)
        if os.environ.get('DFM_SKIP_TORCH') != '1':
            elig = cv.MfmaConv3dG.eligible
            cv.MfmaConv3dG.eligible = lambda self, x: False
            ms2, y2 = run(m, x, iters)
            cv.MfmaConv3dG.eligible = elig
            d = float((y.float() - y2.float()).abs().max())
            print(f'{name:20s} bf16 NDHWC  torch conv + BN + ReLU : {ms2:8.2f} ms  {FLOPS[name] / ms2 / 1e9:7.1f} TFLOP/s'
                  f'   max |diff| {d:.3g} (max |y| {float(y2.float().abs().max()):.3g})', flush=True)
        del m, x


if __name__ == '__main__':
    main()
