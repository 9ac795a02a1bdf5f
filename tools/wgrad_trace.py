#!/usr/bin/env python
"""s_memtime stamps of workgroup (0, 0) of conv3d_wgrad_kernel (debug build: DFM_HIP_LIB=.../libdfm_hip_dbg.so from
build_hip(debug_hooks=True, out=...)): per wave, the median ticks between consecutive stamps over the traced tiles.
stamps: 0 tile start, 1 first batch of staging loads issued, 2 ... arrived, 3 staging done, 4 after barrier,
        5 MFMA phase done, 6 after barrier"""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('depth-from-motion_amd')
cv = importlib.import_module('depth-from-motion_amd.conv3d')
lib = pkg._capi.lib()
dev = torch.device('cuda:0')
cl = torch.channels_last_3d
for (a, b, size, stride) in ((32, 32, (72, 80, 320), 1), (64, 32, (72, 80, 320), 2)):
    x = torch.randn(1, b, *size, device=dev).bfloat16().contiguous(memory_format=cl)
    osz = tuple((s - 1) // stride + 1 for s in size)
    gy = torch.randn(1, a, *osz, device=dev).bfloat16().contiguous(memory_format=cl)
    for _ in range(2):
        cv.conv3d_weight_grad(x, gy, stride, 1)
    torch.cuda.synchronize()
    buf = torch.zeros(3 * 16 * 8, dtype=torch.int64, device=dev)
    lib.dfm_debug_set_wg_trace.argtypes = [ctypes.c_void_p]
    lib.dfm_debug_set_wg_trace(ctypes.c_void_p(buf.data_ptr()))
    cv.conv3d_weight_grad(x, gy, stride, 1)
    torch.cuda.synchronize()
    lib.dfm_debug_set_wg_trace(None)
    t = buf.cpu().numpy().reshape(3, 16, 8).astype(np.float64)
    print(f'wgrad {a}<-{b} {size} stride {stride}')
    for w in range(3):
        tiles = [k for k in range(1, 15) if t[w, k, 0] > 0 and t[w, k + 1, 0] > 0]
        seg = np.array([[t[w, k, i + 1] - t[w, k, i] for i in range(6)] + [t[w, k + 1, 0] - t[w, k, 6]] for k in tiles])
        per = np.array([t[w, k + 1, 0] - t[w, k, 0] for k in tiles])
        print(f'  wave {w}: tile period median {np.median(per):8.0f} ticks; load issue, load wait, rest of staging, barrier, MFMA phase, '
              f'barrier, loop: ', ' '.join(f'{v:7.0f}' for v in np.median(seg, axis=0)))
