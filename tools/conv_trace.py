#!/usr/bin/env python
"""s_memtime stamps of workgroup (0, 0, 0) of conv3d_k3_c32_kernel (debug build: DFM_HIP_LIB=.../libdfm_hip_dbg.so from
build_hip(debug_hooks=True, out=...)): per wave, the median ticks between consecutive stamps over the traced planes.
stamps: 0 plane start, 1 next slab's LDS-DMA issued, 2 the plane's 216 MFMAs done, 3 epilogue done, 4 after the barrier"""
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
x = torch.randn(1, 32, 72, 80, 320, device=dev).bfloat16().contiguous(memory_format=torch.channels_last_3d)
w = (torch.randn(32, 32, 3, 3, 3, device=dev) * 0.03)
pk = cv.pack_conv3d_weights(w)
for stats in (False, True):
    run = lambda: cv.conv3d_k3_c32(x, pk, stats=stats)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    buf = torch.zeros(4 * 16 * 8, dtype=torch.int64, device=dev)
    lib.dfm_debug_set_cv_trace.argtypes = [ctypes.c_void_p]
    lib.dfm_debug_set_cv_trace(ctypes.c_void_p(buf.data_ptr()))
    run()
    torch.cuda.synchronize()
    lib.dfm_debug_set_cv_trace(None)
    t = buf.cpu().numpy().reshape(4, 16, 8).astype(np.float64)
    print(f'conv3d_k3_c32 32->32 72x80x320, statistics epilogue: {stats}')
    for wv in range(4):
        planes = [k for k in range(1, 13) if t[wv, k, 0] > 0 and t[wv, k + 1, 0] > 0]
        seg = np.array([[t[wv, k, i + 1] - t[wv, k, i] for i in range(4)] + [t[wv, k + 1, 0] - t[wv, k, 4]] for k in planes])
        per = np.array([t[wv, k + 1, 0] - t[wv, k, 0] for k in planes])
        print(f'  wave {wv}: plane period median {np.median(per):8.0f} ticks; DMA issue, MFMA steps, epilogue, barrier, loop: ',
              ' '.join(f'{v:7.0f}' for v in np.median(seg, axis=0)))
