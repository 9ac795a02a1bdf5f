"""Host-side mirror of ``DepthHead.forward`` with ``with_convs=False``
(mmdet3d/models/dense_heads/depth_head.py:190-212): one HIP launch
(``dfm_depth_head_fwd``) instead of Upsample + softmax + weighted sum."""
import ctypes

import torch

from . import _capi
from .plane_sweep import _DTYPES, _ptr, _require_gpu, _stream_ptr


def depth_head_forward(stereo_features, depth_samples, downsample_factor=4):
    """(B, 1, D, H, W) -> depth_volumes, depth_volumes_softmax (B, 1, sD, sH, sW),
    depth_preds (B, 1, sH, sW)."""
    _require_gpu(stereo_features, 'stereo_features')
    if stereo_features.dtype not in _DTYPES:
        raise TypeError('stereo_features must be float32 or bfloat16')
    assert stereo_features.dim() == 5 and stereo_features.shape[1] == 1, \
        'with_convs=False expects a single-channel cost volume'
    lib = _capi.lib()
    device = stereo_features.device
    x = stereo_features.contiguous()
    B, _, D, H, W = x.shape
    s = int(downsample_factor)
    ds = depth_samples.to(device=device, dtype=torch.float32).contiguous()
    assert ds.numel() == s * D
    vol = torch.empty((B, 1, s * D, s * H, s * W), dtype=x.dtype, device=device)
    soft = torch.empty_like(vol)
    pred = torch.empty((B, 1, s * H, s * W), dtype=x.dtype, device=device)
    with torch.cuda.device(device):
        _capi.check(
            lib.dfm_depth_head_fwd(B, D, H, W, s, _DTYPES[x.dtype], _ptr(x), _ptr(ds), _ptr(vol),
                                   _ptr(soft), _ptr(pred), _stream_ptr(device)))
    return vol, soft, pred
