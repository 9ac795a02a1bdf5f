"""Host-side mirror of ``DepthHead.forward`` with ``with_convs=False``
(mmdet3d/models/dense_heads/depth_head.py:190-212): one HIP launch
(``dfm_depth_head_fwd``) instead of Upsample + softmax + weighted sum."""
import ctypes

import torch

from . import _capi
from .plane_sweep import _DTYPES, _ptr, _require_gpu, _stream_ptr


class _DepthHeadFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, ds, s):
        lib = _capi.lib()
        device = x.device
        B, _, D, H, W = x.shape
        vol = torch.empty((B, 1, s * D, s * H, s * W), dtype=x.dtype, device=device)
        soft = torch.empty_like(vol)
        pred = torch.empty((B, 1, s * H, s * W), dtype=x.dtype, device=device)
        with torch.cuda.device(device):
            _capi.check(
                lib.dfm_depth_head_fwd(B, D, H, W, s, _DTYPES[x.dtype], _ptr(x), _ptr(ds), _ptr(vol),
                                       _ptr(soft), _ptr(pred), _stream_ptr(device)))
        ctx.save_for_backward(x, ds)
        ctx.s = s
        return vol, soft, pred

    @staticmethod
    def backward(ctx, g_vol, g_soft, g_pred):
        x, ds = ctx.saved_tensors
        lib = _capi.lib()
        device = x.device
        B, _, D, H, W = x.shape
        gs = [None if g is None else g.contiguous().to(x.dtype) for g in (g_vol, g_soft, g_pred)]
        gx = torch.zeros(x.shape, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            _capi.check(
                lib.dfm_depth_head_bwd(B, D, H, W, ctx.s, _DTYPES[x.dtype], _ptr(x), _ptr(ds),
                                       *(None if g is None else _ptr(g) for g in gs), _ptr(gx),
                                       _stream_ptr(device)))
        return gx.to(x.dtype), None, None


def depth_head_forward(stereo_features, depth_samples, downsample_factor=4):
    """(B, 1, D, H, W) -> depth_volumes, depth_volumes_softmax (B, 1, sD, sH, sW),
    depth_preds (B, 1, sH, sW).  Differentiable w.r.t. stereo_features."""
    _require_gpu(stereo_features, 'stereo_features')
    if stereo_features.dtype not in _DTYPES:
        raise TypeError('stereo_features must be float32 or bfloat16')
    assert stereo_features.dim() == 5 and stereo_features.shape[1] == 1, \
        'with_convs=False expects a single-channel cost volume'
    x = stereo_features.contiguous()
    s = int(downsample_factor)
    ds = depth_samples.to(device=x.device, dtype=torch.float32).contiguous()
    assert ds.numel() == s * x.shape[2]
    return _DepthHeadFn.apply(x, ds, s)
