"""The plane sweep fused into the first aggregation convolutions (csrc/sweep_conv.hip):
``build_dfm_cost`` -> ``dres0.conv`` / ``dres0_mono.conv`` in ONE kernel, the (B, 2C, D, H, W) cost
volume never written (reference mmdet3d/models/backbones/dfm_backbone.py:161-176,189).

Inference path of ``DfMBackbone`` for 32-channel bf16 feature maps (config K).  Training keeps the
materialised volume: the weight gradients of dres0 / dres0_mono contract it with the output gradient.
"""
import ctypes

import torch

from . import _capi
from .plane_sweep import _make_desc, _nhwc, _ptr, _require_gpu, _stream_ptr, camera_matrices


def pack_sweep_conv_weights(w_stereo, w_mono):
    """(32, 64, 3, 3, 3) and (32, 32, 3, 3, 3) GPU weights (fp32 / bf16) -> the register-fragment buffer
    of the fused kernel's four wave roles (+ its zero pixel)"""
    assert w_stereo.is_cuda and tuple(w_stereo.shape) == (32, 64, 3, 3, 3) and tuple(w_mono.shape) == (32, 32, 3, 3, 3)
    ws, wm = w_stereo.detach().contiguous(), w_mono.detach().contiguous()
    if ws.dtype not in (torch.float32, torch.bfloat16) or wm.dtype != ws.dtype:
        ws, wm = ws.float(), wm.float()
    lib = _capi.lib()
    packed = torch.empty(lib.dfm_sweep_conv_weight_bytes(), dtype=torch.uint8, device=ws.device)
    with torch.cuda.device(ws.device):
        _capi.check(lib.dfm_sweep_conv_pack_weights(_ptr(ws), _ptr(wm), _capi.DFM_F32 if ws.dtype == torch.float32
                                                    else _capi.DFM_BF16, _ptr(packed), _stream_ptr(ws.device)))
    return packed


def sweep_conv_supported(cur_feats):
    return (cur_feats.is_cuda and cur_feats.dtype == torch.bfloat16 and cur_feats.dim() == 4 and
            cur_feats.shape[1] == 32)


def sweep_dres0(cur_feats, prev_feats, depths, feat_sample_factor, cost_sample_factor, cam2imgs, cur2prevs, img_shape,
                packed, flip=False, img_crop_offset=(0, 0), img_scale_factor=1.0, depth_chunk=0, cam2img_inv=None):
    """Arguments as ``build_dfm_cost`` (+ ``packed`` from ``pack_sweep_conv_weights``).  Returns
    ``(y_stereo, partials_stereo, y_mono, partials_mono)``: the outputs of ``dres0.conv(cost_raw)`` and
    ``dres0_mono.conv(cost_raw[:, :32])`` as (B, 32, D, H, W) bf16 channels_last_3d tensors BEFORE
    GroupNorm / ReLU, and their per-channel moment partials (B, 32, splits, 3) for
    ``group_norm(..., partials=...)``."""
    _require_gpu(cur_feats, 'cur_feats')
    _require_gpu(prev_feats, 'prev_feats')
    if not sweep_conv_supported(cur_feats) or prev_feats.dtype != cur_feats.dtype or prev_feats.shape != cur_feats.shape:
        raise TypeError('sweep_dres0 takes two (B, 32, H, W) bfloat16 feature maps')
    device = cur_feats.device
    if not _nhwc(cur_feats):
        cur_feats = cur_feats.contiguous(memory_format=torch.channels_last)
    if not _nhwc(prev_feats):
        prev_feats = prev_feats.contiguous(memory_format=torch.channels_last)
    depths = depths.reshape(-1).to(device=device, dtype=torch.float32).contiguous()
    B = cur_feats.shape[0]
    desc = _make_desc(cur_feats, depths.numel(), feat_sample_factor, cost_sample_factor, img_shape, flip,
                      img_crop_offset, img_scale_factor)
    P, Pinv, T = camera_matrices(cam2imgs, cur2prevs, B, device, cam2img_inv)
    lib = _capi.lib()
    splits = lib.dfm_sweep_conv_stats_splits(ctypes.byref(desc), depth_chunk)
    if splits <= 0:
        _capi.check(lib.dfm_sweep_conv_fwd(ctypes.byref(desc), None, None, None, None, None, None, None, None, None,
                                           None, None, depth_chunk, None))  # raises with the library's reason
    shape = (B, desc.num_depths, desc.h_out, desc.w_out, 32)
    ys = torch.empty(shape, dtype=torch.bfloat16, device=device)
    ym = torch.empty(shape, dtype=torch.bfloat16, device=device)
    ps = torch.empty((B, 32, splits, 3), dtype=torch.float32, device=device)
    pm = torch.empty((B, 32, splits, 3), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _capi.check(lib.dfm_sweep_conv_fwd(ctypes.byref(desc), _ptr(cur_feats), _ptr(prev_feats), _ptr(depths), _ptr(P),
                                           _ptr(Pinv), _ptr(T), _ptr(packed), _ptr(ys), _ptr(ym), _ptr(ps), _ptr(pm),
                                           depth_chunk, _stream_ptr(device)))
    return ys.permute(0, 4, 1, 2, 3), ps, ym.permute(0, 4, 1, 2, 3), pm
