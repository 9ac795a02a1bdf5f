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


class LazyDepthDistribution:
    """softmax(Upsample_x4(cost), dim=depth) WITHOUT the tensor: the low-resolution cost and the
    per-column softmax statistics (``depth_head_statistics``).  ``FrustumToVoxel`` /
    ``frustum_to_voxel_sample`` accept it in place of ``stereo_feat_softmax`` and evaluate the
    distribution at the voxels' corners on the fly -- bit-identical to sampling the materialised
    tensor (inference; the reference detaches the distribution anyway, feature_transformation.py:136).
    ``materialize()`` returns the (B, 1, sD, sH, sW) tensor if somebody needs it after all.

    Training (``cost_with_grad``: the cost as it sits in the autograd graph): ``DepthHead.loss`` accepts
    the object in place of ``depth_volumes`` and evaluates the logits of the valid pixels from the cost
    (``fused_depth_distribution_loss``); its backward and FrustumToVoxel's need no volume either."""

    def __init__(self, cost, col_max, col_sum, depth_samples, scale, cost_with_grad=None):
        self.cost, self.col_max, self.col_sum = cost, col_max, col_sum
        self.depth_samples, self.scale = depth_samples, scale
        self.cost_with_grad = cost_with_grad
        B, _, D, H, W = cost.shape
        self.shape = (B, 1, scale * D, scale * H, scale * W)
        self.dtype, self.device = cost.dtype, cost.device

    def detach(self):
        return self

    def flatten(self, start_dim=0, end_dim=-1):  # the detector folds (B, N) before the loss: N = 1 here
        return self

    def materialize(self):
        return depth_head_forward(self.cost, self.depth_samples, self.scale)[1]


def depth_head_statistics(stereo_features, depth_samples, downsample_factor=4, need_preds=True, keep_graph=False):
    """The inference form of ``depth_head_forward``: (LazyDepthDistribution, depth_preds) from one
    statistics pass over the low-resolution cost -- none of the three (B, 1, sD, sH, sW) tensors
    (472 MB each per sample at config K) is written.  ``need_preds=False`` skips the expectation
    pass (depth_preds is None): the detector's inference path does not read it.
    ``keep_graph=True`` (training): the distribution remembers ``stereo_features`` as it sits in the
    autograd graph, for ``DepthHead.loss`` (depth_preds itself carries no gradient here)."""
    _require_gpu(stereo_features, 'stereo_features')
    if stereo_features.dtype not in _DTYPES:
        raise TypeError('stereo_features must be float32 or bfloat16')
    assert stereo_features.dim() == 5 and stereo_features.shape[1] == 1
    x = stereo_features.detach().contiguous()
    s = int(downsample_factor)
    ds = depth_samples.to(device=x.device, dtype=torch.float32).contiguous()
    B, _, D, H, W = x.shape
    assert ds.numel() == s * D
    cmax = torch.empty((B, s * H, s * W), dtype=torch.float32, device=x.device)
    csum = torch.empty_like(cmax)
    pred = torch.empty((B, 1, s * H, s * W), dtype=x.dtype, device=x.device) if need_preds else None
    with torch.cuda.device(x.device):
        _capi.check(_capi.lib().dfm_depth_head_stats_fwd(B, D, H, W, s, _DTYPES[x.dtype], _ptr(x), _ptr(ds),
                                                         _ptr(cmax), _ptr(csum),
                                                         _ptr(pred) if need_preds else None,
                                                         _stream_ptr(x.device)))
    return LazyDepthDistribution(x, cmax, csum, ds, s, cost_with_grad=stereo_features if keep_graph else None), pred


# ---------------------------------------------------------------------------
# DepthHead.loss (mmdet3d/models/dense_heads/depth_head.py:75-188)
# ---------------------------------------------------------------------------
class _DepthLossFn(torch.autograd.Function):
    """per-pixel -sum_d p_d f(log_softmax_d) over the valid pixels (dfm_depth_loss_fwd/bwd)"""

    @staticmethod
    def forward(ctx, volumes, depth_img, ds, desc):
        lib = _capi.lib()
        device = volumes.device
        B, D, H, W = volumes.shape
        loss = torch.empty((B, H, W), dtype=torch.float32, device=device)
        valid = torch.empty((B, H, W), dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            _capi.check(lib.dfm_depth_loss_fwd(ctypes.byref(desc), _ptr(volumes), _ptr(depth_img),
                                               _ptr(ds), _ptr(loss), _ptr(valid), _stream_ptr(device)))
        ctx.save_for_backward(volumes, depth_img, ds)
        ctx.desc = desc
        ctx.mark_non_differentiable(valid)
        return loss, valid

    @staticmethod
    def backward(ctx, g_loss, _g_valid):
        volumes, depth_img, ds = ctx.saved_tensors
        lib = _capi.lib()
        device = volumes.device
        g = g_loss.contiguous().float()
        gv = torch.empty_like(volumes)
        with torch.cuda.device(device):
            _capi.check(lib.dfm_depth_loss_bwd(ctypes.byref(ctx.desc), _ptr(volumes), _ptr(depth_img),
                                               _ptr(ds), _ptr(g), _ptr(gv), _stream_ptr(device)))
        return gv, None, None, None


def _loss_desc(B, D, H, W, dtype, depth_samples, loss_type, min_depth, max_depth, alpha, gamma):
    d = _capi.DepthLossDesc()
    d.batch, d.num_depths, d.h, d.w = B, D, H, W
    d.focal = 1 if loss_type in ('focal', 'balanced_focal') else 0
    d.alpha, d.gamma = float(alpha), float(gamma)
    d.min_depth, d.max_depth = float(min_depth), float(max_depth)
    d.sigma = 0.0
    if loss_type in ('ce', 'balanced_ce', 'focal', 'balanced_focal'):
        d.target = _capi.DL_LINEAR
    elif loss_type == 'hard_ce':
        d.target = _capi.DL_HARD
    elif loss_type.startswith('gaussian'):
        d.target, d.sigma = _capi.DL_GAUSSIAN, float(loss_type.split('_')[1])
    elif loss_type.startswith('laplacian'):
        d.target, d.sigma = _capi.DL_LAPLACIAN, float(loss_type.split('_')[1])
    else:
        raise NotImplementedError(loss_type)
    # depth_interval = depth_samples[1] - depth_samples[0] in fp32 (depth_head.py:95); taken from
    # the host copy of two scalars, not a device sync on the hot tensor
    two = depth_samples[:2].detach().to('cpu', torch.float32)
    d.interval = float(two[1] - two[0])
    d.dtype = _DTYPES[dtype]
    return d


def depth_distribution_loss(depth_volumes, depth_img, depth_samples, loss_type, min_depth, max_depth,
                            alpha=1.0, gamma=2.0):
    """Unreduced depth-distribution loss per pixel and the validity mask.

    depth_volumes [B*N, D, H, W] logits, depth_img [B*N, H, W]; returns (pixel_loss [B*N,H,W]
    fp32 -- 0 at invalid pixels --, valid [B*N,H,W] bool).  ``loss_type`` as in the reference
    config: ce | balanced_ce | focal | balanced_focal | hard_ce | gaussian_<s> | laplacian_<s>.
    ``depth_volumes`` may be a ``LazyDepthDistribution`` (``fused_depth_distribution_loss``).
    """
    if isinstance(depth_volumes, LazyDepthDistribution):
        return fused_depth_distribution_loss(depth_volumes, depth_img, depth_samples, loss_type, min_depth,
                                             max_depth, alpha, gamma)
    _require_gpu(depth_volumes, 'depth_volumes')
    if depth_volumes.dtype not in _DTYPES:
        raise TypeError('depth_volumes must be float32 or bfloat16')
    vol = depth_volumes.contiguous()
    B, D, H, W = vol.shape
    img = depth_img.to(device=vol.device, dtype=torch.float32).contiguous()
    assert img.shape == (B, H, W), (img.shape, vol.shape)
    ds = depth_samples.to(device=vol.device, dtype=torch.float32).contiguous()
    assert ds.numel() == D
    d = _loss_desc(B, D, H, W, vol.dtype, depth_samples, loss_type, min_depth, max_depth, alpha, gamma)
    loss, valid = _DepthLossFn.apply(vol, img, ds, d)
    return loss, valid.bool()


class _FusedDepthLossFn(torch.autograd.Function):
    """the same per-pixel loss from the LOW-RESOLUTION cost (dfm_depth_loss_fused_fwd/bwd): the logits of
    a valid pixel's column are evaluated on the fly, the backward goes through the transposed upsample
    straight into the cost's gradient -- no (B, 1, sD, sH, sW) tensor in either direction"""

    @staticmethod
    def forward(ctx, cost, depth_img, ds, desc, scale):
        lib = _capi.lib()
        device = cost.device
        x = cost.contiguous()
        B, H, W = desc.batch, desc.h, desc.w
        loss = torch.empty((B, H, W), dtype=torch.float32, device=device)
        valid = torch.empty((B, H, W), dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            _capi.check(lib.dfm_depth_loss_fused_fwd(ctypes.byref(desc), _ptr(x), scale, _ptr(depth_img), _ptr(ds),
                                                     _ptr(loss), _ptr(valid), _stream_ptr(device)))
        ctx.save_for_backward(x, depth_img, ds)
        ctx.desc, ctx.scale = desc, scale
        ctx.mark_non_differentiable(valid)
        return loss, valid

    @staticmethod
    def backward(ctx, g_loss, _g_valid):
        x, depth_img, ds = ctx.saved_tensors
        lib = _capi.lib()
        device = x.device
        g = g_loss.contiguous().float()
        gx = torch.zeros(x.shape, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            _capi.check(lib.dfm_depth_loss_fused_bwd(ctypes.byref(ctx.desc), _ptr(x), ctx.scale, _ptr(depth_img),
                                                     _ptr(ds), _ptr(g), _ptr(gx), _stream_ptr(device)))
        return gx.to(x.dtype), None, None, None, None


def fused_depth_distribution_loss(dist, depth_img, depth_samples, loss_type, min_depth, max_depth,
                                  alpha=1.0, gamma=2.0):
    """``depth_distribution_loss`` on a ``LazyDepthDistribution`` (DepthHead fused, training): pixel_loss is
    bit-identical to the loss on the materialised ``depth_volumes``; differentiable w.r.t.
    ``dist.cost_with_grad``."""
    cost = dist.cost_with_grad if dist.cost_with_grad is not None else dist.cost
    _require_gpu(cost, 'cost')
    B, _, sD, sH, sW = dist.shape
    img = depth_img.to(device=cost.device, dtype=torch.float32).contiguous()
    assert img.shape == (B, sH, sW), (img.shape, dist.shape)
    ds = depth_samples.to(device=cost.device, dtype=torch.float32).contiguous()
    assert ds.numel() == sD
    d = _loss_desc(B, sD, sH, sW, cost.dtype, depth_samples, loss_type, min_depth, max_depth, alpha, gamma)
    loss, valid = _FusedDepthLossFn.apply(cost, img, ds, d, int(dist.scale))
    return loss, valid.bool()
