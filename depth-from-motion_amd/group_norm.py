"""Fused GroupNorm(+ReLU) for the cost-volume aggregation stacks: one HIP
reduction pass + one apply pass (``dfm_group_norm_fwd/bwd``) instead of torch's
GroupNorm kernel followed by a separate ReLU.

``HipGroupNorm`` subclasses ``nn.GroupNorm`` (same parameters, same
``state_dict`` keys), so it drops into ``ConvModule`` / ``hourglass`` without
touching checkpoints.
"""
import ctypes
import os
import weakref

import torch
from torch import nn

from . import _capi
from .plane_sweep import _DTYPES, _Workspace, _ptr, _stream_ptr


_F32_CACHE = {}


def _f32_params(weight, bias):
    """fp32 copies of the affine parameters the kernels read; bf16 modules would otherwise launch two
    conversion kernels per call (36 per DfMBackbone forward).  Cached per parameter and version."""
    if weight.dtype == torch.float32 and bias.dtype == torch.float32:
        return weight.detach().contiguous(), bias.detach().contiguous()
    key = (id(weight), id(bias))
    ver = (weight._version, bias._version, weight.device, weight.data_ptr(), bias.data_ptr())
    hit = _F32_CACHE.get(key)
    # the entry holds weak references: an id() reused by a new tensor never matches a dead one
    if hit is None or hit[0] != ver or hit[3]() is not weight or hit[4]() is not bias:
        if len(_F32_CACHE) > 1024:
            _F32_CACHE.clear()
        hit = (ver, weight.detach().float().contiguous(), bias.detach().float().contiguous(),
               weakref.ref(weight), weakref.ref(bias))
        _F32_CACHE[key] = hit
    return hit[1], hit[2]


_XMASK = os.environ.get('DFM_GN_KEEP_Y') != '1'   # (A/B runs: the backward reads the ReLU mask from the kept output)


class _GroupNormFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, relu, partials=None, residual=None, stats_out=None):
        lib = _capi.lib()
        device = x.device
        n, c = x.shape[:2]
        # channels-last input (what the NDHWC convolutions hand over) stays channels-last
        vec = 16 // x.element_size()
        cl = (x.dim() in (4, 5) and c % vec == 0 and c <= 256 and ((c // vec) & (c // vec - 1)) == 0
              and not x.is_contiguous()
              and x.is_contiguous(memory_format=torch.channels_last_3d if x.dim() == 5
                                  else torch.channels_last))
        if not cl:
            x = x.contiguous()
        spatial = x.numel() // (n * c)
        y = torch.empty_like(x)  # preserves the memory format
        mean = torch.empty(n * groups, dtype=torch.float32, device=device)
        rstd = torch.empty_like(mean)
        w32, b32 = _f32_params(weight, bias)
        nbytes = lib.dfm_group_norm_workspace_bytes(n, c, spatial, groups)
        ws = _Workspace.get(device, nbytes)
        fn = lib.dfm_group_norm_fwd_channels_last if cl else lib.dfm_group_norm_fwd
        fused_res = residual is not None and cl and residual.dtype == x.dtype and \
            residual.shape == x.shape and residual.stride() == x.stride()
        rp = _ptr(residual) if fused_res else None
        with torch.cuda.device(device):
            if partials is not None:
                # statistics came from the producer (MFMA conv epilogue): normalisation pass only
                assert cl and partials.shape[:2] == (n, groups) and partials.is_contiguous()
                _capi.check(lib.dfm_group_norm_apply_channels_last_res(
                    n, c, spatial, groups, eps, _DTYPES[x.dtype], int(relu and (fused_res or residual is None)),
                    _ptr(x), _ptr(w32), _ptr(b32), rp, _ptr(y), _ptr(mean), _ptr(rstd), _ptr(partials),
                    partials.shape[2], _ptr(ws), nbytes, _stream_ptr(device)))
            elif cl:
                _capi.check(lib.dfm_group_norm_fwd_channels_last_res(
                    n, c, spatial, groups, eps, _DTYPES[x.dtype], int(relu and (fused_res or residual is None)),
                    _ptr(x), _ptr(w32), _ptr(b32), rp, _ptr(y), _ptr(mean), _ptr(rstd), _ptr(ws), nbytes,
                    _stream_ptr(device)))
            else:
                _capi.check(fn(n, c, spatial, groups, eps, _DTYPES[x.dtype], int(relu and residual is None),
                               _ptr(x), _ptr(w32), _ptr(b32), _ptr(y), _ptr(mean), _ptr(rstd), _ptr(ws), nbytes,
                               _stream_ptr(device)))
        if residual is not None and not fused_res:
            # a layout the kernel does not fuse (NCDHW, mismatched strides): plain torch ops
            y = y + residual
            if relu:
                y = torch.relu_(y)
        # y = relu(gn(x)) of a channels-last tensor without a residual: the backward recomputes the ReLU mask from x
        # (dfm_group_norm_bwd_channels_last_xmask) -- y is not kept for it, one activation less per layer in the graph
        xmask = bool(relu) and cl and residual is None and _XMASK
        ctx.save_for_backward(x, y if (relu and not xmask) else x, mean, rstd, w32, b32)
        ctx.cfg = (groups, bool(relu), weight.dtype, bias.dtype, cl, residual is not None, xmask)
        if stats_out is not None:  # (mean, rstd) per (sample, group): HipBatchNorm3d's running statistics
            stats_out.append((mean, rstd))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, mean, rstd, w32, b32 = ctx.saved_tensors
        groups, relu, wdt, bdt, cl, has_res, xmask = ctx.cfg
        lib = _capi.lib()
        device = x.device
        n, c = x.shape[:2]
        spatial = x.numel() // (n * c)
        want_res = has_res and ctx.needs_input_grad[7]
        # (the channels-last kernels overwrite the parameter gradients: no zero fill -- 2 launches per layer saved)
        gw = (torch.empty if cl else torch.zeros)(c, dtype=torch.float32, device=device)
        gb = (torch.empty if cl else torch.zeros)(c, dtype=torch.float32, device=device)
        nbytes = lib.dfm_group_norm_workspace_bytes(n, c, spatial, groups)
        ws = _Workspace.get(device, nbytes)
        if cl:
            # channels-last kernels: no layout round trip; the masked gradient of a fused residual
            # input is a second output of the same pass
            fmt = torch.channels_last_3d if x.dim() == 5 else torch.channels_last
            gy = gy.to(x.dtype).contiguous(memory_format=fmt)
            gx = torch.empty_like(x)
            gres = torch.empty_like(x) if want_res and relu else None
            if xmask:
                with torch.cuda.device(device):
                    _capi.check(lib.dfm_group_norm_bwd_channels_last_xmask(
                        n, c, spatial, groups, _DTYPES[x.dtype], _ptr(gy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(w32),
                        _ptr(b32), _ptr(gx), _ptr(gw), _ptr(gb), _ptr(ws), nbytes, _stream_ptr(device)))
                return gx, gw.to(wdt), gb.to(bdt), None, None, None, None, None, None
            with torch.cuda.device(device):
                _capi.check(lib.dfm_group_norm_bwd_channels_last(
                    n, c, spatial, groups, _DTYPES[x.dtype], int(relu), _ptr(gy), _ptr(x), _ptr(y), _ptr(mean),
                    _ptr(rstd), _ptr(w32), _ptr(gx), _ptr(gres) if gres is not None else None, _ptr(gw),
                    _ptr(gb), _ptr(ws), nbytes, _stream_ptr(device)))
            if want_res and not relu:
                gres = gy
            return gx, gw.to(wdt), gb.to(bdt), None, None, None, None, gres, None
        # y = relu?(gn(x) + residual): the residual's gradient is the incoming one behind the ReLU mask
        gres = None
        if want_res:
            gres = gy * (y > 0).to(gy.dtype) if relu else gy
        gy = gy.contiguous().to(x.dtype)
        gx = torch.empty_like(x)
        with torch.cuda.device(device):
            _capi.check(
                lib.dfm_group_norm_bwd(n, c, spatial, groups, _DTYPES[x.dtype], int(relu), _ptr(gy),
                                       _ptr(x), _ptr(y), _ptr(mean), _ptr(rstd), _ptr(w32), _ptr(gx),
                                       _ptr(gw), _ptr(gb), _ptr(ws), nbytes, _stream_ptr(device)))
        return gx, gw.to(wdt), gb.to(bdt), None, None, None, None, gres, None


def group_norm(x, num_groups, weight, bias, eps=1e-5, relu=False, partials=None, residual=None):
    """torch.nn.functional.group_norm(+relu) on the GPU through the fused kernels.
    ``partials`` (N, groups, splits, 3): count / mean / M2 moment partials of ``x`` from its
    producer (``MfmaConv3d.forward_with_stats``); the statistics pass over ``x`` is skipped.
    ``residual``: added after the affine map, before the ReLU (fused into the channels-last pass)."""
    if not x.is_cuda or x.dtype not in _DTYPES:
        raise RuntimeError('fused group_norm needs a float32/bfloat16 GPU tensor '
                           '(depth-from-motion_amd has no CPU path)')
    return _GroupNormFn.apply(x, weight, bias, int(num_groups), float(eps), bool(relu), partials, residual)


_cpu_reference = False


def allow_cpu_reference(flag=True):
    """TEST-ONLY switch: lets HipGroupNorm run torch's nn.GroupNorm on CPU tensors so that the
    module wiring / the gloo data-parallel glue can be exercised on a box without a GPU
    (tests/test_distributed_cpu.py, tests/test_modules.py).  Off by default: the product has no
    CPU path and a CPU tensor raises.  Returns the previous setting."""
    global _cpu_reference
    prev, _cpu_reference = _cpu_reference, bool(flag)
    return prev


class HipGroupNorm(nn.GroupNorm):
    """nn.GroupNorm whose forward/backward run in the fused HIP kernels (float32 / bfloat16,
    affine).  Anything else raises -- there is no silent eager fallback: a CPU tensor, or a GPU
    tensor the kernels do not cover (fp16, affine=False), is an error.  (Tests that exercise
    module wiring on a CPU-only box opt in with ``allow_cpu_reference(True)``.)"""

    def forward(self, x, relu=False, partials=None, residual=None):
        if x.is_cuda:
            if x.dtype not in _DTYPES or not self.affine:
                raise RuntimeError(
                    f'HipGroupNorm: unsupported GPU input (dtype {x.dtype}, affine={self.affine}); '
                    'the fused kernels cover float32 / bfloat16 with affine parameters')
            return group_norm(x, self.num_groups, self.weight, self.bias, self.eps, relu, partials, residual)
        if not _cpu_reference:
            raise RuntimeError('HipGroupNorm got a CPU tensor: depth-from-motion_amd has no CPU path '
                               '(tests opt in with group_norm.allow_cpu_reference(True))')
        y = super().forward(x)
        if residual is not None:
            y = y + residual
        return torch.relu_(y) if relu else y


def batch_norm_train_channels_last(bn, x, relu=False, residual=None):
    """Training-mode BatchNorm (nn.BatchNorm2d / 3d / SyncBatchNorm in a single-process job) of a channels-last GPU
    tensor through the fused GroupNorm kernels, or None when it does not apply.  The batch statistics of a
    channels-last (N, C, *spatial) tensor are the per-channel GroupNorm statistics of the same memory viewed as ONE
    sample (1, C, N * s0, ...): normalisation (+ residual) (+ ReLU) is one statistics pass and one apply pass, the
    backward the channels-last GroupNorm backward; running statistics are updated as torch does (unbiased variance,
    momentum / cumulative average).  y = relu?(bn(x) + residual)."""
    vec = 16 // x.element_size() if x.dtype in _DTYPES else 0
    c = x.shape[1]
    fmt = torch.channels_last_3d if x.dim() == 5 else torch.channels_last
    if not (bn.training and x.is_cuda and vec and x.dim() in (4, 5) and bn.affine and bn.track_running_stats and
            c % vec == 0 and c <= 256 and ((c // vec) & (c // vec - 1)) == 0 and
            not x.is_contiguous() and x.is_contiguous(memory_format=fmt)):
        return None
    if isinstance(bn, nn.SyncBatchNorm) and torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size() > 1:
        return None   # statistics across ranks: torch's implementation

    def one_sample(t):   # (N, C, s0, ...) channels-last -> (1, C, N * s0, ...) channels-last, the same memory
        n = t.shape[0]
        perm = (0,) + tuple(range(2, t.dim())) + (1,)
        back = (0, t.dim() - 1) + tuple(range(1, t.dim() - 1))
        v = t.permute(*perm)
        return v.reshape(1, n * v.shape[1], *v.shape[2:]).permute(*back)

    res = None
    if residual is not None:
        res = one_sample(residual if residual.is_contiguous(memory_format=fmt) else residual.contiguous(memory_format=fmt))
    stats = []
    y = _GroupNormFn.apply(one_sample(x), bn.weight, bn.bias, c, float(bn.eps), bool(relu), None, res, stats)
    mean, rstd = stats[0]
    with torch.no_grad():
        m = x.numel() // c
        var = (1.0 / (rstd * rstd) - bn.eps).clamp_min_(0.0) * (m / max(m - 1, 1))  # unbiased
        bn.num_batches_tracked += 1
        f = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
        bn.running_mean.mul_(1 - f).add_(mean.to(bn.running_mean.dtype), alpha=f)
        bn.running_var.mul_(1 - f).add_(var.to(bn.running_var.dtype), alpha=f)
    perm = (0,) + tuple(range(2, x.dim())) + (1,)
    back = (0, x.dim() - 1) + tuple(range(1, x.dim() - 1))
    return y.permute(*perm).reshape(x.shape[0], *x.shape[2:], c).permute(*back)


class HipBatchNorm3d(nn.BatchNorm3d):
    """nn.BatchNorm3d (same parameters, buffers and ``state_dict`` keys) whose TRAINING forward /
    backward on channels-last GPU tensors run in the fused HIP kernels: the batch statistics of a
    channels-last (N, C, D, H, W) tensor are the per-channel GroupNorm statistics of the same memory
    viewed as ONE sample (1, C, N*D, H, W), so normalisation (+ residual) (+ ReLU) is one statistics
    pass and one apply pass, and the backward the channels-last GroupNorm backward (torch's
    BatchNorm backward on this layout: 2.3 ms per layer of the voxel neck, 42 % of its training step).
    Eval mode / other inputs: torch (the necks fold eval-mode BatchNorm into the conv epilogue anyway).
    ``forward(x, relu=False, residual=None)``: y = relu?(bn(x) + residual)."""

    def _fusable(self, x):
        vec = 16 // x.element_size() if x.dtype in _DTYPES else 0
        c = x.shape[1]
        return (self.training and x.is_cuda and vec and x.dim() == 5 and self.affine and
                self.track_running_stats and c % vec == 0 and c <= 256 and ((c // vec) & (c // vec - 1)) == 0 and
                not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last_3d))

    @staticmethod
    def _as_one_sample(t):
        N, C, D, H, W = t.shape
        return t.permute(0, 2, 3, 4, 1).reshape(1, N * D, H, W, C).permute(0, 4, 1, 2, 3)

    def forward(self, x, relu=False, residual=None):
        if not self._fusable(x):
            y = super().forward(x)
            if residual is not None:
                y = y + residual
            return torch.relu_(y) if relu else y
        return batch_norm_train_channels_last(self, x, relu, residual)
