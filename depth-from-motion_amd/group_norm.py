"""Fused GroupNorm(+ReLU) for the cost-volume aggregation stacks: one HIP
reduction pass + one apply pass (``dfm_group_norm_fwd/bwd``) instead of torch's
GroupNorm kernel followed by a separate ReLU.

``HipGroupNorm`` subclasses ``nn.GroupNorm`` (same parameters, same
``state_dict`` keys), so it drops into ``ConvModule`` / ``hourglass`` without
touching checkpoints.
"""
import ctypes

import torch
from torch import nn

from . import _capi
from .plane_sweep import _DTYPES, _Workspace, _ptr, _stream_ptr


class _GroupNormFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, relu, partials=None):
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
        w32 = weight.detach().float().contiguous()
        b32 = bias.detach().float().contiguous()
        nbytes = lib.dfm_group_norm_workspace_bytes(n, c, spatial, groups)
        ws = _Workspace.get(device, nbytes)
        fn = lib.dfm_group_norm_fwd_channels_last if cl else lib.dfm_group_norm_fwd
        with torch.cuda.device(device):
            if partials is not None:
                # statistics came from the producer (MFMA conv epilogue): normalisation pass only
                assert cl and partials.shape[:2] == (n, groups) and partials.is_contiguous()
                _capi.check(lib.dfm_group_norm_apply_channels_last(
                    n, c, spatial, groups, eps, _DTYPES[x.dtype], int(relu), _ptr(x), _ptr(w32), _ptr(b32),
                    _ptr(y), _ptr(mean), _ptr(rstd), _ptr(partials), partials.shape[2], _ptr(ws), nbytes,
                    _stream_ptr(device)))
            else:
                _capi.check(fn(n, c, spatial, groups, eps, _DTYPES[x.dtype], int(relu), _ptr(x), _ptr(w32),
                               _ptr(b32), _ptr(y), _ptr(mean), _ptr(rstd), _ptr(ws), nbytes,
                               _stream_ptr(device)))
        ctx.save_for_backward(x, y if relu else x, mean, rstd, w32)
        ctx.cfg = (groups, bool(relu), weight.dtype, bias.dtype, cl)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, mean, rstd, w32 = ctx.saved_tensors
        groups, relu, wdt, bdt, cl = ctx.cfg
        if cl:  # the backward kernels are NC(D)HW: convert (training through channels-last
            x, y = x.contiguous(), y.contiguous()  # stacks pays two extra copies here)
        lib = _capi.lib()
        device = x.device
        n, c = x.shape[:2]
        spatial = x.numel() // (n * c)
        gy = gy.contiguous().to(x.dtype)
        gx = torch.empty_like(x)
        gw = torch.zeros(c, dtype=torch.float32, device=device)
        gb = torch.zeros(c, dtype=torch.float32, device=device)
        nbytes = lib.dfm_group_norm_workspace_bytes(n, c, spatial, groups)
        ws = _Workspace.get(device, nbytes)
        with torch.cuda.device(device):
            _capi.check(
                lib.dfm_group_norm_bwd(n, c, spatial, groups, _DTYPES[x.dtype], int(relu), _ptr(gy),
                                       _ptr(x), _ptr(y), _ptr(mean), _ptr(rstd), _ptr(w32), _ptr(gx),
                                       _ptr(gw), _ptr(gb), _ptr(ws), nbytes, _stream_ptr(device)))
        return gx, gw.to(wdt), gb.to(bdt), None, None, None, None


def group_norm(x, num_groups, weight, bias, eps=1e-5, relu=False, partials=None):
    """torch.nn.functional.group_norm(+relu) on the GPU through the fused kernels.
    ``partials`` (N, groups, splits, 3): count / mean / M2 moment partials of ``x`` from its
    producer (``MfmaConv3d.forward_with_stats``); the statistics pass over ``x`` is skipped."""
    if not x.is_cuda or x.dtype not in _DTYPES:
        raise RuntimeError('fused group_norm needs a float32/bfloat16 GPU tensor '
                           '(depth-from-motion_amd has no CPU path)')
    return _GroupNormFn.apply(x, weight, bias, int(num_groups), float(eps), bool(relu), partials)


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

    def forward(self, x, relu=False, partials=None):
        if x.is_cuda:
            if x.dtype not in _DTYPES or not self.affine:
                raise RuntimeError(
                    f'HipGroupNorm: unsupported GPU input (dtype {x.dtype}, affine={self.affine}); '
                    'the fused kernels cover float32 / bfloat16 with affine parameters')
            return group_norm(x, self.num_groups, self.weight, self.bias, self.eps, relu, partials)
        if not _cpu_reference:
            raise RuntimeError('HipGroupNorm got a CPU tensor: depth-from-motion_amd has no CPU path '
                               '(tests opt in with group_norm.allow_cpu_reference(True))')
        y = super().forward(x)
        return torch.relu_(y) if relu else y
