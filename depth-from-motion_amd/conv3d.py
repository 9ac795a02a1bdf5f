"""Hand-written MFMA Conv3d 3x3x3 (stride 1, pad 1, 32 -> 32 channels, NDHWC bf16) of the 3-D
aggregation stacks: host-side wrapper of ``dfm_conv3d_k3_c32_fwd`` (csrc/conv3d.hip).

Reference call sites: ``ConvModule(..., conv_cfg=dict(type='Conv3d'))`` in
mmdet3d/models/backbones/dfm_backbone.py:50-128 and ``convbn_3d`` in
mmdet3d/models/utils/conv_modules.py:27-43 (every full-resolution convolution of config K is
32 -> 32, or 64 -> 32 = two 32-channel halves accumulated in fp32).

``MfmaConv3d`` is an ``nn.Conv3d`` (same parameters, same ``state_dict`` keys).  Its forward runs
the MFMA kernel when the input is a bfloat16 ``channels_last_3d`` GPU tensor (what a backbone
converted with ``.to(torch.bfloat16, memory_format=torch.channels_last_3d)`` produces, fed by the
channels-last cost volume); any other dtype / layout takes torch's convolution (MIOpen) exactly as
before -- that is the module's other documented path, not a fallback of a failed launch.
Backward: the input gradient runs in the same MFMA kernel (transposed, mirrored weight
fragments); the weight gradient is torch's convolution backward (MIOpen).
"""
import ctypes

import torch
from torch import nn

from . import _capi
from .plane_sweep import _ptr, _stream_ptr

_WDT = {torch.float32: _capi.DFM_F32, torch.bfloat16: _capi.DFM_BF16}


def pack_conv3d_weights(weight, cin_offset=0, transposed=False):
    """(32, C_in >= 32, 3, 3, 3) fp32/bf16 GPU weight -> MFMA A-operand fragments (+ zero page) for
    the 32 input channels starting at ``cin_offset``.  ``transposed``: fragments of the
    backward-data convolution (grad_in[:, cin_offset:cin_offset+32] = conv(grad_out, W'))."""
    assert weight.is_cuda and weight.dim() == 5 and weight.shape[0] == 32 and tuple(weight.shape[2:]) == (3, 3, 3)
    w = weight.detach().contiguous()
    if w.dtype not in _WDT:
        w = w.float()
    lib = _capi.lib()
    packed = torch.empty(lib.dfm_conv3d_k3_c32_weight_bytes(), dtype=torch.uint8, device=w.device)
    with torch.cuda.device(w.device):
        _capi.check(lib.dfm_conv3d_k3_c32_pack_weights(_ptr(w), _WDT[w.dtype], w.shape[1], cin_offset,
                                                       1 if transposed else 0, _ptr(packed),
                                                       _stream_ptr(w.device)))
    return packed


def _is_ndhwc(x):
    return x.dim() == 5 and x.is_contiguous(memory_format=torch.channels_last_3d)


def conv3d_k3_c32(x, packed, relu=False, acc_in=None, out_f32=False, depth_chunk=0, stats=False):
    """x: (N, 32, D, H, W) bf16 channels_last_3d.  Returns (N, 32, D, H, W) channels_last_3d bf16,
    or the fp32 partial (N, D, H, W, 32) when ``out_f32``; ``acc_in``: fp32 partial to start from.
    ``stats``: also return the per-channel moment partials (N, 32, splits, 3) of the stored values
    (the GroupNorm statistics of the layer that follows, see ``group_norm_from_partials``)."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.shape[1] == 32 and _is_ndhwc(x)
    N, _, D, H, W = x.shape
    lib = _capi.lib()
    dev = x.device
    if out_f32:
        out = torch.empty((N, D, H, W, 32), dtype=torch.float32, device=dev)
    else:
        out = torch.empty((N, D, H, W, 32), dtype=torch.bfloat16, device=dev)
    if acc_in is not None:
        assert acc_in.dtype == torch.float32 and acc_in.shape == (N, D, H, W, 32) and acc_in.is_contiguous()
    part = None
    if stats:
        assert not out_f32 and not relu, 'statistics are taken of the plain bf16 output'
        splits = lib.dfm_conv3d_k3_c32_stats_splits(N, D, H, W, depth_chunk)
        part = torch.empty((N, 32, splits, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _capi.check(lib.dfm_conv3d_k3_c32_fwd(N, D, H, W, _ptr(x), _ptr(packed),
                                              _ptr(acc_in) if acc_in is not None else None, _ptr(out),
                                              1 if out_f32 else 0, 1 if relu else 0, depth_chunk,
                                              _ptr(part) if part is not None else None,
                                              _stream_ptr(dev)))
    if out_f32:
        return out
    out = out.permute(0, 4, 1, 2, 3)
    return (out, part) if stats else out


class _MfmaConvFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, weight, packs, want_stats=False):
        ctx.save_for_backward(x, weight)
        # C_in = 32 * k: halves accumulated through the fp32 partial
        part = None
        for i, pk in enumerate(packs):
            xi = x if len(packs) == 1 else x[:, 32 * i:32 * (i + 1)]
            if not _is_ndhwc(xi):
                xi = xi.contiguous(memory_format=torch.channels_last_3d)
            last = i == len(packs) - 1
            part = conv3d_k3_c32(xi, pk, acc_in=part, out_f32=not last, stats=want_stats and last)
        if want_stats:
            ctx.mark_non_differentiable(part[1])
            return part
        return part

    @staticmethod
    def backward(ctx, gy, *_unused):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous(memory_format=torch.channels_last_3d)
        gx = None
        if ctx.needs_input_grad[0]:
            # backward-data = the same MFMA kernel on the gradient with transposed, mirrored weights
            halves = [conv3d_k3_c32(gy, pack_conv3d_weights(weight, 32 * i, transposed=True))
                      for i in range(weight.shape[1] // 32)]
            gx = halves[0] if len(halves) == 1 else \
                torch.cat(halves, dim=1).contiguous(memory_format=torch.channels_last_3d)
        gw = None
        if ctx.needs_input_grad[1]:  # backward-weight: MIOpen through torch
            _, gw, _ = torch.ops.aten.convolution_backward(
                gy, x, weight.to(x.dtype), None, [1, 1, 1], [1, 1, 1], [1, 1, 1], False, [0, 0, 0], 1,
                [False, True, False])
            gw = gw.to(weight.dtype)
        return gx, gw, None, None


class MfmaConv3d(nn.Conv3d):
    """nn.Conv3d(C_in in {32, 64, ...}, 32, 3, stride=1, padding=1, bias=False) whose bf16 / NDHWC
    forward is the hand-written MFMA kernel.  Packed weight fragments are cached and rebuilt when
    the parameter changes (version counter)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._packs, self._pack_key = None, None

    def eligible(self, x):
        return (x.is_cuda and x.dtype == torch.bfloat16 and _is_ndhwc(x) and self.out_channels == 32 and
                self.in_channels % 32 == 0 and self.kernel_size == (3, 3, 3) and self.stride == (1, 1, 1) and
                self.padding == (1, 1, 1) and self.dilation == (1, 1, 1) and self.groups == 1 and
                self.bias is None)

    def _packed(self):
        key = (self.weight._version, self.weight.data_ptr(), self.weight.device)
        if self._pack_key != key:
            self._packs = [pack_conv3d_weights(self.weight, 32 * i) for i in range(self.in_channels // 32)]
            self._pack_key = key
        return self._packs

    def forward(self, x):
        if self.eligible(x):
            return _MfmaConvFn.apply(x, self.weight, self._packed())
        return super().forward(x)

    def forward_with_stats(self, x):
        """(y, moment partials): the convolution plus the per-channel GroupNorm statistics of y
        from the kernel's epilogue (``x`` must be eligible)."""
        return _MfmaConvFn.apply(x, self.weight, self._packed(), True)
