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
import os
import warnings

import torch
import torch.nn.functional as F
from torch import nn

from . import _capi
from .plane_sweep import _Workspace, _ptr, _stream_ptr

_WDT = {torch.float32: _capi.DFM_F32, torch.bfloat16: _capi.DFM_BF16}

# ---------------------------------------------------------------------------------------------
# What an ``Mfma*`` module does with a GPU input its kernel does not take (fp32, NCDHW, a shape no
# tiling fits ...).  The module is an nn.Conv* and CAN run torch's convolution (MIOpen) -- that is how
# the reference's fp32 pipeline keeps working after ``patch_reference()`` -- but it must not do so
# silently: 'warn' (default) says so once per (module class, reason), 'raise' (strict mode) makes it
# an error, 'silent' restores round 2's behaviour.  ``integration.enable_fast_path(model)`` converts a
# model so that every convolution of the path IS eligible.  CPU tensors never warn: there is no
# kernel to miss, only the module-wiring tests run there.
# ---------------------------------------------------------------------------------------------
_POLICY = {'mode': 'warn'}
_WARNED = set()


class MfmaPathError(RuntimeError):
    """strict mode: an Mfma* module was handed an input its MFMA kernel does not take"""


def set_fallback_policy(mode):
    """'warn' | 'raise' | 'silent'; returns the previous mode"""
    if mode not in ('warn', 'raise', 'silent'):
        raise ValueError(mode)
    prev, _POLICY['mode'] = _POLICY['mode'], mode
    return prev


def fallback_policy():
    return _POLICY['mode']


def module_fallback_policy(module):
    """the policy that governs ``module``: its own ``fallback_policy`` attribute (what
    ``integration.enable_fast_path`` sets on the Mfma* modules of the model it converts) or, without one,
    the process-wide mode"""
    return getattr(module, 'fallback_policy', None) or _POLICY['mode']


def _torch_path(module, x, why):
    """called by every Mfma* module right before it runs torch's convolution instead of its kernel"""
    mode = module_fallback_policy(module)
    if not x.is_cuda or mode == 'silent':
        return
    msg = (f'{type(module).__name__}({module.in_channels}->{module.out_channels}): {why}; running torch\'s '
           'convolution (MIOpen) instead of the MFMA kernel.  depth-from-motion_amd.enable_fast_path(model) '
           'converts the path to bf16 / channels-last; set_fallback_policy("raise") makes this an error.')
    if mode == 'raise':
        raise MfmaPathError(msg)
    key = (type(module).__name__, why)
    if key not in _WARNED:
        _WARNED.add(key)
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def _why_not_bf16_cl(x, dims):
    if x.dtype != torch.bfloat16:
        return f'input dtype {str(x.dtype).replace("torch.", "")} (the kernel takes bfloat16)'
    if x.dim() != dims:
        return f'{x.dim()}-D input'
    return None


def pack_conv3d_weights(weight, cin_offset=0, transposed=False):
    """(32, C_in >= 32, 3, 3, 3) fp32/bf16 GPU weight -> MFMA A-operand fragments (+ zero page) for
    the 32 input channels starting at ``cin_offset``.  ``transposed``: fragments of the
    backward-data convolution (grad_in[:, cin_offset:cin_offset+32] = conv(grad_out, W'))."""
    note_derived_build()
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
    cstride = _ndhwc_channel_stride(x)
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.shape[1] == 32 and cstride, \
        'bf16 channels_last_3d with 32 channels (or a 32-channel slice of a wider NDHWC tensor)'
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
        _capi.check(lib.dfm_conv3d_k3_c32_fwd_strided(
            N, D, H, W, _ptr(x), cstride, _ptr(packed), _ptr(acc_in) if acc_in is not None else None,
            _ptr(out), 1 if out_f32 else 0, 1 if relu else 0, depth_chunk,
            _ptr(part) if part is not None else None, _stream_ptr(dev)))
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
            xi = x if len(packs) == 1 else x[:, 32 * i:32 * (i + 1)]  # read in place (pixel stride)
            if not _ndhwc_channel_stride(xi):
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
            k = weight.shape[1] // 32
            if k == 1:
                gx = conv3d_k3_c32(gy, pack_conv3d_weights(weight, 0, transposed=True))
            elif os.environ.get('DFM_C32_CAT') == '1':   # (A/B runs: the round-5 form)
                halves = [conv3d_k3_c32(gy, pack_conv3d_weights(weight, 32 * i, transposed=True)) for i in range(k)]
                gx = torch.cat(halves, dim=1).contiguous(memory_format=torch.channels_last_3d)
            else:
                # the k halves written straight into the (N, D, H, W, 32 k) gradient (dfm_conv3d_k3_c32_fwd_slices)
                N, _, D, H, W = gy.shape
                buf = torch.empty((N, D, H, W, 32 * k), dtype=torch.bfloat16, device=gy.device)
                lib = _capi.lib()
                with torch.cuda.device(gy.device):
                    for i in range(k):
                        pk = pack_conv3d_weights(weight, 32 * i, transposed=True)
                        _capi.check(lib.dfm_conv3d_k3_c32_fwd_slices(
                            N, D, H, W, _ptr(gy), 32, _ptr(pk), buf.data_ptr() + 64 * i, 32 * k, 0, 0,
                            _stream_ptr(gy.device)))
                gx = buf.permute(0, 4, 1, 2, 3)
        gw = None
        if ctx.needs_input_grad[1]:  # backward-weight: chunked implicit-im2col GEMM (see above)
            gw = conv3d_weight_grad(x, gy, 1, 1, out_dtype=weight.dtype)
        return gx, gw, None, None


# Per-module state DERIVED from parameters or from tensors the detector injects: packed weight fragments, folded
# norms, device copies of host tensors (some entries hold weak references, which do not pickle).  None of it is part
# of a module's identity: `torch.save(model)`, `pickle` and `copy.deepcopy` (EMA hooks, `mp.spawn` arguments) leave
# it behind and the copy rebuilds it on its first forward.  Names initialised in __init__ go back to their initial
# value, the rest is dropped.
_DERIVED_RESET = {'_packs': None, '_pack_key': None, '_sweep_conv_pack': (None, None)}
_DERIVED_DROP = frozenset((
    '_split_packs', '_split_key', '_pack2d', '_pack2d_key', '_gate_pack', '_dev_cache', '_coords_ref', '_coords_key',
    '_coords_dev', '_spp_params', '_spp_key', '_fold', '_fold_key'))

# Derived state is built lazily by the FIRST call that needs it, with kernels on that call's HIP stream.  A module
# that runs on two streams (DfMStereoPath: the same 2-D neck for the previous frame on a side stream and for the
# current frame on the main stream) would let the second stream read packed weights the first has not finished
# writing.  Every build site bumps this counter; a caller that forks streams compares it around the first call and
# makes the other stream wait when anything was built (integration.DfMStereoPath.forward).
_DERIVED_BUILDS = [0]


def note_derived_build():
    _DERIVED_BUILDS[0] += 1


def derived_builds():
    return _DERIVED_BUILDS[0]


class DerivedStateMixin:
    """first base of the path's modules: their pickled state carries parameters, buffers and configuration only"""

    def __getstate__(self):
        state = super().__getstate__()
        state = {k: v for k, v in state.items() if k not in _DERIVED_DROP}
        for k, v in _DERIVED_RESET.items():
            if k in state:
                state[k] = v
        return state


class MfmaConv3d(DerivedStateMixin, nn.Conv3d):
    """nn.Conv3d(C_in in {32, 64, ...}, 32, 3, stride=1, padding=1, bias=False) whose bf16 / NDHWC
    forward is the hand-written MFMA kernel.  Packed weight fragments are cached and rebuilt when
    the parameter changes (version counter)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._packs, self._pack_key = None, None

    def why_not(self, x):
        """None when the MFMA kernel takes ``x``, else the reason it does not"""
        if not x.is_cuda:
            return 'CPU tensor'
        if not (self.out_channels == 32 and self.in_channels % 32 == 0 and self.kernel_size == (3, 3, 3) and
                self.stride == (1, 1, 1) and self.padding == (1, 1, 1) and self.dilation == (1, 1, 1) and
                self.groups == 1 and self.bias is None):
            return 'convolution configuration outside the 32-channel kernel\'s coverage'
        why = _why_not_bf16_cl(x, 5)
        if why is None and not _ndhwc_channel_stride(x):
            why = 'input is not channels_last_3d (nor a channel slice of an NDHWC tensor)'
        return why

    def eligible(self, x):
        return self.why_not(x) is None

    def _packed(self):
        key = (self.weight._version, self.weight.data_ptr(), self.weight.device)
        if self._pack_key != key:
            self._packs = [pack_conv3d_weights(self.weight, 32 * i) for i in range(self.in_channels // 32)]
            self._pack_key = key
        return self._packs

    def forward(self, x):
        why = self.why_not(x)
        if why is None:
            return _MfmaConvFn.apply(x, self.weight, self._packed())
        if x.is_cuda and x.dtype == torch.float32 and 'coverage' not in why:
            why32 = _split_why_not(self, x, 'conv')
            if why32 is None:  # an fp32 model: the general kernel in split precision
                return _ConvGSplitFn.apply(x, self.weight, _split_packs(self, self.in_channels, 32, False), 'conv',
                                           self.stride, self.padding)
            why = f'{why}; split precision: {why32}'
        _torch_path(self, x, why)
        return super().forward(x)

    def forward_with_stats(self, x):
        """(y, moment partials): the convolution plus the per-channel GroupNorm statistics of y
        from the kernel's epilogue (``x`` must be eligible)."""
        return _MfmaConvFn.apply(x, self.weight, self._packed(), True)


# ---------------------------------------------------------------------------------------------
# backward-weight: MIOpen's untuned bf16 NDHWC kernels for these shapes are 84 ms .. 1.26 s PER
# CONVOLUTION (naive fallbacks; profiles/archive/r02_c31_train_step_kernel_stats.txt: 2.1 s per training
# step of DfMBackbone).  Until the MFMA weight-gradient kernel exists, the gradient is a chunked
# implicit-im2col GEMM: a strided view of the padded input gives the (rows, 27 C_in) patch matrix of
# a depth chunk, one library GEMM (hipBLASLt, fp32 accumulation over the chunk) contracts it with
# the output gradient, chunks are summed in fp32.
# ---------------------------------------------------------------------------------------------
_OUT_DTYPE_OK = {}  # (op name, device type) -> does op(..., out_dtype=torch.float32) work on this build / backend?
_SPLIT_LONG_AXIS = os.environ.get('DFM_PLAIN_WGRAD_1X1') != '1'


def _product_f32(op, a, b):
    """``op(a, b)`` (torch.mm / torch.bmm) with an fp32 result: ``out_dtype=torch.float32`` where the backend of the
    operands has it (the CUDA / HIP backend of this torch does, its CPU backend does not -- asked per device type, not
    per process: a CPU call after a GPU call must not inherit the GPU's answer), else the product converted"""
    if a.dtype == torch.float32:
        return op(a, b)
    key = (op.__name__, a.device.type)
    ok = _OUT_DTYPE_OK.get(key)
    if ok is None:
        try:
            r = op(a, b, out_dtype=torch.float32)
            _OUT_DTYPE_OK[key] = True
            return r
        except (RuntimeError, TypeError, NotImplementedError):
            ok = _OUT_DTYPE_OK[key] = False
    if ok:
        return op(a, b, out_dtype=torch.float32)
    return op(a, b).float()


def _mm_f32(a, b):
    return _product_f32(torch.mm, a, b)


def _bmm_f32(a, b):
    return _product_f32(torch.bmm, a, b)


def long_axis_gram(a, b, rows_per_batch=2048):
    """``a``: (P, M), ``b``: (P, N), any strides, P long and M, N small -> ``a.t() @ b`` as (M, N) float32.
    The weight gradient of a 1x1 convolution (P = pixels) is such a product; as ONE library GEMM it is a
    handful of output tiles each walking the whole P axis (M = N = 32, P = 409 600 -- ``lastconv`` of
    SPPUNetNeck at config K: hipBLASLt MT16x32x512, 2 workgroups, 0.36 ms per call, two calls per training
    step).  Here P is cut into batches of ~``rows_per_batch`` rows -- a batched GEMM of S x (M x N) tiles, fp32
    partial products where the build has ``out_dtype`` -- and the S partials are summed in fp32."""
    P = a.shape[0]
    S = max(1, min(512, P // max(1, rows_per_batch))) if _SPLIT_LONG_AXIS else 1  # (DFM_PLAIN_WGRAD_1X1=1: A/B runs)
    while S > 1 and P % S:
        S -= 1
    if S < 4:
        return _mm_f32(a.t(), b) if a.dtype != torch.float32 else torch.mm(a.t(), b)
    ab = a.reshape(S, P // S, a.shape[1]).transpose(1, 2)   # (S, M, P/S)
    bb = b.reshape(S, P // S, b.shape[1])                   # (S, P/S, N)
    return _bmm_f32(ab, bb).sum(0)


class _PixelLinearFn(torch.autograd.Function):
    """``F.linear`` over the pixel rows of an NHWC tensor (a 1x1 convolution), with the weight gradient as a
    batched product over slices of the pixel axis (``long_axis_gram``)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.matmul(gy, weight)
        g2 = gy.reshape(-1, gy.shape[-1])
        if ctx.needs_input_grad[1]:
            gw = long_axis_gram(g2, x.reshape(-1, x.shape[-1])).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum(0, dtype=torch.float32).to(weight.dtype)
        return gx, gw, gb


def _ndhwc_strides(t):
    """(n, d, h, w) element strides of a (N, C, D, H, W) tensor whose channels are contiguous, or None"""
    if t.dim() != 5 or t.stride(1) != 1 or t.storage_offset() % 8:
        return None
    st = (t.stride(0), t.stride(2), t.stride(3), t.stride(4))
    # the stride of a singleton dimension is arbitrary: any multiple of 8 will do
    st = tuple(s if t.shape[i] > 1 else 8 * max(1, s // 8) for s, i in zip(st, (0, 2, 3, 4)))
    return st if all(s > 0 and s % 8 == 0 for s in st) else None


def conv3d_weight_grad(x_in, g_out, stride, padding, out_dtype=torch.float32):
    """Weight gradient of a 3x3x3 convolution,
        out[a][b][kd][kh][kw] = sum_o g_out[:, a, o] * x_in[:, b, o * stride - padding + k],
    for bf16 channels-last x_in (N, B, D, H, W) and g_out (N, A, Do, Ho, Wo); (A, B, 3, 3, 3) in ``out_dtype``
    (fp32, or bf16: the kernel's reduction pass rounds its fp32 sums once -- a bf16 parameter's gradient without a
    conversion launch).  The hand-written MFMA kernel (csrc/conv3d_wgrad.hip) when the channel counts are multiples
    of 32, else the chunked implicit-im2col GEMM below."""
    stride, padding = _triple(stride), _triple(padding)
    A, B = g_out.shape[1], x_in.shape[1]
    gs, xs = _ndhwc_strides(g_out), _ndhwc_strides(x_in)
    if (x_in.is_cuda and x_in.dtype == torch.bfloat16 and g_out.dtype == torch.bfloat16 and A % 32 == 0 and
            B % 32 == 0 and gs is not None and xs is not None and all(s in (1, 2) for s in stride) and
            all(0 <= p <= 2 for p in padding)):
        d = _capi.Conv3dWgradDesc()
        d.n, d.a, d.b = x_in.shape[0], A, B
        for i in range(3):
            d.g_size[i], d.x_size[i] = g_out.shape[2 + i], x_in.shape[2 + i]
            d.stride[i], d.padding[i] = stride[i], padding[i]
        for i in range(4):
            d.g_stride[i], d.x_stride[i] = gs[i], xs[i]
        lib = _capi.lib()
        nbytes = lib.dfm_conv3d_wgrad_workspace_bytes(ctypes.byref(d))
        if nbytes:
            direct = out_dtype in (torch.float32, torch.bfloat16)
            out = torch.empty((A, B, 3, 3, 3), dtype=out_dtype if direct else torch.float32, device=x_in.device)
            ws = _Workspace.get(x_in.device, nbytes)
            with torch.cuda.device(x_in.device):
                rc = lib.dfm_conv3d_wgrad_to(ctypes.byref(d), _ptr(g_out), _ptr(x_in), _ptr(out),
                                             _capi.DFM_BF16 if out.dtype == torch.bfloat16 else _capi.DFM_F32,
                                             _ptr(ws), nbytes, _stream_ptr(x_in.device))
            if rc == 0:
                return out if direct else out.to(out_dtype)
            if rc != _capi.DFM_ERR_UNSUPPORTED:  # a tile that does not fit the LDS falls through to the GEMM
                _capi.check(rc)
    if x_in.is_cuda and _POLICY['mode'] == 'raise':
        raise MfmaPathError(f'weight gradient of a {B}->{A} convolution outside the MFMA kernel\'s coverage '
                            '(channels not multiples of 32, layout, or a tile that does not fit the LDS)')
    return _weight_grad_gemm(x_in, g_out, stride, padding).to(out_dtype)


def _weight_grad_gemm(x_in, g_out, stride, padding, chunk_bytes=256 << 20):
    """out[a][b][kd][kh][kw] = sum_o g_out[:, a, o] * x_in[:, b, o * stride - padding + k]
    for NDHWC bf16 tensors x_in (N, B, D, H, W) and g_out (N, A, Do, Ho, Wo); fp32 result (A, B, 3, 3, 3).
    nn.Conv3d: x_in = input, g_out = grad_output -> grad_weight (C_out, C_in, 3, 3, 3);
    nn.ConvTranspose3d (k 3, s 2, p 1, op 1): x_in = grad_output, g_out = input, stride 2, padding 1
    -> grad_weight (C_in, C_out, 3, 3, 3)."""
    stride, padding = _triple(stride), _triple(padding)
    N, B = x_in.shape[:2]
    A = g_out.shape[1]
    Do, Ho, Wo = g_out.shape[2:]
    xl = x_in.permute(0, 2, 3, 4, 1)   # (N, D, H, W, B): a view of the channels-last tensor
    gl = g_out.permute(0, 2, 3, 4, 1)
    if not xl.is_contiguous():
        xl = xl.contiguous()
    if not gl.is_contiguous():
        gl = gl.contiguous()
    # pad so that every tap of every output position is in bounds (high side: what the strides leave)
    need = [(o - 1) * s - p + 3 for o, s, p in zip((Do, Ho, Wo), stride, padding)]
    hi = [max(0, n - d) for n, d in zip(need, xl.shape[1:4])]
    xp = torch.nn.functional.pad(xl, (0, 0, padding[2], hi[2], padding[1], hi[1], padding[0], hi[0]))
    sN, sD, sH, sW, _ = xp.stride()
    cols = xp.as_strided((N, Do, Ho, Wo, 3, 3, 3, B),
                         (sN, sD * stride[0], sH * stride[1], sW * stride[2], sD, sH, sW, 1))
    rows_per_plane = Ho * Wo
    planes = max(1, int(chunk_bytes // (rows_per_plane * 27 * B * 2)))
    acc = torch.zeros((A, 27 * B), dtype=torch.float32, device=x_in.device)
    for n in range(N):
        for d0 in range(0, Do, planes):
            d1 = min(Do, d0 + planes)
            c = cols[n, d0:d1].reshape(-1, 27 * B)            # the only materialised patch matrix
            g = gl[n, d0:d1].reshape(-1, A)
            acc += _mm_f32(g.t(), c)
    return acc.view(A, 3, 3, 3, B).permute(0, 4, 1, 2, 3).contiguous()


class _ChannelSliceFn(torch.autograd.Function):
    """x[:, lo:hi] of a channels-last tensor whose gradient comes back channels-last too (autograd's
    own slice backward allocates a contiguous NCDHW zero tensor: the accumulation with the
    channels-last gradients of x's other consumers then runs as a strided add, 1.1 ms at config K)"""

    @staticmethod
    def forward(ctx, x, lo, hi):
        ctx.cfg = (x.shape, lo, hi)
        return x[:, lo:hi]

    @staticmethod
    def backward(ctx, gy):
        shape, lo, hi = ctx.cfg
        N, C, D, H, W = shape
        g = torch.zeros((N, D, H, W, C), dtype=gy.dtype, device=gy.device).permute(0, 4, 1, 2, 3)
        g[:, lo:hi] = gy
        return g, None, None


class _ChannelSplitFn(torch.autograd.Function):
    """(x, x[:, lo:hi]) for a tensor with two consumers -- the whole tensor and a channel slice of it (DfMBackbone: the
    cost volume feeds dres0 whole and dres0_mono by its first C channels, dfm_backbone.py:175,189; DfMNeck likewise,
    dfm_neck.py:78-84).  As two separate uses the slice's gradient came back as a zero-filled tensor of x's size with
    the slice copied in (_ChannelSliceFn) and the engine added the two full-size gradients: a fill, a copy and a
    full-size addition (0.2 ms per training step at config K).  Here the backward receives both gradients and adds
    the slice's INTO the whole tensor's, in place, over the slice's channels only.  (The whole tensor's gradient is
    the fresh result of its single consumer's backward; nothing else holds it.)"""

    @staticmethod
    def forward(ctx, x, lo, hi):
        ctx.cfg = (x.shape, lo, hi)
        return x.view_as(x), x[:, lo:hi]

    @staticmethod
    def backward(ctx, g_full, g_part):
        shape, lo, hi = ctx.cfg
        if g_full is None:
            if g_part is None:
                return None, None, None
            N, C, D, H, W = shape   # only the slice was used: its gradient in a zero tensor of x's layout
            g_full = torch.zeros((N, D, H, W, C), dtype=g_part.dtype, device=g_part.device).permute(0, 4, 1, 2, 3)
            g_full[:, lo:hi] = g_part
            return g_full, None, None
        if g_part is not None:
            g_full[:, lo:hi] += g_part.to(g_full.dtype)
        return g_full, None, None


def channel_split(x, lo, hi):
    """``(x, x[:, lo:hi])`` for the two consumers of a channels-last 5-D GPU tensor under autograd: one backward node
    that adds the slice's gradient into the whole tensor's in place (see _ChannelSplitFn); plain views otherwise"""
    if x.is_cuda and x.dim() == 5 and x.requires_grad and torch.is_grad_enabled() and \
            x.is_contiguous(memory_format=torch.channels_last_3d) and not x.is_contiguous():
        return _ChannelSplitFn.apply(x, lo, hi)
    return x, x[:, lo:hi]


def channel_slice(x, lo, hi):
    """x[:, lo:hi]; channels-last 5-D GPU tensors get a channels-last gradient (see _ChannelSliceFn)"""
    if x.is_cuda and x.dim() == 5 and x.requires_grad and torch.is_grad_enabled() and \
            x.is_contiguous(memory_format=torch.channels_last_3d) and not x.is_contiguous():
        return _ChannelSliceFn.apply(x, lo, hi)
    return x[:, lo:hi]


class _PackCache:
    """packed weight fragments of a module's parameter, rebuilt when the parameter changes"""

    def __init__(self):
        self._pack, self._key = None, None

    def __getstate__(self):  # a pickled / deep-copied module packs again on its first forward
        return {'_pack': None, '_key': None}

    def get(self, weight, make):
        key = (weight._version, weight.data_ptr(), weight.device)
        if self._key != key:
            self._pack, self._key = make(), key
        return self._pack


def conv3d_to1_norm(y, partials, gamma, beta, eps, weight, relu=True, depth_chunk=0):
    """GroupNorm(32 groups of one channel)(+ReLU) applied ON LOAD inside the Conv3d(32 -> 1, 3, 1, 1) that consumes it
    (csrc/conv3d_to1n.hip; dfm_backbone.py:120-127, inference).  ``y``: (N, 32, D, H, W) bf16 channels_last_3d, the RAW
    output of the 32 -> 32 convolution; ``partials``: its moment partials (N, 32, splits, 3) from
    ``MfmaConv3d.forward_with_stats``; ``gamma`` / ``beta``: the norm's fp32 parameters; ``weight``: (1, 32, 3, 3, 3).
    Returns (N, 1, D, H, W) bf16 -- the normalised volume is never written."""
    assert y.is_cuda and y.dtype == torch.bfloat16 and y.shape[1] == 32 and _is_ndhwc(y)
    N, _, D, H, W = y.shape
    assert partials.shape[:2] == (N, 32) and partials.is_contiguous() and partials.dtype == torch.float32
    w = weight.detach().contiguous()
    if w.dtype not in _WDT:
        w = w.float()
    lib = _capi.lib()
    dev = y.device
    coef = torch.empty((N, 32, 2), dtype=torch.float32, device=dev)
    out = torch.empty((N, 1, D, H, W), dtype=torch.bfloat16, device=dev)
    with torch.cuda.device(dev):
        st = _stream_ptr(dev)
        _capi.check(lib.dfm_group_norm_coefficients(N, 32, 32, float(eps), _ptr(partials), partials.shape[2],
                                                    _ptr(gamma), _ptr(beta), _ptr(coef), st))
        _capi.check(lib.dfm_conv3d_to1_norm_fwd(N, D, H, W, _ptr(y), _ptr(coef), _ptr(w), _WDT[w.dtype],
                                                1 if relu else 0, 0, _ptr(out), int(depth_chunk), st))
    return out


_IDENTITY_COEF = {}


def conv3d_to1(x, weight, depth_chunk=0):
    """Conv3d(32 -> 1, 3, 1, 1) of a bf16 NDHWC tensor through the lean 32 -> 1 kernel (csrc/conv3d_to1n.hip: two MFMAs
    per 32 pixels with the 27 taps as a matrix dimension) with the IDENTITY as its on-load map (a = 1, b = 0, no
    ReLU: x * 1 + 0 is x): what the training path and a head without a fusable norm run instead of the 32 -> 32 kernel
    on a weight zero-padded to 32 output channels (122-150 us against ~50 at config K)."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.shape[1] == 32 and _is_ndhwc(x)
    N, _, D, H, W = x.shape
    w = weight.detach().contiguous()
    if w.dtype not in _WDT:
        w = w.float()
    key = (str(x.device), N)
    coef = _IDENTITY_COEF.get(key)
    if coef is None:
        coef = torch.tensor([1.0, 0.0], dtype=torch.float32, device=x.device).repeat(N * 32).view(N, 32, 2).contiguous()
        if len(_IDENTITY_COEF) > 64:
            _IDENTITY_COEF.clear()
        _IDENTITY_COEF[key] = coef
        note_derived_build()
    out = torch.empty((N, 1, D, H, W), dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        _capi.check(_capi.lib().dfm_conv3d_to1_norm_fwd(N, D, H, W, _ptr(x), _ptr(coef), _ptr(w), _WDT[w.dtype], 0, 0,
                                                        _ptr(out), int(depth_chunk), _stream_ptr(x.device)))
    return out


class _MfmaConvTo1Fn(torch.autograd.Function):
    """Conv3d(32, 1, 3, 1, 1) through the MFMA kernel (weight rows 1..31 zero, channel 0 stored);
    backward is torch's convolution backward (MIOpen) -- the op is memory-bound either way."""

    @staticmethod
    def forward(ctx, x, weight, packed):
        ctx.save_for_backward(x, weight)
        if packed is None:   # the lean 32 -> 1 kernel (round 6)
            return conv3d_to1(x, weight)
        N, _, D, H, W = x.shape
        out = torch.empty((N, 1, D, H, W), dtype=torch.bfloat16, device=x.device)
        with torch.cuda.device(x.device):
            _capi.check(_capi.lib().dfm_conv3d_k3_c32_to1_fwd(N, D, H, W, _ptr(x), _ptr(packed), _ptr(out), 0, 0,
                                                              _stream_ptr(x.device)))
        return out

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gw = None
        gy = gy.contiguous()
        N, _, D, H, W = gy.shape
        lib = _capi.lib()
        direct = (gy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and _is_ndhwc(x) and
                  _ndhwc_channel_stride(x) == 32 and weight.dtype in _WDT and os.environ.get('DFM_TO1_PADDED_BWD') != '1')
        if direct:
            # round 6 (csrc/conv3d_to1_bwd.hip): both gradients as matrix products over the 27 taps -- no gradient padded
            # to 32 channels (a 118 MB fill + copy), no 32 -> 32 convolution / weight gradient for one useful row
            wc = weight.detach().contiguous()
            with torch.cuda.device(gy.device):
                if ctx.needs_input_grad[0]:
                    gxb = torch.empty((N, D, H, W, 32), dtype=torch.bfloat16, device=gy.device)
                    _capi.check(lib.dfm_conv3d_to1_bwd_data(N, D, H, W, _ptr(gy), _ptr(wc), _WDT[wc.dtype], _ptr(gxb),
                                                            _stream_ptr(gy.device)))
                    gx = gxb.permute(0, 4, 1, 2, 3)
                if ctx.needs_input_grad[1]:
                    gw = torch.empty((1, 32, 3, 3, 3), dtype=weight.dtype, device=gy.device)
                    nbytes = lib.dfm_conv3d_to1_wgrad_workspace_bytes()
                    ws = _Workspace.get(gy.device, nbytes)
                    _capi.check(lib.dfm_conv3d_to1_wgrad(N, D, H, W, _ptr(x), _ptr(gy), _ptr(gw), _WDT[gw.dtype],
                                                         _ptr(ws), nbytes, _stream_ptr(gy.device)))
            return gx, gw, None
        g32 = None
        if ctx.needs_input_grad[0]:
            # (the former route, other dtypes / layouts: the one gradient channel zero-padded to 32 through the
            # 32 -> 32 MFMA kernel with transposed / mirrored fragments of the zero-padded weight)
            g32 = torch.zeros((N, D, H, W, 32), dtype=gy.dtype, device=gy.device)
            g32[..., 0] = gy[:, 0]
            w32 = torch.zeros((32, 32, 3, 3, 3), dtype=torch.float32, device=weight.device)
            w32[0] = weight.detach().float()[0]
            gx = conv3d_k3_c32(g32.permute(0, 4, 1, 2, 3), pack_conv3d_weights(w32, 0, transposed=True))
        if ctx.needs_input_grad[1]:
            if g32 is None:
                g32 = torch.zeros((N, D, H, W, 32), dtype=gy.dtype, device=gy.device)
                g32[..., 0] = gy[:, 0]
            # rows 1..31 of the padded gradient are zero: row 0 is the (1, 32, 3, 3, 3) gradient
            gw = conv3d_weight_grad(x, g32.permute(0, 4, 1, 2, 3), 1, 1)[:1].to(weight.dtype)
        return gx, gw, None


class MfmaConv3dTo1(DerivedStateMixin, nn.Conv3d):
    """nn.Conv3d(32, 1, 3, 1, 1, bias=False): the prediction convolutions of DfMBackbone
    (dfm_backbone.py:120-127).  bf16 / NDHWC input: the 32 -> 32 MFMA kernel with a zero-padded
    weight, storing channel 0 only (MIOpen's untuned kernel for this shape takes 3.4 ms at config K,
    this one 0.12 ms)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._cache = _PackCache()

    def why_not(self, x):
        if not x.is_cuda:
            return 'CPU tensor'
        if not (self.in_channels == 32 and self.out_channels == 1 and self.kernel_size == (3, 3, 3) and
                self.stride == (1, 1, 1) and self.padding == (1, 1, 1) and self.dilation == (1, 1, 1) and
                self.groups == 1 and self.bias is None):
            return 'convolution configuration outside the 32 -> 1 kernel\'s coverage'
        why = _why_not_bf16_cl(x, 5)
        if why is None and not _is_ndhwc(x):
            why = 'input is not channels_last_3d'
        return why

    def eligible(self, x):
        return self.why_not(x) is None

    def _packed(self):
        def make():
            w = torch.zeros((32, 32, 3, 3, 3), dtype=torch.float32, device=self.weight.device)
            w[0] = self.weight.detach().float()[0]
            return pack_conv3d_weights(w)
        return self._cache.get(self.weight, make)

    def forward(self, x):
        why = self.why_not(x)
        if why is None:
            # DFM_TO1_C32_FWD=1: the 32 -> 32 kernel on the zero-padded weight (A/B runs)
            lean = os.environ.get('DFM_TO1_C32_FWD') != '1' and _ndhwc_channel_stride(x) == 32
            return _MfmaConvTo1Fn.apply(x, self.weight, None if lean else self._packed())
        if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and 'coverage' not in why and
                _FP32_MODE['mode'] != 'torch' and x.shape[1] == 32 and
                conv3d_g_plannable(x.shape[0], 32, 32, tuple(x.shape[2:]), 1, 1)):
            # an fp32 model: the general kernel in split precision on the weight zero-padded to 32 output
            # channels (MIOpen's naive kernel takes 1.4 s for this convolution at config K)
            w32 = torch.cat([self.weight, self.weight.new_zeros((31, 32, 3, 3, 3))], 0)
            return _ConvGSplitFn.apply(x, w32, None, 'conv', (1, 1, 1), (1, 1, 1))[:, :1].contiguous()
        _torch_path(self, x, why)
        return super().forward(x)


# ---------------------------------------------------------------------------------------------
# General MFMA Conv3d / ConvTranspose3d 3x3x3 (csrc/conv3d_g.hip): channels = 32 k, stride 1 | 2,
# padding 0..2, x2 transposed axes; epilogue scale/shift (+ residual) (+ ReLU).
#   hourglass conv1..conv6            mmdet3d/models/utils/conv_modules.py:73-149
#   ResModule / OutdoorImVoxelNeck    mmdet3d/models/necks/imvoxel_neck.py:26-55,85-117
#   DfMNeck                           mmdet3d/models/necks/dfm_neck.py:29-95
# ---------------------------------------------------------------------------------------------
def _triple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)


def pack_conv3d_g_weights(weight, cin, cout, swap=False, flip=0):
    """torch weight (dim0, dim1, 3, 3, 3) fp32/bf16 on the GPU -> MFMA A-operand fragments (+ zero
    page) for a convolution with ``cin`` input and ``cout`` output channels.  ``swap``: dim0 is the
    input-channel axis of that convolution; ``flip``: bit mask (4 = d, 2 = h, 1 = w) of mirrored
    kernel axes (see include/dfm_hip.h)."""
    note_derived_build()
    two_d = weight.dim() == 4   # a 2-D weight (dim0, dim1, 3, 3): packed into the centre depth slice, no embedding copy
    assert weight.is_cuda and weight.dim() in (4, 5) and tuple(weight.shape[2:]) == ((3, 3) if two_d else (3, 3, 3))
    assert tuple(weight.shape[:2]) == ((cin, cout) if swap else (cout, cin))
    w = weight.detach().contiguous()
    if w.dtype not in _WDT:
        w = w.float()
    lib = _capi.lib()
    packed = torch.empty(lib.dfm_conv3d_g_weight_bytes(cin, cout), dtype=torch.uint8, device=w.device)
    entry = lib.dfm_conv3d_g_pack_weights_2d if two_d else lib.dfm_conv3d_g_pack_weights
    with torch.cuda.device(w.device):
        _capi.check(entry(_ptr(w), _WDT[w.dtype], cin, cout, 1 if swap else 0, int(flip), _ptr(packed),
                          _stream_ptr(w.device)))
    return packed


def _ndhwc_channel_stride(x):
    """x: (N, C, D, H, W).  The pixel stride (elements) when x is a channels-last tensor or a channel
    slice of one (x[:, a:b] of an NDHWC tensor: dense pixels of C_total channels), else 0."""
    if x.dim() != 5 or x.stride(1) != 1 or x.storage_offset() % 8:
        return 0
    C = x.shape[1]
    ps, inner = 0, 1  # pixel stride, pixels spanned by the dimensions further in
    for size, st in zip((x.shape[4], x.shape[3], x.shape[2], x.shape[0]),
                        (x.stride(4), x.stride(3), x.stride(2), x.stride(0))):
        if size == 1:
            continue  # the stride of a singleton dimension is arbitrary
        if ps == 0:
            if st % inner:
                return 0
            ps = st // inner
        elif st != ps * inner:
            return 0
        inner *= size
    ps = ps or C
    return ps if ps >= C and ps % 8 == 0 else 0


def _conv_desc(n, cin, cout, in_size, out_size, stride, padding, transposed, relu, in_channel_stride=0,
               kernel1=(False, False, False)):
    d = _capi.Conv3dDesc()
    d.n, d.cin, d.cout, d.relu = n, cin, cout, 1 if relu else 0
    d.in_channel_stride = 0 if in_channel_stride == cin else in_channel_stride
    for i in range(3):
        d.in_size[i], d.out_size[i] = in_size[i], out_size[i]
        d.stride[i], d.padding[i], d.transposed[i] = stride[i], padding[i], 1 if transposed[i] else 0
        d.kernel1[i] = 1 if kernel1[i] else 0
    return d


def conv3d_g_out_size(in_size, stride, padding, transposed, kernel1=(False, False, False)):
    return tuple(2 * s if t else ((s - 1) // st + 1 if k1 else (s + 2 * p - 3) // st + 1)
                 for s, st, p, t, k1 in zip(in_size, stride, padding, transposed, kernel1))


def conv3d_g_plan(n, cin, cout, in_size, stride=1, padding=1, transposed=False):
    """The tiling the kernel picks: dict(pfw, cw, tile, block_px, lds, workgroups)."""
    stride, padding = _triple(stride), _triple(padding)
    transposed = _triple(transposed)
    out_size = conv3d_g_out_size(in_size, stride, padding, transposed)
    d = _conv_desc(n, cin, cout, in_size, out_size, stride, padding, transposed, False)
    plan = (ctypes.c_int64 * 8)()
    _capi.check(_capi.lib().dfm_conv3d_g_plan(ctypes.byref(d), plan))
    return dict(pfw=plan[0], cw=plan[1], tile=(plan[2], plan[3], plan[4]), block_px=plan[5], lds=plan[6],
                workgroups=plan[7])


_PLAN_OK = {}


def conv3d_g_plannable(n, cin, cout, in_size, stride, padding, transposed=False, kernel1=False, in_channel_stride=0):
    """does ``dfm_conv3d_g_plan`` find a tiling for this problem (sample < 2^31 bytes, a block that
    fits the LDS, a grid within the launch limits)?  Cached per problem; the Mfma* modules ask before
    they take the MFMA path, so a shape the kernel rejects runs torch's convolution (with the
    fallback policy's warning) instead of raising DfmHipError from inside the launch."""
    stride, padding, transposed, kernel1 = _triple(stride), _triple(padding), _triple(transposed), _triple(kernel1)
    key = (n, cin, cout, tuple(in_size), stride, padding, transposed, kernel1, in_channel_stride)
    ok = _PLAN_OK.get(key)
    if ok is None:
        out_size = conv3d_g_out_size(in_size, stride, padding, transposed, kernel1)
        d = _conv_desc(n, cin, cout, in_size, out_size, stride, padding, transposed, False, in_channel_stride, kernel1)
        plan = (ctypes.c_int64 * 8)()
        ok = all(o > 0 for o in out_size) and _capi.lib().dfm_conv3d_g_plan(ctypes.byref(d), plan) == 0
        if len(_PLAN_OK) > 4096:
            _PLAN_OK.clear()
        _PLAN_OK[key] = ok
    return ok


def conv3d_g(x, packed, cout, stride=1, padding=1, transposed=False, relu=False, scale=None, shift=None,
             residual=None, kernel1=False):
    """x: (N, C_in, D, H, W) bf16 channels_last_3d.  Returns (N, cout, D', H', W') bf16
    channels_last_3d = relu?(conv(x) * scale + shift + residual).  ``transposed``: per-axis flags of
    the x2 transposed convolution (kernel 3, stride 2, padding 1, output_padding 1); ``kernel1``:
    per-axis flags of kernel extent 1 (padding 0; the packed 27-tap weights' centre index is used)."""
    cstride = _ndhwc_channel_stride(x)
    assert x.is_cuda and x.dtype == torch.bfloat16 and cstride, 'bf16 channels_last_3d (or a channel slice of it)'
    stride, padding, transposed = _triple(stride), _triple(padding), _triple(transposed)
    kernel1 = _triple(kernel1)
    N, cin = x.shape[:2]
    in_size = tuple(x.shape[2:])
    out_size = conv3d_g_out_size(in_size, stride, padding, transposed, kernel1)
    out = torch.empty((N, *out_size, cout), dtype=torch.bfloat16, device=x.device)
    d = _conv_desc(N, cin, cout, in_size, out_size, stride, padding, transposed, relu, cstride, kernel1)
    if scale is not None:
        scale, shift = scale.float().contiguous(), shift.float().contiguous()
        assert scale.numel() == cout and shift.numel() == cout
    if residual is not None:
        assert residual.dtype == torch.bfloat16 and tuple(residual.shape) == (N, cout, *out_size) and \
            _is_ndhwc(residual)
    with torch.cuda.device(x.device):
        _capi.check(_capi.lib().dfm_conv3d_g_fwd(
            ctypes.byref(d), _ptr(x), _ptr(packed), _ptr(scale) if scale is not None else None,
            _ptr(shift) if shift is not None else None, _ptr(residual) if residual is not None else None,
            _ptr(out), _stream_ptr(x.device)))
    return out.permute(0, 4, 1, 2, 3)


def conv3d_g_f32(x, packed, cout, stride=1, padding=1, transposed=False, kernel1=False, acc=None):
    """The general kernel with its fp32 accumulators stored as they are (``dfm_conv3d_g_fwd_f32``):
    x (N, C_in, D, H, W) bf16 channels_last_3d -> fp32 (N, D', H', W', cout), plus ``acc`` (same shape;
    accumulated in place when given)."""
    cstride = _ndhwc_channel_stride(x)
    assert x.is_cuda and x.dtype == torch.bfloat16 and cstride, 'bf16 channels_last_3d (or a channel slice of it)'
    stride, padding, transposed, kernel1 = _triple(stride), _triple(padding), _triple(transposed), _triple(kernel1)
    N, cin = x.shape[:2]
    in_size = tuple(x.shape[2:])
    out_size = conv3d_g_out_size(in_size, stride, padding, transposed, kernel1)
    if acc is not None:
        assert acc.dtype == torch.float32 and tuple(acc.shape) == (N, *out_size, cout) and acc.is_contiguous()
        out = acc
    else:
        out = torch.empty((N, *out_size, cout), dtype=torch.float32, device=x.device)
    d = _conv_desc(N, cin, cout, in_size, out_size, stride, padding, transposed, False, cstride, kernel1)
    with torch.cuda.device(x.device):
        _capi.check(_capi.lib().dfm_conv3d_g_fwd_f32(ctypes.byref(d), _ptr(x), _ptr(packed),
                                                     _ptr(acc) if acc is not None else None, _ptr(out),
                                                     _stream_ptr(x.device)))
    return out


# ---------------------------------------------------------------------------------------------
# fp32 models (the reference's default precision) on the same MFMA kernels: split precision.
#   x = x0 + x1 + x2, w = w0 + w1 + w2: bf16 pieces, 8 + 8 + 8 = every significand bit of an fp32 value
#   conv(x, w) = sum over i + j <= 2 of conv(x_i, w_j)        (dropped: products of order 2^-27)
# six launches whose exact bf16 x bf16 products accumulate in fp32 (conv3d_g_f32) -- fp32-equivalent; with
# two pieces / three launches ('split2') 2^-17 of the sum of |products|.  torch's own fp32 convolution on
# this stack is MIOpen's naive kernel: 1.4 s per 32 -> 1 Conv3d, 31 ms per 2-D convolution, 2.9 s per
# DfMStereoPath training step (profiles/archive/r03_c46_*, r04_c13_*).  Forward, backward-data, backward-weight.
# ---------------------------------------------------------------------------------------------
_FP32_MODE = {'mode': 'split'}
_SPLIT_PIECES = {'split': 3, 'split2': 2}


def set_fp32_mode(mode):
    """How fp32 CUDA inputs of the Mfma* convolutions run.  'split' (default): the MFMA kernels on THREE
    bf16 pieces per operand (8 + 8 + 8 = all 24 significand bits; the six products of total order <= 2,
    dropped terms 2^-27): fp32-equivalent results.  'split2': two pieces, three products (2^-17 of the sum of
    |products|, half the launches).  'torch': torch's convolution (MIOpen) as in rounds 1-3.  Returns the
    previous mode."""
    if mode not in ('split', 'split2', 'torch'):
        raise ValueError(mode)
    prev, _FP32_MODE['mode'] = _FP32_MODE['mode'], mode
    return prev


def split_pieces(t, n=None):
    """fp32 tensor -> n bf16 tensors of the same memory format whose sum is t to 2^-(9 n): each piece is the
    bf16 rounding of what the previous ones left (the remainders are exact in fp32)"""
    n = n or _SPLIT_PIECES.get(_FP32_MODE['mode'], 3)
    pieces, r = [], t
    for i in range(n):
        p = r.to(torch.bfloat16)
        pieces.append(p)
        if i + 1 < n:
            # (a non-finite value -- or one that rounds to bf16's infinity -- lives in the first piece alone: its
            # remainder Inf - Inf = NaN would turn torch's Inf / finite result into NaN)
            r = torch.nan_to_num_(r - p, nan=0.0, posinf=0.0, neginf=0.0)
    return pieces


def _split_pairs(n):
    """(operand piece, weight piece) index pairs of total order < n, largest products first"""
    return [(i, j) for order in range(n) for i in range(order, -1, -1) for j in (order - i,)]


def conv3d_g_split(x, weight, cin, cout, swap=False, flip=0, stride=1, padding=1, transposed=False, kernel1=False,
                   packs=None):
    """fp32 convolution through 6 (or 3) bf16 launches accumulated in fp32.  x: fp32 (N, cin, D, H, W), any
    layout; weight: fp32 5-D torch weight (as ``pack_conv3d_g_weights`` takes it).  Returns fp32
    (N, cout, D', H', W'), a channels_last_3d view.  ``packs``: cached fragment buffers of the weight's
    pieces."""
    xs = split_pieces(x.contiguous(memory_format=torch.channels_last_3d))
    n = len(xs)
    if packs is None or len(packs) != n:
        packs = [pack_conv3d_g_weights(w, cin, cout, swap=swap, flip=flip)
                 for w in split_pieces(weight.detach().float(), n)]
    kw = dict(stride=stride, padding=padding, transposed=transposed, kernel1=kernel1)
    y = None
    for i, j in _split_pairs(n):
        y = conv3d_g_f32(xs[i], packs[j], cout, acc=y, **kw)
    return y.permute(0, 4, 1, 2, 3)


def conv3d_weight_grad_split(x_in, g_out, stride, padding):
    """``conv3d_weight_grad`` of fp32 operands: the significant pairings of their bf16 pieces, summed in fp32"""
    xs = split_pieces(x_in.contiguous(memory_format=torch.channels_last_3d))
    gs = split_pieces(g_out.contiguous(memory_format=torch.channels_last_3d), len(xs))
    out = None
    for i, j in _split_pairs(len(xs)):
        t = conv3d_weight_grad(xs[i], gs[j], stride, padding)
        out = t if out is None else out + t
    return out


class _ConvGSplitFn(torch.autograd.Function):
    """nn.Conv3d / nn.ConvTranspose3d (k3 s2 p1 op1) of an fp32 model through the MFMA kernels in split
    precision; input and output keep the caller's layout (NCDHW for the reference's pipeline).
    ``two_d``: x is a depth-1 view of an NCHW / NHWC tensor and weight a 2-D kernel embedded in the centre
    depth slice (``_embed2d``): kernel extent 1 along depth, stride / transposition on (h, w) only."""

    @staticmethod
    def forward(ctx, x, weight, packs, kind, stride, padding, two_d=False):
        cl = torch.channels_last_3d
        keep_cl = x.is_contiguous(memory_format=cl) and not x.is_contiguous()
        xd = x.detach()
        k1 = (True, False, False) if two_d else False
        if kind == 'conv':
            cout, cin = weight.shape[:2]
            y = conv3d_g_split(xd, weight, cin, cout, stride=stride, padding=padding, kernel1=k1, packs=packs)
        else:
            cin, cout = weight.shape[:2]
            y = conv3d_g_split(xd, weight, cin, cout, swap=True, stride=1, padding=(0, 1, 1) if two_d else 1,
                               transposed=(False, True, True) if two_d else True, kernel1=k1, packs=packs)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (kind, stride, padding, keep_cl, two_d)
        return y if keep_cl else y.contiguous()

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        kind, stride, padding, keep_cl, two_d = ctx.cfg
        gx = gw = None
        in_size = tuple(x.shape[2:])
        w = weight.detach().float()
        k1 = (True, False, False) if two_d else False
        wpad = (1, 1, 1) if two_d else None   # weight gradient of a depth-1 volume: the 27-tap kernel, depth padded
        if kind == 'conv':
            cout, cin = weight.shape[:2]
            stride, padding = _triple(stride), _triple(padding)
            if ctx.needs_input_grad[0]:
                if _bwd_data_supported(in_size, stride, (1, 1, 1) if two_d else padding):
                    up = tuple(st == 2 for st in stride)
                    flip = sum(b for b, st in zip((4, 2, 1), stride) if st == 1)
                    bpad = (0, 1, 1) if two_d else tuple(2 - p for p in padding)
                    gx = conv3d_g_split(gy, w, cout, cin, swap=True, flip=flip, stride=1, padding=bpad,
                                        transposed=up, kernel1=k1)
                else:
                    gx = torch.ops.aten.convolution_backward(
                        gy, x, w, None, list(stride), list((1, 1, 1) if two_d else padding), [1, 1, 1], False,
                        [0, 0, 0], 1, [True, False, False])[0]
            if ctx.needs_input_grad[1]:
                gw = conv3d_weight_grad_split(x, gy, stride, wpad or padding).to(weight.dtype)
        else:
            cin, cout = weight.shape[:2]
            if ctx.needs_input_grad[0]:
                gx = conv3d_g_split(gy, w, cout, cin, swap=False, flip=0, stride=(1, 2, 2) if two_d else 2,
                                    padding=(0, 1, 1) if two_d else 1, kernel1=k1)
            if ctx.needs_input_grad[1]:
                gw = conv3d_weight_grad_split(gy, x, (1, 2, 2) if two_d else 2, wpad or 1).to(weight.dtype)
        if gx is not None and not keep_cl:
            gx = gx.contiguous()
        return gx, gw, None, None, None, None, None


def _split_why_not(module, x, kind):
    """why an fp32 CUDA input does NOT take the split-precision MFMA path (None: it does)"""
    if _FP32_MODE['mode'] == 'torch':
        return 'set_fp32_mode("torch")'
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and x.shape[1] == module.in_channels):
        return 'not an fp32 GPU call of this module'
    n, size = x.shape[0], tuple(x.shape[2:])
    if kind == 'conv':
        if not all(s + 2 * p >= 3 for s, p in zip(size, module.padding)):
            return 'input smaller than the kernel'
        ok = conv3d_g_plannable(n, module.in_channels, module.out_channels, size, module.stride, module.padding)
    else:
        ok = conv3d_g_plannable(n, module.in_channels, module.out_channels, size, 1, 1, transposed=True)
    return None if ok else 'no tiling of the general kernel fits this shape'


def _split_packs(module, cin, cout, swap):
    """(hi, lo) fragment buffers of the module's fp32 weight, cached per weight version"""
    key = (module.weight._version, module.weight.data_ptr(), str(module.weight.device), _FP32_MODE['mode'])
    if module.__dict__.get('_split_key') != key:
        module.__dict__['_split_packs'] = [pack_conv3d_g_weights(w, cin, cout, swap=swap)
                                           for w in split_pieces(module.weight.detach().float())]
        module.__dict__['_split_key'] = key
    return module.__dict__['_split_packs']


# ---------------------------------------------------------------------------------------------
# 2-D 3x3 convolutions through the same kernel: an NHWC (channels_last) tensor is a depth-1 NDHWC
# volume and the kernel is (1, 3, 3).  The 2-D producers / consumers either side of the path
# (SURVEY.md 8f rank 3): SPPUNetNeck (necks/spp_unet_neck.py), BEVHourglass / hourglass2d
# (backbones/bev_hourglass.py, utils/conv_modules.py).  Inference (no autograd through these).
# ---------------------------------------------------------------------------------------------
def pack_conv2d_g_weights(weight, cin, cout, swap=False):
    """torch 2-D weight (dim0, dim1, 3, 3) -> the 27-tap fragment buffer with the 2-D kernel in its
    centre depth slice (the depth axis runs with kernel extent 1)"""
    assert weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3)
    return pack_conv3d_g_weights(weight, cin, cout, swap=swap)


def conv2d_g_why_not(x, cin, cout):
    """inference path: a bf16 channels_last input under no_grad (training goes through
    ``_Conv2dGFn``, which takes any layout: ``MfmaConv2d.train_why_not``)"""
    if not x.is_cuda:
        return 'CPU tensor'
    if cin % 32 or cout % 32:
        return 'channel counts are not multiples of 32'
    why = _why_not_bf16_cl(x, 4)
    if why is not None:
        return why
    if x.shape[1] != cin:
        return 'channel count differs from the module\'s'
    if torch.is_grad_enabled():
        return 'autograd is recording (inference path)'
    if not (x.is_contiguous(memory_format=torch.channels_last) and x.stride(1) == 1):
        return 'input is not channels_last'
    return None


def conv2d_g_eligible(x, cin, cout):
    return conv2d_g_why_not(x, cin, cout) is None


def conv2d_g(x, packed, cout, stride=1, transposed=False, relu=False, scale=None, shift=None, residual=None):
    """x: (N, C_in, H, W) bf16 channels_last.  3x3 convolution, padding 1, stride 1 | 2 -- or the x2
    transposed convolution (kernel 3, stride 2, padding 1, output_padding 1) -- with the fused epilogue
    of ``conv3d_g``; returns (N, cout, H', W') bf16 channels_last."""
    sh, sw = (stride, stride) if not isinstance(stride, (tuple, list)) else tuple(stride)[-2:]
    x5 = x.unsqueeze(2)
    res5 = residual.unsqueeze(2) if residual is not None else None
    y = conv3d_g(x5, packed, cout, stride=(1, sh, sw), padding=(0, 1, 1),
                 transposed=(False, bool(transposed), bool(transposed)), relu=relu, scale=scale, shift=shift,
                 residual=res5, kernel1=(True, False, False))
    return y.squeeze(2)


def _embed2d(w):
    """(a, b, 3, 3) -> (a, b, 3, 3, 3) with the 2-D kernel in the centre depth slice"""
    w3 = w.new_zeros((*w.shape[:2], 3, 3, 3))
    w3[:, :, 1] = w
    return w3


class _Conv2dGFn(torch.autograd.Function):
    """Training path of the 2-D 3x3 convolutions of SPPUNetNeck / BEVHourglass (spp_unet_neck.py:93-119,
    bev_hourglass.py:36-137, conv_modules.py:152-214): forward, backward-data and backward-weight in the
    hand-written MFMA kernels (csrc/conv3d_g.hip with a (1, 3, 3) kernel, csrc/conv3d_wgrad.hip) on NHWC
    copies of the operands; input and output keep the CALLER's layout, so the torch ops either side
    (BatchNorm, bilinear resize, concatenation: NCHW while training) are untouched.
    kind 'conv': nn.Conv2d k3 p1 stride 1 | 2 (+bias); 'convT': nn.ConvTranspose2d k3 s2 p1 op1."""

    @staticmethod
    def forward(ctx, x, weight, bias, kind, stride):
        cl = torch.channels_last
        keep_cl = x.is_contiguous(memory_format=cl) and not x.is_contiguous()
        xcl = x.detach().contiguous(memory_format=cl)
        cin = weight.shape[1] if kind == 'conv' else weight.shape[0]
        cout = weight.shape[0] if kind == 'conv' else weight.shape[1]
        w = weight.detach()
        if cin < 32:  # the 3-channel image skip: zero channels up to one 32-channel chunk
            B, C, H, W = xcl.shape
            xp = xcl.new_zeros((B, H, W, 32))
            xp[..., :C] = xcl.permute(0, 2, 3, 1)
            xcl = xp.permute(0, 3, 1, 2)
            w = torch.cat([w, w.new_zeros(w.shape[0], 32 - cin, 3, 3)], 1)
        cin_p = max(cin, 32)
        scale = shift = None
        if bias is not None:
            shift = bias.detach().float()
            scale = torch.ones_like(shift)
        if kind == 'conv':
            packed = pack_conv2d_g_weights(w, cin_p, cout)
            y = conv2d_g(xcl, packed, cout, stride=stride, scale=scale, shift=shift)
        else:
            packed = pack_conv2d_g_weights(w, cin_p, cout, swap=True)
            y = conv2d_g(xcl, packed, cout, transposed=True)
        ctx.save_for_backward(xcl, weight)
        ctx.cfg = (kind, stride, cin, cout, bias is not None, keep_cl)
        return y if keep_cl else y.contiguous()

    @staticmethod
    def backward(ctx, gy):
        xcl, weight = ctx.saved_tensors
        kind, stride, cin, cout, has_bias, keep_cl = ctx.cfg
        cin_p = max(cin, 32)
        gcl = gy.contiguous(memory_format=torch.channels_last)
        g5, x5 = gcl.unsqueeze(2), xcl.unsqueeze(2)
        k1 = (True, False, False)
        gx = gw = gb = None
        w = weight.detach()
        if kind == 'conv':
            if cin < 32:
                w = torch.cat([w, w.new_zeros(w.shape[0], 32 - cin, 3, 3)], 1)
            if ctx.needs_input_grad[0]:
                # backward-data = the same kernel on the mirrored, channel-swapped weights; a stride-2 axis
                # becomes a transposed axis (even extents, padding 1: MfmaConv2d.train_why_not)
                up = stride == 2
                pk = pack_conv3d_g_weights(w, cout, cin_p, swap=True, flip=4 if up else 7)
                gx = conv3d_g(g5, pk, cin_p, stride=1, padding=(0, 1, 1), transposed=(False, up, up),
                              kernel1=k1).squeeze(2)[:, :cin]
            if ctx.needs_input_grad[1]:
                gw = conv3d_weight_grad(x5, g5, (1, stride, stride), (1, 1, 1))[:, :cin, 1].to(weight.dtype)
            if has_bias and ctx.needs_input_grad[2]:
                gb = gcl.float().sum((0, 2, 3)).to(weight.dtype)
        else:
            if ctx.needs_input_grad[0]:
                pk = pack_conv3d_g_weights(w, cout, cin, swap=False, flip=0)
                gx = conv3d_g(g5, pk, cin, stride=(1, 2, 2), padding=(0, 1, 1), kernel1=k1).squeeze(2)
            if ctx.needs_input_grad[1]:
                gw = conv3d_weight_grad(g5, x5, (1, 2, 2), (1, 1, 1))[:, :, 1].to(weight.dtype)
        if gx is not None and not keep_cl:
            gx = gx.contiguous()
        return gx, gw, gb, None, None


class _Mfma2dMixin:
    """packs the 2-D weight once per version (inference: the weights do not change between calls)"""

    def _packed2d(self, cin, cout, swap):
        key = (self.weight._version, self.weight.data_ptr(), str(self.weight.device))
        if self.__dict__.get('_pack2d_key') != key:
            self.__dict__['_pack2d'] = pack_conv2d_g_weights(self.weight, cin, cout, swap=swap)
            self.__dict__['_pack2d_key'] = key
        return self.__dict__['_pack2d']


class MfmaConv2d(DerivedStateMixin, nn.Conv2d, _Mfma2dMixin):
    """nn.Conv2d (same parameters / state_dict keys).  kernel 3, padding 1, stride 1 | 2, dilation 1,
    groups 1, channels = 32 k, bf16 channels_last input under no_grad: the hand-written MFMA kernel
    (csrc/conv3d_g.hip with a (1, 3, 3) kernel); anything else: torch's convolution, the module's other
    documented path.  ``forward_fused`` = relu?(conv(x) * scale + shift + residual) in one launch
    (eval-mode BatchNorm / bias folded into the epilogue)."""

    @staticmethod
    def covers(in_channels, out_channels, kernel_size, stride=1, padding=0):
        """would an ``MfmaConv2d`` of this configuration ever reach an MFMA / matrix-product path?  (what
        ``modules.ConvModule`` asks before choosing this class over a plain nn.Conv2d: 3x3 / padding 1 /
        stride 1 | 2 with whole 32-channel chunks -- or a narrow input padded to one chunk -- and 1x1 / stride 1 /
        padding 0; any other Conv2d of a 2-D neck stays torch's, silently, as it always was)"""
        pair = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)  # noqa: E731
        k, st, pd = pair(kernel_size), pair(stride), pair(padding)
        if k == (1, 1):
            return st == (1, 1) and pd == (0, 0)
        return (k == (3, 3) and pd == (1, 1) and st in ((1, 1), (2, 2)) and out_channels % 32 == 0 and
                (in_channels % 32 == 0 or in_channels < 32))

    def _cin_padded(self):
        """input channels as the kernel sees them: a narrow input (the 3-channel image of
        upconv_module's last skip, spp_unet_neck.py:51-56) is zero-padded to one 32-channel chunk"""
        return 32 if self.in_channels < 32 else self.in_channels

    def why_not(self, x):
        if not x.is_cuda:
            return 'CPU tensor'
        if not (self.kernel_size == (3, 3) and self.padding == (1, 1) and self.dilation == (1, 1) and
                self.groups == 1 and self.stride in ((1, 1), (2, 2)) and self.padding_mode == 'zeros' and
                self.out_channels % 32 == 0 and (self.in_channels % 32 == 0 or self.in_channels < 32)):
            return 'convolution configuration outside the kernel\'s coverage (3x3, padding 1, stride 1 | 2)'
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            return 'input shape does not match the module'
        if self.in_channels >= 32:
            why = conv2d_g_why_not(x, self.in_channels, self.out_channels)
        else:
            # narrow input: the zero-padded NHWC copy is made here, whatever the caller's layout
            why = _why_not_bf16_cl(x, 4) or \
                ('autograd is recording (the 2-D MFMA path is inference-only)' if torch.is_grad_enabled() else None)
        if why is None and not conv3d_g_plannable(x.shape[0], self._cin_padded(), self.out_channels,
                                                  (1, x.shape[2], x.shape[3]), (1,) + self.stride, (0, 1, 1),
                                                  kernel1=(True, False, False)):
            why = 'no tiling of the general kernel fits this shape'
        return why

    def eligible(self, x):
        return self.why_not(x) is None

    def train_why_not(self, x):
        """why a call with autograd recording does NOT take the MFMA kernels (None: it does)"""
        if not x.is_cuda:
            return 'CPU tensor'
        if not (self.kernel_size == (3, 3) and self.padding == (1, 1) and self.dilation == (1, 1) and
                self.groups == 1 and self.stride in ((1, 1), (2, 2)) and self.padding_mode == 'zeros' and
                self.out_channels % 32 == 0 and (self.in_channels % 32 == 0 or self.in_channels < 32)):
            return 'convolution configuration outside the kernel\'s coverage (3x3, padding 1, stride 1 | 2)'
        if x.dim() != 4 or x.shape[1] != self.in_channels or x.dtype != torch.bfloat16 or \
                self.weight.dtype != torch.bfloat16:
            return 'not a bf16 call of this module'
        n, _, h, w = x.shape
        st = self.stride[0]
        if st == 2 and (h % 2 or w % 2):
            return 'odd extent under stride 2 (backward-data is a transposed convolution)'
        cin, cout = self._cin_padded(), self.out_channels
        ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
        if not (conv3d_g_plannable(n, cin, cout, (1, h, w), (1, st, st), (0, 1, 1), kernel1=(True, False, False)) and
                conv3d_g_plannable(n, cout, cin, (1, ho, wo), (1, 1, 1), (0, 1, 1),
                                   transposed=(False, st == 2, st == 2), kernel1=(True, False, False))):
            return 'no tiling of the general kernel fits this shape'
        return None

    def _packed2d_padded(self):
        key = (self.weight._version, self.weight.data_ptr(), str(self.weight.device))
        if self.__dict__.get('_pack2d_key') != key:
            w = self.weight.detach()
            if self.in_channels < 32:
                w = torch.cat([w, w.new_zeros(w.shape[0], 32 - self.in_channels, 3, 3)], 1)
            self.__dict__['_pack2d'] = pack_conv2d_g_weights(w, self._cin_padded(), self.out_channels)
            self.__dict__['_pack2d_key'] = key
        return self.__dict__['_pack2d']

    def forward_fused(self, x, scale=None, shift=None, residual=None, relu=False):
        if self.bias is not None:
            b = self.bias.float()
            shift = b if shift is None else shift + b * (scale if scale is not None else 1.0)
            if scale is None:
                scale = torch.ones_like(b)
        if self.in_channels < 32:  # NHWC, zero channels up to 32: one small copy (26 MB at 320 x 1280)
            B, C, H, W = x.shape
            xp = x.new_zeros((B, H, W, 32))
            xp[..., :C] = x.permute(0, 2, 3, 1)
            x = xp.permute(0, 3, 1, 2)
        return conv2d_g(x, self._packed2d_padded(), self.out_channels, stride=self.stride, relu=relu,
                        scale=scale, shift=shift, residual=residual)

    def forward(self, x):
        if (self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0) and self.groups == 1 and
                x.is_cuda and x.dim() == 4 and not x.is_contiguous() and
                x.is_contiguous(memory_format=torch.channels_last) and x.dtype == self.weight.dtype):
            # a 1x1 convolution of an NHWC tensor IS a matrix product over its pixel rows: hipBLASLt forward
            # and backward instead of MIOpen's NHWC kernels (naive on this stack: 14 ms per call,
            # profiles/archive/r03_c43_*); the result is the same channels_last tensor torch would return
            w2 = self.weight.view(self.out_channels, self.in_channels)
            if torch.is_grad_enabled() and self.weight.requires_grad:
                y = _PixelLinearFn.apply(x.permute(0, 2, 3, 1), w2, self.bias)
            else:
                y = F.linear(x.permute(0, 2, 3, 1), w2, self.bias)
            return y.permute(0, 3, 1, 2)
        recording = torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad)
        if recording:
            why = self.train_why_not(x)
            if why is None:
                return _Conv2dGFn.apply(x, self.weight, self.bias, 'conv', self.stride[0])
        else:
            why = self.why_not(x)
            if why is None:
                return self.forward_fused(x)
        if self.kernel_size == (3, 3) and x.is_cuda and x.dtype == torch.float32 and 'coverage' not in why:
            y = self._forward_fp32_split(x)
            if y is not None:
                return y
        # (the 1x1 convolutions built through convbn() are torch's by design)
        if self.kernel_size == (3, 3):
            _torch_path(self, x, why)
        return super().forward(x)

    def _forward_fp32_split(self, x):
        """an fp32 model: the depth-1 form of the general kernel in split precision (None: not applicable)"""
        if _FP32_MODE['mode'] == 'torch' or x.dim() != 4 or x.shape[1] != self.in_channels:
            return None
        n, cin, h, w_ = x.shape
        st, cout, cin_p = self.stride[0], self.out_channels, self._cin_padded()
        if not conv3d_g_plannable(n, cin_p, cout, (1, h, w_), (1, st, st), (0, 1, 1), kernel1=(True, False, False)):
            return None
        w = self.weight
        if cin < 32:  # the 3-channel image skip: zero channels up to one 32-channel chunk
            x = torch.cat([x, x.new_zeros((n, 32 - cin, h, w_))], 1)
            w = torch.cat([w, w.new_zeros((cout, 32 - cin, 3, 3))], 1)
        keep_cl = x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
        y = _ConvGSplitFn.apply(x.unsqueeze(2), _embed2d(w), None, 'conv', (1, st, st), (0, 1, 1), True).squeeze(2)
        if self.bias is not None:
            y = y + self.bias.view(1, -1, 1, 1)
        return y if keep_cl else y.contiguous()


class MfmaConvTranspose2d(DerivedStateMixin, nn.ConvTranspose2d, _Mfma2dMixin):
    """nn.ConvTranspose2d kernel 3, stride 2, padding 1, output_padding 1 (hourglass2d's up-convs,
    conv_modules.py:196-214) through the MFMA kernel under the conditions of ``MfmaConv2d``."""

    def why_not(self, x):
        if not x.is_cuda:
            return 'CPU tensor'
        if not (self.kernel_size == (3, 3) and self.padding == (1, 1) and self.stride == (2, 2) and
                self.output_padding == (1, 1) and self.dilation == (1, 1) and self.groups == 1 and
                self.bias is None):
            return 'transposed-convolution configuration outside the kernel\'s coverage'
        why = conv2d_g_why_not(x, self.in_channels, self.out_channels)
        if why is None and not conv3d_g_plannable(x.shape[0], self.in_channels, self.out_channels,
                                                  (1, x.shape[2], x.shape[3]), (1, 1, 1), (0, 1, 1),
                                                  transposed=(False, True, True), kernel1=(True, False, False)):
            why = 'no tiling of the general kernel fits this shape'
        return why

    def eligible(self, x):
        return self.why_not(x) is None

    def train_why_not(self, x):
        if not x.is_cuda:
            return 'CPU tensor'
        if not (self.kernel_size == (3, 3) and self.padding == (1, 1) and self.stride == (2, 2) and
                self.output_padding == (1, 1) and self.dilation == (1, 1) and self.groups == 1 and
                self.bias is None and self.in_channels % 32 == 0 and self.out_channels % 32 == 0):
            return 'transposed-convolution configuration outside the kernel\'s coverage'
        if x.dim() != 4 or x.shape[1] != self.in_channels or x.dtype != torch.bfloat16 or \
                self.weight.dtype != torch.bfloat16:
            return 'not a bf16 call of this module'
        n, _, h, w = x.shape
        k1 = (True, False, False)
        if not (conv3d_g_plannable(n, self.in_channels, self.out_channels, (1, h, w), (1, 1, 1), (0, 1, 1),
                                   transposed=(False, True, True), kernel1=k1) and
                conv3d_g_plannable(n, self.out_channels, self.in_channels, (1, 2 * h, 2 * w), (1, 2, 2), (0, 1, 1),
                                   kernel1=k1)):
            return 'no tiling of the general kernel fits this shape'
        return None

    def forward(self, x, output_size=None):
        recording = torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad)
        if output_size is not None:
            why = 'explicit output_size'
        elif recording:
            why = self.train_why_not(x)
            if why is None:
                return _Conv2dGFn.apply(x, self.weight, None, 'convT', 2)
        else:
            why = self.why_not(x)
            if why is None:
                return conv2d_g(x, self._packed2d(self.in_channels, self.out_channels, True), self.out_channels,
                                transposed=True)
        if (output_size is None and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and
                'coverage' not in why and _FP32_MODE['mode'] != 'torch' and x.shape[1] == self.in_channels and
                conv3d_g_plannable(x.shape[0], self.in_channels, self.out_channels, (1, x.shape[2], x.shape[3]),
                                   (1, 1, 1), (0, 1, 1), transposed=(False, True, True), kernel1=(True, False, False))):
            keep_cl = x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
            y = _ConvGSplitFn.apply(x.unsqueeze(2), _embed2d(self.weight), None, 'convT', 2, 1, True).squeeze(2)
            return y if keep_cl else y.contiguous()
        _torch_path(self, x, why)
        return super().forward(x, output_size)


def _bwd_data_supported(in_size, stride, padding):
    """backward-data of an nn.Conv3d runs in the same kernel when every stride-2 axis has padding 1
    and an even extent (it becomes a transposed axis)"""
    return all(st == 1 or (st == 2 and p == 1 and s % 2 == 0) for s, st, p in zip(in_size, stride, padding))


class _ConvGFn(torch.autograd.Function):
    """nn.Conv3d (kind 'conv') / nn.ConvTranspose3d k3 s2 p1 op1 (kind 'convT') through the MFMA kernel.
    Backward-data is another launch of the same kernel; backward-weight is torch's (MIOpen)."""

    @staticmethod
    def forward(ctx, x, weight, packed, kind, stride, padding):
        ctx.save_for_backward(x, weight)
        ctx.cfg = (kind, stride, padding)
        if kind == 'conv':
            return conv3d_g(x, packed, weight.shape[0], stride, padding)
        return conv3d_g(x, packed, weight.shape[1], transposed=True)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        kind, stride, padding = ctx.cfg
        gy = gy.contiguous(memory_format=torch.channels_last_3d)
        gx = gw = None
        in_size = tuple(x.shape[2:])
        if kind == 'conv':
            cout, cin = weight.shape[:2]
            if ctx.needs_input_grad[0]:
                if _bwd_data_supported(in_size, stride, padding):
                    up = tuple(st == 2 for st in stride)
                    flip = sum(b for b, st in zip((4, 2, 1), stride) if st == 1)
                    pk = pack_conv3d_g_weights(weight, cout, cin, swap=True, flip=flip)
                    gx = conv3d_g(gy, pk, cin, stride=1, padding=tuple(2 - p for p in padding), transposed=up)
                else:
                    gx = torch.ops.aten.convolution_backward(
                        gy, x, weight.to(x.dtype), None, list(stride), list(padding), [1, 1, 1], False,
                        [0, 0, 0], 1, [True, False, False])[0]
            if ctx.needs_input_grad[1]:
                gw = conv3d_weight_grad(x, gy, stride, padding, out_dtype=weight.dtype)
        else:
            cin, cout = weight.shape[:2]
            if ctx.needs_input_grad[0]:
                pk = pack_conv3d_g_weights(weight, cout, cin, swap=False, flip=0)
                gx = conv3d_g(gy, pk, cin, stride=2, padding=1)
            if ctx.needs_input_grad[1]:
                gw = conv3d_weight_grad(gy, x, 2, 1, out_dtype=weight.dtype)
        return gx, gw, None, None, None, None


class MfmaConv3dG(DerivedStateMixin, nn.Conv3d):
    """nn.Conv3d(32 j, 32 k, 3, stride in {1, 2}, padding in {0, 1, 2}, bias=False) whose bf16 /
    NDHWC forward is the general MFMA kernel; any other input takes torch's convolution (MIOpen),
    the module's other documented path.  ``forward_fused`` folds a per-channel scale / shift (an
    eval-mode BatchNorm3d), a residual and the ReLU into the epilogue (inference)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._cache = _PackCache()

    def why_not(self, x):
        if not x.is_cuda:
            return 'CPU tensor'
        if not (self.in_channels % 32 == 0 and self.out_channels % 32 == 0 and self.kernel_size == (3, 3, 3) and
                all(s in (1, 2) for s in self.stride) and all(0 <= p <= 2 for p in self.padding) and
                self.dilation == (1, 1, 1) and self.groups == 1 and self.bias is None and
                self.padding_mode == 'zeros'):
            return 'convolution configuration outside the general kernel\'s coverage'
        why = _why_not_bf16_cl(x, 5)
        if why is not None:
            return why
        cs = _ndhwc_channel_stride(x)
        if not cs:
            return 'input is not channels_last_3d (nor a channel slice of an NDHWC tensor)'
        if not all(s + 2 * p >= 3 for s, p in zip(x.shape[2:], self.padding)):
            return 'input smaller than the kernel'
        if not conv3d_g_plannable(x.shape[0], self.in_channels, self.out_channels, tuple(x.shape[2:]), self.stride,
                                  self.padding, in_channel_stride=0 if cs == self.in_channels else cs):
            return 'no tiling of the general kernel fits this shape'
        return None

    def eligible(self, x):
        return self.why_not(x) is None

    def _packed(self):
        return self._cache.get(self.weight, lambda: pack_conv3d_g_weights(
            self.weight, self.in_channels, self.out_channels))

    def forward(self, x):
        why = self.why_not(x)
        if why is None:
            return _ConvGFn.apply(x, self.weight, self._packed(), 'conv', self.stride, self.padding)
        if x.is_cuda and x.dtype == torch.float32 and 'coverage' not in why:
            why32 = _split_why_not(self, x, 'conv')
            if why32 is None:  # an fp32 model: the same kernel in split precision
                return _ConvGSplitFn.apply(x, self.weight,
                                           _split_packs(self, self.in_channels, self.out_channels, False), 'conv',
                                           self.stride, self.padding)
            why = f'{why}; split precision: {why32}'
        _torch_path(self, x, why)
        return super().forward(x)

    def forward_fused(self, x, scale=None, shift=None, residual=None, relu=False):
        """inference only (no autograd): relu?(conv(x) * scale + shift + residual)"""
        assert self.eligible(x)
        return conv3d_g(x, self._packed(), self.out_channels, self.stride, self.padding, relu=relu,
                        scale=scale, shift=shift, residual=residual)


class MfmaConvTranspose3d(DerivedStateMixin, nn.ConvTranspose3d):
    """nn.ConvTranspose3d(32 j, 32 k, 3, stride=2, padding=1, output_padding=1, bias=False) of the
    hourglass (conv_modules.py:101-117): evaluated per output parity class on the low-resolution
    input by the general MFMA kernel when the input is bf16 / NDHWC."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._cache = _PackCache()

    def why_not(self, x):
        if not x.is_cuda:
            return 'CPU tensor'
        if not (self.in_channels % 32 == 0 and self.out_channels % 32 == 0 and self.kernel_size == (3, 3, 3) and
                self.stride == (2, 2, 2) and self.padding == (1, 1, 1) and self.output_padding == (1, 1, 1) and
                self.dilation == (1, 1, 1) and self.groups == 1 and self.bias is None):
            return 'transposed-convolution configuration outside the general kernel\'s coverage'
        why = _why_not_bf16_cl(x, 5)
        if why is None and not _is_ndhwc(x):
            why = 'input is not channels_last_3d'
        if why is None and not conv3d_g_plannable(x.shape[0], self.in_channels, self.out_channels,
                                                  tuple(x.shape[2:]), 1, 1, transposed=True):
            why = 'no tiling of the general kernel fits this shape'
        return why

    def eligible(self, x):
        return self.why_not(x) is None

    def _packed(self):
        return self._cache.get(self.weight, lambda: pack_conv3d_g_weights(
            self.weight, self.in_channels, self.out_channels, swap=True))

    def forward(self, x, output_size=None):
        why = self.why_not(x) if output_size is None else 'explicit output_size'
        if why is None:
            return _ConvGFn.apply(x, self.weight, self._packed(), 'convT', self.stride, self.padding)
        if output_size is None and x.is_cuda and x.dtype == torch.float32 and 'coverage' not in why:
            why32 = _split_why_not(self, x, 'convT')
            if why32 is None:  # an fp32 model: the same kernel in split precision
                return _ConvGSplitFn.apply(x, self.weight,
                                           _split_packs(self, self.in_channels, self.out_channels, True), 'convT',
                                           self.stride, self.padding)
            why = f'{why}; split precision: {why32}'
        _torch_path(self, x, why)
        return super().forward(x, output_size)
