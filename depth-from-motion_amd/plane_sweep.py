"""Host-side mirror of the reference's ``build_dfm_cost``
(mmdet3d/models/backbones/dfm_backbone.py:217-314): same name, arguments and
argument meaning; the body is one call into the HIP library.

Differences from the reference, all deliberate and documented in DESIGN.md:
  * B > 1 is supported with the obviously intended per-sample semantics (the
    reference loop is only correct for B == 1, dfm_backbone.py:257-275).
  * non-finite sampling coordinates (a plane exactly through the camera
    centre) give 0 like torch's GPU grid_sample; torch's CPU kernel gives NaN.
"""
import contextlib
import os
import ctypes
import threading

import numpy as np
import torch

from . import _capi

_DTYPES = {torch.float32: _capi.DFM_F32, torch.bfloat16: _capi.DFM_BF16}


def _require_gpu(t, name):
    if not t.is_cuda:
        raise RuntimeError(
            f'{name} must live on the GPU: depth-from-motion_amd has no CPU path '
            '(the HIP kernels are the product; the CPU oracle is test-only)')


def _pad4x4_batch(m, batch_size):
    """(>=B, 3|4, 3|4) -> (B,4,4): the padding points_img2cam / points_cam2img apply
    (utils.py:199-203, 239-240): rows :3 of a 4-row matrix, identity elsewhere."""
    m = m[:batch_size]
    rows = 3 if m.shape[-2] == 4 else m.shape[-2]
    out = torch.eye(4, dtype=torch.float32, device=m.device).repeat(batch_size, 1, 1)
    out[:, :rows, :m.shape[-1]] = m[:, :rows]
    return out


def camera_matrices(cam2imgs, cur2prevs, batch_size, device, cam2img_inv=None):
    """(B,16) fp32 device tensors: padded cam2img, its fp32 inverse, cur2prev.

    Inputs that already live on ``device`` (the reference pipeline hands device tensors,
    dfm_backbone.py:151-154) never leave it: ``dfm_camera_prepare`` pads and inverts them in one
    small kernel on the current stream, with no host round trip or synchronisation.  Host
    inputs (lists / numpy / CPU tensors, what ``img_metas`` carries) are inverted on the host
    with ``torch.inverse`` in fp32 -- exactly the op the reference's PyTorch-CPU path runs
    (utils.py:241), which is what the bit-exact parity tests replay -- and uploaded once.
    ``cam2img_inv``: a precomputed (B,4,4) inverse (bit-exact replay of another backend).
    """
    def as_f32(x):
        if isinstance(x, (list, tuple)):
            if len(x) and torch.is_tensor(x[0]):
                x = torch.stack(list(x))
            else:
                x = np.asarray(x, dtype=np.float32)
        return torch.as_tensor(x).detach().to(torch.float32)

    cam2imgs, cur2prevs = as_f32(cam2imgs), as_f32(cur2prevs)
    if cam2imgs.device == device and device.type == 'cuda':
        # device-resident: one small kernel pads + inverts, nothing touches the host
        lib = _capi.lib()
        k = cam2imgs[:batch_size].contiguous()
        P = torch.empty((batch_size, 16), dtype=torch.float32, device=device)
        Pinv = torch.empty_like(P)
        with torch.cuda.device(device):
            _capi.check(lib.dfm_camera_prepare(_ptr(k), k.shape[-2], k.shape[-1], batch_size, _ptr(P),
                                               _ptr(Pinv), _stream_ptr(device)))
        if cam2img_inv is not None:
            Pinv = as_f32(cam2img_inv)[:batch_size].to(device).reshape(batch_size, 16).contiguous()
        T = cur2prevs[:batch_size].to(device).reshape(batch_size, 16).contiguous()
        return P, Pinv, T
    P = _pad4x4_batch(cam2imgs.cpu(), batch_size)
    if cam2img_inv is not None:
        Pinv = as_f32(cam2img_inv)[:batch_size].cpu()
    else:
        Pinv = torch.stack([torch.inverse(P[i]) for i in range(batch_size)])
    if cur2prevs.device.type == 'cuda':
        # the detector moves cur2prevs to the device (dfm.py:288-293) while the intrinsics stay
        # img_meta lists: no device -> host round trip for the poses, one pinned upload for P / Pinv
        pack = _upload(torch.stack([P, Pinv]).reshape(2, batch_size, 16), device)
        return pack[0], pack[1], cur2prevs[:batch_size].to(device).reshape(batch_size, 16).contiguous()
    T = cur2prevs[:batch_size]
    pack = _upload(torch.stack([P, Pinv, T]).reshape(3, batch_size, 16), device)
    return pack[0], pack[1], pack[2]


def _make_desc(cur_feats, num_depths, feat_sample_factor, cost_sample_factor, img_shape, flip,
               img_crop_offset, img_scale_factor):
    batch_size, channels, h_in, w_in = cur_feats.shape
    desc = _capi.SweepDesc()
    desc.batch, desc.channels, desc.h_in, desc.w_in = batch_size, channels, h_in, w_in
    desc.num_depths = num_depths
    # Python round (banker's), dfm_backbone.py:242-243
    desc.h_out = round(h_in / cost_sample_factor)
    desc.w_out = round(w_in / cost_sample_factor)
    desc.feat_sample_factor = float(feat_sample_factor)
    desc.cost_sample_factor = float(cost_sample_factor)
    desc.img_scale_factor = float(img_scale_factor)
    desc.crop_x = float(img_crop_offset[0])
    desc.crop_y = float(img_crop_offset[1])
    desc.org_w = float(img_shape[1])
    desc.flip = 1 if flip else 0
    desc.dtype = _DTYPES[cur_feats.dtype]
    return desc


def _upload(t, device):
    """small host tensor -> device without blocking the host on the stream: a pageable H2D copy
    waits for everything queued before it, which serialises the Python launch loop of the next
    step with the GPU work of the previous one (profiles/archive/r02_c26_*: 2x on the multi-view path)"""
    if t.device.type != 'cpu':
        return t.to(device)
    return t.contiguous().pin_memory().to(device, non_blocking=True)


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class _Workspace:
    """Scratch (blocked feature copies, GroupNorm partials, pixel-major staging), grown on
    demand and kept so that the steady state allocates nothing.  One buffer per (device,
    stream): two ops issued on different streams never share scratch."""
    _bufs = {}

    @classmethod
    def get(cls, device, nbytes):
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        buf = cls._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            cls._bufs[key] = buf
        return buf


# ---------------------------------------------------------------------------
# launch options: per call, never process state inside the library
# ---------------------------------------------------------------------------
_tls = threading.local()
_OPT_FIELDS = {'kernel': 'kernel', 'lanes': 'lanes_per_workgroup', 'lds_kib': 'lds_kib',
               'blocks_per_group': 'blocks_per_group', 'planes': 'planes_per_workgroup',
               'bands_per_chunk': 'bands_per_chunk', 'points_per_lane': 'points_per_lane',
               'pipeline': 'pipeline', 'store_align': 'store_align_points',
               'pair_stores': 'pair_stores', 'unpack': 'unpack'}


def make_opts(**kw):
    """dfm_sweep_opts from keywords (kernel, lanes, lds_kib, blocks_per_group, planes,
    bands_per_chunk, points_per_lane, pipeline, store_align, pair_stores); unspecified fields = library default."""
    o = _capi.SweepOpts()
    for k, v in kw.items():
        setattr(o, _OPT_FIELDS[k], int(v or 0))
    return o


@contextlib.contextmanager
def launch_options(**kw):
    """A/B runs and the kernel-mode parity matrix: every plane-sweep call made by THIS thread
    inside the block (forward and backward, also through ``build_dfm_cost``) carries these
    options to the C ABI.  Nothing is stored in the library."""
    prev = getattr(_tls, 'opts', None)
    _tls.opts = make_opts(**kw) if kw else None  # no keywords: back to tuned / default
    try:
        yield _tls.opts
    finally:
        _tls.opts = prev


# strided fp32 sweeps: the prev map's gradient by the gather kernel (True) or the LDS-atomic tile kernel (A/B runs,
# tests: ``prev_gather(False)``)
_PREV_GATHER = {'on': os.environ.get('DFM_NO_PREV_GATHER') != '1'}


@contextlib.contextmanager
def prev_gather(on):
    prev, _PREV_GATHER['on'] = _PREV_GATHER['on'], bool(on)
    try:
        yield
    finally:
        _PREV_GATHER['on'] = prev


_bwd_kernel = None  # process-wide (autograd runs backward functions on its own threads)
_bwd_kernel_lock = threading.RLock()  # one override at a time: nested use is fine, two threads take turns


@contextlib.contextmanager
def backward_kernel(kernel):
    """Force the kernel of every plane-sweep BACKWARD launched inside the block, from any thread
    (``dfm_plane_sweep_bwd_opts``: 1 = lane-per-point scatter, 5 = LDS-atomic tile kernel for both
    maps, 6 / None = default: the matrix-product kernel where it applies).  ``launch_options`` is
    thread-local and never reaches the autograd threads."""
    global _bwd_kernel
    with _bwd_kernel_lock:
        prev = _bwd_kernel
        _bwd_kernel = kernel
        try:
            yield
        finally:
            _bwd_kernel = prev


def _current_opts(schedule=None):
    o = getattr(_tls, 'opts', None)
    if schedule:
        o2 = _capi.SweepOpts()
        if o is not None:
            ctypes.pointer(o2)[0] = o
        o2.bands_per_chunk = int(schedule)
        return o2
    return o


def _nhwc(t):
    """(B, C, H, W) stored (B, H, W, C): torch channels_last, 16-byte aligned whole channel blocks"""
    return (t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last) and
            t.shape[1] % (16 // t.element_size()) == 0 and t.data_ptr() % 16 == 0)


def plane_sweep_forward(desc, cur_feats, prev_feats, depths, P, Pinv, T, out=None,
                        channels_last=False, schedule=None):
    """Raw launch: everything already on the device.  ``channels_last``: write the volume
    as (B, D, H, W, 2C) and return it as a (B, 2C, D, H, W) channels_last_3d tensor; channels_last
    (NHWC) feature maps are then sampled in place (``dfm_plane_sweep_fwd_nhwc``: no pack pass).
    ``schedule``: bands_per_chunk of this call (None: tuned / default)."""
    lib = _capi.lib()
    device = cur_feats.device
    if channels_last and _nhwc(cur_feats) and _nhwc(prev_feats):
        if out is None:
            out = torch.empty((desc.batch, desc.num_depths, desc.h_out, desc.w_out, 2 * desc.channels),
                              dtype=cur_feats.dtype, device=device).permute(0, 4, 1, 2, 3)
        assert out.is_contiguous(memory_format=torch.channels_last_3d)
        nbytes = 256 + 4 * desc.channels
        ws = _Workspace.get(device, nbytes)
        with torch.cuda.device(device):
            _capi.check(
                lib.dfm_plane_sweep_fwd_nhwc(
                    ctypes.byref(desc), _ptr(cur_feats), _ptr(prev_feats), _ptr(depths), _ptr(P),
                    _ptr(Pinv), _ptr(T), _ptr(out), _ptr(ws), nbytes, _stream_ptr(device)))
        return out
    if (not channels_last and _nhwc(cur_feats) and _nhwc(prev_feats) and _current_opts(schedule) is None and
            (desc.h_out * desc.w_out) % (16 // cur_feats.element_size()) == 0):
        # NHWC maps, reference-layout volume: the transpose kernel on the caller's maps (no pack pass)
        if out is None:
            out = torch.empty((desc.batch, 2 * desc.channels, desc.num_depths, desc.h_out, desc.w_out),
                              dtype=cur_feats.dtype, device=device)
        nbytes = 256 + 4 * desc.channels
        ws = _Workspace.get(device, nbytes)
        with torch.cuda.device(device):
            rc = lib.dfm_plane_sweep_fwd_from_nhwc(
                ctypes.byref(desc), _ptr(cur_feats), _ptr(prev_feats), _ptr(depths), _ptr(P), _ptr(Pinv),
                _ptr(T), _ptr(out), _ptr(ws), nbytes, _stream_ptr(device))
        if rc == 0:
            return out
        if rc != _capi.DFM_ERR_UNSUPPORTED:
            _capi.check(rc)
    cur_feats, prev_feats = cur_feats.contiguous(), prev_feats.contiguous()
    if channels_last:
        if out is None:
            out = torch.empty((desc.batch, desc.num_depths, desc.h_out, desc.w_out, 2 * desc.channels),
                              dtype=cur_feats.dtype, device=device).permute(0, 4, 1, 2, 3)
        assert out.is_contiguous(memory_format=torch.channels_last_3d)
        nbytes = lib.dfm_plane_sweep_cl_workspace_bytes(ctypes.byref(desc))
        ws = _Workspace.get(device, nbytes)
        with torch.cuda.device(device):
            _capi.check(
                lib.dfm_plane_sweep_fwd_channels_last(
                    ctypes.byref(desc), _ptr(cur_feats), _ptr(prev_feats), _ptr(depths), _ptr(P),
                    _ptr(Pinv), _ptr(T), _ptr(out), _ptr(ws), nbytes, _stream_ptr(device)))
        return out
    if out is None:
        out = torch.empty((desc.batch, 2 * desc.channels, desc.num_depths, desc.h_out, desc.w_out),
                          dtype=cur_feats.dtype, device=device)
    nbytes = lib.dfm_plane_sweep_workspace_bytes(ctypes.byref(desc))
    ws = _Workspace.get(device, nbytes)
    opts = _current_opts(schedule)
    with torch.cuda.device(device):
        _capi.check(
            lib.dfm_plane_sweep_fwd_opts(ctypes.byref(desc), _ptr(cur_feats), _ptr(prev_feats),
                                         _ptr(depths), _ptr(P), _ptr(Pinv), _ptr(T), _ptr(out),
                                         _ptr(ws), nbytes, _stream_ptr(device),
                                         ctypes.byref(opts) if opts is not None else None))
    return out


def plane_sweep_autotune(desc, cur_feats, prev_feats, depths, P, Pinv, T, out):
    """Times the candidate launch shapes / workgroup schedules of the LDS-staged kernel on these
    tensors and caches the fastest for this (device, shape) inside the library
    (``dfm_plane_sweep_autotune``; synchronous; ``out`` holds valid results afterwards).
    Returns the choice as a dict.  ``dfm_plane_sweep_fwd`` does this by itself on the first
    launch of a volume >= 1 GB."""
    lib = _capi.lib()
    device = cur_feats.device
    nbytes = lib.dfm_plane_sweep_workspace_bytes(ctypes.byref(desc))
    ws = _Workspace.get(device, nbytes)
    chosen = _capi.SweepOpts()
    with torch.cuda.device(device):
        _capi.check(
            lib.dfm_plane_sweep_autotune(ctypes.byref(desc), _ptr(cur_feats), _ptr(prev_feats),
                                         _ptr(depths), _ptr(P), _ptr(Pinv), _ptr(T), _ptr(out), _ptr(ws),
                                         nbytes, _stream_ptr(device), ctypes.byref(chosen)))
    return chosen.as_dict()


def plane_sweep_tuning(desc):
    """The cached launch options for this shape on the current device, or None."""
    lib = _capi.lib()
    o = _capi.SweepOpts()
    rc = lib.dfm_plane_sweep_tuning(ctypes.byref(desc), ctypes.byref(o))
    if rc < 0:
        _capi.check(rc)
    return o.as_dict() if rc == 1 else None


class _PlaneSweepFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, cur_feats, prev_feats, depths, P, Pinv, T, desc, channels_last=False):
        ctx.desc = desc
        ctx.save_for_backward(depths, P, Pinv, T)
        ctx.in_dtype = cur_feats.dtype
        return plane_sweep_forward(desc, cur_feats, prev_feats, depths, P, Pinv, T,
                                   channels_last=channels_last)

    @staticmethod
    def backward(ctx, grad_out):
        depths, P, Pinv, T = ctx.saved_tensors
        g_cur, g_prev = plane_sweep_backward(ctx.desc, grad_out, depths, P, Pinv, T)
        return g_cur.to(ctx.in_dtype), g_prev.to(ctx.in_dtype), None, None, None, None, None, None


def plane_sweep_backward(desc, grad_out, depths, P, Pinv, T):
    """Gradients of the plane sweep with respect to the two feature maps, in **fp32** --
    ``(grad_cur, grad_prev)``, each ``(B, C, H, W)`` -- from the gradient of the volume
    (``dfm_plane_sweep_bwd``; autograd of ``F.grid_sample`` at dfm_backbone.py:296-311).  The autograd
    function casts them to the maps' dtype; tests and mixed-precision trainers that keep fp32 master
    gradients take them from here."""
    lib = _capi.lib()
    device = grad_out.device
    _require_gpu(grad_out, 'grad_out')
    shape = (desc.batch, desc.channels, desc.h_in, desc.w_in)
    opts = _current_opts()
    if _bwd_kernel is not None:
        opts = make_opts(kernel=_bwd_kernel)
    if (opts is None and not grad_out.is_contiguous() and grad_out.dim() == 5 and
            grad_out.is_contiguous(memory_format=torch.channels_last_3d)):
        # the NDHWC stack's gradient is read where it lies (the 236 MB conversion to the reference
        # layout cost 2.2 ms of a 20 ms training step at config K)
        # up to 2 GB: re-laid by the library's LDS-tile transpose (copy speed) into a scratch of the
        # volume's size; larger volumes are read in place (no extra memory)
        if (_PREV_GATHER['on'] and desc.cost_sample_factor >= 1.5 and desc.channels % 32 == 0 and
                _DTYPES.get(grad_out.dtype) == desc.dtype):
            # strided sweeps (config K): both maps by the gather kernel, the volume read where it lies -- a hit is
            # one contiguous run of 32 channels -- and the map gradients written pixel-major (returned as
            # channels_last (B, C, H, W) tensors: the layout the NHWC necks' backward wants).  No re-layout
            # pass, no scratch of the volume's size, no atomics, no zero-filled maps.
            nb = lib.dfm_plane_sweep_bwd_prev_gather_workspace_bytes(ctypes.byref(desc))
            gws = _Workspace.get(device, nb)
            maps = [torch.empty((desc.batch, desc.h_in, desc.w_in, desc.channels), dtype=torch.float32,
                                device=device) for _ in range(2)]
            rc = 0
            with torch.cuda.device(device):
                for half in (0, 1):
                    if rc == 0:
                        rc = lib.dfm_plane_sweep_bwd_gather(ctypes.byref(desc), half, _ptr(grad_out), 1, _ptr(depths),
                                                            _ptr(P), _ptr(Pinv), _ptr(T), _ptr(maps[half]), 1,
                                                            _ptr(gws), nb, _stream_ptr(device))
            if rc == 0:
                return maps[0].permute(0, 3, 1, 2), maps[1].permute(0, 3, 1, 2)
            if rc != _capi.DFM_ERR_UNSUPPORTED:
                _capi.check(rc)
        nbytes = grad_out.numel() * grad_out.element_size()
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device) if nbytes <= (2 << 30) else None
        g_cur = torch.zeros(shape, dtype=torch.float32, device=device)
        g_prev = torch.zeros(shape, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            rc = lib.dfm_plane_sweep_bwd_channels_last(ctypes.byref(desc), _ptr(grad_out), _ptr(depths), _ptr(P),
                                                       _ptr(Pinv), _ptr(T), _ptr(g_cur), _ptr(g_prev),
                                                       _ptr(ws) if ws is not None else None,
                                                       nbytes if ws is not None else 0, _stream_ptr(device))
        if rc == 0:
            return g_cur, g_prev
        if rc != _capi.DFM_ERR_UNSUPPORTED:
            _capi.check(rc)
    grad_out = grad_out.contiguous()
    if (opts is None and desc.cost_sample_factor >= 1.5 and grad_out.dtype == torch.float32 and
            desc.channels % 32 == 0 and (desc.h_out * desc.w_out) % 16 == 0):
        # strided fp32 sweeps (config K): the cur map's taps of a lattice point stay inside one 3x3 pixel
        # window over all depth planes -- a wave keeps it in registers and writes it out once, into a
        # PIXEL-MAJOR gradient map (returned as a channels_last (B, C, H, W) tensor, the layout the NHWC
        # necks' backward wants); the prev map stays with the LDS-atomic tile kernel
        g_cur = torch.zeros((desc.batch, desc.h_in, desc.w_in, desc.channels), dtype=torch.float32,
                            device=device).permute(0, 3, 1, 2)
        g_prev = torch.empty(shape, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            # the prev map: a lane per map pixel gathers its contributions plane by plane through the planes'
            # inverse homographies and STORES the sums -- no zero-filled map (csrc/plane_sweep_bwd_gather.hip;
            # round 5) -- the LDS-atomic tile kernel (kernel 8, prev map only) where that form does not apply
            rc = _capi.DFM_ERR_UNSUPPORTED
            if _PREV_GATHER['on']:
                nb = lib.dfm_plane_sweep_bwd_prev_gather_workspace_bytes(ctypes.byref(desc))
                gws = _Workspace.get(device, nb)
                rc = lib.dfm_plane_sweep_bwd_prev_gather(ctypes.byref(desc), _ptr(grad_out), _ptr(depths), _ptr(P),
                                                         _ptr(Pinv), _ptr(T), _ptr(g_prev), _ptr(gws), nb,
                                                         _stream_ptr(device))
            if rc == _capi.DFM_ERR_UNSUPPORTED:
                g_prev.zero_()
                prev_only = make_opts(kernel=8)
                rc = lib.dfm_plane_sweep_bwd_opts(ctypes.byref(desc), _ptr(grad_out), _ptr(depths), _ptr(P),
                                                  _ptr(Pinv), _ptr(T), _ptr(g_prev), _ptr(g_prev), _stream_ptr(device),
                                                  ctypes.byref(prev_only))
            if rc == 0:
                rc = lib.dfm_plane_sweep_bwd_cur_nhwc(ctypes.byref(desc), _ptr(grad_out), _ptr(depths), _ptr(P),
                                                      _ptr(Pinv), _ptr(T), _ptr(g_cur), _stream_ptr(device))
        if rc == 0:
            return g_cur, g_prev
        if rc != _capi.DFM_ERR_UNSUPPORTED:
            _capi.check(rc)
    g_cur = torch.zeros(shape, dtype=torch.float32, device=device)
    g_prev = torch.zeros(shape, dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _capi.check(
            lib.dfm_plane_sweep_bwd_opts(ctypes.byref(desc), _ptr(grad_out), _ptr(depths),
                                         _ptr(P), _ptr(Pinv), _ptr(T), _ptr(g_cur), _ptr(g_prev),
                                         _stream_ptr(device),
                                         ctypes.byref(opts) if opts is not None else None))
    return g_cur, g_prev


def build_dfm_cost(cur_feats,
                   prev_feats,
                   depths,
                   feat_sample_factor,
                   cost_sample_factor,
                   cam2imgs,
                   cur2prevs,
                   img_shape,
                   flip=False,
                   img_crop_offset=(0, 0),
                   img_scale_factor=1.0,
                   memory_format=torch.contiguous_format,
                   cam2img_inv=None):
    """Plane-sweep cost volume, drop-in for the reference function.

    ``memory_format`` (extension): ``torch.channels_last_3d`` returns the same tensor
    (same shape, same values bit for bit) laid out (B, D, H, W, 2C) in memory -- the
    layout MIOpen's bf16 Conv3d runs in and the faster one to write (one contiguous run
    per lattice point).  Needs C to be a multiple of 16 bytes.

    Args:
        cur_feats/prev_feats: [B, C, H, W] fp32 or bf16, on the GPU
        depths: [D] or [1, D]
        cam2imgs: [B, 4, 4] original intrinsics (``ori_cam2img``)
        cur2prevs: [>=B, 4, 4]; indexed by the batch index like the reference
            (dfm_backbone.py:270)
        img_shape: (org_h, org_w) of the original image (flip only)
        cam2img_inv (extension): precomputed fp32 inverse of the padded intrinsics, (B,4,4);
            default: computed where ``cam2imgs`` lives (device: batched ``torch.linalg.inv``,
            no host sync; host: ``torch.inverse`` like the reference's CPU path)

    Returns:
        cost_volume: [B, 2C, D, H_out, W_out], same dtype as the inputs
    """
    _require_gpu(cur_feats, 'cur_feats')
    _require_gpu(prev_feats, 'prev_feats')
    if cur_feats.dtype not in _DTYPES or prev_feats.dtype != cur_feats.dtype:
        raise TypeError('cur_feats/prev_feats must both be float32 or bfloat16')
    assert cur_feats.dim() == 4 and cur_feats.shape == prev_feats.shape
    device = cur_feats.device
    if not (_nhwc(cur_feats) and _nhwc(prev_feats)):
        cur_feats = cur_feats.contiguous()   # (NHWC maps stay as they are: sampled in place where a kernel can)
        prev_feats = prev_feats.contiguous()
    depths = depths.reshape(-1).to(device=device, dtype=torch.float32).contiguous()
    batch_size = cur_feats.shape[0]
    desc = _make_desc(cur_feats, depths.numel(), feat_sample_factor, cost_sample_factor, img_shape,
                      flip, img_crop_offset, img_scale_factor)
    P, Pinv, T = camera_matrices(cam2imgs, cur2prevs, batch_size, device, cam2img_inv)
    if memory_format not in (torch.contiguous_format, torch.channels_last_3d):
        raise ValueError('memory_format must be contiguous_format or channels_last_3d')
    return _PlaneSweepFn.apply(cur_feats, prev_feats, depths, P, Pinv, T, desc,
                               memory_format == torch.channels_last_3d)


def plane_sweep_grid(cur_feats, depths, feat_sample_factor, cost_sample_factor, cam2imgs, cur2prevs,
                     img_shape, flip=False, img_crop_offset=(0, 0), img_scale_factor=1.0, sample=0):
    """Parity aid: the normalised (cur_grid, prev_grid), each (D*H_out*W_out, 2),
    that the reference feeds to F.grid_sample (dfm_backbone.py:291-294)."""
    _require_gpu(cur_feats, 'cur_feats')
    lib = _capi.lib()
    device = cur_feats.device
    depths = depths.reshape(-1).to(device=device, dtype=torch.float32).contiguous()
    desc = _make_desc(cur_feats, depths.numel(), feat_sample_factor, cost_sample_factor, img_shape,
                      flip, img_crop_offset, img_scale_factor)
    P, Pinv, T = camera_matrices(cam2imgs, cur2prevs, cur_feats.shape[0], device)
    n = desc.num_depths * desc.h_out * desc.w_out
    cur_grid = torch.empty((n, 2), dtype=torch.float32, device=device)
    prev_grid = torch.empty((n, 2), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _capi.check(
            lib.dfm_plane_sweep_grid(ctypes.byref(desc), sample, _ptr(depths), _ptr(P), _ptr(Pinv),
                                     _ptr(T), _ptr(cur_grid), _ptr(prev_grid), _stream_ptr(device)))
    return cur_grid, prev_grid
