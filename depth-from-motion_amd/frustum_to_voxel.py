"""Host-side mirror of the sampling stage of ``FrustumToVoxel.forward``
(mmdet3d/models/necks/feature_transformation.py:82-158): one HIP launch
(``dfm_frustum_to_voxel_fwd``) produces cat(Voxel, Voxel_2D), the input of
``voxel_convs``."""
import contextlib
import ctypes
import os
import weakref

import numpy as np
import torch

from . import _capi
from .depth_head import LazyDepthDistribution
from .geometry import stack_meta
from .plane_sweep import _DTYPES, _Workspace, _ptr, _require_gpu, _stream_ptr, _upload


class _F2vFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, stereo, sem, soft, coords, cam4, desc):
        lib = _capi.lib()
        device = stereo.device
        out = _alloc_out(desc, stereo)
        nbytes = lib.dfm_frustum_to_voxel_workspace_bytes(ctypes.byref(desc))
        ws = _Workspace.get(device, nbytes)
        with torch.cuda.device(device):
            _capi.check(
                lib.dfm_frustum_to_voxel_fwd(ctypes.byref(desc), _ptr(stereo),
                                             _ptr(soft) if soft is not None else None,
                                             _ptr(sem) if sem is not None else None, _ptr(coords),
                                             _ptr(cam4), _ptr(out), _ptr(ws), nbytes,
                                             _stream_ptr(device)))
        ctx.desc = desc
        ctx.has_sem = sem is not None
        ctx.shapes = (stereo.shape, None if sem is None else sem.shape, stereo.dtype)
        ctx.save_for_backward(coords, cam4, *(() if soft is None else (soft,)))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        coords, cam4, *rest = ctx.saved_tensors
        soft = rest[0] if rest else None
        lib = _capi.lib()
        desc = ctx.desc
        st_shape, sem_shape, dtype = ctx.shapes
        device = grad_out.device
        go = _grad_in_output_layout(grad_out, desc, dtype)
        g_sem = torch.zeros(sem_shape, dtype=torch.float32, device=device) if ctx.has_sem else None
        g_st = _try_gather_backward(desc, go, soft, None, 0, coords, cam4, st_shape, dtype, g_sem, device)
        if g_st is not None:
            return g_st.to(dtype), (g_sem.to(dtype) if g_sem is not None else None), None, None, None, None
        g_st = torch.zeros(st_shape, dtype=torch.float32, device=device)
        nbytes = lib.dfm_frustum_to_voxel_bwd_workspace_bytes(ctypes.byref(desc))
        ws = _Workspace.get(device, nbytes)
        with torch.cuda.device(device):
            _capi.check(
                lib.dfm_frustum_to_voxel_bwd(ctypes.byref(desc), _ptr(go),
                                             _ptr(soft) if soft is not None else None, _ptr(coords),
                                             _ptr(cam4), _ptr(g_st),
                                             _ptr(g_sem) if g_sem is not None else None, _ptr(ws),
                                             nbytes, _stream_ptr(device)))
        return g_st.to(dtype), (g_sem.to(dtype) if g_sem is not None else None), None, None, None, None


class _F2vFusedFn(torch.autograd.Function):
    """the DepthHead fused into the sampling, with autograd (training): forward =
    dfm_frustum_to_voxel_fused_fwd, backward = dfm_frustum_to_voxel_fused_bwd -- the depth distribution
    (detached in the reference, feature_transformation.py:136) is evaluated from the low-resolution cost in
    both directions"""

    @staticmethod
    def forward(ctx, stereo, sem, cost, col_max, col_sum, scale, coords, cam4, desc):
        lib = _capi.lib()
        device = stereo.device
        out = _alloc_out(desc, stereo)
        nbytes = lib.dfm_frustum_to_voxel_workspace_bytes(ctypes.byref(desc))
        ws = _Workspace.get(device, nbytes)
        with torch.cuda.device(device):
            _capi.check(lib.dfm_frustum_to_voxel_fused_fwd(
                ctypes.byref(desc), _ptr(stereo), _ptr(cost), _ptr(col_max), _ptr(col_sum), scale,
                _ptr(sem) if sem is not None else None, _ptr(coords), _ptr(cam4), _ptr(out), _ptr(ws), nbytes,
                _stream_ptr(device)))
        ctx.desc, ctx.scale = desc, scale
        ctx.has_sem = sem is not None
        ctx.shapes = (stereo.shape, None if sem is None else sem.shape, stereo.dtype)
        ctx.save_for_backward(coords, cam4, cost, col_max, col_sum)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        coords, cam4, cost, col_max, col_sum = ctx.saved_tensors
        lib = _capi.lib()
        desc = ctx.desc
        st_shape, sem_shape, dtype = ctx.shapes
        device = grad_out.device
        go = _grad_in_output_layout(grad_out, desc, dtype)
        g_sem = torch.zeros(sem_shape, dtype=torch.float32, device=device) if ctx.has_sem else None
        bdesc = desc
        g_st = _try_gather_backward(desc, go, None, (cost, col_max, col_sum), ctx.scale, coords, cam4, st_shape, dtype,
                                    g_sem, device)
        if g_st is not None:
            return (g_st.to(dtype), (g_sem.to(dtype) if g_sem is not None else None), None, None, None, None, None,
                    None, None)
        g_st = torch.zeros(st_shape, dtype=torch.float32, device=device)
        nbytes = lib.dfm_frustum_to_voxel_bwd_workspace_bytes(ctypes.byref(bdesc))
        ws = _Workspace.get(device, nbytes)
        with torch.cuda.device(device):
            _capi.check(lib.dfm_frustum_to_voxel_fused_bwd(
                ctypes.byref(bdesc), _ptr(go), _ptr(cost), _ptr(col_max), _ptr(col_sum), ctx.scale, _ptr(coords),
                _ptr(cam4), _ptr(g_st), _ptr(g_sem) if g_sem is not None else None, _ptr(ws), nbytes,
                _stream_ptr(device)))
        return (g_st.to(dtype), (g_sem.to(dtype) if g_sem is not None else None), None, None, None, None, None,
                None, None)


_GRID_CACHE = {}
_BWD_GATHER = {'on': os.environ.get('DFM_NO_F2V_GATHER') != '1',
               # the gradient of a channels-last cost volume in that layout and type (DFM_F2V_PLANAR_GRAD=1: A/B runs)
               'native': os.environ.get('DFM_F2V_PLANAR_GRAD') != '1'}


@contextlib.contextmanager
def bwd_gather(on, native=None):
    """backward by the gather kernel (default) or the pixel-major scatter (A/B runs, tests); ``native=False``: the
    gather writes the planar fp32 gradient also for a channels-last cost volume"""
    prev = dict(_BWD_GATHER)
    _BWD_GATHER['on'] = bool(on)
    if native is not None:
        _BWD_GATHER['native'] = bool(native)
    try:
        yield
    finally:
        _BWD_GATHER['on'], _BWD_GATHER['native'] = prev['on'], prev['native']


def _regular_grid(coords, desc):
    """(x0, dx, y0, dy, z0, dz) if ``coords`` (nz * ny * nx, 3) is the regular grid prepare_coordinates_3d builds
    (x fastest, centres origin + index * step), else None.  Checked ON THE DEVICE once per tensor (one small
    reduction and one host read, cached on the tensor's identity and version): the gather form of the backward
    enumerates voxels by index arithmetic."""
    # the entry is tied to the tensor OBJECT that owns the memory (a view's base; views of one tensor share it): the
    # caching allocator hands a freed address to the next coordinate tensor of the same shape, version 0 again, and
    # an address-only key would serve it the previous tensor's verdict
    owner = coords._base if coords._base is not None else coords
    key = (id(owner), coords.data_ptr(), coords._version, tuple(coords.shape), desc.nz, desc.ny, desc.nx)
    hit = _GRID_CACHE.get(key)
    if hit is not None and hit[1]() is owner:
        return hit[0]
    nz, ny, nx = desc.nz, desc.ny, desc.nx
    grid = None
    if coords.numel() == 3 * nz * ny * nx and min(nz, ny, nx) >= 1:
        c = coords.reshape(nz, ny, nx, 3)
        o = c[0, 0, 0]
        dx = (c[0, 0, nx - 1, 0] - o[0]) / max(nx - 1, 1)
        dy = (c[0, ny - 1, 0, 1] - o[1]) / max(ny - 1, 1)
        dz = (c[nz - 1, 0, 0, 2] - o[2]) / max(nz - 1, 1)
        ix = torch.arange(nx, device=c.device, dtype=torch.float32)
        iy = torch.arange(ny, device=c.device, dtype=torch.float32)
        iz = torch.arange(nz, device=c.device, dtype=torch.float32)
        err = torch.stack([(c[..., 0] - (o[0] + ix.view(1, 1, nx) * dx)).abs().max(),
                           (c[..., 1] - (o[1] + iy.view(1, ny, 1) * dy)).abs().max(),
                           (c[..., 2] - (o[2] + iz.view(nz, 1, 1) * dz)).abs().max()])
        vals = torch.cat([torch.stack([o[0], dx, o[1], dy, o[2], dz]), err]).tolist()
        step = min(abs(vals[1]) if nx > 1 else 1.0, abs(vals[3]) if ny > 1 else 1.0, abs(vals[5]) if nz > 1 else 1.0)
        if max(vals[6:]) <= 1e-3 * step and all(np.isfinite(vals)) and step > 0:
            # (an axis of extent 1 has no step: any non-zero value serves the index arithmetic)
            grid = tuple(v if (i % 2 == 0 or v != 0.0) else 1.0 for i, v in enumerate(vals[:6]))
    if len(_GRID_CACHE) > 16:
        _GRID_CACHE.clear()
    _GRID_CACHE[key] = (grid, weakref.ref(owner))
    return grid


def _try_gather_backward(desc, go, soft, fused, scale, coords, cam4, st_shape, dtype, g_sem, device):
    """the gather form of the backward (csrc/frustum_to_voxel.hip: f2v_bwd_gather_kernel) -> the gradient of the cost
    volume (the kernel overwrites it: no zero fill), None: not applicable.  A channels-last cost volume (the NDHWC
    stack) gets its gradient in its own layout and type -- what the prediction convolution's backward produces for
    the same tensor, so that autograd's accumulation is one contiguous addition; a planar volume gets planar fp32."""
    if not _BWD_GATHER['on']:
        return None
    grid = _regular_grid(coords, desc)
    if grid is None:
        return None
    lib = _capi.lib()
    nbytes = lib.dfm_frustum_to_voxel_bwd_gather_workspace_bytes(ctypes.byref(desc))
    ws = _Workspace.get(device, nbytes)
    g6 = (ctypes.c_float * 6)(*grid)
    cost, cmax, csum = fused if fused is not None else (None, None, None)
    native = bool(desc.stereo_channels_last) and _BWD_GATHER['native']
    if native:
        B, C, D, H, W = st_shape
        g_st = torch.empty((B, D, H, W, C), dtype=dtype, device=device).permute(0, 4, 1, 2, 3)
        entry = lib.dfm_frustum_to_voxel_bwd_gather_cl
    else:
        g_st = torch.empty(st_shape, dtype=torch.float32, device=device)
        entry = lib.dfm_frustum_to_voxel_bwd_gather
    with torch.cuda.device(device):
        rc = entry(ctypes.byref(desc), _ptr(go), _ptr(soft) if soft is not None else None,
                   _ptr(cost) if cost is not None else None, _ptr(cmax) if cmax is not None else None,
                   _ptr(csum) if csum is not None else None, int(scale), _ptr(coords), g6, _ptr(cam4), _ptr(g_st),
                   _ptr(g_sem) if g_sem is not None else None, _ptr(ws), nbytes, _stream_ptr(device))
    if rc == _capi.DFM_ERR_UNSUPPORTED:
        return None
    _capi.check(rc)
    _BWD_GATHER['calls'] = _BWD_GATHER.get('calls', 0) + 1   # (tests read it: which form took the call)
    return g_st


def _grad_in_output_layout(grad_out, desc, dtype):
    """the backward kernels read ``grad_out`` in the layout the forward wrote its output in: an NDHWC
    voxel_convs backward hands a channels_last_3d gradient over, which is then passed as it is (torch's strided
    re-layout to the planar form cost 2.1 ms of a 31 ms training step)"""
    fmt = torch.channels_last_3d if desc.out_channels_last else torch.contiguous_format
    return grad_out.contiguous(memory_format=fmt).to(dtype)


def _alloc_out(desc, stereo):
    ctot = desc.channels + desc.sem_channels
    if desc.out_channels_last:  # (B, Nz, Ny, Nx, C) in memory = channels_last_3d
        return torch.empty((desc.batch, desc.nz, desc.ny, desc.nx, ctot), dtype=stereo.dtype,
                           device=stereo.device).permute(0, 4, 1, 2, 3)
    return torch.empty((desc.batch, ctot, desc.nz, desc.ny, desc.nx), dtype=stereo.dtype, device=stereo.device)


def frustum_to_voxel_sample(stereo_feat, stereo_feat_softmax, img_metas, cur_sem_feats,
                            coordinates_3d, depth_cfg, memory_format=None, sem_atten_feat=True,
                            stereo_atten_feat=False):
    """
    Args:
        stereo_feat: (B, C, D, H, W) cost-volume features
        stereo_feat_softmax: (B, 1, Ds, Hs, Ws) depth distribution (used detached,
            like the reference, feature_transformation.py:136), or a ``LazyDepthDistribution``
            (``depth_head_statistics``): the DepthHead fused into this kernel, inference only
        img_metas: list of dicts with 'cam2img' (4x4) and 'pad_shape'
        cur_sem_feats: (B, Cs, H, W) or None (cat_img_feature=False)
        coordinates_3d: (Nz, Ny, Nx, 3) voxel centres in pseudo-LiDAR coordinates
        depth_cfg: dict with 'depth_min', 'depth_max'
        sem_atten_feat, stereo_atten_feat: the module's switches (feature_transformation.py:141,154):
            weight Voxel_2D / Voxel by the sampled depth distribution; with neither on the
            distribution is not touched (``stereo_feat_softmax`` may be None)
        memory_format: layout of the result; default: channels_last_3d when stereo_feat is (the
            NDHWC stack: voxel_convs' MFMA convolution reads it in place), else contiguous
    Returns:
        (B, C + Cs, Nz, Ny, Nx), same dtype as stereo_feat
    """
    _require_gpu(stereo_feat, 'stereo_feat')
    lib = _capi.lib()
    device = stereo_feat.device
    if stereo_feat.dtype not in _DTYPES:
        raise TypeError('stereo_feat must be float32 or bfloat16')
    # a channels_last_3d cost volume (NDHWC conv stack) is sampled where it lies: it IS the
    # pixel-major layout the kernel stages an NCDHW volume into
    vec = 16 // stereo_feat.element_size()
    cs = 0 if cur_sem_feats is None else cur_sem_feats.shape[1]
    in_place = (not stereo_feat.is_contiguous() and stereo_feat.shape[1] % vec == 0 and cs % vec == 0
                and stereo_feat.is_contiguous(memory_format=torch.channels_last_3d))
    stereo = stereo_feat if in_place else stereo_feat.contiguous()
    B, C, D, H, W = stereo.shape
    desc = _capi.F2vDesc()
    desc.batch, desc.channels, desc.d, desc.h, desc.w = B, C, D, H, W
    desc.stereo_channels_last = 1 if in_place else 0
    if memory_format is None:
        memory_format = torch.channels_last_3d if in_place else torch.contiguous_format
    desc.out_channels_last = 1 if (memory_format == torch.channels_last_3d and C % vec == 0 and
                                   cs % vec == 0) else 0
    sem = soft = lazy = None
    desc.stereo_atten = 1 if stereo_atten_feat else 0
    desc.no_sem_atten = 0 if sem_atten_feat else 1
    if cur_sem_feats is not None:
        sem = cur_sem_feats.to(stereo.dtype)
        # an NHWC semantic map (channels_last 2-D neck) is sampled in place by the 16-byte-block kernel
        sem_cl = (C % vec == 0 and cs % vec == 0 and not sem.is_contiguous() and
                  sem.is_contiguous(memory_format=torch.channels_last) and sem.data_ptr() % 16 == 0)
        desc.sem_channels_last = 1 if sem_cl else 0
        if not sem_cl:
            sem = sem.contiguous()
        desc.sem_channels, desc.hsem, desc.wsem = sem.shape[1:]
    if stereo_atten_feat or (sem is not None and sem_atten_feat):  # pred_disp is wanted (:133)
        if isinstance(stereo_feat_softmax, LazyDepthDistribution):
            lazy = stereo_feat_softmax
            if lazy.dtype != stereo.dtype:
                raise TypeError('the fused depth head needs cost and stereo_feat in the same dtype')
            desc.ds, desc.hs, desc.ws = lazy.shape[2:]
        else:
            soft = stereo_feat_softmax.detach().to(stereo.dtype).contiguous()
            desc.ds, desc.hs, desc.ws = soft.shape[2:]
    else:
        desc.ds = desc.hs = desc.ws = 1
    coords = coordinates_3d.to(device=device, dtype=torch.float32).contiguous()
    desc.nz, desc.ny, desc.nx = coords.shape[:3]
    pad_shape = img_metas[0]['pad_shape']  # the reference uses sample 0's for all (:101)
    desc.pad_h, desc.pad_w = float(pad_shape[0]), float(pad_shape[1])
    desc.depth_min = float(depth_cfg['depth_min'])
    desc.depth_span = float(depth_cfg['depth_max'] - depth_cfg['depth_min'])
    desc.dtype = _DTYPES[stereo.dtype]
    cam = stack_meta(img_metas, 'cam2img')  # (staged on the device by stage_geometry: padded where it lies)
    cam4 = torch.eye(4, device=cam.device).repeat(B, 1, 1)
    cam4[:, :cam.shape[1], :cam.shape[2]] = cam
    cam4 = cam4.reshape(B, 16).contiguous() if cam.is_cuda and cam.device == device else \
        _upload(cam4.cpu().reshape(B, 16), device)
    if lazy is not None and torch.is_grad_enabled() and (stereo.requires_grad or
                                                         (sem is not None and sem.requires_grad)):
        # training with the depth head fused (the backward ignores the input layouts, like _F2vFn's)
        return _F2vFusedFn.apply(stereo, sem, lazy.cost, lazy.col_max, lazy.col_sum, int(lazy.scale), coords, cam4,
                                 desc)
    if lazy is not None:
        lib = _capi.lib()
        out = _alloc_out(desc, stereo)
        nbytes = lib.dfm_frustum_to_voxel_workspace_bytes(ctypes.byref(desc))
        ws = _Workspace.get(device, nbytes)
        with torch.cuda.device(device):
            _capi.check(lib.dfm_frustum_to_voxel_fused_fwd(
                ctypes.byref(desc), _ptr(stereo.detach()), _ptr(lazy.cost), _ptr(lazy.col_max),
                _ptr(lazy.col_sum), lazy.scale, _ptr(sem.detach()) if sem is not None else None, _ptr(coords), _ptr(cam4), _ptr(out),
                _ptr(ws), nbytes, _stream_ptr(device)))
        return out
    return _F2vFn.apply(stereo, sem, soft, coords, cam4, desc)
