"""Host-side mirrors of the reference's multi-view voxel lifting:

* ``point_sample``  -- mmdet3d/models/fusion_layers/point_fusion.py:14-106
* ``mv_feature_transformation`` -- the sampling/reduction part of
  ``MultiViewDfM.feature_transformation`` (mmdet3d/models/detectors/
  multiview_dfm.py:119-208, ``valid_sample=True``)
* ``voxel_centers`` -- the sample points the detector takes from
  ``AlignedAnchor3DRangeGenerator`` (core/anchor/anchor_3d_generator.py:285-332)

One HIP launch (``dfm_point_sample_mv_fwd``) replaces the per-(frame, view)
``point_sample`` calls, the stack/sum/count/divide passes and the permute.
"""
import ctypes

import numpy as np
import torch

from . import _capi
from .plane_sweep import _DTYPES, _Workspace, _ptr, _require_gpu, _stream_ptr, _upload


def voxel_centers(voxel_range, n_voxels):
    """(Nz*Ny*Nx, 3) fp32 voxel centres, ordered z-major then y then x, computed
    with the torch ops of AlignedAnchor3DRangeGenerator.anchors_single_range
    (align_corner=False) on the CPU."""
    nx, ny, nz = (int(v) for v in n_voxels)
    rng = torch.tensor(voxel_range, dtype=torch.float32)
    z = torch.linspace(rng[2], rng[5], nz + 1)
    y = torch.linspace(rng[1], rng[4], ny + 1)
    x = torch.linspace(rng[0], rng[3], nx + 1)
    z = z + (z[1] - z[0]) / 2
    y = y + (y[1] - y[0]) / 2
    x = x + (x[1] - x[0]) / 2
    zz, yy, xx = torch.meshgrid(z[:nz], y[:ny], x[:nx], indexing='ij')
    return torch.stack([xx, yy, zz], dim=-1).reshape(-1, 3).contiguous()


_CENTERS = {}


def _device_voxel_centers(voxel_range, n_voxels, device):
    """voxel centres on the device, built and uploaded once per (range, grid, device): the
    reference regenerates the anchor grid every forward (multiview_dfm.py:122-123); 9.5 MB per
    call through a blocking H2D copy is a pipeline stall"""
    key = (tuple(float(v) for v in voxel_range), tuple(int(v) for v in n_voxels), str(device))
    pts = _CENTERS.get(key)
    if pts is None:
        if len(_CENTERS) > 16:
            _CENTERS.clear()
        pts = _CENTERS[key] = voxel_centers(voxel_range, n_voxels).to(device)
    return pts


def _scale_xy(img_scale_factor):
    if torch.is_tensor(img_scale_factor) or isinstance(img_scale_factor, np.ndarray):
        s = np.asarray(torch.as_tensor(img_scale_factor).detach().cpu(), dtype=np.float32).reshape(-1)
        return (float(s[0]), float(s[1])) if s.size >= 2 else (float(s[0]), float(s[0]))
    return float(img_scale_factor), float(img_scale_factor)


def _crop_xy(img_crop_offset):
    if torch.is_tensor(img_crop_offset) or isinstance(img_crop_offset, (np.ndarray, list, tuple)):
        c = np.asarray(torch.as_tensor(img_crop_offset).detach().cpu(), dtype=np.float32).reshape(-1)
        return float(c[0]), float(c[1])
    return float(img_crop_offset), float(img_crop_offset)


def _views_channels_last(x):
    """(B, F*Nv, C, H, W) whose every view is stored (H, W, C) -- what ``.view(B, F*Nv, C, H, W)`` of
    a channels_last image backbone / neck output is -- with whole 16-byte channel blocks"""
    if x.dim() != 5 or x.is_contiguous():
        return False
    B, V, C, H, W = x.shape
    return (x.stride() == (V * H * W * C, H * W * C, 1, W * C, C) and C % (16 // x.element_size()) == 0 and
            x.data_ptr() % 16 == 0)


def _make_desc(feats, npoints, nxyz, num_views, num_frames, scale, crop, flip, pad_shape, aligned,
               aggregate, valid_sample):
    nvf, C, Hf, Wf = feats.shape[-4:]
    assert nvf == num_views * num_frames
    desc = _capi.MvDesc()
    desc.num_views, desc.num_frames, desc.channels = num_views, num_frames, C
    desc.feat_h, desc.feat_w = Hf, Wf
    desc.nx, desc.ny, desc.nz = nxyz if nxyz is not None else (0, 0, 0)
    desc.num_points = npoints
    desc.scale_x, desc.scale_y = scale
    desc.crop_x, desc.crop_y = crop
    desc.flip = 1 if flip else 0
    desc.pad_h, desc.pad_w = float(pad_shape[0]), float(pad_shape[1])
    desc.mode = 1 if aligned else 0
    desc.aggregate = 1 if aggregate == 'concat' else 0
    desc.valid_sample = 1 if valid_sample else 0
    desc.dtype = _DTYPES[feats.dtype]
    return desc


def _reverse_3d_flow(points, coord_type, img_meta):
    """``apply_3d_transformation(points, coord_type, img_meta, reverse=True)``
    (fusion_layers/coord_transform.py:9-95) on the device, operation by operation in the
    reference's order: the recorded ``transformation_3d_flow`` undone back to front -- 'T' adds
    -pcd_trans, 'S' multiplies by 1/pcd_scale_factor, 'R' multiplies by inverse(pcd_rotation) from
    the right (as an explicit fp32 multiply-add chain over x, y, z, not a BLAS call), 'HF' / 'VF'
    negate the BEV axes of the coordinate type (core/points/{lidar,cam,depth}_points.py flip).
    Returns ``points`` itself when there is nothing to undo."""
    flow = (img_meta or {}).get('transformation_3d_flow') or []
    if not flow:
        return points
    axes = {'LIDAR': (1, 0), 'CAMERA': (0, 2), 'DEPTH': (0, 1)}[coord_type]  # (horizontal, vertical)
    p = points.clone()
    for op in flow[::-1]:
        if op == 'T':
            t = torch.as_tensor(img_meta.get('pcd_trans', [0.0, 0.0, 0.0]), dtype=torch.float32)
            p[:, :3] += (-t).to(p.device)
        elif op == 'S':
            p[:, :3] *= 1.0 / img_meta.get('pcd_scale_factor', 1.0)
        elif op == 'R':
            r = torch.as_tensor(img_meta['pcd_rotation'], dtype=torch.float32) if 'pcd_rotation' in img_meta \
                else torch.eye(3)
            ri = torch.inverse(r).to(p.device)
            x, y, z = p[:, 0].clone(), p[:, 1].clone(), p[:, 2].clone()
            for j in range(3):
                p[:, j] = torch.addcmul(torch.addcmul(x * ri[0, j], y, ri[1, j]), z, ri[2, j])
        elif op == 'HF':
            if img_meta.get('pcd_horizontal_flip', False):
                p[:, axes[0]] = -p[:, axes[0]]
        elif op == 'VF':
            if img_meta.get('pcd_vertical_flip', False):
                p[:, axes[1]] = -p[:, axes[1]]
        else:
            raise KeyError(f'unknown transformation_3d_flow entry {op!r}')
    return p


class _MvFn(torch.autograd.Function):
    """feats (B, F*Nv, C, Hf, Wf); one launch per sample (its own image transform in
    ``descs[b]``) straight into one (B, ...) output; proj (B, F*Nv, 16), ori_w (B, F*Nv);
    points (N, 3) shared by the batch or (B, N, 3) per sample (3-D augmentation flows)."""

    @staticmethod
    def forward(ctx, feats, points, proj, ori_w, descs, nxyz, want_valid, channels_last=False):
        lib = _capi.lib()
        device = feats.device
        B = feats.shape[0]
        d0 = descs[0]
        c_out = d0.channels * (d0.num_frames if d0.aggregate else 1)
        if nxyz is not None and channels_last:
            # (B, Nx, Ny, Nz, C) in memory = (B, C, Nx, Ny, Nz) channels_last_3d
            out = torch.empty((B,) + tuple(nxyz) + (c_out,), dtype=feats.dtype, device=device).permute(0, 4, 1, 2, 3)
            for d in descs:
                d.out_channels_last = 1
        elif nxyz is not None:
            out = torch.empty((B, c_out) + tuple(nxyz), dtype=feats.dtype, device=device)
        else:
            out = torch.empty((B, points.shape[-2], c_out), dtype=feats.dtype, device=device)
        valid = torch.empty((B, points.shape[-2]), dtype=torch.uint8, device=device) if want_valid else None
        nbytes = lib.dfm_point_sample_mv_workspace_bytes(ctypes.byref(d0))
        ws = _Workspace.get(device, nbytes)
        with torch.cuda.device(device):
            # the whole batch in one launch when the lanes-per-voxel kernel covers the call (channels-last
            # views and volume, nearest sampling); DFM_ERR_UNSUPPORTED: one launch per sample below
            rc = _capi.DFM_ERR_UNSUPPORTED
            if d0.feats_channels_last and d0.mode == 0 and d0.valid_sample and \
                    (nxyz is None or channels_last) and proj.is_contiguous() and ori_w.is_contiguous():
                arr = (_capi.MvDesc * B)(*descs)
                rc = lib.dfm_point_sample_mv_fwd_batched(
                    arr, B, _ptr(feats), _ptr(points), 1 if points.dim() == 3 else 0, _ptr(proj), _ptr(ori_w),
                    _ptr(out), _ptr(valid) if want_valid else None, _stream_ptr(device))
                if rc not in (0, _capi.DFM_ERR_UNSUPPORTED):
                    _capi.check(rc)
            for b in range(B if rc != 0 else 0):
                _capi.check(
                    lib.dfm_point_sample_mv_fwd(ctypes.byref(descs[b]), _ptr(feats[b]),
                                                _ptr(points[b] if points.dim() == 3 else points),
                                                _ptr(proj[b]), _ptr(ori_w[b]), _ptr(out[b]),
                                                _ptr(valid[b]) if want_valid else None, _ptr(ws),
                                                nbytes, _stream_ptr(device)))
        ctx.descs = descs  # (the backward reads a (C, N) gradient: grad_out.contiguous() below)
        ctx.meta = (feats.shape, feats.dtype)
        ctx.save_for_backward(points, proj, ori_w)
        if want_valid:
            ctx.mark_non_differentiable(valid)
        return out, valid

    @staticmethod
    def backward(ctx, grad_out, _grad_valid):
        points, proj, ori_w = ctx.saved_tensors
        lib = _capi.lib()
        shape, dtype = ctx.meta
        device = grad_out.device
        go = grad_out.contiguous().to(dtype)
        gf = torch.zeros(shape, dtype=torch.float32, device=device)
        nbytes = lib.dfm_point_sample_mv_bwd_workspace_bytes(ctypes.byref(ctx.descs[0]))
        ws = _Workspace.get(device, nbytes)
        with torch.cuda.device(device):
            for b in range(shape[0]):
                _capi.check(
                    lib.dfm_point_sample_mv_bwd(ctypes.byref(ctx.descs[b]), _ptr(go[b]),
                                                _ptr(points[b] if points.dim() == 3 else points),
                                                _ptr(proj[b]), _ptr(ori_w[b]), _ptr(gf[b]), _ptr(ws),
                                                nbytes, _stream_ptr(device)))
        return gf.to(dtype), None, None, None, None, None, None, None


def point_sample(img_meta,
                 img_features,
                 points,
                 proj_mat,
                 coord_type,
                 img_scale_factor,
                 img_crop_offset,
                 img_flip,
                 img_pad_shape,
                 img_shape,
                 aligned=True,
                 padding_mode='zeros',
                 align_corners=True,
                 valid_flag=False):
    """Drop-in for the reference ``point_sample``: (N, C) features of one view
    [+ (N,) bool validity when ``valid_flag``]."""
    if padding_mode != 'zeros' or not align_corners:
        raise NotImplementedError('only padding_mode="zeros", align_corners=True (what DfM uses)')
    _require_gpu(img_features, 'img_features')
    assert img_features.dim() == 4 and img_features.shape[0] == 1
    device = img_features.device
    feats = img_features.contiguous()
    pts = torch.as_tensor(points, dtype=torch.float32).to(device).contiguous()
    pts = _reverse_3d_flow(pts, coord_type, img_meta).contiguous()  # point_fusion.py:57-58
    proj = torch.as_tensor(proj_mat, dtype=torch.float32).reshape(1, 16).to(device).contiguous()
    ori_w = torch.tensor([float(img_shape[1])], dtype=torch.float32, device=device)
    desc = _make_desc(feats, pts.shape[0], None, 1, 1, _scale_xy(img_scale_factor),
                      _crop_xy(img_crop_offset), img_flip, img_pad_shape, aligned, 'mean', valid_flag)
    out, valid = _MvFn.apply(feats[None], pts, proj[None], ori_w[None], [desc], None, valid_flag)
    if valid_flag:
        return out[0], valid[0].bool()
    return out[0]


def mv_feature_transformation(batch_feats, img_metas, num_views, num_frames, voxel_range, n_voxels,
                              temporal_aggregate='mean', points=None, valid_sample=True,
                              memory_format=torch.contiguous_format):
    """(B, F*Nv, C, Hf, Wf) view features -> (B, C or C*F, Nx, Ny, Nz) voxel volume,
    the tensor the reference hands to ``neck_3d`` (multiview_dfm.py:206-209).
    ``memory_format=torch.channels_last_3d`` (extension): the same tensor stored (B, Nx, Ny, Nz, C),
    what the NDHWC / MFMA neck convolutions read -- written directly, no conversion copy."""
    _require_gpu(batch_feats, 'batch_feats')
    device = batch_feats.device
    if points is None:
        points = _device_voxel_centers(voxel_range, n_voxels, device)
    points = torch.as_tensor(points, dtype=torch.float32).to(device).contiguous()
    nxyz = tuple(int(v) for v in n_voxels)
    nvf = num_views * num_frames
    # channels_last view features are sampled in place (they are the kernel's pixel-major layout)
    feats_cl = _views_channels_last(batch_feats)
    feats = batch_feats if feats_cl else batch_feats.contiguous()
    descs, proj, ori_w = [], [], []
    for img_meta in img_metas:
        if 'scale_factor' in img_meta:
            sf = img_meta['scale_factor']
            scale = _scale_xy(sf[:2] if isinstance(sf, np.ndarray) and len(sf) >= 2 else sf)
        else:
            scale = (1.0, 1.0)
        flip = img_meta.get('flip', False)
        crop = _crop_xy(img_meta['img_crop_offset']) if 'img_crop_offset' in img_meta else (0.0, 0.0)
        l2i = img_meta['ori_lidar2img']
        if torch.is_tensor(l2i):   # staged by data_geometry.stage_geometry: read where it lies
            proj.append(l2i[:nvf].to(torch.float32).reshape(nvf, 16))
        else:
            proj.append(np.asarray(l2i[:nvf], dtype=np.float32).reshape(nvf, 16))
        ori_w.append([float(img_meta['img_shape'][i][1]) for i in range(nvf)])
        descs.append(_make_desc(feats, points.shape[0], nxyz, num_views, num_frames, scale, crop,
                                flip, img_meta['input_shape'], False, temporal_aggregate,
                                valid_sample))
        descs[-1].feats_channels_last = 1 if feats_cl else 0
    # one upload for the whole batch's matrices
    if all(torch.is_tensor(p_) for p_ in proj):
        proj = torch.stack([p_.to(device) for p_ in proj]).contiguous()   # device tensors: no host round trip
    else:
        proj = _upload(torch.from_numpy(np.stack([p_.detach().cpu().numpy() if torch.is_tensor(p_) else p_
                                                  for p_ in proj])), device)
    ori_w = _upload(torch.tensor(ori_w, dtype=torch.float32), device)
    if any(m.get('transformation_3d_flow') for m in img_metas):
        # point_sample undoes each sample's own 3-D augmentation first (point_fusion.py:57-58)
        points = torch.stack([_reverse_3d_flow(points, 'LIDAR', m) for m in img_metas]).contiguous()
    out, _ = _MvFn.apply(feats, points, proj, ori_w, descs, nxyz, False,
                         memory_format == torch.channels_last_3d)
    return out


def voxel_sample(voxel_features,
                 voxel_range,
                 voxel_size,
                 depth_samples,
                 proj_mat,
                 downsample_factor,
                 img_scale_factor,
                 img_crop_offset,
                 img_flip,
                 img_pad_shape,
                 img_shape,
                 aligned=True,
                 padding_mode='zeros',
                 align_corners=True,
                 proj_inv=None):
    """Drop-in for the reference ``voxel_sample`` (point_fusion.py:324-410):
    (1, C, Nx, Ny, Nz) voxel features -> (1, C, D, H_out, W_out) frustum features.

    ``proj_inv`` (extension): a precomputed fp32 inverse of ``proj_mat``.  By default
    it is taken with torch.inverse on the host like the reference does
    (utils.py:241); for a general 4x4 LAPACK's result can differ in the last bits
    between CPU models, so bit-exact replays of a fixture pass the stored inverse."""
    if padding_mode != 'zeros' or not align_corners:
        raise NotImplementedError('only padding_mode="zeros", align_corners=True')
    _require_gpu(voxel_features, 'voxel_features')
    assert voxel_features.dim() == 5 and voxel_features.shape[0] == 1
    lib = _capi.lib()
    device = voxel_features.device
    vox = voxel_features.contiguous()
    depths = torch.as_tensor(depth_samples, dtype=torch.float32)[::downsample_factor]
    depths = depths.to(device).contiguous()
    desc = _capi.VsDesc()
    desc.channels, desc.nx, desc.ny, desc.nz = vox.shape[1:]
    desc.num_depths = depths.numel()
    desc.h_out = round(img_pad_shape[0] / downsample_factor)
    desc.w_out = round(img_pad_shape[1] / downsample_factor)
    desc.downsample_factor = float(downsample_factor)
    desc.scale_x, desc.scale_y = _scale_xy(img_scale_factor)
    desc.crop_x, desc.crop_y = _crop_xy(img_crop_offset)
    desc.flip, desc.ori_w = (1 if img_flip else 0), float(img_shape[1])
    for i, v in enumerate(np.asarray(voxel_range, dtype=np.float32).reshape(6)):
        desc.voxel_range[i] = float(v)
    for i, v in enumerate(np.asarray(voxel_size, dtype=np.float32).reshape(3)):
        desc.voxel_size[i] = float(v)
    inv = torch.as_tensor(proj_inv, dtype=torch.float32) if proj_inv is not None else \
        torch.inverse(torch.as_tensor(proj_mat, dtype=torch.float32).detach().cpu())  # utils.py:241
    for i, v in enumerate(inv.reshape(16).tolist()):
        desc.proj_inv[i] = v
    desc.mode = 1 if aligned else 0
    desc.dtype = _DTYPES[vox.dtype]
    return _VoxelSampleFn.apply(vox, depths, desc)


class _VoxelSampleFn(torch.autograd.Function):
    """dfm_voxel_sample_fwd / _bwd (gradient w.r.t. the voxel features)"""

    @staticmethod
    def forward(ctx, vox, depths, desc):
        lib = _capi.lib()
        device = vox.device
        out = torch.empty((1, desc.channels, desc.num_depths, desc.h_out, desc.w_out), dtype=vox.dtype,
                          device=device)
        with torch.cuda.device(device):
            _capi.check(lib.dfm_voxel_sample_fwd(ctypes.byref(desc), _ptr(vox), _ptr(depths), _ptr(out),
                                                 _stream_ptr(device)))
        ctx.desc, ctx.meta = desc, (vox.shape, vox.dtype)
        ctx.save_for_backward(depths)
        return out

    @staticmethod
    def backward(ctx, gout):
        (depths,) = ctx.saved_tensors
        shape, dtype = ctx.meta
        lib = _capi.lib()
        device = gout.device
        go = gout.contiguous().to(dtype)
        gv = torch.zeros(shape, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            _capi.check(lib.dfm_voxel_sample_bwd(ctypes.byref(ctx.desc), _ptr(go), _ptr(depths), _ptr(gv),
                                                 _stream_ptr(device)))
        return gv.to(dtype), None, None
