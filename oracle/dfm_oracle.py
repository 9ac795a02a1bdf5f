"""numpy front-end of libdfm_oracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The C file restates the reference's fp32 algorithm
(oracle/dfm_oracle.c cites file:line); this wrapper only marshals arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libdfm_oracle.so')


def build(force=False):
    """gcc -O2 -ffp-contract=off ... (oracle/Makefile)."""
    src = os.path.join(_HERE, 'dfm_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B', 'libdfm_oracle.so'])
    return _SO


class SweepParams(ctypes.Structure):
    """struct dfm_oracle_sweep_params"""
    _fields_ = [('D', ctypes.c_int32), ('h_out', ctypes.c_int32), ('w_out', ctypes.c_int32),
                ('h_in', ctypes.c_int32), ('w_in', ctypes.c_int32), ('fsf', ctypes.c_float),
                ('csf', ctypes.c_float), ('scale', ctypes.c_float), ('crop_x', ctypes.c_float),
                ('crop_y', ctypes.c_float), ('flip', ctypes.c_int32), ('org_w', ctypes.c_float),
                ('P', ctypes.c_float * 16), ('Pinv', ctypes.c_float * 16),
                ('T', ctypes.c_float * 16)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.dfm_oracle_version.restype = ctypes.c_int
    return _lib


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def sweep_params(h_in, w_in, num_depths, fsf, csf, P, Pinv, T, img_shape, flip, crop, scale):
    p = SweepParams()
    p.D, p.h_in, p.w_in = int(num_depths), int(h_in), int(w_in)
    p.h_out, p.w_out = round(h_in / csf), round(w_in / csf)  # dfm_backbone.py:242-243
    p.fsf, p.csf, p.scale = float(fsf), float(csf), float(scale)
    p.crop_x, p.crop_y = float(crop[0]), float(crop[1])
    p.flip, p.org_w = int(bool(flip)), float(img_shape[1])
    for name, m in (('P', P), ('Pinv', Pinv), ('T', T)):
        arr = getattr(p, name)
        for i, v in enumerate(_f32(m).reshape(16)):
            arr[i] = float(v)
    return p


def plane_sweep_grid(p, depths):
    n = p.D * p.h_out * p.w_out
    cg = np.empty((n, 2), np.float32)
    pg = np.empty((n, 2), np.float32)
    depths = _f32(depths)
    lib().dfm_oracle_plane_sweep_grid(ctypes.byref(p), _vp(depths), _vp(cg), _vp(pg))
    return cg, pg


def build_dfm_cost(cur, prev, depths, fsf, csf, P, Pinv, T, img_shape, flip=False, crop=(0, 0),
                   scale=1.0):
    """cur/prev (B,C,H,W) fp32; P/Pinv/T (B,4,4); -> (B,2C,D,h_out,w_out).
    The batch is looped with single-sample reference semantics."""
    cur, prev, depths = _f32(cur), _f32(prev), _f32(depths).reshape(-1)
    B, C, H, W = cur.shape
    outs = []
    for b in range(B):
        p = sweep_params(H, W, depths.size, fsf, csf, P[b], Pinv[b], T[b], img_shape, flip, crop,
                         scale)
        out = np.empty((2 * C, p.D, p.h_out, p.w_out), np.float32)
        lib().dfm_oracle_build_dfm_cost(ctypes.byref(p), _vp(depths), _vp(cur[b]), _vp(prev[b]),
                                        ctypes.c_int(C), _vp(out))
        outs.append(out)
    return np.stack(outs)


def grid_sample2d(inp, grid, mode='bilinear'):
    """inp (C,H,W), grid (N,2) normalised -> (C,N); zeros, align_corners=True."""
    inp, grid = _f32(inp), _f32(grid)
    C, H, W = inp.shape
    out = np.empty((C, grid.shape[0]), np.float32)
    lib().dfm_oracle_grid_sample2d(_vp(inp), C, H, W, _vp(grid), ctypes.c_int64(grid.shape[0]),
                                   0 if mode == 'bilinear' else 1, _vp(out))
    return out


class PSParams(ctypes.Structure):
    """struct dfm_oracle_ps_params"""
    _fields_ = [('C', ctypes.c_int32), ('Hf', ctypes.c_int32), ('Wf', ctypes.c_int32),
                ('scale_x', ctypes.c_float), ('scale_y', ctypes.c_float),
                ('crop_x', ctypes.c_float), ('crop_y', ctypes.c_float), ('flip', ctypes.c_int32),
                ('ori_w', ctypes.c_float), ('pad_h', ctypes.c_float), ('pad_w', ctypes.c_float),
                ('mode', ctypes.c_int32), ('proj', ctypes.c_float * 16)]


def point_sample(feat, points, proj, scale=(1.0, 1.0), crop=(0.0, 0.0), flip=False, ori_w=0.0,
                 pad_shape=(1, 1), aligned=True, valid_flag=False):
    """feat (C,Hf,Wf), points (N,3) -> (N,C) [, valid (N,) bool]; reference
    point_fusion.py:14-106 with identity 3-D transformation."""
    feat, points = _f32(feat), _f32(points)
    p = PSParams()
    p.C, p.Hf, p.Wf = feat.shape
    p.scale_x, p.scale_y = float(scale[0]), float(scale[1])
    p.crop_x, p.crop_y = float(crop[0]), float(crop[1])
    p.flip, p.ori_w = int(bool(flip)), float(ori_w)
    p.pad_h, p.pad_w = float(pad_shape[0]), float(pad_shape[1])
    p.mode = 1 if aligned else 0
    for i, v in enumerate(_f32(proj).reshape(16)):
        p.proj[i] = float(v)
    n = points.shape[0]
    out = np.empty((n, p.C), np.float32)
    valid = np.empty(n, np.uint8) if valid_flag else None
    lib().dfm_oracle_point_sample(ctypes.byref(p), _vp(feat), _vp(points), ctypes.c_int64(n),
                                  _vp(out), _vp(valid) if valid_flag else None)
    return (out, valid.astype(bool)) if valid_flag else out


def mv_feature_transformation(feats, points, lidar2img, n_voxels, num_views, num_frames,
                              input_shape, img_shape, scale=None, flip=False, crop=None,
                              aggregate='mean'):
    """One sample of MultiViewDfM.feature_transformation (detectors/multiview_dfm.py:119-208,
    valid_sample=True): feats (F*Nv, C, Hf, Wf) -> (C or C*F, Nx, Ny, Nz)."""
    sc = (1.0, 1.0) if scale is None or len(scale) == 0 else (scale[0], scale[1])
    cr = (0.0, 0.0) if crop is None or len(crop) == 0 else (crop[0], crop[1])
    frame_vol, frame_valid = [], []
    for f in range(num_frames):
        vol, cnt = None, None
        for v in range(num_views):
            i = f * num_views + v
            o, ok = point_sample(feats[i], points, lidar2img[i], sc, cr, flip, img_shape[1],
                                 input_shape, aligned=False, valid_flag=True)
            vol = o.copy() if vol is None else vol + o          # stack().sum(0): in view order
            cnt = ok.astype(np.int64) if cnt is None else cnt + ok
        vol[cnt == 0] = 0
        frame_vol.append(vol)
        frame_valid.append(cnt)
    if aggregate == 'mean':
        vol, cnt = frame_vol[0], frame_valid[0]
        for f in range(1, num_frames):
            vol, cnt = vol + frame_vol[f], cnt + frame_valid[f]
        vol[cnt == 0] = 0
        vol = vol / np.maximum(cnt, 1)[:, None].astype(np.float32)
    else:
        vol = np.concatenate([fv / np.maximum(fc, 1)[:, None].astype(np.float32)
                              for fv, fc in zip(frame_vol, frame_valid)], axis=1)
    nx, ny, nz = (int(v) for v in n_voxels)
    return np.ascontiguousarray(vol.astype(np.float32).reshape(nz, ny, nx, -1).transpose(3, 2, 1, 0))


class F2VParams(ctypes.Structure):
    """struct dfm_oracle_f2v_params"""
    _fields_ = [('C', ctypes.c_int32), ('D', ctypes.c_int32), ('H', ctypes.c_int32),
                ('W', ctypes.c_int32), ('Ds', ctypes.c_int32), ('Hs', ctypes.c_int32),
                ('Ws', ctypes.c_int32), ('Csem', ctypes.c_int32), ('Hsem', ctypes.c_int32),
                ('Wsem', ctypes.c_int32), ('Nz', ctypes.c_int32), ('Ny', ctypes.c_int32),
                ('Nx', ctypes.c_int32), ('pad_h', ctypes.c_float), ('pad_w', ctypes.c_float),
                ('depth_min', ctypes.c_float), ('depth_span', ctypes.c_float),
                ('P', ctypes.c_float * 12), ('stereo_atten', ctypes.c_int32),
                ('no_sem_atten', ctypes.c_int32)]


def frustum_to_voxel(stereo, softmax, sem, coordinates_3d, cam2img, pad_shape, depth_min,
                     depth_max, sem_atten_feat=True, stereo_atten_feat=False):
    """Sampling stage of FrustumToVoxel.forward (feature_transformation.py:82-158):
    stereo (B,C,D,H,W), softmax (B,1,Ds,Hs,Ws), sem (B,Cs,H,W) or None, cam2img (B,4,4)
    -> (B, C+Cs, Nz, Ny, Nx).  pad_shape of sample 0 is used for all (reference :101)."""
    stereo, softmax = _f32(stereo), _f32(softmax)
    coords = _f32(coordinates_3d)
    B, C, D, H, W = stereo.shape
    nz, ny, nx = coords.shape[:3]
    cs = 0 if sem is None else sem.shape[1]
    outs = []
    for b in range(B):
        p = F2VParams()
        p.C, p.D, p.H, p.W = C, D, H, W
        p.Ds, p.Hs, p.Ws = softmax.shape[2:]
        p.Csem = cs
        p.stereo_atten, p.no_sem_atten = int(bool(stereo_atten_feat)), int(not sem_atten_feat)
        if cs:
            p.Hsem, p.Wsem = sem.shape[2:]
        p.Nz, p.Ny, p.Nx = nz, ny, nx
        p.pad_h, p.pad_w = float(pad_shape[0]), float(pad_shape[1])
        p.depth_min = float(depth_min)
        p.depth_span = float(depth_max) - float(depth_min)  # python double, then fp32
        for i, v in enumerate(_f32(cam2img[b])[:3].reshape(12)):
            p.P[i] = float(v)
        out = np.empty((C + cs, nz, ny, nx), np.float32)
        s = _f32(sem[b]) if cs else None
        lib().dfm_oracle_frustum_to_voxel(ctypes.byref(p), _vp(stereo[b]), _vp(softmax[b]),
                                          _vp(s) if cs else None, _vp(coords), _vp(out))
        outs.append(out)
    return np.stack(outs)


def depth_head(stereo_features, depth_samples, scale=4):
    """DepthHead.forward with with_convs=False (depth_head.py:205-210):
    (B,1,D,H,W) -> depth_volumes, softmax (B,1,sD,sH,sW), depth_preds (B,1,sH,sW)."""
    x = _f32(stereo_features)
    ds = _f32(depth_samples).reshape(-1)
    B, one, D, H, W = x.shape
    assert one == 1 and ds.size == scale * D
    vol = np.empty((B, 1, scale * D, scale * H, scale * W), np.float32)
    soft = np.empty_like(vol)
    pred = np.empty((B, 1, scale * H, scale * W), np.float32)
    for b in range(B):
        lib().dfm_oracle_depth_head(_vp(x[b, 0]), D, H, W, int(scale), _vp(ds), _vp(vol[b, 0]),
                                    _vp(soft[b, 0]), _vp(pred[b, 0]))
    return vol, soft, pred


class VSParams(ctypes.Structure):
    """struct dfm_oracle_vs_params"""
    _fields_ = [('C', ctypes.c_int32), ('Nx', ctypes.c_int32), ('Ny', ctypes.c_int32),
                ('Nz', ctypes.c_int32), ('D', ctypes.c_int32), ('h_out', ctypes.c_int32),
                ('w_out', ctypes.c_int32), ('ds', ctypes.c_float), ('scale_x', ctypes.c_float),
                ('scale_y', ctypes.c_float), ('crop_x', ctypes.c_float), ('crop_y', ctypes.c_float),
                ('flip', ctypes.c_int32), ('ori_w', ctypes.c_float), ('range', ctypes.c_float * 6),
                ('vsize', ctypes.c_float * 3), ('mode', ctypes.c_int32),
                ('Minv', ctypes.c_float * 16)]


def voxel_sample(voxel_features, voxel_range, voxel_size, depth_samples, proj_inv,
                 downsample_factor, scale=(1.0, 1.0), crop=(0.0, 0.0), flip=False,
                 img_pad_shape=(1, 1), img_shape=(1, 1), aligned=True):
    """(1,C,Nx,Ny,Nz) -> (1,C,D,h_out,w_out); reference point_fusion.py:324-410.
    proj_inv = torch.inverse(proj_mat) in fp32 (utils.py:241)."""
    vox = _f32(voxel_features)[0]
    p = VSParams()
    p.C, p.Nx, p.Ny, p.Nz = vox.shape
    depths = _f32(depth_samples)[::int(downsample_factor)].copy()
    p.D = depths.size
    p.h_out = round(img_pad_shape[0] / downsample_factor)
    p.w_out = round(img_pad_shape[1] / downsample_factor)
    p.ds = float(downsample_factor)
    p.scale_x, p.scale_y = float(scale[0]), float(scale[1])
    p.crop_x, p.crop_y = float(crop[0]), float(crop[1])
    p.flip, p.ori_w = int(bool(flip)), float(img_shape[1])
    for i, v in enumerate(_f32(voxel_range)):
        p.range[i] = float(v)
    for i, v in enumerate(_f32(voxel_size)):
        p.vsize[i] = float(v)
    p.mode = 1 if aligned else 0
    for i, v in enumerate(_f32(proj_inv).reshape(16)):
        p.Minv[i] = float(v)
    out = np.empty((p.C, p.D, p.h_out, p.w_out), np.float32)
    lib().dfm_oracle_voxel_sample(ctypes.byref(p), _vp(vox), _vp(depths), _vp(out))
    return out[None]


def bf16_round(a):
    """fp32 -> bf16 (round-to-nearest-even) -> fp32, numpy."""
    u = _f32(a).view(np.uint32)
    nan = (u & 0x7fffffff) > 0x7f800000
    r = (u + (0x7fff + ((u >> 16) & 1))) & 0xffff0000
    r = np.where(nan, np.uint32(0x7fc00000), r).astype(np.uint32)
    return r.view(np.float32)
