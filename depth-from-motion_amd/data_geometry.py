"""Data-side half of the path's geometry: the pose bookkeeping the
reference's loading pipelines do per sample on the host, restated with the same numpy operations
(bit-identical matrices), and a collate-time stager that hands the model DEVICE-READY tensors so a
training / inference step uploads no matrices.

  select_ref_frames        LoadMultiViewImageFromFiles.__call__   datasets/pipelines/loading.py:67-96
  fold_ref_frame_matrices  ... lidar2img / lidar2cam of previous frames become
                           [cur lidar] -> [prev img / cam]        loading.py:122-142
  video_cur2prevs          VideoPipeline.__call__                 loading.py:474-541
  stage_geometry           what DfM.extract_feat (detectors/dfm.py:286-293) and
                           MultiViewDfM.feature_transformation (multiview_dfm.py:161-162) otherwise
                           convert and upload every step

The loaders themselves (image decoding, augmentation) stay the reference's: this module is what a
maintainer calls from them -- or right after them -- see INTEGRATION.md.
"""
import numpy as np
import torch


def _pad4(m):
    """np.eye(4) with ``m`` in its top-left corner (loading.py:126-136, 475-478, 527-530)"""
    m = np.asarray(m)
    out = np.eye(4)
    out[:m.shape[0], :m.shape[1]] = m
    return out


def select_ref_frames(num_prev_frames, num_ref_frames, test_mode, rng=np.random):
    """Indices (into the per-frame lists, 0 = current frame) of the frames a multi-sweep sample uses:
    [0] + ``num_ref_frames`` previous ones (loading.py:67-96; the info lists run from the latest to
    earlier frames).  ``rng``: the module / RandomState whose ``choice`` the training branch draws from."""
    if num_ref_frames <= 0:
        return np.array([0], dtype=np.int64)
    n = int(num_prev_frames)
    if n == 0:  # no previous frame: copy the current one
        choices = rng.choice(1, num_ref_frames, replace=True)
    elif n >= num_ref_frames:
        choices = np.arange(n - num_ref_frames, n) + 1 if test_mode else \
            rng.choice(n, num_ref_frames, replace=False) + 1
    else:
        if test_mode:
            choices = np.concatenate([np.arange(n) + 1, rng.choice(n, num_ref_frames - n, replace=True) + 1])
        else:
            choices = rng.choice(n, num_ref_frames, replace=True) + 1
    return np.concatenate([np.array([0], dtype=np.int64), choices])


def fold_ref_frame_matrices(mats, ego2global, num_views):
    """``mats``: per-(frame, view) 4x4 ``lidar2img`` (or ``lidar2cam``) of the SELECTED frames, frame 0
    the current one; ``ego2global``: one matrix per selected frame.  Returns the list with every
    previous frame's matrices right-multiplied by ``inv(prev_ego2global) @ cur_ego2global`` -- the
    current LiDAR frame then projects straight into the previous images (loading.py:122-142), which
    is what ``point_sample`` receives as ``proj_mat``."""
    mats = [np.asarray(m) for m in mats]
    num_frames = len(mats) // num_views
    assert len(ego2global) >= num_frames
    out = list(mats)
    cur = _pad4(ego2global[0])
    for f in range(1, num_frames):
        cur2prev = np.linalg.inv(_pad4(ego2global[f])).dot(cur)
        for i in range(f * num_views, (f + 1) * num_views):
            out[i] = mats[i].dot(cur2prev)
    return out


def video_cur2prevs(cur_cam2global, prev_cam2globals):
    """(N-1, 4, 4) ``cur2prevs`` of a video sample: ``inv(prev_cam2global) @ cur_cam2global`` per
    reference frame (loading.py:531-541).  (0, 4, 4) without reference frames -- the reference's
    ``np.stack`` of an empty list raises there."""
    cur = _pad4(cur_cam2global)
    mats = [np.linalg.inv(_pad4(p)).dot(cur) for p in prev_cam2globals]
    return np.stack(mats, axis=0) if mats else np.zeros((0, 4, 4))


def _pinned_upload(arr, device):
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))
    if device.type == 'cuda':
        return t.pin_memory().to(device, non_blocking=True)
    return t


def stage_geometry(img_metas, device):
    """Collate-time: every matrix of a batch the path reads -- ``ori_cam2img`` and ``cur2prevs`` (the
    plane sweep), ``ori_lidar2img`` (the multi-view lifting), ``cam2img`` (FrustumToVoxel) -- goes to
    ``device`` in ONE pinned, non-blocking upload, and each ``img_meta`` gets views of that buffer
    under the same keys (fp32 tensors).  The path's functions recognise device tensors and neither
    convert nor upload anything in the step (``camera_matrices``: device-side pad + inverse;
    ``mv_feature_transformation``: the projection matrices are read where they lie).  Returns the
    number of bytes staged.  Matrices that are already device tensors are left alone."""
    device = torch.device(device)
    keys = ('ori_cam2img', 'cur2prevs', 'ori_lidar2img', 'cam2img')
    parts, where = [], []
    for bi, meta in enumerate(img_metas):
        for k in keys:
            v = meta.get(k)
            if v is None or (torch.is_tensor(v) and v.device == device):
                continue
            a = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(
                [np.asarray(m, dtype=np.float64) for m in v] if isinstance(v, (list, tuple)) and len(v) and
                np.ndim(v[0]) == 2 else v, dtype=np.float64)
            a = a.astype(np.float32)
            where.append((bi, k, a.shape, sum(p.size for p in parts)))
            parts.append(a.reshape(-1))
    if not parts:
        return 0
    flat = _pinned_upload(np.concatenate(parts), device)
    for bi, k, shape, off in where:
        n = int(np.prod(shape))
        img_metas[bi][k] = flat[off:off + n].view(*shape)
    return flat.numel() * 4
