"""Init-time geometry tables of the DfM detector, restated with the same
Python / torch ops so the kernels receive bit-identical inputs:

* ``prepare_depth``           mmdet3d/models/detectors/dfm.py:152-172
* ``prepare_coordinates_3d``  mmdet3d/models/detectors/dfm.py:174-211
"""
import numpy as np
import torch


def stack_meta(img_metas, key, dtype=torch.float32):
    """``[m[key] for m in img_metas]`` as ONE tensor of ``dtype``.  Matrices that a collate step put on
    the device (``data_geometry.stage_geometry``) are stacked where they lie -- no host round trip, no
    sync; lists / numpy arrays give a host tensor like the reference's ``torch.tensor([...])``
    (dfm.py:288-291, dfm_backbone.py:236).  A batch that mixes the two is gathered on the device of its
    tensors."""
    vals = [m[key] for m in img_metas]
    if any(torch.is_tensor(v) for v in vals):
        dev = next(v.device for v in vals if torch.is_tensor(v))
        return torch.stack([(v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v, dtype=np.float64))).to(
            device=dev, dtype=dtype) for v in vals])
    return torch.as_tensor(np.asarray(vals), dtype=dtype)


def prepare_depth(depth_cfg, downsampled_depth_offset=0.5):
    """-> (downsampled_depth (num_bins // ds,), depth (num_bins,)) fp32 plane depths /
    bin centres: d_i = (i + offset) * ds * interval + depth_min."""
    ds = depth_cfg['downsample_factor']
    assert depth_cfg['depth_min'] >= 0 and depth_cfg['depth_max'] > depth_cfg['depth_min']
    interval = (depth_cfg['depth_max'] - depth_cfg['depth_min']) / depth_cfg['num_bins']
    n_low = depth_cfg['num_bins'] // ds
    low = torch.zeros(n_low, dtype=torch.float32)
    for i in range(n_low):
        low[i] = (i + downsampled_depth_offset) * ds * interval + depth_cfg['depth_min']
    full = torch.zeros(depth_cfg['num_bins'], dtype=torch.float32)
    for i in range(depth_cfg['num_bins']):
        full[i] = (i + 0.5) * interval + depth_cfg['depth_min']
    return low, full


def prepare_coordinates_3d(voxel_cfg, sample_rate=(1, 1, 1)):
    """-> (Nz, Ny, Nx, 3) fp32 voxel centres (x, y, z) in pseudo-LiDAR coordinates."""
    pcr, vs = voxel_cfg['point_cloud_range'], list(voxel_cfg['voxel_size'])
    grid = (np.array(pcr[3:6], dtype=np.float32) - np.array(pcr[0:3], dtype=np.float32)) / np.array(vs)
    gx, gy, gz = np.round(grid).astype(np.int64).tolist()
    vs = [v / r for v, r in zip(vs, sample_rate)]
    gx, gy, gz = gx * sample_rate[0], gy * sample_rate[1], gz * sample_rate[2]
    zs = torch.linspace(pcr[2] + vs[2] / 2., pcr[5] - vs[2] / 2., gz, dtype=torch.float32)
    ys = torch.linspace(pcr[1] + vs[1] / 2., pcr[4] - vs[1] / 2., gy, dtype=torch.float32)
    xs = torch.linspace(pcr[0] + vs[0] / 2., pcr[3] - vs[0] / 2., gx, dtype=torch.float32)
    zs, ys, xs = torch.meshgrid(zs, ys, xs, indexing='ij')
    return torch.stack([xs, ys, zs], dim=-1).float()
