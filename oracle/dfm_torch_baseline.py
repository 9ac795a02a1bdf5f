"""TEST / BENCH INFRASTRUCTURE -- never imported by the product path.

The reference's ``build_dfm_cost`` (mmdet3d/models/backbones/dfm_backbone.py:217-314) re-stated as the
SEQUENCE OF TORCH LIBRARY CALLS it issues, so that ``bench.py``'s ``cpu_baseline`` leg can time the
reference's whole compute -- lattice construction, un-projection / re-projection, the two ``grid_sample``
calls and the channel ``cat`` -- through the reference's own library (PyTorch-CPU) on the GPU box's host
cores, where ``/root/reference`` does not exist.  It follows the reference op for op (same tensor shapes,
same in-place updates, same matmuls), because the baseline is about what that op sequence costs; the
C oracle (dfm_oracle.c) stays the parity checker.

Pinned: ``tests/test_oracle_golden.py`` compares its output with the fixtures the reference's own code
generated (tests/golden/plane_sweep_*.npz: ``ref_out``, ``ref_cur_grid``, ``ref_prev_grid``) bit for bit.

  lattice + meshgrid + stack + repeat            dfm_backbone.py:247-255
  undo augmentation (crop, scale, flip)          :257-263
  points_img2cam / points_cam2img                core/bbox/structures/utils.py:175-248
  prev = homo @ cur2prev^T                       :266-271
  redo augmentation, / fsf, normalise            :277-294
  grid_sample x 2, view, cat                     :296-313
"""
import torch
import torch.nn.functional as F


def _pad_to_4x4(m):
    # utils.py:196-200 / 232-236: a 3x3 / 3x4 / 4x4 projection padded into an identity 4x4
    out = torch.eye(4, dtype=m.dtype, device=m.device)
    out[:m.shape[0], :m.shape[1]] = m
    return out


def cam2img(points, proj):
    """utils.py:175-213: [p, 1] @ P^T, divide by the third component (no clamp)"""
    homo = torch.cat([points, points.new_ones(points.shape[0], 1)], dim=-1)
    res = homo @ _pad_to_4x4(proj).T
    return res[..., :2] / res[..., 2:3]


def img2cam(points, proj):
    """utils.py:216-248: [u d, v d, d, 1] @ inverse(P_pad)^T"""
    depth = points[:, 2:3]
    unnorm = torch.cat([points[:, :2] * depth, depth], dim=1)
    inv = torch.inverse(_pad_to_4x4(proj)).transpose(0, 1)
    homo = torch.cat([unnorm, unnorm.new_ones(unnorm.shape[0], 1)], dim=1)
    return (homo @ inv)[:, :3]


def sampling_grids(h_in, w_in, depths, fsf, csf, cam2imgs, cur2prevs, img_shape, flip=False,
                   crop=(0, 0), scale=1.0, batch=1):
    """(cur_grid, prev_grid), each (batch, 1, D * h_out * w_out, 2) normalised for ``grid_sample`` --
    with the reference's batch semantics (the loop keeps the LAST sample's grids, SURVEY.md appendix A.1:
    call it with batch = 1)."""
    crop_t = torch.tensor(crop)
    h_out, w_out = round(h_in / csf), round(w_in / csf)
    ws = torch.linspace(0, w_out - 1, w_out) * fsf * csf
    hs = torch.linspace(0, h_out - 1, h_out) * fsf * csf
    dd, yy, xx = torch.meshgrid(depths, hs, ws, indexing='ij')
    lattice = torch.stack([xx, yy, dd], dim=-1)[None].repeat(batch, 1, 1, 1, 1)
    for b in range(batch):
        lattice[..., :2] += crop_t
        lattice[..., :2] /= scale
        if flip:
            lattice[..., 0] = img_shape[1] - lattice[..., 0]
        cam = img2cam(lattice[b].view(-1, 3), cam2imgs[b][:3])
        homo = torch.cat([cam, cam.new_ones(cam.shape[0], 1)], dim=1)
        cur = cam2img(cam, cam2imgs[b])[:, :2]
        prev = cam2img((homo @ cur2prevs[b].transpose(0, 1))[:, :3], cam2imgs[b])[:, :2]
    cur, prev = cur.view(batch, 1, -1, 2), prev.view(batch, 1, -1, 2)
    if flip:
        cur[..., 0] = img_shape[1] - cur[..., 0]
        prev[..., 0] = img_shape[1] - prev[..., 0]
    for g in (cur, prev):
        g *= scale
        g -= crop_t
        g /= fsf
        g[..., 0] = g[..., 0] / (w_in - 1) * 2 - 1
        g[..., 1] = g[..., 1] / (h_in - 1) * 2 - 1
    return cur, prev


def build_dfm_cost(cur_feats, prev_feats, depths, fsf, csf, cam2imgs, cur2prevs, img_shape, flip=False,
                   crop=(0, 0), scale=1.0):
    """(B, 2C, D, h_out, w_out) fp32 on the CPU -- every library call of the reference function"""
    B, _, h_in, w_in = cur_feats.shape
    D = depths.shape[-1]
    h_out, w_out = round(h_in / csf), round(w_in / csf)
    cg, pg = sampling_grids(h_in, w_in, depths, fsf, csf, cam2imgs, cur2prevs, img_shape, flip, crop, scale, B)
    kw = dict(mode='bilinear', padding_mode='zeros', align_corners=True)
    a = F.grid_sample(cur_feats, cg, **kw).view(B, -1, D, h_out, w_out)
    b = F.grid_sample(prev_feats, pg, **kw).view(B, -1, D, h_out, w_out)
    return torch.cat([a, b], dim=1)
