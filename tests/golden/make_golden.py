"""Generate golden fixtures by running the REFERENCE's own code on PyTorch-CPU.

Run in the build container only (needs /root/reference, which does not exist
on the GPU box):   python tests/golden/make_golden.py

The reference cannot be imported as a package (mmcv / mmdet are not
installed), so its functions are lifted from their source files by AST and
exec'd UNMODIFIED; nothing of the reference is copied into this repo -- only
inputs and the outputs it produced are stored (tests/golden/*.npz).

Pinned here:
  * plane_sweep_*.npz : build_dfm_cost (dfm_backbone.py:217-314) outputs AND
    the normalised grids it passes to F.grid_sample, on seeded inputs with
    flip / crop / scale / cost_sample_factor variants and a pose that sends
    part of the sweep behind the camera.
  * helpers.npz : the reference tests' own golden vectors for points_cam2img
    (tests/test_utils/test_box3d.py:1653-1680), points_img2cam
    (tests/test_utils/test_utils.py:186-193) and point_sample
    (tests/test_models/test_fusion/test_point_fusion.py:13-61), re-derived by
    running the reference functions so the fixture carries inputs + outputs.
"""
import ast
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REF = '/root/reference/mmdet3d/'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def extract(path, names, glb):
    """exec the named top-level functions of a reference file, decorators off"""
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            node.decorator_list = []
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, 'exec'), glb)
    return glb


def extract_method(path, cls, name, glb, rename=None):
    """exec method `cls.name` of a reference file as a plain function"""
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == name:
                    sub.decorator_list = []
                    if rename:
                        sub.name = rename
                    exec(compile(ast.Module(body=[sub], type_ignores=[]), path, 'exec'), glb)
                    return glb
    raise KeyError(f'{cls}.{name} not found in {path}')


def load_reference():
    g = {'torch': torch, 'F': F, 'np': np, 'nn': torch.nn}
    extract(REF + 'core/bbox/structures/utils.py', ['points_cam2img', 'points_img2cam'], g)
    extract(REF + 'models/backbones/dfm_backbone.py', ['build_dfm_cost'], g)
    g['apply_3d_transformation'] = lambda pts, coord_type, img_meta, reverse=False: pts
    extract(REF + 'models/fusion_layers/point_fusion.py', ['point_sample', 'voxel_sample'], g)
    extract_method(REF + 'models/detectors/multiview_dfm.py', 'MultiViewDfM',
                   'feature_transformation', g, rename='mv_feature_transformation')
    extract_method(REF + 'core/anchor/anchor_3d_generator.py', 'AlignedAnchor3DRangeGenerator',
                   'anchors_single_range', g, rename='aligned_anchors_single_range')
    return g


KITTI_P2 = [[721.5377, 0, 609.5593, 44.85728], [0, 721.5377, 172.854, 0.2163791],
            [0, 0, 1, 0.002745884], [0, 0, 0, 1]]


def pose(yaw_deg, tx, ty, tz):
    c, s = math.cos(math.radians(yaw_deg)), math.sin(math.radians(yaw_deg))
    return [[c, 0, s, tx], [0, 1, 0, ty], [-s, 0, c, tz], [0, 0, 0, 1]]


# name, C, H, W, D, fsf, csf, flip, crop, scale, pose, depth range
SWEEP_CASES = [
    ('nstar_like', 5, 24, 78, 6, 16, 1, False, (0, 0), 1.0, pose(1.3, 0.05, 0.01, -1.1), (2, 59.6)),
    ('kitti_like', 4, 48, 160, 5, 1, 4, False, (0, 55), 1.0, pose(-0.7, -0.03, 0.0, -0.8), (2, 59.6)),
    ('flip_crop_scale', 3, 24, 78, 4, 16, 1, True, (11, 55), 1.03, pose(2.0, 0.1, -0.02, -1.4),
     (2, 59.6)),
    ('kitti_flip_f32scale', 3, 48, 160, 4, 1, 4, True, (7, 55), np.float32(0.97),
     pose(0.4, 0.0, 0.0, -0.5), (2, 59.6)),
    # cur2prev pushes the nearest planes behind the previous camera (z <= 0
    # after the warp): unguarded divide in the reference, zeros from padding
    ('behind_camera', 3, 24, 78, 6, 16, 1, False, (0, 0), 1.0, pose(5.0, 0.3, 0.0, -9.0), (2, 20)),
    ('odd_channels_bigshift', 9, 20, 64, 3, 8, 2, False, (3, 5), 1.0, pose(-8.0, 1.5, 0.2, 2.5),
     (2, 40)),
]


def make_sweep(g):
    captured = []
    orig = F.grid_sample

    def capture(inp, grid, **kw):
        captured.append(grid.clone())
        return orig(inp, grid, **kw)

    for i, (name, C, H, W, D, fsf, csf, flip, crop, scale, T, (dmin, dmax)) in enumerate(SWEEP_CASES):
        gen = torch.Generator().manual_seed(100 + i)
        cur = torch.randn(1, C, H, W, generator=gen)
        prev = torch.randn(1, C, H, W, generator=gen)
        depths = torch.tensor([dmin + (k + 0.5) * ((dmax - dmin) / D) for k in range(D)],
                              dtype=torch.float32)
        P = torch.tensor(KITTI_P2, dtype=torch.float32)
        Tm = torch.tensor(T, dtype=torch.float32)
        captured.clear()
        F.grid_sample = capture
        try:
            out = g['build_dfm_cost'](cur, prev, depths, fsf, csf, P[None], Tm[None], (375, 1242),
                                      flip, crop, scale)
        finally:
            F.grid_sample = orig
        Ppad = torch.eye(4)
        Ppad[:3, :4] = P[:3]
        np.savez_compressed(
            os.path.join(HERE, f'plane_sweep_{name}.npz'),
            cur=cur.numpy(), prev=prev.numpy(), depths=depths.numpy(), P=P.numpy(),
            Pinv=torch.inverse(Ppad).numpy(), T=Tm.numpy(), fsf=np.float64(fsf),
            csf=np.float64(csf), flip=np.bool_(flip), crop=np.asarray(crop, np.float64),
            scale=np.float64(scale), img_shape=np.asarray((375, 1242)),
            ref_out=out.numpy(), ref_cur_grid=captured[0].numpy().reshape(-1, 2),
            ref_prev_grid=captured[1].numpy().reshape(-1, 2))
        print(name, tuple(out.shape), 'nan:', int(torch.isnan(out).sum()))


def make_helpers(g):
    d = {}
    # points_cam2img golden (reference tests/test_utils/test_box3d.py:1653-1680)
    torch.manual_seed(0)  # the reference test calls set_random_seed(0)
    np.random.seed(0)
    points = torch.rand([5, 3])
    proj_mat = torch.rand([4, 4])
    d['cam2img_points'] = points.numpy()
    d['cam2img_proj'] = proj_mat.numpy()
    d['cam2img_out'] = g['points_cam2img'](points, proj_mat).numpy()
    d['cam2img_out_depth'] = g['points_cam2img'](points, proj_mat, with_depth=True).numpy()
    d['cam2img_expected'] = np.array([[0.5832, 0.6496], [0.6146, 0.7910], [0.6994, 0.7782],
                                      [0.5623, 0.6303], [0.4359, 0.6532]], np.float32)
    # points_img2cam golden (reference tests/test_utils/test_utils.py:186-193)
    pts = torch.tensor([[0.5764, 0.9109, 0.7576], [0.6656, 0.5498, 0.9813]])
    cam2img = torch.tensor([[700., 0., 450., 0.], [0., 700., 200., 0.], [0., 0., 1., 0.]])
    d['img2cam_points'] = pts.numpy()
    d['img2cam_cam2img'] = cam2img.numpy()
    d['img2cam_out'] = g['points_img2cam'](pts, cam2img).numpy()
    d['img2cam_expected'] = np.array([[-0.4864, -0.2155, 0.7576], [-0.6299, -0.2796, 0.9813]],
                                     np.float32)
    # point_sample golden (reference tests/test_models/test_fusion/test_point_fusion.py:13-40)
    img_meta = {'img_shape': (370, 1224), 'pad_shape': (370, 1224), 'ori_shape': (370, 1224)}
    lidar2img = torch.tensor([[6.0294e+02, -7.0791e+02, -1.2275e+01, -1.7094e+02],
                              [1.7678e+02, 8.8088e+00, -7.0794e+02, -1.0257e+02],
                              [9.9998e-01, -1.5283e-03, -5.2907e-03, -3.2757e-01],
                              [0.0, 0.0, 0.0, 1.0]])
    img = torch.arange(370 * 1224, dtype=torch.float32).reshape(1, 1, 370, 1224) / (370 * 1224)
    pts = torch.tensor([[8.356, -4.312, -0.445], [11.777, -6.724, -0.564], [6.453, 2.53, -1.612],
                        [6.227, -3.839, -0.563]])
    # PointFusion.sample_single defaults (point_fusion.py:298-320): scale 1,
    # crop 0, no flip, aligned (bilinear), zeros, align_corners=True
    out = g['point_sample'](img_meta, img, pts, lidar2img, 'LIDAR', 1, 0, False, (370, 1224),
                            (370, 1224), aligned=True, padding_mode='zeros', align_corners=True)
    # the ramp image is arange(370*1224)/(370*1224): rebuilt by the test, not stored
    d['ps_points'] = pts.numpy()
    d['ps_lidar2img'] = lidar2img.numpy()
    d['ps_out'] = out.numpy()
    d['ps_expected'] = np.array([0.5560822, 0.5476625, 0.9687978, 0.6241757], np.float32)
    np.savez_compressed(os.path.join(HERE, 'helpers.npz'), **d)
    print('cam2img', np.abs(d['cam2img_out'] - d['cam2img_expected']).max(), 'img2cam',
          np.abs(d['img2cam_out'] - d['img2cam_expected']).max(), 'point_sample',
          np.abs(d['ps_out'].reshape(-1) - d['ps_expected']).max())


def waymo_like_cameras(num_views, num_frames, seed):
    """lidar2img (4x4) per (frame, view): cameras at yaw 0, +-45, +-90 deg around
    the ego vehicle, Waymo-like intrinsics scaled to the 1/1 padded input, and a
    per-frame ego motion folded in (loading.py:122-142 pre-folds it the same way)."""
    rng = np.random.RandomState(seed)
    yaws = [0.0, 45.0, -45.0, 90.0, -90.0][:num_views]
    mats = []
    for f in range(num_frames):
        ego = np.eye(4)
        if f > 0:
            a = np.radians(rng.uniform(-3, 3))
            ego[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
            ego[:3, 3] = [rng.uniform(0.5, 2.0), rng.uniform(-0.2, 0.2), 0.0]
        for yaw in yaws:
            a = np.radians(yaw)
            # lidar (x fwd, y left, z up) -> camera (x right, y down, z fwd), camera yawed by a
            R_yaw = np.array([[np.cos(a), np.sin(a), 0], [-np.sin(a), np.cos(a), 0], [0, 0, 1]])
            R_axes = np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0]], float)
            ext = np.eye(4)
            ext[:3, :3] = R_axes @ R_yaw
            ext[:3, 3] = R_axes @ R_yaw @ -np.array([1.5, 0.0, 2.0])
            K = np.eye(4)
            K[0, 0] = K[1, 1] = 130.0 + rng.uniform(-2, 2)
            K[0, 2], K[1, 2] = 78.0, 50.0
            mats.append((K @ ext @ ego).astype(np.float32))
    return np.stack(mats)


# name, num_views, num_frames, C, feature HxW, input (pad) shape, n_voxels (x,y,z), range,
# scale_factor, flip, crop offset, temporal aggregate
MV_CASES = [
    ('mv_mean_1frame', 5, 1, 6, (26, 39), (104, 156), (22, 30, 4),
     [-11.0, -15.0, -2.0, 11.0, 15.0, 2.0], None, False, None, 'mean'),
    ('mv_concat_2frames_aug', 5, 2, 5, (26, 39), (104, 156), (22, 30, 4),
     [-11.0, -15.0, -2.0, 11.0, 15.0, 2.0], np.array([0.95, 1.05, 0.95, 1.05], np.float32), True,
     np.array([3.0, 2.0], np.float32), 'concat'),
    ('mv_mean_2frames', 3, 2, 4, (26, 39), (104, 156), (20, 28, 8),
     [-10.0, -14.0, -2.0, 10.0, 14.0, 2.0], None, False, None, 'mean'),
]


def make_mv(g):
    from types import SimpleNamespace
    for i, (name, nv, nf, C, (hf, wf), pad, nvox, rng_, scale, flip, crop, agg) in enumerate(MV_CASES):
        gen = torch.Generator().manual_seed(200 + i)
        feats = torch.randn(1, nv * nf, C, hf, wf, generator=gen)
        lidar2img = waymo_like_cameras(nv, nf, 300 + i)
        gen_self = SimpleNamespace(align_corner=False, custom_values=[])

        def grid_anchors(featmap_sizes, device='cpu'):
            a = g['aligned_anchors_single_range'](gen_self, featmap_sizes[0], rng_, 1,
                                                  sizes=[[0.0, 0.0, 0.0]], rotations=[0.0],
                                                  device=device)
            return [a.reshape(-1, a.size(-1))]

        self_ = SimpleNamespace(
            anchor_generator=SimpleNamespace(grid_anchors=grid_anchors), n_voxels=list(nvox),
            valid_sample=True, temporal_aggregate=agg, with_backbone_3d=False,
            with_depth_head=False, with_neck_3d=False)
        meta = {'ori_lidar2img': [m for m in lidar2img], 'input_shape': pad,
                'img_shape': [(pad[0] - 4, pad[1] - 6, 3)] * (nv * nf)}
        if scale is not None:
            meta['scale_factor'] = scale
        if flip:
            meta['flip'] = True
        if crop is not None:
            meta['img_crop_offset'] = crop
        out = g['mv_feature_transformation'](self_, feats, [meta], nv, nf)[0]
        points = grid_anchors([list(nvox)[::-1]])[0][:, :3]
        np.savez_compressed(
            os.path.join(HERE, f'{name}.npz'), feats=feats.numpy(), lidar2img=lidar2img,
            points=points.numpy(), n_voxels=np.asarray(nvox), voxel_range=np.asarray(rng_),
            input_shape=np.asarray(pad), img_shape=np.asarray(meta['img_shape'][0][:2]),
            scale=np.zeros(0, np.float32) if scale is None else scale, flip=np.bool_(flip),
            crop=np.zeros(0, np.float32) if crop is None else crop, aggregate=agg,
            num_views=nv, num_frames=nf, ref_out=out.numpy())
        nz = (out != 0).float().mean().item()
        print(name, tuple(out.shape), f'nonzero {nz:.3f}')


# name, C, (D,H,W) of the cost volume, point cloud range, voxel size, depth range, batch
F2V_CASES = [
    ('f2v_small', 4, (6, 10, 32), [2, -6.0, -3, 14.0, 6.0, 1], [0.5, 0.5, 0.5], (2, 14.0), 1),
    ('f2v_batch2', 3, (5, 12, 24), [1, -5.0, -2, 11.0, 5.0, 2], [0.5, 0.5, 1.0], (1, 11.0), 2),
]


def make_f2v():
    import ref_stubs
    from types import SimpleNamespace
    ref_stubs.install()
    ft = ref_stubs.load_file('mmdet3d/models/necks/feature_transformation.py', 'ref_ft')
    g = {'torch': torch, 'np': np}
    extract_method(REF + 'models/detectors/dfm.py', 'DfM', 'prepare_coordinates_3d', g)
    torch.Tensor.cuda = lambda self, *a, **k: self  # feature_transformation.py:82,93 hard-code .cuda()
    for i, (name, C, (D, H, W), pcr, vs, (dmin, dmax), B) in enumerate(F2V_CASES):
        gen = torch.Generator().manual_seed(400 + i)
        m = ft.FrustumToVoxel(cv_channels=C, out_channels=C, in_sem_channels=C,
                              norm_cfg=dict(type='GN', num_groups=1, requires_grad=True))
        m.voxel_convs = torch.nn.Identity()   # capture the sampled (B,2C,Nz,Ny,Nx) volume
        m.voxel_pool = torch.nn.Identity()
        holder = SimpleNamespace()
        g['prepare_coordinates_3d'](holder, dict(point_cloud_range=pcr, voxel_size=vs))
        m.coordinates_3d = holder.coordinates_3d
        m.depth_cfg = dict(depth_min=dmin, depth_max=dmax)
        stereo = torch.randn(B, C, D, H, W, generator=gen)
        soft = torch.softmax(torch.randn(B, 1, 4 * D, 4 * H, 4 * W, generator=gen), dim=2)
        sem = torch.randn(B, C, H, W, generator=gen)
        pad = (4 * H, 4 * W)
        metas = []
        for b in range(B):
            K = np.eye(4, dtype=np.float32)
            K[0, 0] = K[1, 1] = 70.0 + 3 * b
            K[0, 2], K[1, 2] = pad[1] / 2 + b, pad[0] / 2 - b
            K[0, 3], K[1, 3], K[2, 3] = 4.5, 0.2, 0.003
            metas.append({'cam2img': K.tolist(), 'pad_shape': pad + (3,)})
        out = m(stereo, soft, metas, sem)
        np.savez_compressed(
            os.path.join(HERE, f'{name}.npz'), stereo=stereo.numpy(), softmax=soft.numpy(),
            sem=sem.numpy(), coordinates_3d=holder.coordinates_3d.numpy(),
            cam2img=np.stack([np.asarray(mm['cam2img'], np.float32) for mm in metas]),
            pad_shape=np.asarray(pad), depth_min=np.float64(dmin), depth_max=np.float64(dmax),
            ref_out=out.numpy())
        print(name, tuple(out.shape), 'nonzero', float((out != 0).float().mean()))


def make_depth_head():
    import ref_stubs
    ref_stubs.install()
    dh = ref_stubs.load_file('mmdet3d/models/dense_heads/depth_head.py', 'ref_depth_head')
    for i, (name, B, D, H, W) in enumerate([('depth_head_small', 2, 6, 5, 9),
                                            ('depth_head_wide', 1, 18, 4, 40)]):
        gen = torch.Generator().manual_seed(500 + i)
        m = dh.DepthHead(depth_cfg=dict(mode='UD', num_bins=4 * D, min_depth=2, max_depth=59.6),
                         with_convs=False, depth_loss=dict(type='ce', loss_weight=1.0),
                         downsample_factor=4, num_views=1)
        interval = (59.6 - 2) / (4 * D)
        m.depth_samples = torch.tensor([(k + 0.5) * interval + 2 for k in range(4 * D)],
                                       dtype=torch.float32)  # DfM.prepare_depth, dfm.py:170-172
        x = torch.randn(B, 1, D, H, W, generator=gen) * 3
        vol, soft, pred = m(x)
        np.savez_compressed(os.path.join(HERE, f'{name}.npz'), x=x.numpy(),
                            depth_samples=m.depth_samples.numpy(), ref_vol=vol.numpy(),
                            ref_soft=soft.numpy(), ref_pred=pred.numpy())
        print(name, tuple(vol.shape), tuple(pred.shape))


def make_modules():
    """Whole-module fixtures: the reference modules (executed unmodified under
    ref_stubs) with the deterministic weights of tests/util.synthetic_state_dict."""
    import ref_stubs
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tests import util
    ref = ref_stubs.load_hot_path_modules()
    torch.Tensor.cuda = lambda self, *a, **k: self
    out = {}
    # --- DfMBackbone: 2 x (1,4,16,32) feats, csf 4, 8 planes
    depth_cfg = dict(mode='UD', num_bins=32, depth_min=2, depth_max=59.6, downsample_factor=4)
    m = ref['dfm_backbone'].DfMBackbone(in_channels=4, cv_channels=32, cost_sample_factor=4,
                                        depth_cfg=depth_cfg).eval()
    m.load_state_dict(util.synthetic_state_dict(m, 11))
    interval = (59.6 - 2) / 32
    m.downsampled_depth = torch.tensor([(i + 0.5) * 4 * interval + 2 for i in range(8)])
    gen = torch.Generator().manual_seed(12)
    cur, prev = torch.randn(1, 4, 16, 32, generator=gen), torch.randn(1, 4, 16, 32, generator=gen)
    K = KITTI_P2
    meta = dict(ori_cam2img=K, cur2prevs=torch.tensor(pose(1.0, 0.05, 0.0, -0.9))[None],
                ori_shape=(375, 1242, 3), pad_shape=(16, 32, 3), crop_offset=[600, 150],
                flip=False, scale_factor=[1.0])
    with torch.no_grad():
        cost, sfeat, mfeat = m(cur, prev, [meta])
    out.update(bb_cur=cur.numpy(), bb_prev=prev.numpy(), bb_cost=cost.numpy(),
               bb_stereo=sfeat.numpy(), bb_mono=mfeat.numpy(),
               bb_depths=m.downsampled_depth.numpy())
    # --- FrustumToVoxel with its conv + pool
    z = np.load(os.path.join(HERE, 'f2v_small.npz'))
    C = z['stereo'].shape[1]
    m = ref['feature_transformation'].FrustumToVoxel(
        cv_channels=C, out_channels=8, in_sem_channels=C,
        norm_cfg=dict(type='GN', num_groups=4, requires_grad=True)).eval()
    m.load_state_dict(util.synthetic_state_dict(m, 21))
    m.coordinates_3d = torch.from_numpy(z['coordinates_3d'])
    m.depth_cfg = dict(depth_min=float(z['depth_min']), depth_max=float(z['depth_max']))
    metas = [{'cam2img': c.tolist(), 'pad_shape': tuple(z['pad_shape']) + (3,)} for c in z['cam2img']]
    with torch.no_grad():
        out['f2v_out'] = m(torch.from_numpy(z['stereo']), torch.from_numpy(z['softmax']), metas,
                           torch.from_numpy(z['sem'])).numpy()
    # --- voxel necks, eval-mode BN
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(2, 8, 6, 5, 12, generator=gen)
    m = ref['imvoxel_neck'].OutdoorImVoxelNeck(in_channels=8, out_channels=16).eval()
    m.load_state_dict(util.synthetic_state_dict(m, 32))
    with torch.no_grad():
        out.update(neck_x=x.numpy(), imvoxel_out=m(x)[0].numpy())
    m = ref['dfm_neck'].DfMNeck(in_channels=4, out_channels=16, num_frames=2).eval()
    m.load_state_dict(util.synthetic_state_dict(m, 33))
    with torch.no_grad():
        out['dfmneck_out'] = m(x)[0].numpy()
    np.savez_compressed(os.path.join(HERE, 'modules.npz'), **out)
    print('modules:', {k: v.shape for k, v in out.items()})


def make_voxel_sample(g):
    gen = torch.Generator().manual_seed(600)
    vox = torch.randn(1, 5, 22, 30, 4, generator=gen)
    rng_, vsz = [-11.0, -15.0, -2.0, 11.0, 15.0, 2.0], [1.0, 1.0, 1.0]
    cam = torch.from_numpy(waymo_like_cameras(1, 1, 7)[0])
    depth_samples = torch.tensor([1.0 + 0.5 * k for k in range(24)])
    out = {}
    for tag, kw in (('plain', dict(img_scale_factor=torch.tensor([1.0, 1.0]),
                                   img_crop_offset=torch.tensor([0.0, 0.0]), img_flip=False)),
                    ('aug', dict(img_scale_factor=torch.tensor([0.95, 1.05]),
                                 img_crop_offset=torch.tensor([3.0, 2.0]), img_flip=True))):
        for aligned in (True, False):
            o = g['voxel_sample'](vox, rng_, vsz, depth_samples, cam, 4, img_pad_shape=(104, 156),
                                  img_shape=(100, 150), aligned=aligned, **kw)
            out[f'out_{tag}_{"tri" if aligned else "near"}'] = o.numpy()
    np.savez_compressed(os.path.join(HERE, 'voxel_sample.npz'), vox=vox.numpy(),
                        voxel_range=np.asarray(rng_, np.float32), voxel_size=np.asarray(vsz, np.float32),
                        depth_samples=depth_samples.numpy(), proj=cam.numpy(),
                        proj_inv=torch.inverse(cam).numpy(), **out)
    print('voxel_sample', {k: (v.shape, float((v != 0).mean())) for k, v in out.items()})


if __name__ == '__main__':
    if not os.path.isdir(REF):
        sys.exit('reference not mounted; fixtures are committed, nothing to do')
    torch.set_num_threads(1)
    ref = load_reference()
    make_sweep(ref)
    make_helpers(ref)
    make_mv(ref)
    make_voxel_sample(ref)
    make_f2v()
    make_depth_head()
    make_modules()
