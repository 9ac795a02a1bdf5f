"""Round-2 golden fixtures, produced by running the REFERENCE's own code on PyTorch-CPU.

Build container only (needs /root/reference):   python tests/golden/make_golden_r02.py
Same rules as make_golden.py: reference files are executed unmodified from where they lie (AST
lift / ref_stubs); only inputs and the outputs the reference produced are stored.

  depth_loss.npz       DepthHead.loss (dense_heads/depth_head.py:75-188) for every loss type the
                       class implements: loss value and d loss / d (depth_volumes, depth_preds)
                       from torch autograd.
  configs_dfm.json     the model sub-dicts (backbone_stereo, feature_transformation, depth_head,
                       neck_3d, backbone_3d, voxel / depth settings) of configs/dfm/*.py, obtained
                       by exec'ing the config files (with their _base_ chain) -- what
                       tests/test_config_build.py builds through the registry.
  backbone_cfg1.npz    BASELINE.json configs[0] ("one synthetic KITTI pair 375x1242, D=4"):
                       DfMBackbone forward (dfm_backbone.py:143-214) on seeded stereo features of
                       the padded 384x1248 image, D=4 planes (num_bins=16, downsample_factor=4).
                       Inputs are regenerated from the seeds by the test; stored are the cost
                       volume and a strided sample of the two feature volumes.
  modules_wide.npz     the 3-D aggregation modules at their REAL channel widths (what the MFMA
                       convolution kernels cover): hourglass(32) (utils/conv_modules.py:73-149),
                       OutdoorImVoxelNeck(64 -> 256) (necks/imvoxel_neck.py) and
                       DfMNeck(64 -> 256, 2 frames) (necks/dfm_neck.py) in eval mode on small seeded
                       volumes with synthetic weights; inputs are regenerated from the seeds.
  bev_spp.npz          BEVHourglass (backbones/bev_hourglass.py) and SPPUNetNeck
                       (necks/spp_unet_neck.py) forward on seeded inputs with synthetic weights.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
REF_ROOT = '/root/reference'

LOSS_TYPES = ['ce', 'balanced_ce', 'focal', 'balanced_focal', 'hard_ce', 'gaussian_1.5',
              'laplacian_2.0', 'l1', 'purel1']


def depth_loss_inputs(seed=700, B=2, D=24, H=10, W=14):
    """shared by the generator and tests/test_depth_loss.py"""
    gen = torch.Generator().manual_seed(seed)
    interval = (59.6 - 2) / D
    ds = torch.tensor([(k + 0.5) * interval + 2 for k in range(D)], dtype=torch.float32)
    vol = torch.randn(B, D, H, W, generator=gen) * 3
    pred = torch.rand(B, H, W, generator=gen) * 57.6 + 2
    img = torch.rand(B, H, W, generator=gen) * 70.0            # some beyond max_depth
    img[torch.rand(B, H, W, generator=gen) < 0.45] = 0.0       # no LiDAR return
    img.view(-1)[:3] = torch.tensor([2.0, 59.6, 30.0])         # the strict-inequality edges
    fg = (torch.rand(B, H, W, generator=gen) * 3).floor()      # box ids 0 (bg), 1, 2
    return ds, vol, pred, img, fg


def make_depth_loss():
    import ref_stubs
    ref_stubs.install()
    dh = ref_stubs.load_file('mmdet3d/models/dense_heads/depth_head.py', 'ref_depth_head_r02')
    dh.dist = types.SimpleNamespace(get_rank=lambda: 1)  # gaussian/laplacian print on rank 0
    ds, vol, pred, img, fg = depth_loss_inputs()
    out = dict(depth_samples=ds.numpy(), volumes=vol.numpy(), preds=pred.numpy(), depth_img=img.numpy(),
               fgmask=fg.numpy())
    for t in LOSS_TYPES:
        cfg = dict(type=t, loss_weight=0.7)
        if 'balanced' in t:
            cfg.update(fg_weight=5, bg_weight=1)
        if 'focal' in t:
            cfg.update(alpha=0.75, gamma=2)
        m = dh.DepthHead(depth_cfg=dict(mode='UD', num_bins=len(ds), min_depth=2, max_depth=59.6),
                         with_convs=False, depth_loss=cfg, downsample_factor=4, num_views=1)
        m.depth_samples = ds
        v, p = vol.clone().requires_grad_(True), pred.clone().requires_grad_(True)
        loss = m.loss(p, v, img, depth_fgmask_img=fg)
        loss.backward()
        key = t.replace('.', 'p')
        out[f'{key}_loss'] = loss.detach().numpy()
        out[f'{key}_gvol'] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy()
        out[f'{key}_gpred'] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
        print(f'depth loss {t:16s} {float(loss):.6f}')
    # gamma = 3 (generic pow path) for the focal form
    m = dh.DepthHead(depth_cfg=dict(mode='UD', num_bins=len(ds), min_depth=2, max_depth=59.6),
                     with_convs=False, depth_loss=dict(type='focal', loss_weight=1.0, alpha=1, gamma=3),
                     downsample_factor=4, num_views=1)
    m.depth_samples = ds
    v = vol.clone().requires_grad_(True)
    loss = m.loss(pred, v, img)
    loss.backward()
    out['focal_g3_loss'], out['focal_g3_gvol'] = loss.detach().numpy(), v.grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'depth_loss.npz'), **out)


def _jsonable(x):
    if isinstance(x, dict):
        return {k: _jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    if isinstance(x, (int, float, str, bool)) or x is None:
        return x
    return repr(x)


def exec_config(path):
    """mmcv.Config.fromfile for plain-Python configs: exec the _base_ chain, then the file
    (later files override earlier keys; dicts are merged recursively like mmcv does)."""
    def merge(base, new):
        for k, v in new.items():
            if isinstance(v, dict) and isinstance(base.get(k), dict) and not v.pop('_delete_', False):
                merge(base[k], v)
            else:
                base[k] = v
        return base

    src = open(path).read()
    glb = {}
    exec(compile(src, path, 'exec'), glb)
    cfg = {}
    bases = glb.get('_base_', [])
    for b in ([bases] if isinstance(bases, str) else bases):
        merge(cfg, exec_config(os.path.normpath(os.path.join(os.path.dirname(path), b))))
    merge(cfg, {k: v for k, v in glb.items() if not k.startswith('__') and k != '_base_' and
                not isinstance(v, types.ModuleType)})
    return cfg


def make_configs():
    names = ['dfm_r34_1x8_kitti-3d-3class.py']
    cfg_dir = os.path.join(REF_ROOT, 'configs', 'dfm')
    names += sorted(f for f in os.listdir(cfg_dir) if f.startswith('multiview-dfm_') and f.endswith('.py'))
    keep = ('backbone_stereo', 'feature_transformation', 'depth_head', 'neck_3d', 'backbone_3d', 'neck',
            'neck_stereo', 'neck_bev', 'depth_cfg', 'voxel_cfg', 'n_voxels', 'anchor_generator',
            'depth_head', 'temporal_aggregate', 'num_ref_frames', 'type', 'voxel_size')
    out = {}
    for n in names:
        cfg = exec_config(os.path.join(cfg_dir, n))
        model = cfg['model']
        out[n] = {'model': {k: _jsonable(v) for k, v in model.items() if k in keep},
                  'file': f'configs/dfm/{n}'}
        print(n, sorted(out[n]['model']))
    with open(os.path.join(HERE, 'configs_dfm.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)


CFG1 = dict(C=32, H=384, W=1248, D=4, csf=4, seed=41, wseed=42,
            ori_shape=(375, 1242, 3), pad_shape=(384, 1248, 3))


def cfg1_inputs():
    """shared by the generator and tests/test_modules.py"""
    c = CFG1
    gen = torch.Generator().manual_seed(c['seed'])
    cur = torch.randn(1, c['C'], c['H'], c['W'], generator=gen)
    prev = torch.randn(1, c['C'], c['H'], c['W'], generator=gen)
    interval = (59.6 - 2) / 16  # num_bins=16, downsample_factor=4 -> D=4
    depths = torch.tensor([(i + 0.5) * 4 * interval + 2 for i in range(c['D'])])
    from make_golden import KITTI_P2, pose
    meta = dict(ori_cam2img=KITTI_P2, cur2prevs=torch.tensor(pose(-0.7, 0.03, 0.0, -1.1))[None],
                ori_shape=c['ori_shape'], pad_shape=c['pad_shape'], crop_offset=[0, 0], flip=False,
                scale_factor=[1.0])
    return cur, prev, depths, meta


def make_backbone_cfg1():
    import ref_stubs
    from tests import util
    ref = ref_stubs.load_hot_path_modules()
    torch.Tensor.cuda = lambda self, *a, **k: self
    c = CFG1
    depth_cfg = dict(mode='UD', num_bins=16, depth_min=2, depth_max=59.6, downsample_factor=4)
    m = ref['dfm_backbone'].DfMBackbone(in_channels=c['C'], cv_channels=32, num_hg=1,
                                        cost_sample_factor=c['csf'], depth_cfg=depth_cfg,
                                        norm_cfg=dict(type='GN', num_groups=32, requires_grad=True)).eval()
    m.load_state_dict(util.synthetic_state_dict(m, c['wseed']))
    cur, prev, depths, meta = cfg1_inputs()
    m.downsampled_depth = depths
    torch.set_num_threads(8)
    with torch.no_grad():
        cost, sfeat, mfeat = m(cur, prev, [meta])
    print('cfg1', tuple(cost.shape), tuple(sfeat.shape), float(cost.abs().mean()))
    np.savez_compressed(os.path.join(HERE, 'backbone_cfg1.npz'), cost=cost.numpy(),
                        stereo_s8=sfeat[..., ::8, ::8].numpy(), mono_s8=mfeat[..., ::8, ::8].numpy(),
                        stereo_abs_mean=sfeat.abs().mean((0, 2, 3, 4)).numpy(),
                        mono_abs_mean=mfeat.abs().mean((0, 2, 3, 4)).numpy(), depths=depths.numpy())


def bev_spp_inputs():
    """shared by the generator and tests/test_modules.py"""
    gen = torch.Generator().manual_seed(51)
    bev = torch.randn(2, 16, 24, 32, generator=gen)
    # image + four backbone levels (1/2, 1/4, 1/4, 1/4) of a 256x256 crop: the 64x64 average
    # pool of the SPP branch needs >= 64 pixels at 1/4 resolution
    feats = [torch.randn(1, c, h, w, generator=gen) for c, h, w in
             ((3, 256, 256), (8, 128, 128), (16, 64, 64), (32, 64, 64), (32, 64, 64))]
    return bev, feats


BEV_CFG = dict(in_channels=16, out_channels=32, norm_cfg=dict(type='GN', num_groups=32, requires_grad=True))
SPP_CFG = dict(in_channels=[3, 8, 16, 32, 32], start_level=2, sem_channels=[16, 12],
               stereo_channels=[32, 12], with_upconv=True, cat_img_feature=True,
               norm_cfg=dict(type='GN', num_groups=4, requires_grad=True))


def make_bev_spp():
    import ref_stubs
    from tests import util
    ref_stubs.install()
    bh = ref_stubs.load_file('mmdet3d/models/backbones/bev_hourglass.py', 'ref_bev_hourglass')
    sp = ref_stubs.load_file('mmdet3d/models/necks/spp_unet_neck.py', 'ref_spp_unet')
    bev, feats = bev_spp_inputs()
    out = {}
    m = bh.BEVHourglass(**BEV_CFG).eval()
    m.load_state_dict(util.synthetic_state_dict(m, 52))
    with torch.no_grad():
        pre, post = m(bev)
    out['bev_prehg'], out['bev_out'] = pre.numpy(), post.numpy()
    out['bev_keys'] = np.array(list(m.state_dict().keys()))
    m = sp.SPPUNetNeck(**SPP_CFG).eval()
    m.load_state_dict(util.synthetic_state_dict(m, 53))
    with torch.no_grad():
        stereo, sem = m(feats)
    out['spp_stereo_s4'], out['spp_sem'] = stereo[..., ::4, ::4].numpy(), sem.numpy()
    out['spp_stereo_abs_mean'] = stereo.abs().mean((0, 2, 3)).numpy()
    out['spp_keys'] = np.array(list(m.state_dict().keys()))
    np.savez_compressed(os.path.join(HERE, 'bev_spp.npz'), **out)
    print('bev', out['bev_out'].shape, 'spp', tuple(stereo.shape), out['spp_sem'].shape)


def wide_inputs():
    """shared by the generator and tests/test_modules.py"""
    gen = torch.Generator().manual_seed(60)
    return dict(hg=torch.randn(1, 32, 8, 12, 16, generator=gen),
                neck=torch.randn(1, 64, 10, 12, 12, generator=gen),
                dfmneck=torch.randn(1, 128, 10, 12, 12, generator=gen))


def make_wide_modules():
    import ref_stubs
    from tests import util
    ref = ref_stubs.load_hot_path_modules()
    cm = sys.modules['mmdet3d.models.utils.conv_modules'] if 'mmdet3d.models.utils.conv_modules' in sys.modules \
        else ref_stubs.load_file('mmdet3d/models/utils/conv_modules.py', 'ref_conv_modules')
    x = wide_inputs()
    out = {}
    m = cm.hourglass(32, gn=True).eval()
    m.load_state_dict(util.synthetic_state_dict(m, 61))
    with torch.no_grad():
        y, pre, post = m(x['hg'], None, None)
    out.update(hg_out=y.numpy(), hg_pre=pre.numpy(), hg_post=post.numpy(),
               hg_keys=np.array(list(m.state_dict().keys())))
    # hourglass(gn=False): (Sync)BatchNorm3d instead of GroupNorm (conv_modules.py:42,113,126); eval mode.
    # SyncBatchNorm refuses CPU tensors even in eval mode, so its instances are swapped for
    # BatchNorm3d holding the same parameters and buffers (eval forward is the same F.batch_norm).
    m = cm.hourglass(32, gn=False).eval()
    m.load_state_dict(util.synthetic_state_dict(m, 64))
    out['hgbn_keys'] = np.array(list(m.state_dict().keys()))
    for seq in (m.conv1[0], m.conv2, m.conv3[0], m.conv4[0]):
        sbn = seq[1]
        bn = torch.nn.BatchNorm3d(sbn.num_features).eval()
        bn.load_state_dict(sbn.state_dict())
        seq[1] = bn
    with torch.no_grad():
        y, pre, post = m(x['hg'], None, None)
    out.update(hgbn_out=y.numpy(), hgbn_pre=pre.numpy(), hgbn_post=post.numpy())
    m = ref['imvoxel_neck'].OutdoorImVoxelNeck(in_channels=64, out_channels=256).eval()
    m.load_state_dict(util.synthetic_state_dict(m, 62))
    with torch.no_grad():
        out['imvoxel_out'] = m(x['neck'])[0].numpy()
    m = ref['dfm_neck'].DfMNeck(in_channels=64, out_channels=256, num_frames=2).eval()
    m.load_state_dict(util.synthetic_state_dict(m, 63))
    with torch.no_grad():
        out['dfmneck_out'] = m(x['dfmneck'])[0].numpy()
    np.savez_compressed(os.path.join(HERE, 'modules_wide.npz'), **out)
    print('wide modules:', {k: v.shape for k, v in out.items() if k != 'hg_keys'})


def make_f2v_variants():
    """FrustumToVoxel's attention switches (feature_transformation.py:141-142,154-155) on the
    inputs of f2v_small.npz / f2v_batch2.npz: the sampled volume of the REFERENCE module
    (voxel_convs / voxel_pool replaced by identities, like make_golden.py:make_f2v)."""
    import ref_stubs
    ref_stubs.install()
    ft = ref_stubs.load_file('mmdet3d/models/necks/feature_transformation.py', 'ref_ft_variants')
    torch.Tensor.cuda = lambda self, *a, **k: self  # feature_transformation.py:82,93 hard-code .cuda()
    out = {}
    for name in ('f2v_small', 'f2v_batch2'):
        z = np.load(os.path.join(HERE, name + '.npz'))
        C = z['stereo'].shape[1]
        metas = [{'cam2img': c.tolist(), 'pad_shape': tuple(int(v) for v in z['pad_shape']) + (3,)}
                 for c in z['cam2img']]
        for tag, kw in (('stereo_sem', dict(stereo_atten_feat=True, sem_atten_feat=True)),
                        ('none', dict(stereo_atten_feat=False, sem_atten_feat=False)),
                        ('stereo_only', dict(stereo_atten_feat=True, sem_atten_feat=False)),
                        ('stereo_nocat', dict(stereo_atten_feat=True, cat_img_feature=False))):
            m = ft.FrustumToVoxel(cv_channels=C, out_channels=C, in_sem_channels=C,
                                  norm_cfg=dict(type='GN', num_groups=1, requires_grad=True), **kw)
            m.voxel_convs = torch.nn.Identity()
            m.voxel_pool = torch.nn.Identity()
            m.coordinates_3d = torch.from_numpy(z['coordinates_3d'])
            m.depth_cfg = dict(depth_min=float(z['depth_min']), depth_max=float(z['depth_max']))
            with torch.no_grad():
                y = m(torch.from_numpy(z['stereo']), torch.from_numpy(z['softmax']), metas,
                      torch.from_numpy(z['sem']) if kw.get('cat_img_feature', True) else None)
            out[f'{name}__{tag}'] = y.numpy()
    np.savez_compressed(os.path.join(HERE, 'frustum_atten_variants.npz'), **out)
    print('f2v variants:', {k: v.shape for k, v in out.items()})


FLOW_METAS = {  # coord_type, img_meta: every entry of transformation_3d_flow (coord_transform.py:19-24)
    'lidar_rsthf': ('LIDAR', dict(transformation_3d_flow=['R', 'S', 'T', 'HF'],
                                  pcd_rotation=[[0.9801, -0.1987, 0.0], [0.1987, 0.9801, 0.0], [0.0, 0.0, 1.0]],
                                  pcd_scale_factor=1.05, pcd_trans=[0.1, -0.2, 0.05], pcd_horizontal_flip=True)),
    'lidar_hf_vf_st': ('LIDAR', dict(transformation_3d_flow=['HF', 'VF', 'S', 'T'], pcd_scale_factor=0.96,
                                     pcd_trans=[-0.3, 0.4, 0.0], pcd_horizontal_flip=True,
                                     pcd_vertical_flip=True)),
    'lidar_hf_st': ('LIDAR', dict(transformation_3d_flow=['HF', 'S', 'T'], pcd_scale_factor=0.96,
                                  pcd_trans=[-0.3, 0.4, 0.0], pcd_horizontal_flip=True)),
    'camera_hf_vf': ('CAMERA', dict(transformation_3d_flow=['HF', 'VF', 'S'], pcd_scale_factor=1.02,
                                    pcd_horizontal_flip=True, pcd_vertical_flip=True)),
    'depth_vf_unset_hf': ('DEPTH', dict(transformation_3d_flow=['T', 'HF', 'VF'], pcd_trans=[0.5, 0.25, -0.125],
                                        pcd_horizontal_flip=False, pcd_vertical_flip=True)),
}


def make_point_sample_flow():
    """point_sample with a 3-D augmentation flow in img_meta (point_fusion.py:57-58 ->
    coord_transform.py:9-95): the REAL apply_3d_transformation / points classes of the reference
    (make_golden.py stubbed it with the identity: its fixtures carry no flow)."""
    import types
    import importlib.util
    import ref_stubs
    import make_golden as mg
    ref_stubs.install()
    R = ref_stubs.REF_ROOT
    ac = ref_stubs.load_file('mmdet3d/core/utils/array_converter.py', 'mmdet3d.core.utils.array_converter')
    cu = types.ModuleType('mmdet3d.core.utils')
    cu.array_converter, cu.__path__ = ac.array_converter, []
    sys.modules['mmdet3d.core.utils'] = cu
    for n in ('mmdet3d.core.bbox', 'mmdet3d.core.bbox.structures'):
        if n not in sys.modules:
            m = types.ModuleType(n)
            m.__path__ = []
            sys.modules[n] = m
    ref_stubs.load_file('mmdet3d/core/bbox/structures/utils.py', 'mmdet3d.core.bbox.structures.utils')
    spec = importlib.util.spec_from_file_location('mmdet3d.core.points', f'{R}/mmdet3d/core/points/__init__.py',
                                                  submodule_search_locations=[f'{R}/mmdet3d/core/points'])
    m = importlib.util.module_from_spec(spec)
    sys.modules['mmdet3d.core.points'] = m
    spec.loader.exec_module(m)
    ct = ref_stubs.load_file('mmdet3d/models/fusion_layers/coord_transform.py', 'ref_coord_transform')
    g = mg.load_reference()
    g['apply_3d_transformation'] = ct.apply_3d_transformation
    mg.extract(mg.REF + 'models/fusion_layers/point_fusion.py', ['point_sample'], g)
    gen = torch.Generator().manual_seed(70)
    pts = torch.rand(400, 3, generator=gen) * torch.tensor([30.0, 30.0, 3.0]) + torch.tensor([4.0, -15.0, -2.0])
    out = dict(points=pts.numpy())
    for name, (ctype, meta) in FLOW_METAS.items():
        out[f'rev__{name}'] = ct.apply_3d_transformation(pts, ctype, meta, reverse=True).numpy()
    lidar2img = torch.tensor([[6.0294e+02, -7.0791e+02, -1.2275e+01, -1.7094e+02],
                              [1.7678e+02, 8.8088e+00, -7.0794e+02, -1.0257e+02],
                              [9.9998e-01, -1.5283e-03, -5.2907e-03, -3.2757e-01],
                              [0.0, 0.0, 0.0, 1.0]])
    img = torch.randn(1, 5, 46, 153, generator=gen)
    out['ps_img'], out['ps_lidar2img'] = img.numpy(), lidar2img.numpy()
    for name in ('lidar_rsthf', 'lidar_hf_st'):
        meta = FLOW_METAS[name][1]
        for aligned in (True, False):
            feat, valid = g['point_sample'](meta, img, pts, lidar2img, 'LIDAR', pts.new_tensor([0.125, 0.125]),
                                            pts.new_tensor([1.0, 0.5]), False, (46, 153), (46, 153),
                                            aligned=aligned, valid_flag=True)
            out[f'ps__{name}__{int(aligned)}'] = feat.numpy()
            out[f'psvalid__{name}__{int(aligned)}'] = valid.numpy()
    np.savez_compressed(os.path.join(HERE, 'point_sample_flow.npz'), **out)
    print('point_sample flow:', {k: (v.shape, float(np.mean(v != 0))) for k, v in out.items() if k.startswith('ps__')})


if __name__ == '__main__':
    if not os.path.isdir(REF_ROOT):
        sys.exit('reference not mounted; fixtures are committed, nothing to do')
    which = sys.argv[1:] or ['depth_loss', 'configs', 'cfg1', 'bev_spp', 'wide', 'f2v_variants', 'ps_flow']
    torch.set_num_threads(1)
    if 'depth_loss' in which:
        make_depth_loss()
    if 'configs' in which:
        make_configs()
    if 'cfg1' in which:
        make_backbone_cfg1()
    if 'bev_spp' in which:
        make_bev_spp()
    if 'wide' in which:
        make_wide_modules()
    if 'f2v_variants' in which:
        make_f2v_variants()
    if 'ps_flow' in which:
        make_point_sample_flow()
