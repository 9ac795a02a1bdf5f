"""Round-3 golden fixtures, produced by running the REFERENCE's own code on PyTorch-CPU.

Build container only (needs /root/reference):   python tests/golden/make_golden_r03.py
Same rules as make_golden.py / make_golden_r02.py: reference files are executed unmodified from
where they lie (AST lift / ref_stubs); only inputs and the outputs the reference produced are
stored.

  mvpath_*.npz   BASELINE.json configs #4 / #5 end to end: MultiViewDfM.feature_transformation
                  (detectors/multiview_dfm.py:119-268) INCLUDING its neck_3d call (:257-263), with
                  the reference's own OutdoorImVoxelNeck (necks/imvoxel_neck.py, 'mean', F = 1) and
                  DfMNeck (necks/dfm_neck.py, 'concat', F = 2) on a grid with Nz = 12 (what the necks
                  collapse 12 -> 6 -> 3 -> 1).  Narrow variants (C = 8: fp32 parity, eval and
                  training-mode BatchNorm) and wide variants (C = 32, the width the MFMA kernels
                  take: bf16 parity).  Weights are tests/util.synthetic_state_dict(seed) on both
                  sides; features are regenerated from the seed by the test.
  data_geometry.npz  the reference's loaders (datasets/pipelines/loading.py) executed unmodified with
                  stand-ins for file IO only: frame selection and [cur lidar] -> [prev img] folding of
                  LoadMultiViewImageFromFiles, cur2prevs of VideoPipeline.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

# name, views, frames, C, neck type, neck out channels, aggregate, (scale, flip, crop), seed
MV_PATH_CASES = [
    ('mvpath_mean_1f', 5, 1, 8, 'OutdoorImVoxelNeck', 16, 'mean', (None, False, None), 400),
    ('mvpath_concat_2f', 5, 2, 8, 'DfMNeck', 16, 'concat',
     (np.array([0.95, 1.05, 0.95, 1.05], np.float32), True, np.array([3.0, 2.0], np.float32)), 401),
    ('mvpath_wide_mean_1f', 5, 1, 32, 'OutdoorImVoxelNeck', 64, 'mean', (None, False, None), 402),
    ('mvpath_wide_concat_2f', 5, 2, 32, 'DfMNeck', 64, 'concat', (None, False, None), 403),
]
MV_PATH_GRID = dict(n_voxels=(22, 30, 12), voxel_range=[-11.0, -15.0, -3.0, 11.0, 15.0, 3.0],
                    voxel_size=[1.0, 1.0, 0.5], feat_hw=(26, 39), pad=(104, 156))


def mv_path_feats(seed, nv, nf, C):
    """shared by the generator and the tests"""
    gen = torch.Generator().manual_seed(seed)
    hf, wf = MV_PATH_GRID['feat_hw']
    return torch.randn(1, nv * nf, C, hf, wf, generator=gen)


def make_mv_paths():
    import make_golden as g1
    import ref_stubs
    from tests import util
    g = g1.load_reference()
    ref = ref_stubs.load_hot_path_modules()
    nvox, rng_ = MV_PATH_GRID['n_voxels'], MV_PATH_GRID['voxel_range']
    pad = MV_PATH_GRID['pad']
    for name, nv, nf, C, neck_type, cout, agg, (scale, flip, crop), seed in MV_PATH_CASES:
        feats = mv_path_feats(seed, nv, nf, C)
        lidar2img = g1.waymo_like_cameras(nv, nf, seed + 100)
        gen_self = SimpleNamespace(align_corner=False, custom_values=[])

        def grid_anchors(featmap_sizes, device='cpu'):
            a = g['aligned_anchors_single_range'](gen_self, featmap_sizes[0], rng_, 1, sizes=[[0.0, 0.0, 0.0]],
                                                  rotations=[0.0], device=device)
            return [a.reshape(-1, a.size(-1))]

        if neck_type == 'OutdoorImVoxelNeck':
            neck = ref['imvoxel_neck'].OutdoorImVoxelNeck(in_channels=C, out_channels=cout)
        else:
            neck = ref['dfm_neck'].DfMNeck(in_channels=C, out_channels=cout, num_frames=nf)
        neck.load_state_dict(util.synthetic_state_dict(neck, seed + 200))
        self_ = SimpleNamespace(
            anchor_generator=SimpleNamespace(grid_anchors=grid_anchors), n_voxels=list(nvox), valid_sample=True,
            temporal_aggregate=agg, with_backbone_3d=False, with_depth_head=False, with_neck_3d=True, neck_3d=neck)
        meta = {'ori_lidar2img': [m for m in lidar2img], 'input_shape': pad,
                'img_shape': [(pad[0] - 4, pad[1] - 6, 3)] * (nv * nf)}
        if scale is not None:
            meta['scale_factor'] = scale
        if flip:
            meta['flip'] = True
        if crop is not None:
            meta['img_crop_offset'] = crop
        out = {}
        neck.eval()
        with torch.no_grad():
            out['ref_out'] = g['mv_feature_transformation'](self_, feats, [meta], nv, nf)[0].numpy()
        if C < 32:
            # training-mode BatchNorm (batch statistics) as well; the running statistics it updates
            # are stored so the test can check the buffers too
            neck.train()
            with torch.no_grad():
                out['ref_out_train'] = g['mv_feature_transformation'](self_, feats, [meta], nv, nf)[0].numpy()
            sd = neck.state_dict()
            key = next(k for k in sd if k.endswith('running_mean'))
            out['train_running_mean_key'] = np.array(key)
            out['train_running_mean'] = sd[key].numpy()
            self_.with_neck_3d = False
            with torch.no_grad():
                out['ref_volume'] = g['mv_feature_transformation'](self_, feats, [meta], nv, nf)[0].numpy()
        np.savez_compressed(
            os.path.join(HERE, f'{name}.npz'), lidar2img=lidar2img, n_voxels=np.asarray(nvox),
            voxel_range=np.asarray(rng_), voxel_size=np.asarray(MV_PATH_GRID['voxel_size']),
            input_shape=np.asarray(pad), img_shape=np.asarray(meta['img_shape'][0][:2]),
            scale=np.zeros(0, np.float32) if scale is None else scale, flip=np.bool_(flip),
            crop=np.zeros(0, np.float32) if crop is None else crop, aggregate=agg, num_views=nv, num_frames=nf,
            channels=C, neck_type=neck_type, neck_out=cout, seed=seed,
            neck_keys=np.array(list(neck.state_dict().keys())), **out)
        print(name, out['ref_out'].shape, 'nonzero', float((out['ref_out'] != 0).mean()))




# ---------------------------------------------------------------------------------------------
# data_geometry.npz: the reference's LOADERS executed unmodified (datasets/pipelines/loading.py),
# with stand-ins for file IO / image decoding only: LoadMultiViewImageFromFiles.__call__ (frame
# selection + the [cur lidar] -> [prev img] folding, :67-142) and VideoPipeline.__call__
# (cur2prevs, :419-546).  Stored: the synthetic poses that went in, the matrices that came out.
# ---------------------------------------------------------------------------------------------
def _load_reference_loading():
    import types
    import ref_stubs

    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        m.__dict__.update(attrs)
        if not hasattr(m, '__path__'):
            m.__path__ = []
        sys.modules[name] = m
        return m

    class FileClient:
        def __init__(self, **kw):
            pass

        def get(self, name):
            return name

    class _Registry(ref_stubs.Registry):
        def split_scope_key(self, key):
            return None, key

    class _Compose:  # the reference's Compose resolves dict transforms through mmcv; callables pass through
        def __init__(self, transforms):
            self.transforms = list(transforms)

        def __call__(self, data):
            for t in self.transforms:
                data = t(data)
            return data

    reg = _Registry()
    mod('mmcv', FileClient=FileClient, imfrombytes=lambda b, flag=None: np.zeros((8, 12, 3), np.uint8),
        impad=lambda img, shape=None, pad_val=0: img)
    mod('pyquaternion', Quaternion=None)
    mod('mmdet')
    mod('mmdet.datasets')
    mod('mmdet.datasets.pipelines', LoadAnnotations=object, LoadImageFromFile=object)
    mod('mmdet3d')
    mod('mmdet3d.core')
    mod('mmdet3d.core.points', BasePoints=object, get_points_type=None)
    mod('mmdet3d.datasets')
    mod('mmdet3d.datasets.builder', PIPELINES=reg)
    mod('mmdet3d.datasets.pipelines')
    mod('mmdet3d.datasets.pipelines.compose', Compose=_Compose)
    return ref_stubs.load_file('mmdet3d/datasets/pipelines/loading.py', 'mmdet3d.datasets.pipelines.loading')


def _rigid(rng, yaw_deg, t):
    a = np.radians(yaw_deg)
    m = np.eye(4)
    m[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
    m[:3, 3] = t
    return m


def make_data_geometry():
    import make_golden as g1
    loading = _load_reference_loading()
    rng = np.random.RandomState(11)
    out = {}
    # --- multi-view: 4 frames in the info (current + 3 earlier), 5 views each
    nv, nframes = 5, 4
    lidar2img = [m.astype(np.float64) + rng.uniform(-1e-3, 1e-3, (4, 4)) for m in g1.waymo_like_cameras(nv, nframes, 21)]
    ego2global = [_rigid(rng, rng.uniform(-20, 20), rng.uniform(-30, 30, 3)) for _ in range(nframes)]
    ego2global[2] = ego2global[2][:3]  # a 3x4 pose: the loader pads it (loading.py:126-131)
    out.update(mv_lidar2img=np.stack(lidar2img), mv_ego2global_0=ego2global[0], mv_ego2global_1=ego2global[1],
               mv_ego2global_2=ego2global[2], mv_ego2global_3=ego2global[3])
    for name, nref, test_mode in (('mv_test_2ref', 2, True), ('mv_test_1ref', 1, True), ('mv_test_5ref', 5, True),
                                  ('mv_train_2ref', 2, False)):
        np.random.seed(5)
        res = dict(img_filename=[f'f{f}_v{v}.jpg' for f in range(nframes) for v in range(nv)],
                   lidar2img=[m.copy() for m in lidar2img], ego2global=[m.copy() for m in ego2global])
        res = loading.LoadMultiViewImageFromFiles(num_views=nv, num_ref_frames=nref, test_mode=test_mode)(res)
        out[name + '_lidar2img'] = np.stack(res['lidar2img'])
        out[name + '_ori_lidar2img'] = np.stack(res['ori_lidar2img'])
        out[name + '_frames'] = np.array([int(f.split('_')[0][1:]) for f in res['img_filename'][::nv]])
    # --- video (KITTI student): current frame + 3 sweeps with cam2global in the infos
    cam2global = _rigid(rng, 3.0, [1.0, 2.0, 0.5]).astype(np.float32)
    sweeps = [dict(data_path=f's{i}.png', cam2global=_rigid(rng, 3.0 + 2 * i, [1.0 - 1.1 * i, 2.0, 0.5])[:3 + (i % 2)])
              for i in range(1, 4)]
    out.update(video_cam2global=cam2global, video_sweep_cam2global_1=sweeps[0]['cam2global'],
               video_sweep_cam2global_2=sweeps[1]['cam2global'], video_sweep_cam2global_3=sweeps[2]['cam2global'])

    def load(results):  # stand-in for LoadImageFromFileMono3D & co: passes cam2global through
        r = dict(results)
        r['img'] = np.zeros((4, 4, 3), np.float32)
        if 'cam2global' not in r:
            r['cam2global'] = r['img_info']['cam2global']
        return r
    for name, nref in (('video_1ref', 1), ('video_3ref', 3)):
        pipe = loading.VideoPipeline([load], num_ref_imgs=nref, random=False)
        res = pipe(dict(img_info=dict(filename='cur.png', cam2global=cam2global, sweeps=sweeps)))
        out[name + '_cur2prevs'] = res['cur2prevs']
    np.savez_compressed(os.path.join(HERE, 'data_geometry.npz'), **out)
    print('data_geometry:', {k: v.shape for k, v in out.items() if 'lidar2img' in k or 'cur2prevs' in k or 'frames' in k})


if __name__ == '__main__':
    which = sys.argv[1:] or ['mv', 'data']
    if 'mv' in which:
        make_mv_paths()
    if 'data' in which:
        make_data_geometry()
