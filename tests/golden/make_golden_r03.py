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


if __name__ == '__main__':
    make_mv_paths()
