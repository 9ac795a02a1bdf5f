"""patch_reference() against the REFERENCE's own detector and module files (build container only).

tests/golden/ref_stubs.load_detectors() executes mmdet3d/models/detectors/dfm.py and
multiview_dfm.py (and the path's module files, point_fusion.py, coord_transform.py) unmodified
from /root/reference, with stand-ins only for what lies outside the path (mmcv / mmdet base
classes, the 2-D backbone, the detection heads).  The configs are the reference's real files
(configs/dfm/*.py, exec'd with their _base_ chain), built through the reference's own
``DfM.__init__`` / ``MultiViewDfM.__init__`` -> ``build_backbone / build_neck / build_head``.

What is checked: after ``patch_reference()`` those constructors produce THIS package's modules,
the detector's attribute injection (dfm.py:82-100) lands on them with the reference's own values,
a state_dict of the unpatched reference detector loads strictly into the patched one (the
checkpoint contract of SURVEY.md 8b at detector level), the functions the reference looks up by
name are rebound, and the patched voxel necks reproduce the reference necks on CPU inputs
(torch convolutions; the HIP-only stages raise instead of falling back).  The GPU side of the same
configs is tests/test_path_parity_gpu.py (reference-generated fixtures).
"""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from tests import util

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'configs', 'dfm')),
                                reason='reference not mounted (GPU box)')

KITTI = 'dfm_r34_1x8_kitti-3d-3class.py'
WAYMO = 'multiview-dfm_r101_dcn_2x16_waymoD5-3d-3class_camsync.py'
WAYMO10 = 'multiview-dfm_r101_dcn_2x16_waymoD5-3d-3class_camsync_10sweeps.py'
PATH_TYPES = ('DfMBackbone', 'FrustumToVoxel', 'DepthHead', 'OutdoorImVoxelNeck', 'DfMNeck', 'BEVHourglass',
              'SPPUNetNeck')


@pytest.fixture()
def ref_env():
    """the reference's files loaded under the stubs; sys.modules restored afterwards (other tests
    rely on mmdet3d NOT being importable)"""
    before = dict(sys.modules)
    sys.path.insert(0, util.GOLDEN)
    import make_golden_r02 as g
    import ref_stubs
    reg, mods = ref_stubs.load_detectors()
    yield reg, mods, g
    for k in list(sys.modules):
        if k not in before and k.split('.')[0] in ('mmcv', 'mmdet', 'mmdet3d', 'ref_depth_head'):
            del sys.modules[k]
    sys.path.remove(util.GOLDEN)


def _model_cfg(g, name):
    cfg = g.exec_config(os.path.join(REF, 'configs', 'dfm', name))['model']
    cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in cfg.items()}
    kind = cfg.pop('type')
    return kind, cfg


def test_dfm_detector_builds_this_package_through_the_reference_constructor(ref_env):
    reg, mods, g = ref_env
    pkg = importlib.import_module('depth-from-motion_amd')
    ours = importlib.import_module('depth-from-motion_amd.registry').registered()
    kind, cfg = _model_cfg(g, KITTI)
    assert kind == 'DfM'
    ref_det = mods['dfm'].DfM(**_model_cfg(g, KITTI)[1])
    for name in ('neck', 'backbone_stereo', 'depth_head', 'feature_transformation', 'backbone_3d'):
        assert type(getattr(ref_det, name)).__module__.startswith('mmdet3d.'), name

    report = pkg.patch_reference()
    assert set(PATH_TYPES) <= set(report['modules'])
    assert 'mmdet3d.models.backbones.dfm_backbone.build_dfm_cost' in report['functions']
    assert 'mmdet3d.models.detectors.multiview_dfm.point_sample' in report['functions']
    assert 'mmdet3d.models.fusion_layers.point_fusion.point_sample' in report['functions']
    assert mods['dfm_backbone'].build_dfm_cost is pkg.build_dfm_cost
    assert mods['multiview_dfm'].point_sample is pkg.point_sample
    assert mods['multiview_dfm'].voxel_sample is pkg.voxel_sample
    assert mods['point_fusion'].point_sample is pkg.point_sample

    det = mods['dfm'].DfM(**cfg)
    for attr, type_name in (('neck', 'SPPUNetNeck'), ('backbone_stereo', 'DfMBackbone'),
                            ('depth_head', 'DepthHead'), ('feature_transformation', 'FrustumToVoxel'),
                            ('backbone_3d', 'BEVHourglass')):
        assert type(getattr(det, attr)) is ours[type_name], attr
    # dfm.py:56-64,82-100: the detector's own injection code ran against our modules
    assert det.feature_transformation.cat_img_feature == det.neck.cat_img_feature
    assert det.feature_transformation.in_sem_channels == det.neck.sem_channels[-1]
    assert det.backbone_stereo.downsampled_depth is det.downsampled_depth
    assert det.depth_head.depth_samples is det.depth
    assert det.depth_head.downsample_factor == 4
    assert det.feature_transformation.depth_cfg is cfg['depth_cfg'] or \
        det.feature_transformation.depth_cfg == cfg['depth_cfg']
    assert det.feature_transformation.coordinates_3d is det.coordinates_3d
    # ... and the values equal what this package's own injection helper computes
    geo = importlib.import_module('depth-from-motion_amd.geometry')
    ds, depth = geo.prepare_depth(cfg['depth_cfg'])
    assert torch.equal(ds, det.downsampled_depth) and torch.equal(depth, det.depth)
    assert torch.equal(geo.prepare_coordinates_3d(cfg['voxel_cfg']), det.coordinates_3d)

    # checkpoint contract: the unpatched reference detector's state_dict loads strictly
    sd = ref_det.state_dict()
    assert len(sd) > 100
    missing, unexpected = det.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    for k, v in det.state_dict().items():
        assert v.shape == sd[k].shape, k

    # the HIP-only stages refuse CPU tensors instead of falling back
    meta = dict(ori_cam2img=util.KITTI_P2, cur2prevs=torch.eye(4)[None], ori_shape=(375, 1242, 3),
                pad_shape=(32, 64, 3), crop_offset=[0, 0], flip=False, scale_factor=[1.0])
    with pytest.raises(RuntimeError, match='no CPU path'):
        det.backbone_stereo(torch.zeros(1, 32, 32, 64), torch.zeros(1, 32, 32, 64), [meta])


@pytest.mark.parametrize('name,neck_type,frames', [(WAYMO, 'OutdoorImVoxelNeck', 1), (WAYMO10, 'DfMNeck', 2)])
def test_multiview_detector_is_routed_and_its_neck_matches_the_reference(ref_env, name, neck_type, frames):
    reg, mods, g = ref_env
    pkg = importlib.import_module('depth-from-motion_amd')
    ours = importlib.import_module('depth-from-motion_amd.registry').registered()
    kind, cfg = _model_cfg(g, name)
    assert kind == 'MultiViewDfM'
    ref_det = mods['multiview_dfm'].MultiViewDfM(**_model_cfg(g, name)[1])
    ref_method = mods['multiview_dfm'].MultiViewDfM.feature_transformation
    assert type(ref_det.neck_3d).__name__ == neck_type and type(ref_det.neck_3d) is not ours[neck_type]

    report = pkg.patch_reference()
    assert 'MultiViewDfM.feature_transformation' in report['methods']
    det = mods['multiview_dfm'].MultiViewDfM(**cfg)
    assert type(det.neck_3d) is ours[neck_type]
    assert mods['multiview_dfm'].MultiViewDfM.feature_transformation is pkg.MultiViewDfMMixin.feature_transformation
    assert mods['multiview_dfm'].MultiViewDfM.feature_transformation is not ref_method
    assert det.n_voxels == [220, 300, 12] and det.temporal_aggregate == ('concat' if frames == 2 else 'mean')

    sd = util.synthetic_state_dict(ref_det.neck_3d, 90 + frames)
    ref_det.neck_3d.load_state_dict(sd, strict=True)
    det.neck_3d.load_state_dict(sd, strict=True)
    full = ref_det.state_dict()
    missing, unexpected = det.load_state_dict(full, strict=True)
    assert not missing and not unexpected

    gen = torch.Generator().manual_seed(5)
    x = torch.randn(1, 64 * frames, 6, 5, 12, generator=gen)
    for mode in ('eval', 'train'):
        getattr(ref_det.neck_3d, mode)()
        getattr(det.neck_3d, mode)()
        with torch.no_grad():
            want = ref_det.neck_3d(x)[0]
            got = det.neck_3d(x)[0]
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-4, atol=1e-5)
    # the lifting stage is HIP-only: the patched detector method raises on CPU features
    meta = {'ori_lidar2img': [np.eye(4, dtype=np.float32)] * (5 * frames), 'input_shape': (64, 96),
            'img_shape': [(64, 96, 3)] * (5 * frames)}
    with pytest.raises(RuntimeError, match='no CPU path'):
        det.feature_transformation(torch.zeros(1, 5 * frames, 64, 16, 24), [meta], 5, frames)
