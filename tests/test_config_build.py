"""The reference's own config files build through this package's registry.

tests/golden/configs_dfm.json holds the ``model`` sub-dicts of configs/dfm/*.py, extracted by
exec'ing the config files (with their _base_ chain) in tests/golden/make_golden_r02.py -- nothing
is retyped by hand here.  When /root/reference is mounted (build container) the extraction is
re-run and compared with the committed fixture.  (reference: mmdet3d/models/detectors/dfm.py:30-112,
multiview_dfm.py:17-65, tests/test_runtime/test_config.py:20-52)"""
import importlib
import json
import os
import sys
import types

import pytest
import torch

from tests import util

KITTI = 'dfm_r34_1x8_kitti-3d-3class.py'
WAYMO = 'multiview-dfm_r101_dcn_2x16_waymoD5-3d-3class_camsync.py'
WAYMO10 = 'multiview-dfm_r101_dcn_2x16_waymoD5-3d-3class_camsync_10sweeps.py'


@pytest.fixture(scope='module')
def cfgs():
    with open(os.path.join(util.GOLDEN, 'configs_dfm.json')) as f:
        return json.load(f)


@pytest.fixture(scope='module')
def pkg():
    return importlib.import_module('depth-from-motion_amd')


def test_fixture_is_what_the_reference_config_files_say(cfgs):
    if not os.path.isdir('/root/reference/configs/dfm'):
        pytest.skip('reference not mounted (GPU box): the committed extraction is used')
    sys.path.insert(0, util.GOLDEN)
    import make_golden_r02 as g
    for name, entry in cfgs.items():
        model = g.exec_config(os.path.join('/root/reference/configs/dfm', name))['model']
        for key, val in entry['model'].items():
            assert g._jsonable(model[key]) == val, (name, key)


def test_kitti_config_builds_the_stereo_path_with_injected_attributes(pkg, cfgs):
    model = cfgs[KITTI]['model']
    assert model['type'] == 'DfM'
    path = pkg.DfMStereoPath(model)
    mods = importlib.import_module('depth-from-motion_amd.modules')
    assert isinstance(path.neck, mods.SPPUNetNeck)
    assert isinstance(path.backbone_stereo, mods.DfMBackbone)
    assert isinstance(path.depth_head, mods.DepthHead)
    assert isinstance(path.feature_transformation, mods.FrustumToVoxel)
    assert isinstance(path.backbone_3d, mods.BEVHourglass)
    # dfm.py:56-64: the neck's settings flow into feature_transformation
    assert path.feature_transformation.cat_img_feature is True
    assert path.feature_transformation.in_sem_channels == 32
    # dfm.py:82-100: attribute injection
    assert path.backbone_stereo.downsampled_depth.shape == (72,)
    assert path.depth_head.depth_samples.shape == (288,)
    assert path.depth_head.downsample_factor == 4
    assert path.feature_transformation.depth_cfg == model['depth_cfg']
    assert path.feature_transformation.coordinates_3d.shape == (20, 304, 288, 3)
    assert path.depth_head.depth_loss_type == 'balanced_focal'
    assert abs(float(path.backbone_stereo.downsampled_depth[0]) - (2 + 0.5 * 4 * 0.2)) < 1e-6
    # parameter counts of SURVEY.md 8a
    assert sum(p.numel() for p in path.backbone_stereo.parameters()) == 1313344
    # BEVHourglass takes Cv*Nz = 32*5 channels after the (4,1,1) pooling of 20 z-cells
    assert path.backbone_3d.compress_conv.conv.in_channels == 160
    assert path.neck.lastconv[1].out_channels == path.backbone_stereo.in_channels == 32


@pytest.mark.parametrize('name,neck_type,frames', [(WAYMO, 'OutdoorImVoxelNeck', 1),
                                                  (WAYMO10, 'DfMNeck', 2)])
def test_multiview_configs_build_the_voxel_path(pkg, cfgs, name, neck_type, frames):
    model = cfgs[name]['model']
    assert model['type'] == 'MultiViewDfM'
    path = pkg.MultiViewVoxelPath(model)
    assert type(path.neck_3d).__name__ == neck_type
    assert path.n_voxels == [220, 300, 12]  # round(110/0.5): the config comment's "240" is wrong
    assert path.temporal_aggregate == ('concat' if frames == 2 else 'mean')
    n = sum(p.numel() for p in path.neck_3d.parameters())
    assert n == (15932160 if frames == 2 else 7523328)


def test_patch_reference_routes_a_mmdet3d_installation_to_this_package(pkg, monkeypatch):
    """No mmdet3d here: a skeleton with the reference's module / attribute names stands in, so
    the patch points themselves are exercised (what INTEGRATION.md tells a maintainer to call)."""
    with pytest.raises(ImportError):
        pkg.patch_reference()

    class Registry:
        def __init__(self):
            self.module_dict = {}

        def register_module(self, name=None, force=False, module=None):
            assert force
            self.module_dict[name] = module

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__path__ = []
        monkeypatch.setitem(sys.modules, name, m)
        return m

    def ref_fn(*a, **k):
        raise AssertionError('the reference function must have been replaced')

    class RefMultiViewDfM:
        def feature_transformation(self, *a, **k):
            raise AssertionError('not replaced')

    reg = Registry()
    mod('mmdet3d')
    mod('mmdet3d.models')
    mod('mmdet3d.models.builder', MODELS=reg)
    bb = mod('mmdet3d.models.backbones')
    bbf = mod('mmdet3d.models.backbones.dfm_backbone', build_dfm_cost=ref_fn)
    mod('mmdet3d.models.fusion_layers', point_sample=ref_fn, voxel_sample=ref_fn)
    pf = mod('mmdet3d.models.fusion_layers.point_fusion', point_sample=ref_fn, voxel_sample=ref_fn)
    mod('mmdet3d.models.detectors')
    det = mod('mmdet3d.models.detectors.multiview_dfm', point_sample=ref_fn, voxel_sample=ref_fn,
              MultiViewDfM=RefMultiViewDfM)
    report = pkg.patch_reference()
    for name in ('DfMBackbone', 'FrustumToVoxel', 'DepthHead', 'OutdoorImVoxelNeck', 'DfMNeck',
                 'BEVHourglass', 'SPPUNetNeck'):
        assert reg.module_dict[name] is importlib.import_module('depth-from-motion_amd.registry').registered()[name]
        assert name in report['modules']
    assert bbf.build_dfm_cost is pkg.build_dfm_cost
    assert pf.point_sample is pkg.point_sample and pf.voxel_sample is pkg.voxel_sample
    assert det.point_sample is pkg.point_sample
    assert det.MultiViewDfM.feature_transformation is pkg.MultiViewDfMMixin.feature_transformation
    assert 'MultiViewDfM.feature_transformation' in report['methods']
    assert bb is sys.modules['mmdet3d.models.backbones']


# MultiViewVoxelPath end to end (lifting -> neck_3d, BASELINE configs #4 / #5) against the reference's
# detector method + necks: tests/test_path_parity_gpu.py (fixtures of tests/golden/make_golden_r03.py)
