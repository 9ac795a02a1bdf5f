"""CPU: oracle point_sample / multi-view lifting against the reference's golden
vectors and against fixtures produced by the reference code (bit-exact)."""
import glob
import os

import numpy as np
import pytest

from oracle import dfm_oracle as orc
from tests import util


def mv_cases():
    return sorted(glob.glob(os.path.join(util.GOLDEN, 'mv_*.npz')))


def run_oracle_mv(z):
    return orc.mv_feature_transformation(z['feats'][0], z['points'], z['lidar2img'], z['n_voxels'],
                                         int(z['num_views']), int(z['num_frames']),
                                         z['input_shape'], z['img_shape'], z['scale'],
                                         bool(z['flip']), z['crop'], str(z['aggregate']))


@pytest.mark.parametrize('path', mv_cases(), ids=lambda p: os.path.basename(p)[:-4])
def test_mv_feature_transformation_bitexact(path):
    z = np.load(path)
    out = run_oracle_mv(z)
    assert np.array_equal(util.bits(out), util.bits(z['ref_out'][0]))
    # the fixture exercises the reduction: some voxels are seen by >= 2 views
    assert 0.2 < (z['ref_out'] != 0).mean() < 0.9


def test_point_sample_reproduces_the_reference_tests_golden_vector():
    """tests/test_models/test_fusion/test_point_fusion.py:13-40 of the reference"""
    z = np.load(os.path.join(util.GOLDEN, 'helpers.npz'))
    img = (np.arange(370 * 1224, dtype=np.float32).reshape(1, 370, 1224) /
           np.float32(370 * 1224)).astype(np.float32)
    out = orc.point_sample(img, z['ps_points'], z['ps_lidar2img'], pad_shape=(370, 1224),
                           aligned=True).reshape(-1)
    assert np.array_equal(util.bits(out), util.bits(z['ps_out'].reshape(-1)))  # reference run here
    np.testing.assert_allclose(out, z['ps_expected'], rtol=1e-4)               # reference's own bar


def test_helper_golden_vectors_of_the_reference_tests():
    """points_cam2img (test_box3d.py:1653-1680) and points_img2cam (test_utils.py:186-193):
    the fixture stores what the reference functions returned here; they meet the
    reference tests' own tolerances, and the fma-chain restatement matches bitwise."""
    z = np.load(os.path.join(util.GOLDEN, 'helpers.npz'))
    np.testing.assert_allclose(z['cam2img_out'], z['cam2img_expected'], rtol=1e-3)
    np.testing.assert_allclose(z['img2cam_out'], z['img2cam_expected'], atol=1e-3)
    # restate points_cam2img with the k-ordered fma chain, in numpy float64->float32 steps
    pts, P = z['cam2img_points'], z['cam2img_proj']

    def fma(a, b, c):
        return np.float32(np.float64(a) * np.float64(b) + np.float64(c))

    out = np.empty((5, 2), np.float32)
    for i in range(5):
        p4 = [pts[i, 0], pts[i, 1], pts[i, 2], np.float32(1)]
        rows = []
        for r in range(3):
            acc = np.float32(p4[0] * P[r, 0])
            for k in (1, 2, 3):
                acc = fma(p4[k], P[r, k], acc)
            rows.append(acc)
        out[i] = [rows[0] / rows[2], rows[1] / rows[2]]
    assert np.array_equal(util.bits(out), util.bits(z['cam2img_out']))


VS_CASES = [('plain', dict(scale=(1.0, 1.0), crop=(0.0, 0.0), flip=False)),
            ('aug', dict(scale=(0.95, 1.05), crop=(3.0, 2.0), flip=True))]


@pytest.mark.parametrize('aligned', [True, False])
@pytest.mark.parametrize('tag,kw', VS_CASES)
def test_voxel_sample_oracle_bitexact_vs_reference(tag, kw, aligned):
    z = np.load(os.path.join(util.GOLDEN, 'voxel_sample.npz'))
    out = orc.voxel_sample(z['vox'], z['voxel_range'], z['voxel_size'], z['depth_samples'],
                           z['proj_inv'], 4, img_pad_shape=(104, 156), img_shape=(100, 150),
                           aligned=aligned, **kw)
    ref = z[f'out_{tag}_{"tri" if aligned else "near"}']
    assert np.array_equal(util.bits(out), util.bits(ref))


def test_reverse_3d_flow_vs_reference_apply_3d_transformation():
    """host logic of point_sample's first step (point_fusion.py:57-58): the product's
    ``_reverse_3d_flow`` against ``apply_3d_transformation(..., reverse=True)`` of the reference
    (tests/golden/point_sample_flow.npz, make_golden_r02.py) for LIDAR / CAMERA / DEPTH points and
    every flow entry; exact without a rotation, 1 ulp-level with one (BLAS product restated)."""
    import importlib
    import torch
    sys_path_pkg = importlib.import_module('depth-from-motion_amd.point_sample')
    from tests.golden.make_golden_r02 import FLOW_METAS
    z = np.load(os.path.join(util.GOLDEN, 'point_sample_flow.npz'))
    pts = torch.from_numpy(z['points'])
    for name, (ctype, meta) in FLOW_METAS.items():
        got = sys_path_pkg._reverse_3d_flow(pts, ctype, meta).numpy()
        ref = z[f'rev__{name}']
        if 'R' in meta['transformation_3d_flow']:
            np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-6)
        else:
            assert np.array_equal(util.bits(got), util.bits(ref)), name
    assert sys_path_pkg._reverse_3d_flow(pts, 'LIDAR', {}) is pts
