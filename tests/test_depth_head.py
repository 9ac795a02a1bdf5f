"""DepthHead.forward: oracle vs fixtures produced by the reference module (CPU)
and HIP vs oracle / fixtures (GPU).
Bar: upsampled volume bit-exact; softmax rtol 2e-6 (+1e-9 abs) and expectation
rtol 5e-6 -- the only difference is expf vs torch's vectorised Sleef exp."""
import glob
import importlib
import os

import numpy as np
import pytest
import torch

from oracle import dfm_oracle as orc
from tests import util

SOFT_TOL = dict(rtol=2e-6, atol=1e-9)
PRED_TOL = dict(rtol=5e-6, atol=0)


def cases():
    return sorted(glob.glob(os.path.join(util.GOLDEN, 'depth_head_*.npz')))


@pytest.mark.parametrize('path', cases(), ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_vs_reference_module(path):
    z = np.load(path)
    vol, soft, pred = orc.depth_head(z['x'], z['depth_samples'])
    assert np.array_equal(util.bits(vol), util.bits(z['ref_vol']))
    np.testing.assert_allclose(soft, z['ref_soft'], **SOFT_TOL)
    np.testing.assert_allclose(pred, z['ref_pred'], **PRED_TOL)
    np.testing.assert_allclose(z['ref_soft'].sum(2), 1.0, rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('path', cases(), ids=lambda p: os.path.basename(p)[:-4])
def test_hip_vs_reference_fixture(path):
    pkg = importlib.import_module('depth-from-motion_amd')
    z = np.load(path)
    vol, soft, pred = pkg.depth_head_forward(torch.from_numpy(z['x']).cuda(),
                                             torch.from_numpy(z['depth_samples']))
    torch.cuda.synchronize()
    assert np.array_equal(util.bits(vol.cpu().numpy()), util.bits(z['ref_vol']))
    np.testing.assert_allclose(soft.cpu().numpy(), z['ref_soft'], **SOFT_TOL)
    np.testing.assert_allclose(pred.cpu().numpy(), z['ref_pred'], **PRED_TOL)


@pytest.mark.gpu
def test_hip_config_k_shape_properties():
    """(1,1,72,80,320) -> 288x320x1280: checked on sub-columns against the oracle,
    plus softmax sums to one and the expectation stays inside the depth range."""
    pkg = importlib.import_module('depth-from-motion_amd')
    rng = np.random.RandomState(0)
    x = (rng.randn(1, 1, 72, 80, 320) * 4).astype(np.float32)
    ds = np.array([(k + 0.5) * (57.6 / 288) + 2 for k in range(288)], np.float32)
    vol, soft, pred = pkg.depth_head_forward(torch.from_numpy(x).cuda(), torch.from_numpy(ds))
    assert vol.shape == (1, 1, 288, 320, 1280) and pred.shape == (1, 1, 320, 1280)
    s = soft.sum(2)
    assert torch.allclose(s, torch.ones_like(s), rtol=1e-5, atol=0)
    assert float(pred.min()) >= 2.0 and float(pred.max()) <= 59.6
    # a 20-row strip of the cost volume reproduces rows 0..76 of the output exactly
    # only if those rows interpolate inside the strip -> compare against the oracle on
    # the full input but a slice of outputs
    ovol, osoft, opred = orc.depth_head(x[:, :, :, :, :], ds)
    assert np.array_equal(util.bits(vol.cpu().numpy()), util.bits(ovol))
    np.testing.assert_allclose(soft.cpu().numpy(), osoft, **SOFT_TOL)
    np.testing.assert_allclose(pred.cpu().numpy(), opred, **PRED_TOL)


@pytest.mark.gpu
def test_hip_bf16_storage():
    pkg = importlib.import_module('depth-from-motion_amd')
    z = np.load(cases()[0])
    x16 = orc.bf16_round(z['x'])
    vol, soft, pred = pkg.depth_head_forward(torch.from_numpy(x16).cuda().bfloat16(),
                                             torch.from_numpy(z['depth_samples']))
    ovol, _, _ = orc.depth_head(x16, z['depth_samples'])
    # volume: one bf16 rounding of the exact fp32 interpolation
    v16 = orc.bf16_round(ovol)
    assert np.array_equal(util.bits(vol.float().cpu().numpy()), util.bits(v16))
    # softmax of the bf16-STORED logits (what F.softmax of a bf16 tensor sees),
    # computed in fp32, stored as bf16; expectation over the stored probabilities
    e = np.exp(v16 - v16.max(2, keepdims=True))
    p = e / e.sum(2, keepdims=True)
    got_soft = soft.float().cpu().numpy()
    np.testing.assert_allclose(got_soft, orc.bf16_round(p), rtol=8e-3, atol=1e-9)  # <= 2 bf16 ulp
    exp_pred = (got_soft * z['depth_samples'][None, None, :, None, None]).sum(2)
    np.testing.assert_allclose(pred.float().cpu().numpy(), exp_pred, rtol=5e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('scale,shape', [(2, (1, 1, 5, 4, 7)), (3, (2, 1, 4, 3, 5)), (1, (1, 1, 6, 5, 9))])
def test_hip_one_pixel_per_lane_path_vs_oracle(scale, shape):
    """row lengths that are not a multiple of 4 take the scalar-store instantiation"""
    pkg = importlib.import_module('depth-from-motion_amd')
    rng = np.random.RandomState(scale)
    x = (rng.randn(*shape) * 3).astype(np.float32)
    ds = np.linspace(2.0, 59.6, scale * shape[2]).astype(np.float32)
    vol, soft, pred = pkg.depth_head_forward(torch.from_numpy(x).cuda(), torch.from_numpy(ds), scale)
    ovol, osoft, opred = orc.depth_head(x, ds, scale)
    assert np.array_equal(util.bits(vol.cpu().numpy()), util.bits(ovol))
    np.testing.assert_allclose(soft.cpu().numpy(), osoft, **SOFT_TOL)
    np.testing.assert_allclose(pred.cpu().numpy(), opred, **PRED_TOL)
