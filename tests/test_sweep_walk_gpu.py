"""The depth-walking strided-sweep kernel (sweep_cltw_kernel, ``kernel=5``; what strided fp32 sweeps with
whole 32-channel passes and 32-point tiles take by default): bit-exact against the CPU oracle and against
the per-plane kernel (``kernel=4``), through both entry points (NCHW maps packed by the library, NHWC maps
sampled in place), over depth runs that end inside a chunk / a footprint pair, footprints that leave the
map, and several channel passes.  Reference: dfm_backbone.py:217-314 with cost_sample_factor 4."""
import importlib

import numpy as np
import pytest
import torch

from tests import util
from oracle import dfm_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def pkg():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return importlib.import_module('depth-from-motion_amd')


def _case(C, H, W, D, csf, seed, far_pose=False):
    rng = np.random.RandomState(seed)
    cur = rng.randn(1, C, H, W).astype(np.float32)
    prev = rng.randn(1, C, H, W).astype(np.float32)
    P = util.KITTI_P2[None].copy()
    T = util.random_poses(1, seed=seed + 3)
    if far_pose:  # a large lateral motion: prev footprints leave the map on one side
        T[0, 0, 3] = 6.0
    depths = util.depth_planes(D)
    return cur, prev, depths, P, T, (1, csf, P, T, (375, 1242), False, (0, 55 if H == 320 else 0), 1.0)


def _hip(pkg, cur, prev, depths, args, kernel, nhwc=False):
    sweep = importlib.import_module('depth-from-motion_amd.plane_sweep')
    dev = torch.device('cuda:0')
    c = torch.from_numpy(cur).to(dev)
    p = torch.from_numpy(prev).to(dev)
    if nhwc:
        c = c.contiguous(memory_format=torch.channels_last)
        p = p.contiguous(memory_format=torch.channels_last)
    fsf, csf, P, T, img_shape, flip, crop, scale = args
    call = lambda: pkg.build_dfm_cost(c, p, torch.from_numpy(depths).to(dev), fsf, csf, torch.from_numpy(P),
                                      torch.from_numpy(T), img_shape, flip, crop, scale)
    if nhwc:  # (the in-place NHWC entry point takes no launch options)
        out = call()
    else:
        with sweep.launch_options(kernel=kernel):
            out = call()
    torch.cuda.synchronize()
    assert pkg._capi.lib().dfm_plane_sweep_last_kernel() == 4
    return out.cpu().numpy()


@pytest.mark.parametrize('C,H,W,D,far', [(32, 64, 256, 6, False), (32, 64, 256, 7, False), (32, 64, 256, 51, False),
                                         (64, 32, 128, 9, False), (32, 64, 256, 12, True), (96, 32, 128, 2, False),
                                         (32, 64, 256, 1, False)])
@pytest.mark.parametrize('nhwc', [False, True], ids=['nchw', 'nhwc'])
def test_walk_matches_oracle_and_per_plane_kernel(pkg, C, H, W, D, far, nhwc):
    cur, prev, depths, P, T, args = _case(C, H, W, D, 4, seed=C + D, far_pose=far)
    ref = orc.build_dfm_cost(cur, prev, depths, 1, 4, P, util.host_inverse(P), T, (375, 1242), False, args[6], 1.0)
    got = _hip(pkg, cur, prev, depths, args, kernel=5, nhwc=nhwc)
    assert got.shape == (1, 2 * C, D, H // 4, W // 4)
    assert np.array_equal(util.bits(got), util.bits(ref))
    if not nhwc:
        per_plane = _hip(pkg, cur, prev, depths, args, kernel=4)
        assert np.array_equal(util.bits(got), util.bits(per_plane))


def test_walk_config_k_planes(pkg):
    """config K's geometry (320x1280 maps, crop (0, 55), 72 planes) at C=32 for one sample: every plane against
    the per-plane kernel, planes 0 / 23 / 24 / 47 / 71 (chunk ends) against the oracle"""
    cur, prev, depths, P, T, args = _case(32, 320, 1280, 72, 4, seed=5)
    got = _hip(pkg, cur, prev, depths, args, kernel=5)
    per_plane = _hip(pkg, cur, prev, depths, args, kernel=4)
    assert np.array_equal(util.bits(got), util.bits(per_plane))
    sel = [0, 23, 24, 47, 71]
    ref = orc.build_dfm_cost(cur, prev, depths[sel], 1, 4, P, util.host_inverse(P), T, (375, 1242), False,
                             (0, 55), 1.0)
    assert np.array_equal(util.bits(got[:, :, sel]), util.bits(ref))


def test_walk_batched(pkg):
    rng = np.random.RandomState(11)
    B, C, H, W, D = 3, 32, 32, 128, 5
    cur = rng.randn(B, C, H, W).astype(np.float32)
    prev = rng.randn(B, C, H, W).astype(np.float32)
    P = np.stack([util.KITTI_P2] * B).copy()
    P[1, 0, 2] += 3.5
    T = util.random_poses(B, seed=4)
    depths = util.depth_planes(D)
    args = (1, 4, P, T, (375, 1242), False, (0, 0), 1.0)
    ref = orc.build_dfm_cost(cur, prev, depths, 1, 4, P, util.host_inverse(P), T, (375, 1242), False, (0, 0), 1.0)
    got = _hip(pkg, cur, prev, depths, args, kernel=5)
    assert np.array_equal(util.bits(got), util.bits(ref))
