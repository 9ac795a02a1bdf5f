"""Shared helpers for the tests (synthetic cameras / poses, fixture loading)."""
import math
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

KITTI_P2 = np.array([[721.5377, 0, 609.5593, 44.85728], [0, 721.5377, 172.854, 0.2163791],
                     [0, 0, 1, 0.002745884], [0, 0, 0, 1]], np.float32)


def pose(yaw_deg, tx, ty, tz):
    c, s = math.cos(math.radians(yaw_deg)), math.sin(math.radians(yaw_deg))
    return np.array([[c, 0, s, tx], [0, 1, 0, ty], [-s, 0, c, tz], [0, 0, 0, 1]], np.float32)


def random_poses(batch, seed=2):
    """forward t_z ~ U(-1.5,-0.3) m, lateral t_x ~ U(-0.1,0.1), yaw ~ U(-2,2) deg."""
    rng = np.random.RandomState(seed)
    return np.stack([
        pose(rng.uniform(-2, 2), rng.uniform(-0.1, 0.1), 0.0, rng.uniform(-1.5, -0.3))
        for _ in range(batch)
    ])


def depth_planes(num, dmin=2.0, dmax=59.6):
    return np.array([dmin + (i + 0.5) * ((dmax - dmin) / num) for i in range(num)], np.float32)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def sweep_fixture_paths():
    import glob
    return sorted(glob.glob(os.path.join(GOLDEN, 'plane_sweep_*.npz')))


def host_inverse(P):
    """fp32 torch.inverse on the host -- the op the reference runs
    (utils.py:241); used for BOTH the oracle and the HIP path in parity tests."""
    import torch
    P = np.asarray(P, np.float32)
    if P.ndim == 2:
        return torch.inverse(torch.from_numpy(P.copy())).numpy()
    return np.stack([torch.inverse(torch.from_numpy(p.copy())).numpy() for p in P])


def synthetic_state_dict(module, seed):
    """Deterministic weights for a module (shared by the golden generator and the
    tests so no checkpoint has to be stored): keys in state_dict order, values from a
    seeded CPU generator; norm scales near 1, running_var positive."""
    import torch
    gen = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in module.state_dict().items():
        if k.endswith('num_batches_tracked'):
            out[k] = torch.zeros_like(v)
        elif k.endswith('running_var'):
            out[k] = torch.rand(v.shape, generator=gen) + 0.5
        elif k.endswith('running_mean'):
            out[k] = torch.randn(v.shape, generator=gen) * 0.1
        elif (k.endswith('gn.weight') or k.endswith('bn.weight') or
              (v.dim() == 1 and k.endswith('.weight'))):
            out[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=gen)
        elif v.dim() == 1:
            out[k] = 0.1 * torch.randn(v.shape, generator=gen)
        else:
            fan_in = v[0].numel()
            out[k] = torch.randn(v.shape, generator=gen) * (2.0 / fan_in) ** 0.5
    return out


def assert_matches_reference(out, ref):
    """A plane-sweep volume against the reference's: bit-equal wherever the reference is finite.  Where the
    reference is NaN -- F.grid_sample on PyTorch-CPU answers NON-FINITE sampling coordinates (a division by
    z = 0 in the unguarded projection, utils.py:209) with NaN, its bilinear weights being Inf - Inf -- the
    oracle and the HIP kernels give exactly +0, what `padding_mode='zeros'` gives every other coordinate
    outside the map (and what torch's GPU kernel gives).  tests/golden/plane_sweep_zero_depth.npz is the
    reference-generated fixture that reaches such coordinates; every other fixture is finite everywhere."""
    out = np.asarray(out, np.float32)
    ref = np.asarray(ref, np.float32)
    assert out.shape == ref.shape
    nan = np.isnan(ref)
    assert np.array_equal(bits(np.where(nan, np.float32(0), out)), bits(np.where(nan, np.float32(0), ref)))
    assert np.array_equal(bits(out[nan]), np.zeros(int(nan.sum()), np.uint32)), 'reference NaN <-> exactly +0 here'
    return int(nan.sum())


def bf16_end_to_end_error(got, ref, what=''):
    """whole-module bf16 output against the reference's fp32 output: max |d| / max |ref| and rms(d) / rms(ref),
    printed (pytest -s shows what the suite actually measures; the bars below are 2x the largest value seen on
    MI355X, round 5 -- they were 3 % of full scale + 5 % relative before)"""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    d = got - ref
    e_max = float(np.abs(d).max() / max(np.abs(ref).max(), 1e-30))
    e_rms = float(np.sqrt((d * d).mean()) / max(np.sqrt((ref * ref).mean()), 1e-30))
    print(f'[bf16 end-to-end] {what}: max|d|/max|ref| = {e_max:.5f}, rms(d)/rms(ref) = {e_rms:.5f}')
    return e_max, e_rms
