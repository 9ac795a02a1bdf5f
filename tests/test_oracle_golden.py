"""CPU: the oracle (oracle/dfm_oracle.c) against the fixtures the REFERENCE
produced (tests/golden/make_golden.py).  Bar: bit-exact -- the oracle restates
the reference's fp32 op order, including the fma chains of torch.mm and of
ATen's bilinear accumulation."""
import glob
import os

import numpy as np
import pytest

from oracle import dfm_oracle as orc
from tests import util


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def sweep_cases(golden_dir=None):
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    return sorted(glob.glob(os.path.join(here, 'plane_sweep_*.npz')))


@pytest.mark.parametrize('path', sweep_cases(), ids=lambda p: os.path.basename(p)[12:-4])
def test_plane_sweep_grid_bitexact(path):
    z = np.load(path)
    _, _, H, W = z['cur'].shape
    p = orc.sweep_params(H, W, z['depths'].size, float(z['fsf']), float(z['csf']), z['P'],
                         z['Pinv'], z['T'], z['img_shape'], bool(z['flip']), z['crop'],
                         float(z['scale']))
    cg, pg = orc.plane_sweep_grid(p, z['depths'])
    assert np.array_equal(_bits(cg), _bits(z['ref_cur_grid']))
    assert np.array_equal(_bits(pg), _bits(z['ref_prev_grid']))


@pytest.mark.parametrize('path', sweep_cases(), ids=lambda p: os.path.basename(p)[12:-4])
def test_build_dfm_cost_bitexact(path):
    z = np.load(path)
    out = orc.build_dfm_cost(z['cur'], z['prev'], z['depths'], float(z['fsf']), float(z['csf']),
                             z['P'][None], z['Pinv'][None], z['T'][None], z['img_shape'],
                             bool(z['flip']), z['crop'], float(z['scale']))
    ref = z['ref_out']
    assert out.shape == ref.shape
    if 'zero_depth' in path or 'nan_coords' in path:
        # the one fixture that reaches non-finite sampling coordinates: torch-CPU gives NaN there, the oracle 0
        assert util.assert_matches_reference(out, ref) > 0
        return
    assert np.isfinite(ref).all()
    # value-equal everywhere (treats +0 == -0), and bit-equal
    assert np.array_equal(out, ref)
    assert np.array_equal(_bits(out), _bits(ref))


def test_zero_depth_case_reaches_non_finite_coordinates_and_pins_the_deviation():
    """SURVEY 8c's adversarial case taken to its end: plane 1 of this fixture sits at z = 0 in the previous
    camera, the reference's projection divides by zero, its grid is +-Inf there, and F.grid_sample on
    PyTorch-CPU returns NaN for exactly those points (all channels of the prev half); the cur half and the
    other planes are finite.  The oracle reproduces the reference's grid bit for bit -- Inf included -- and
    answers those points with +0 (the chosen behaviour: zeros padding, as torch's GPU kernel)."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    z = np.load(os.path.join(here, 'plane_sweep_zero_depth.npz'))
    C, D = z['cur'].shape[1], z['depths'].size
    ref = z['ref_out'][0]
    hw = ref.shape[2] * ref.shape[3]
    bad = ~np.isfinite(z['ref_prev_grid']).all(1)
    assert bad.sum() == hw and bad.reshape(D, hw)[1].all(), 'every point of plane 1, and only those'
    assert np.isfinite(z['ref_cur_grid']).all() and np.isfinite(ref[:C]).all()
    nan = np.isnan(ref[C:])
    assert np.array_equal(nan, np.broadcast_to(bad.reshape(1, D, *ref.shape[2:]), nan.shape))
    out = orc.build_dfm_cost(z['cur'], z['prev'], z['depths'], float(z['fsf']), float(z['csf']),
                             z['P'][None], z['Pinv'][None], z['T'][None], z['img_shape'],
                             bool(z['flip']), z['crop'], float(z['scale']))[0]
    assert (out[C:, 1] == 0).all() and np.signbit(out[C:, 1]).sum() == 0


def test_nan_coordinate_case_reaches_zero_over_zero_and_pins_the_deviation():
    """round 6: the reference-generated fixture whose previous-frame grid holds NaN coordinates (0 / 0: the
    lattice column / row through the principal point un-project to exactly 0 and plane 1 sits at z = 0), next to
    +-Inf on the rest of that plane.  The reference answers all of them with NaN; the oracle reproduces the
    grid bit for bit (NaN positions included) and answers +0 -- now pinned by a reference output, not only by the
    oracle's own isfinite test."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    z = np.load(os.path.join(here, 'plane_sweep_nan_coords.npz'))
    C, D = z['cur'].shape[1], z['depths'].size
    H, W = z['ref_out'].shape[3:]
    pg = z['ref_prev_grid'].reshape(D, H, W, 2)
    nanx, nany = np.isnan(pg[..., 0]), np.isnan(pg[..., 1])
    assert nanx[1, :, 38].all() and nany[1, 11, :].all() and nanx.sum() == H and nany.sum() == W
    assert nanx[1, 11, 38] and nany[1, 11, 38], '0 / 0 on both axes at the principal point'
    assert np.isinf(pg[1]).any() and np.isfinite(pg[[0, 2, 3]]).all()
    p = orc.sweep_params(z['cur'].shape[2], z['cur'].shape[3], D, float(z['fsf']), float(z['csf']), z['P'], z['Pinv'],
                         z['T'], z['img_shape'], bool(z['flip']), z['crop'], float(z['scale']))
    _, opg = orc.plane_sweep_grid(p, z['depths'])
    assert np.array_equal(np.isnan(opg), np.isnan(z['ref_prev_grid']))
    ref = z['ref_out'][0]
    assert np.isnan(ref[C:, 1]).all() and np.isfinite(ref[:C]).all() and np.isfinite(ref[C:][:, [0, 2, 3]]).all()
    out = orc.build_dfm_cost(z['cur'], z['prev'], z['depths'], float(z['fsf']), float(z['csf']),
                             z['P'][None], z['Pinv'][None], z['T'][None], z['img_shape'],
                             bool(z['flip']), z['crop'], float(z['scale']))[0]
    assert (out[C:, 1] == 0).all() and np.signbit(out[C:, 1]).sum() == 0


def test_behind_camera_case_really_goes_out_of_bounds():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    z = np.load(os.path.join(here, 'plane_sweep_behind_camera.npz'))
    C = z['cur'].shape[1]
    prev_half = z['ref_out'][0, C:]
    # plane 0 (d=3.5) lands BEHIND the previous camera (z' = d-9 < 0: mirrored,
    # still sampled); plane 2 (z' ~ 0.5) leaves the image -> zeros from padding
    zero_frac = [(prev_half[:, d] == 0).mean() for d in range(prev_half.shape[1])]
    assert zero_frac[0] < 0.1 and max(zero_frac) > 0.99
    assert np.abs(z['ref_prev_grid']).max() > 10.0


@pytest.mark.parametrize('path', sweep_cases(), ids=lambda p: os.path.basename(p)[12:-4])
def test_torch_restatement_for_the_cpu_baseline_reproduces_the_reference_fixtures(path):
    """oracle/dfm_torch_baseline.py issues the torch calls of the reference's build_dfm_cost (bench.py times it
    as ``cpu_baseline``): grids and volume must equal what the reference's own code produced, bit for bit
    (NaN where the reference has NaN: the same library)."""
    import torch
    from oracle import dfm_torch_baseline as tb
    z = np.load(path)
    cur, prev = torch.from_numpy(z['cur']), torch.from_numpy(z['prev'])
    args = (torch.from_numpy(z['depths']), float(z['fsf']), float(z['csf']), torch.from_numpy(z['P'])[None],
            torch.from_numpy(z['T'])[None], tuple(int(v) for v in z['img_shape']), bool(z['flip']),
            tuple(z['crop'].tolist()), float(z['scale']))
    cg, pg = tb.sampling_grids(cur.shape[2], cur.shape[3], *args)
    assert np.array_equal(_bits(cg.numpy().reshape(-1, 2)), _bits(z['ref_cur_grid']))
    assert np.array_equal(_bits(pg.numpy().reshape(-1, 2)), _bits(z['ref_prev_grid']))
    out = tb.build_dfm_cost(cur, prev, *args).numpy()
    assert out.shape == z['ref_out'].shape
    assert np.array_equal(np.isnan(out), np.isnan(z['ref_out']))
    assert np.array_equal(_bits(np.nan_to_num(out)), _bits(np.nan_to_num(z['ref_out'])))
