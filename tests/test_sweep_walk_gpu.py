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
    # the dispatcher reports the body that ran: 5 = the depth-walking kernel (every shape of this file is one
    # it covers), 4 = the per-plane kernel where it is pinned
    assert pkg._capi.lib().dfm_plane_sweep_last_kernel() == (4 if (kernel == 4 and not nhwc) else 5)
    return out.cpu().numpy()


@pytest.mark.parametrize('C,H,W,D,far', [(32, 64, 256, 6, False), (32, 64, 256, 7, False), (32, 64, 256, 51, False),
                                         (64, 32, 128, 9, False), (32, 64, 256, 12, True), (96, 32, 128, 2, False),
                                         (32, 64, 256, 1, False), (32, 32, 160, 5, False)])  # (w_out 40: tiles span rows)
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


# ---- backward of strided fp32 sweeps: the cur map's 3x3 windows in registers (dfm_plane_sweep_bwd_cur_nhwc) ----

@pytest.mark.parametrize('B,C,H,W,D,t_z', [(1, 32, 64, 256, 7, None), (2, 32, 32, 128, 51, None),
                                           (1, 64, 32, 128, 9, -4.0), (1, 32, 64, 256, 1, None),
                                           (1, 32, 32, 160, 6, None)])  # (w_out 40: 16-point tiles span rows)
def test_walk_backward_matches_torch_cpu_autograd(pkg, B, C, H, W, D, t_z):
    """gradients of both maps against torch CPU autograd through F.grid_sample on the oracle's grids
    (test_plane_sweep_gpu._check_backward's bar), and the kernel the call took"""
    from tests.test_plane_sweep_gpu import _check_backward
    _check_backward(pkg, B, C, H, W, D, 1, 4, (0, 0), seed=D + C, img_shape=(375, 1242), t_z=t_z)
    assert pkg._capi.lib().dfm_plane_sweep_bwd_last_kernel() == 9  # (the prev map: the gather kernel of round 5)


def test_walk_backward_overlapping_windows(pkg):
    """cost_sample_factor 2: neighbouring points' 3x3 windows share a pixel column / row -- two waves add to the
    same words of the gradient map (the window flush is atomic for that reason)"""
    from tests.test_plane_sweep_gpu import _check_backward
    _check_backward(pkg, 2, 32, 64, 128, 5, 1, 2, (0, 0), seed=21, img_shape=(375, 1242))


def test_walk_backward_equals_tile_kernel_and_is_channels_last(pkg):
    """cur gradient from the window kernel == the tile kernel's (fp32 sums in another order)"""
    sweep = importlib.import_module('depth-from-motion_amd.plane_sweep')
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(3)
    C, H, W, D = 32, 64, 256, 13
    args = (torch.from_numpy(util.depth_planes(D)).to(dev), 1, 4, torch.from_numpy(util.KITTI_P2[None]),
            torch.from_numpy(util.random_poses(1, seed=9)), (375, 1242))
    cur = rng.randn(1, C, H, W).astype(np.float32)
    prev = rng.randn(1, C, H, W).astype(np.float32)
    gout = torch.from_numpy(rng.randn(1, 2 * C, D, H // 4, W // 4).astype(np.float32)).to(dev)
    grads = []
    for kernel in (None, 5):
        c = torch.from_numpy(cur).to(dev).requires_grad_(True)
        p = torch.from_numpy(prev).to(dev).requires_grad_(True)
        with sweep.backward_kernel(kernel):
            pkg.build_dfm_cost(c, p, *args).backward(gout)
        torch.cuda.synchronize()
        grads.append((c.grad, p.grad, pkg._capi.lib().dfm_plane_sweep_bwd_last_kernel()))
    # (autograd re-lays a leaf's gradient to the leaf's strides: the layouts are checked on the raw call below)
    for a, b in zip(grads[0][:2], grads[1][:2]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5 * float(b.abs().max()))


def test_walk_backward_nonfinite_and_zero_gradients(pkg):
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(0)
    C, H, W, D = 32, 32, 128, 4
    args = (torch.from_numpy(util.depth_planes(D)).to(dev), 1, 4, torch.from_numpy(util.KITTI_P2[None]),
            torch.from_numpy(util.random_poses(1, seed=1)), (375, 1242))
    for poison in (None, float('nan'), float('inf')):
        c = torch.from_numpy(rng.randn(1, C, H, W).astype(np.float32)).to(dev).requires_grad_(True)
        p = torch.from_numpy(rng.randn(1, C, H, W).astype(np.float32)).to(dev).requires_grad_(True)
        g = torch.zeros(1, 2 * C, D, H // 4, W // 4, device=dev)
        if poison is not None:
            g[0, 3, 1, 5, 7] = poison
        pkg.build_dfm_cost(c, p, *args).backward(g)
        if poison is None:
            assert float(c.grad.abs().max()) == 0.0 and float(p.grad.abs().max()) == 0.0
        else:
            # the value reaches its (up to) four taps of channel 3 and nothing else
            bad = ~torch.isfinite(c.grad)
            assert bool(bad.any()) and int(bad.sum()) <= 4 and bool(bad[0, 3].any())
            assert not bool((~torch.isfinite(p.grad)).any())


def test_walk_backward_raw_call_returns_a_pixel_major_cur_gradient(pkg):
    """plane_sweep_backward() on a strided fp32 sweep: the cur gradient comes back channels_last (the window
    kernel's pixel-major map), the prev gradient planar (tile kernel); both equal the forced tile kernel's"""
    sweep = importlib.import_module('depth-from-motion_amd.plane_sweep')
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(8)
    B, C, H, W, D = 2, 32, 32, 128, 6
    cur = torch.empty(B, C, H, W, device=dev)
    desc = sweep._make_desc(cur, D, 1, 4, (375, 1242), False, (0, 0), 1.0)
    depths = torch.from_numpy(util.depth_planes(D)).to(dev)
    P, Pinv, T = sweep.camera_matrices(torch.from_numpy(np.stack([util.KITTI_P2] * B)),
                                       torch.from_numpy(util.random_poses(B, seed=2)), B, dev)
    gout = torch.from_numpy(rng.randn(B, 2 * C, D, H // 4, W // 4).astype(np.float32)).to(dev)
    g_cur, g_prev = sweep.plane_sweep_backward(desc, gout, depths, P, Pinv, T)
    assert g_cur.shape == (B, C, H, W) and g_cur.is_contiguous(memory_format=torch.channels_last)
    assert not g_cur.is_contiguous() and g_prev.is_contiguous()
    with sweep.launch_options(kernel=5):
        t_cur, t_prev = sweep.plane_sweep_backward(desc, gout, depths, P, Pinv, T)
    assert t_cur.is_contiguous()
    assert torch.allclose(g_cur, t_cur, rtol=1e-5, atol=1e-5 * float(t_cur.abs().max()))
    assert torch.equal(g_prev, t_prev) or torch.allclose(g_prev, t_prev, rtol=1e-5, atol=1e-6)


# ---- backward of strided fp32 sweeps, PREV map: a lane per map pixel gathers through the planes' inverse
# ---- homographies (dfm_plane_sweep_bwd_prev_gather, csrc/plane_sweep_bwd_gather.hip) ---------------------------

def _prev_grads(pkg, cur, prev, gout, args, gather):
    sweep = importlib.import_module('depth-from-motion_amd.plane_sweep')
    dev = torch.device('cuda:0')
    c = torch.from_numpy(cur).to(dev).requires_grad_(True)
    p = torch.from_numpy(prev).to(dev).requires_grad_(True)
    with sweep.prev_gather(gather):
        pkg.build_dfm_cost(c, p, *args).backward(gout)
    torch.cuda.synchronize()
    return c.grad, p.grad, pkg._capi.lib().dfm_plane_sweep_bwd_last_kernel()


@pytest.mark.parametrize('csf,flip,crop,scale,t_z', [(4, False, (0, 55), 1.0, None), (4, True, (7, 55), 0.97, None),
                                                     (2, False, (0, 0), 1.0, None), (4, False, (0, 0), 1.0, -6.0),
                                                     (8, False, (0, 0), 1.03, 1.5)])
def test_prev_gather_equals_the_tile_kernel(pkg, csf, flip, crop, scale, t_z):
    """the same (point, tap, weight) set summed in another order: gather (stores) against the LDS-atomic tile kernel
    (atomics), augmentation and strong forward / backward motion included; cur gradients are the same call's"""
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(17 + csf)
    B, C, H, W, D = 2, 32, 64, 256, 11
    T = util.random_poses(B, seed=5)
    if t_z is not None:
        T[:, 2, 3] = t_z
    args = (torch.from_numpy(util.depth_planes(D)).to(dev), 1, csf, torch.from_numpy(np.stack([util.KITTI_P2] * B)),
            torch.from_numpy(T), (375, 1242), flip, crop, scale)
    cur = rng.randn(B, C, H, W).astype(np.float32)
    prev = rng.randn(B, C, H, W).astype(np.float32)
    ho, wo = int(round(H / csf)), int(round(W / csf))
    gout = torch.from_numpy(rng.randn(B, 2 * C, D, ho, wo).astype(np.float32)).to(dev)
    gc1, gp1, k1 = _prev_grads(pkg, cur, prev, gout, args, True)
    gc0, gp0, k0 = _prev_grads(pkg, cur, prev, gout, args, False)
    assert (k1, k0) == (9, 5)
    # (the cur map's window kernel flushes with atomics: at csf = 2 neighbouring windows overlap and the order varies)
    assert torch.allclose(gc1, gc0, rtol=1e-5, atol=2e-6 * float(gc0.abs().max()))
    assert float(gp0.abs().max()) > 0
    diff = (gp1 - gp0).abs()
    bar = 1e-5 * gp0.abs() + 2e-6 * float(gp0.abs().max())
    assert bool((diff <= bar).all()), (
        f'max |diff| {float(diff.max()):.3e} at {np.unravel_index(int(diff.argmax()), diff.shape)}, '
        f'{int((diff > bar).sum())} of {diff.numel()} over the bar, max |g| {float(gp0.abs().max()):.3e}')


def test_prev_gather_planes_without_an_inverse_are_scattered(pkg):
    """a plane at z = 0 in the previous camera (non-finite sample positions: nothing to add) and a plane right
    behind it (mirrored, finite, huge coordinates at the corners): the fit does not vouch for them, the second
    kernel of the call scatters what they contribute; the other planes are gathered.  Against the tile kernel."""
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(2)
    B, C, H, W, D = 1, 32, 64, 256, 6
    depths = util.depth_planes(D)
    P = np.array([[720.0, 0, 608.0, 0], [0, 720.0, 176.0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    T = np.eye(4, dtype=np.float32)[None].copy()
    T[0, 2, 3] = -float(depths[1])           # plane 1 lands at z = 0, plane 0 behind the camera
    T[0, 0, 3] = 0.3
    args = (torch.from_numpy(depths).to(dev), 1, 4, torch.from_numpy(P[None]), torch.from_numpy(T), (375, 1242))
    cur = rng.randn(B, C, H, W).astype(np.float32)
    prev = rng.randn(B, C, H, W).astype(np.float32)
    gout = torch.from_numpy(rng.randn(B, 2 * C, D, H // 4, W // 4).astype(np.float32)).to(dev)
    _, gp1, k1 = _prev_grads(pkg, cur, prev, gout, args, True)
    _, gp0, k0 = _prev_grads(pkg, cur, prev, gout, args, False)
    assert (k1, k0) == (9, 5) and torch.isfinite(gp1).all()
    assert torch.allclose(gp1, gp0, rtol=1e-5, atol=2e-6 * float(gp0.abs().max()))


def test_prev_gather_propagates_nonfinite_gradients_like_the_scatter(pkg):
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(4)
    C, H, W, D = 32, 32, 128, 4
    args = (torch.from_numpy(util.depth_planes(D)).to(dev), 1, 4, torch.from_numpy(util.KITTI_P2[None]),
            torch.from_numpy(util.random_poses(1, seed=1)), (375, 1242))
    cur = rng.randn(1, C, H, W).astype(np.float32)
    prev = rng.randn(1, C, H, W).astype(np.float32)
    g = torch.zeros(1, 2 * C, D, H // 4, W // 4, device=dev)
    g[0, C + 3, 1, 4, 16] = float('inf')    # (a lattice point in the middle of the map: its taps are in bounds)
    _, gp1, _ = _prev_grads(pkg, cur, prev, g, args, True)
    _, gp0, _ = _prev_grads(pkg, cur, prev, g, args, False)
    assert not torch.isfinite(gp0).all(), 'the poisoned point must reach the map in the scatter kernel'
    assert torch.equal(torch.isfinite(gp1), torch.isfinite(gp0))
    assert torch.equal(torch.isnan(gp1), torch.isnan(gp0))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('csf,flip,crop,scale', [(4, False, (0, 55), 1.0), (4, True, (7, 55), 0.97), (2, False, (0, 0), 1.0)])
def test_gather_reads_the_channels_last_volume_in_place(pkg, dtype, csf, flip, crop, scale):
    """the NDHWC stack hands the volume's gradient channels-last: both maps by the gather kernel, read where the
    volume lies, map gradients pixel-major -- against the same call on the reference layout"""
    sweep = importlib.import_module('depth-from-motion_amd.plane_sweep')
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(31 + csf)
    B, C, H, W, D = 2, 64, 64, 256, 9
    cur = torch.from_numpy(rng.randn(B, C, H, W).astype(np.float32)).to(dev).to(dtype)
    prev = torch.from_numpy(rng.randn(B, C, H, W).astype(np.float32)).to(dev).to(dtype)
    depths = torch.from_numpy(util.depth_planes(D)).to(dev)
    Pm, Pinv, Tm = sweep.camera_matrices(torch.from_numpy(np.stack([util.KITTI_P2] * B)),
                                         torch.from_numpy(util.random_poses(B, seed=8)), B, dev)
    desc = sweep._make_desc(cur, D, 1, csf, (375, 1242), flip, crop, scale)
    g = torch.from_numpy(rng.randn(B, 2 * C, D, desc.h_out, desc.w_out).astype(np.float32)).to(dev).to(dtype)
    g_cl = g.contiguous(memory_format=torch.channels_last_3d)
    assert not g_cl.is_contiguous()
    ref_c, ref_p = sweep.plane_sweep_backward(desc, g, depths, Pm, Pinv, Tm)
    got_c, got_p = sweep.plane_sweep_backward(desc, g_cl, depths, Pm, Pinv, Tm)
    torch.cuda.synchronize()
    assert pkg._capi.lib().dfm_plane_sweep_bwd_last_kernel() == 9
    assert got_c.dtype == torch.float32 and got_c.shape == (B, C, H, W)
    assert got_c.is_contiguous(memory_format=torch.channels_last) and got_p.is_contiguous(memory_format=torch.channels_last)
    for a, b in ((got_c, ref_c), (got_p, ref_p)):
        assert float(b.abs().max()) > 0
        assert torch.allclose(a, b, rtol=1e-5, atol=3e-6 * float(b.abs().max()))


@pytest.mark.parametrize('yaw_deg,csf', [(20.0, 2), (35.0, 4), (-28.0, 2)])
def test_prev_gather_under_strong_perspective(pkg, yaw_deg, csf):
    """a large yaw between the frames: the lattice -> prev-map homography zooms by a factor that varies strongly
    across the map, so the candidate box the gather needs differs from pixel to pixel.  The fit kernel vouches for
    a plane only if its bound on the box over EVERY map pixel stays below 4 x 4 (round 6, ADVICE round 5: it used to
    look at five sample points while the gather cut larger boxes silently); the other planes are scattered.  Either
    way the result must be the tile kernel's."""
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(23)
    B, C, H, W, D = 1, 32, 96, 320, 9     # (feat_sample_factor 4: the maps span the whole image; h_out * w_out % 16 == 0)
    a = np.radians(yaw_deg)
    T = np.eye(4, dtype=np.float32)[None].copy()
    T[0, 0, 0], T[0, 0, 2], T[0, 2, 0], T[0, 2, 2] = np.cos(a), np.sin(a), -np.sin(a), np.cos(a)
    T[0, 2, 3] = -1.0
    args = (torch.from_numpy(util.depth_planes(D)).to(dev), 4, csf, torch.from_numpy(util.KITTI_P2[None]),
            torch.from_numpy(T), (375, 1242))
    cur = rng.randn(B, C, H, W).astype(np.float32)
    prev = rng.randn(B, C, H, W).astype(np.float32)
    gout = torch.from_numpy(rng.randn(B, 2 * C, D, H // csf, W // csf).astype(np.float32)).to(dev)
    _, gp1, k1 = _prev_grads(pkg, cur, prev, gout, args, True)
    _, gp0, k0 = _prev_grads(pkg, cur, prev, gout, args, False)
    assert (k1, k0) == (9, 5)
    assert float((gp0 != 0).float().mean()) > 0.2, 'the rotated view must still see a good part of the map'
    diff = (gp1 - gp0).abs()
    bar = 1e-5 * gp0.abs() + 2e-6 * float(gp0.abs().max())
    assert bool((diff <= bar).all()), f'max |diff| {float(diff.max()):.3e}, {int((diff > bar).sum())} over the bar'
