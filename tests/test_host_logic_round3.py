"""Host-side logic added in round 3 that needs no GPU: the backward-kernel override (process-wide, because
autograd runs backward functions on its own threads), the lazy depth distribution's stand-ins for tensor
methods the detector calls (dfm.py:348-356 flattens (B, N) before the loss), the loss-type -> descriptor
mapping of DepthHead.loss (depth_head.py:111-183), and the layout decision of the 2-D necks."""
import importlib
import threading

import pytest
import torch


@pytest.fixture(scope='module')
def pkg():
    importlib.import_module('depth-from-motion_amd.build').build_hip()
    return importlib.import_module('depth-from-motion_amd')


def test_backward_kernel_override_is_visible_from_other_threads(pkg):
    ps = pkg.plane_sweep
    seen = []
    assert ps._bwd_kernel is None
    with ps.backward_kernel(5):
        t = threading.Thread(target=lambda: seen.append(ps._bwd_kernel))   # what an autograd thread reads
        t.start()
        t.join()
        with ps.backward_kernel(1):
            assert ps._bwd_kernel == 1
        assert ps._bwd_kernel == 5
    assert seen == [5] and ps._bwd_kernel is None
    # launch_options, by contrast, is thread-local by design
    with ps.launch_options(kernel=2):
        t = threading.Thread(target=lambda: seen.append(ps._current_opts()))
        t.start()
        t.join()
    assert seen[-1] is None


def test_lazy_depth_distribution_stands_in_for_the_volume(pkg):
    dh = importlib.import_module('depth-from-motion_amd.depth_head')
    cost = torch.zeros(2, 1, 3, 4, 5)
    d = dh.LazyDepthDistribution(cost, torch.zeros(2, 16, 20), torch.ones(2, 16, 20), torch.arange(12.0), 4,
                                 cost_with_grad=cost)
    assert d.shape == (2, 1, 12, 16, 20) and d.dtype == torch.float32
    assert d.flatten(start_dim=0, end_dim=1) is d and d.detach() is d and d.cost_with_grad is cost


@pytest.mark.parametrize('loss_type, target, focal, sigma', [
    ('ce', 'DL_LINEAR', 0, 0.0), ('balanced_ce', 'DL_LINEAR', 0, 0.0), ('focal', 'DL_LINEAR', 1, 0.0),
    ('balanced_focal', 'DL_LINEAR', 1, 0.0), ('hard_ce', 'DL_HARD', 0, 0.0),
    ('gaussian_0.5', 'DL_GAUSSIAN', 0, 0.5), ('laplacian_2', 'DL_LAPLACIAN', 0, 2.0)])
def test_loss_descriptor_mapping(pkg, loss_type, target, focal, sigma):
    dh = importlib.import_module('depth-from-motion_amd.depth_head')
    samples = torch.tensor([2.1, 2.3, 2.5, 2.7])
    d = dh._loss_desc(3, 4, 8, 16, torch.bfloat16, samples, loss_type, 2.0, 59.6, 0.75, 2.0)
    assert (d.batch, d.num_depths, d.h, d.w) == (3, 4, 8, 16)
    assert d.target == getattr(pkg._capi, target) and d.focal == focal and d.sigma == pytest.approx(sigma)
    assert d.interval == pytest.approx(float(samples[1] - samples[0])) and d.dtype == pkg._capi.DFM_BF16
    with pytest.raises(NotImplementedError):
        dh._loss_desc(3, 4, 8, 16, torch.float32, samples, 'l2', 2.0, 59.6, 1.0, 2.0)


def test_two_d_necks_leave_cpu_tensors_alone(pkg):
    mods = importlib.import_module('depth-from-motion_amd.modules')
    m = torch.nn.Conv2d(4, 4, 3)
    feats = [torch.zeros(1, 4, 8, 8)]
    out = mods._channels_last_2d(m, feats)
    assert out is feats and '_weights_format' not in m.__dict__


def test_regular_grid_detection_for_the_gather_backward():
    """FrustumToVoxel's gather backward enumerates voxels by index arithmetic: the coordinate tensor must be the
    regular grid prepare_coordinates_3d builds (frustum_to_voxel._regular_grid, cached per tensor)"""
    import importlib
    import types
    f2v = importlib.import_module('depth-from-motion_amd.frustum_to_voxel')
    geo = importlib.import_module('depth-from-motion_amd.geometry')
    voxel_cfg = dict(point_cloud_range=[2, -30.4, -3, 59.6, 30.4, 1], voxel_size=[0.2, 0.2, 0.2])
    coords = geo.prepare_coordinates_3d(voxel_cfg)          # (nz, ny, nx, 3), config K's grid
    nz, ny, nx = coords.shape[:3]
    desc = types.SimpleNamespace(nz=nz, ny=ny, nx=nx)
    flat = coords.reshape(-1, 3).float().contiguous()
    g = f2v._regular_grid(flat, desc)
    assert g is not None
    x0, dx, y0, dy, z0, dz = g
    assert abs(dx - 0.2) < 1e-5 and abs(dy - 0.2) < 1e-5 and abs(dz - 0.2) < 1e-5
    assert abs(x0 - 2.1) < 1e-4 and abs(y0 + 30.3) < 1e-4 and abs(z0 + 2.9) < 1e-4
    assert f2v._regular_grid(flat, desc) is g or f2v._regular_grid(flat, desc) == g      # cached
    bent = flat.clone()
    bent[nx * ny + 7, 2] += 0.01                                                       # one voxel off its plane
    assert f2v._regular_grid(bent, desc) is None
    swapped = flat.clone().reshape(nz, ny, nx, 3).transpose(1, 2).reshape(-1, 3).contiguous()   # y fastest
    assert f2v._regular_grid(swapped, types.SimpleNamespace(nz=nz, ny=ny, nx=nx)) is None
    assert f2v._regular_grid(flat[:-3], desc) is None                                  # wrong element count
