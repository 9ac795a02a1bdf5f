"""The plane sweep fused into dres0 / dres0_mono (csrc/sweep_conv.hip, ``dfm_sweep_conv_fwd``)
against the unfused sequence it replaces (mmdet3d/models/backbones/dfm_backbone.py:161-176,189):

  cost_raw = build_dfm_cost(cur, prev, ...)      (bit-exact to the reference: tests/test_plane_sweep_gpu.py)
  dres0.conv(cost_raw), dres0_mono.conv(cost_raw[:, :32])

* delta-kernel weights turn the fused kernel into a shifted copy of the bf16 volume: the sampler
  (coordinates, footprints, blend, bf16 rounding), the halo, the convolution's zero padding and the
  depth-chunk seams must reproduce the unfused volume BIT FOR BIT;
* random weights: fp32 torch convolution of the unfused bf16 volume, per-layer bf16 bar
  (|got - ref| <= 2^-7 |ref| + 2e-3, the bar of tests/test_conv3d_g_gpu.py);
* the GroupNorm moment partials merge to the moments of the stored tensors;
* DfMBackbone: fused vs ``fuse_sweep_dres0 = False`` at the per-layer bar, and vs the reference
  module's fixture of BASELINE.json configs[0] (tests/golden/backbone_cfg1.npz).
"""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import util

pytestmark = pytest.mark.gpu
RTOL, ATOL = 2.0 ** -7, 2e-3


@pytest.fixture(scope='module')
def pkg():
    assert torch.cuda.is_available()
    return importlib.import_module('depth-from-motion_amd')


@pytest.fixture(scope='module')
def sc():
    return importlib.import_module('depth-from-motion_amd.sweep_conv')


# name: B, (H_in, W_in), fsf, csf, D, flip, crop, scale, pose, depth range, depth_chunk
CASES = {
    'kitti_like': (1, (64, 256), 1, 4, 8, False, (0, 55), 1.0, util.pose(-0.7, -0.03, 0.0, -0.8), (2, 59.6), 0),
    'ragged_two_samples': (2, (52, 148), 1, 4, 7, False, (3, 5), 1.0, util.pose(1.3, 0.05, 0.01, -1.1), (2, 40.0), 3),
    'flip_crop_scale': (1, (48, 160), 1, 4, 5, True, (7, 55), 0.97, util.pose(0.4, 0.0, 0.0, -0.5), (2, 59.6), 2),
    'nstar_like_fsf4': (1, (22, 70), 4, 1, 6, False, (0, 0), 1.0, util.pose(2.0, 0.1, -0.02, -1.4), (2, 59.6), 1),
    'behind_camera': (1, (40, 132), 1, 4, 6, False, (0, 0), 1.0, util.pose(5.0, 0.3, 0.0, -9.0), (2, 20.0), 0),
    # a single depth plane (both depth neighbours are the convolution's zero padding), a volume smaller
    # than one 8 x 32 tile, three samples with one plane per workgroup
    'one_plane': (1, (32, 96), 1, 4, 1, False, (0, 0), 1.0, util.pose(0.5, 0.02, 0.0, -0.8), (2, 59.6), 0),
    'two_planes_tiny': (1, (12, 40), 1, 4, 2, False, (0, 0), 1.0, util.pose(0.5, 0.02, 0.0, -0.8), (2, 59.6), 0),
    'three_samples_chunk1': (3, (32, 132), 1, 4, 4, False, (0, 5), 1.0, util.pose(-1.0, 0.0, 0.0, -1.2), (2, 30.0), 1),
}


def _inputs(name):
    B, (H, W), fsf, csf, D, flip, crop, scale, pose, (dmin, dmax), dchunk = CASES[name]
    g = torch.Generator().manual_seed(100 + list(CASES).index(name))
    cur = torch.randn(B, 32, H, W, generator=g).bfloat16().cuda()
    prev = torch.randn(B, 32, H, W, generator=g).bfloat16().cuda()
    depths = torch.from_numpy(util.depth_planes(D, dmin, dmax)).cuda()
    P = torch.from_numpy(util.KITTI_P2)[None].repeat(B, 1, 1)
    T = torch.from_numpy(np.stack([pose] * B))
    args = (depths, fsf, csf, P, T, (375, 1242))
    kw = dict(flip=flip, img_crop_offset=crop, img_scale_factor=scale)
    return cur, prev, args, kw, dchunk


def _weights(seed):
    g = torch.Generator().manual_seed(seed)
    ws = (torch.randn(32, 64, 3, 3, 3, generator=g) * (2.0 / (27 * 64)) ** 0.5).bfloat16().cuda()
    wm = (torch.randn(32, 32, 3, 3, 3, generator=g) * (2.0 / (27 * 32)) ** 0.5).bfloat16().cuda()
    return ws, wm


def _fused(sc, cur, prev, args, kw, ws, wm, dchunk):
    depths, fsf, csf, P, T, shape = args
    return sc.sweep_dres0(cur, prev, depths, fsf, csf, P, T, shape, sc.pack_sweep_conv_weights(ws, wm),
                          depth_chunk=dchunk, **kw)


@pytest.mark.parametrize('name', list(CASES))
def test_delta_weights_reproduce_the_unfused_volume_bit_for_bit(pkg, sc, name):
    cur, prev, args, kw, dchunk = _inputs(name)
    vol = pkg.build_dfm_cost(cur, prev, *args, memory_format=torch.channels_last_3d, **kw)   # (B, 64, D, H, W)
    volf = vol.float()
    for layout in (False, True):   # NCHW maps (converted by the wrapper) and NHWC maps sampled in place
        c, p = (cur.contiguous(memory_format=torch.channels_last), prev.contiguous(memory_format=torch.channels_last)) \
            if layout else (cur, prev)
        for tap in ((1, 1, 1), (0, 0, 0), (2, 2, 2), (0, 1, 2), (2, 0, 1)):
            # stereo output channel o copies volume channel 2o (cur half) for even o ... a permutation that
            # reaches both halves and all 32 channel lanes; mono copies cur channel 31 - o
            src_s = [(3 * o + 5) % 64 for o in range(32)]
            src_m = [31 - o for o in range(32)]
            ws = torch.zeros(32, 64, 3, 3, 3)
            wm = torch.zeros(32, 32, 3, 3, 3)
            for o in range(32):
                ws[o, src_s[o], tap[0], tap[1], tap[2]] = 1.0
                wm[o, src_m[o], tap[0], tap[1], tap[2]] = 1.0
            ys, ps, ym, pm = _fused(sc, c, p, args, kw, ws.cuda(), wm.cuda(), dchunk)
            pad = F.pad(volf, (1, 1, 1, 1, 1, 1))
            D, H, W = vol.shape[2:]
            sh = pad[:, :, tap[0]:tap[0] + D, tap[1]:tap[1] + H, tap[2]:tap[2] + W]
            assert torch.equal(ys.float(), sh[:, src_s]), (name, layout, tap, 'stereo')
            assert torch.equal(ym.float(), sh[:, src_m]), (name, layout, tap, 'mono')


@pytest.mark.parametrize('name', list(CASES))
def test_random_weights_match_the_unfused_convolutions(pkg, sc, name):
    cur, prev, args, kw, dchunk = _inputs(name)
    ws, wm = _weights(11)
    vol = pkg.build_dfm_cost(cur, prev, *args, memory_format=torch.channels_last_3d, **kw).float()
    ref_s = F.conv3d(vol, ws.float(), padding=1)
    ref_m = F.conv3d(vol[:, :32], wm.float(), padding=1)
    ys, ps, ym, pm = _fused(sc, cur, prev, args, kw, ws, wm, dchunk)
    assert ys.dtype == torch.bfloat16 and ys.is_contiguous(memory_format=torch.channels_last_3d)
    np.testing.assert_allclose(ys.float().cpu().numpy(), ref_s.cpu().numpy(), rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(ym.float().cpu().numpy(), ref_m.cpu().numpy(), rtol=RTOL, atol=ATOL)
    for y, part in ((ys, ps), (ym, pm)):
        p = part.double()
        cnt, mean, m2 = p[..., 0], p[..., 1], p[..., 2]
        n = cnt.sum(-1)
        mu = (cnt * mean).sum(-1) / n
        var = (m2.sum(-1) + (cnt * (mean - mu[..., None]) ** 2).sum(-1)) / n
        yd = y.double().flatten(2)
        assert float((n - yd.shape[2]).abs().max()) == 0
        np.testing.assert_allclose(mu.cpu().numpy(), yd.mean(-1).cpu().numpy(), atol=1e-5 * float(yd.std()) + 1e-7)
        np.testing.assert_allclose(var.cpu().numpy(), yd.var(-1, unbiased=False).cpu().numpy(), rtol=1e-4)


def test_unsupported_inputs_raise(sc):
    cur = torch.randn(1, 64, 16, 32).bfloat16().cuda()
    with pytest.raises(TypeError):
        sc.sweep_dres0(cur, cur, torch.ones(4).cuda(), 1, 4, torch.eye(4)[None], torch.eye(4)[None], (375, 1242), None)
    assert not sc.sweep_conv_supported(cur) and not sc.sweep_conv_supported(cur[:, :32].float())


def _backbone(mods, seed, depth_cfg, csf=4):
    m = mods.DfMBackbone(in_channels=32, cv_channels=32, num_hg=1, cost_sample_factor=csf, depth_cfg=depth_cfg,
                         norm_cfg=dict(type='GN', num_groups=32, requires_grad=True))
    m.load_state_dict(util.synthetic_state_dict(m, seed), strict=True)
    m = m.eval().cuda().to(torch.bfloat16)
    m.volume_memory_format = torch.channels_last_3d
    return m


def test_backbone_fused_matches_unfused_and_counts_launches(pkg, monkeypatch):
    mods = importlib.import_module('depth-from-motion_amd.modules')
    cv = importlib.import_module('depth-from-motion_amd.conv3d')
    depth_cfg = dict(mode='UD', num_bins=32, depth_min=2, depth_max=59.6, downsample_factor=4)
    m = _backbone(mods, 17, depth_cfg)
    m.downsampled_depth = torch.from_numpy(util.depth_planes(8))
    g = torch.Generator().manual_seed(3)
    cur = torch.randn(2, 32, 96, 256, generator=g).cuda().bfloat16()
    prev = torch.randn(2, 32, 96, 256, generator=g).cuda().bfloat16()
    metas = [dict(ori_cam2img=util.KITTI_P2, cur2prevs=torch.from_numpy(util.random_poses(2)[i])[None],
                  ori_shape=(375, 1242, 3), pad_shape=(96, 256, 3), crop_offset=[0, 55], flip=False,
                  scale_factor=[1.0]) for i in range(2)]
    calls = {'c32': 0, 'sweep': 0}
    real_c, real_s = cv.conv3d_k3_c32, mods.sweep_dres0
    monkeypatch.setattr(cv, 'conv3d_k3_c32', lambda *a, **k: (calls.__setitem__('c32', calls['c32'] + 1), real_c(*a, **k))[1])
    monkeypatch.setattr(mods, 'sweep_dres0', lambda *a, **k: (calls.__setitem__('sweep', calls['sweep'] + 1), real_s(*a, **k))[1])
    with torch.no_grad():
        fused = m(cur, prev, metas)
    assert calls == {'c32': 4, 'sweep': 1}, calls      # dres1 x 2, pred.0 x 2; dres0 / dres0_mono are in the sweep
    m.fuse_sweep_dres0 = False
    calls.update(c32=0, sweep=0)
    with torch.no_grad():
        plain = m(cur, prev, metas)
    assert calls == {'c32': 7, 'sweep': 0}, calls
    for a, b, name in zip(fused, plain, ('cost', 'stereo_feat', 'mono_feat')):
        a, b = a.float(), b.float()
        # 9 bf16 layers downstream of two convolutions that differ in their last bf16 bit
        assert float((a - b).abs().max()) <= 4 * 2.0 ** -7 * float(b.abs().max()), name
        assert float((a - b).abs().mean()) <= 2.0 ** -9 * float(b.abs().max()), name
    # with autograd recording the module keeps the materialised volume (weight gradients need it)
    m.fuse_sweep_dres0 = True
    calls.update(c32=0, sweep=0)
    m(cur, prev, metas)
    assert calls['sweep'] == 0


def test_backbone_fused_vs_reference_module_fixture_config1(pkg):
    """BASELINE.json configs[0] (KITTI pair, D = 4) through the fused kernel in bf16 against the
    reference module's fp32 output (tests/golden/backbone_cfg1.npz)"""
    mods = importlib.import_module('depth-from-motion_amd.modules')
    sys.path.insert(0, util.GOLDEN)
    try:
        from make_golden_r02 import CFG1, cfg1_inputs
    finally:
        sys.path.remove(util.GOLDEN)
    z = np.load(os.path.join(util.GOLDEN, 'backbone_cfg1.npz'))
    depth_cfg = dict(mode='UD', num_bins=16, depth_min=2, depth_max=59.6, downsample_factor=4)
    m = _backbone(mods, CFG1['wseed'], depth_cfg, CFG1['csf'])
    cur, prev, depths, meta = cfg1_inputs()
    m.downsampled_depth = depths
    assert m._sweep_dres0_fusable(cur.cuda().bfloat16()) is False or not torch.is_grad_enabled()
    with torch.no_grad():
        assert m._sweep_dres0_fusable(cur.cuda().bfloat16())
        cost, sfeat, mfeat = m(cur.cuda().bfloat16(), prev.cuda().bfloat16(), [meta])
    assert cost.shape == (1, 1, 4, 96, 312) and sfeat.shape == (1, 32, 4, 96, 312)
    for got, ref in ((cost, z['cost']), (sfeat[..., ::8, ::8], z['stereo_s8']), (mfeat[..., ::8, ::8], z['mono_s8'])):
        np.testing.assert_allclose(got.float().cpu().numpy(), ref, rtol=5e-2, atol=0.03 * float(np.abs(ref).max()))
    np.testing.assert_allclose(sfeat.float().abs().mean((0, 2, 3, 4)).cpu().numpy(), z['stereo_abs_mean'], rtol=2e-2)
    np.testing.assert_allclose(mfeat.float().abs().mean((0, 2, 3, 4)).cpu().numpy(), z['mono_abs_mean'], rtol=2e-2)
