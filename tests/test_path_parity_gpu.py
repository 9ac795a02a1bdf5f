"""End-to-end and layer-wise parity of the MFMA module path.

1. BASELINE.json configs #4 / #5 end to end: ``MultiViewVoxelPath`` (voxel lifting -> neck_3d)
   against fixtures produced by the REFERENCE's ``MultiViewDfM.feature_transformation`` including its
   ``neck_3d`` call with the reference's own OutdoorImVoxelNeck / DfMNeck
   (tests/golden/make_golden_r03.py; detectors/multiview_dfm.py:119-268, necks/imvoxel_neck.py:26-73,
   necks/dfm_neck.py:97-122): fp32 within the conv-stack tolerance of SURVEY.md 8c, training-mode
   BatchNorm included, and bf16 / channels-last with every 3x3x3 convolution in the MFMA kernels.
2. Layer by layer: EVERY MFMA convolution launch and EVERY fused GroupNorm pass issued by
   ``DfMStereoPath`` (SPPUNetNeck, DfMBackbone + hourglass, FrustumToVoxel, BEVHourglass) and by the
   voxel necks in bf16 is replayed in place -- the op the reference module runs there (torch's
   conv3d / conv_transpose3d / group_norm in fp32) on the launch's OWN bf16 input and the bf16-rounded
   weights -- and compared at SURVEY.md 8c's bf16 bar, ``atol = 2^-7 * max|ref|``.
3. The general kernel at the voxel necks' real extent (220 x 300 x {12, 6, 3}, 64 / 128 / 256
   channels) against an fp64 evaluation of a voxel subset.
"""
import importlib
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import util

pytestmark = pytest.mark.gpu
CONV_TOL = dict(rtol=1e-3, atol=1e-4)
BF16_BAR = 2.0 ** -7
# whole MultiViewVoxelPath in bf16 vs the reference's fp32 output: 2x what the suite measures on MI355X
# (round 5: max |d| / max |ref| 0.0098, rms 0.0075; the bar was rtol 5e-2 + 3 % of full scale)
BF16_PATH_MAX, BF16_PATH_RMS = 0.02, 0.016


@pytest.fixture(scope='module')
def pkg():
    assert torch.cuda.is_available()
    return importlib.import_module('depth-from-motion_amd')


@pytest.fixture(scope='module')
def cv():
    return importlib.import_module('depth-from-motion_amd.conv3d')


@pytest.fixture(scope='module')
def cfgs():
    with open(os.path.join(util.GOLDEN, 'configs_dfm.json')) as f:
        return json.load(f)


WAYMO = 'multiview-dfm_r101_dcn_2x16_waymoD5-3d-3class_camsync.py'
WAYMO10 = 'multiview-dfm_r101_dcn_2x16_waymoD5-3d-3class_camsync_10sweeps.py'


def _mv_feats(z):
    sys.path.insert(0, util.GOLDEN)
    try:
        import make_golden_r03 as g3
    finally:
        sys.path.remove(util.GOLDEN)
    return g3.mv_path_feats(int(z['seed']), int(z['num_views']), int(z['num_frames']), int(z['channels']))


def _mv_path(pkg, cfgs, z):
    """MultiViewVoxelPath from the reference's config dict, narrowed to the fixture's grid / widths"""
    from tests.test_point_sample_gpu import meta_from_fixture
    nf = int(z['num_frames'])
    model = dict(cfgs[WAYMO10 if nf == 2 else WAYMO]['model'])
    assert model['neck_3d']['type'] == str(z['neck_type'])
    model['neck_3d'] = dict(model['neck_3d'], in_channels=int(z['channels']), out_channels=int(z['neck_out']))
    model['anchor_generator'] = dict(model['anchor_generator'], ranges=[[float(v) for v in z['voxel_range']]])
    model['voxel_size'] = [float(v) for v in z['voxel_size']]
    path = pkg.MultiViewVoxelPath(model)
    assert path.n_voxels == [int(v) for v in z['n_voxels']] and path.n_voxels[2] == 12
    assert path.temporal_aggregate == str(z['aggregate'])
    assert list(path.neck_3d.state_dict().keys()) == list(z['neck_keys'])
    path.neck_3d.load_state_dict(util.synthetic_state_dict(path.neck_3d, int(z['seed']) + 200), strict=True)
    return path.cuda(), meta_from_fixture(z)


@pytest.mark.parametrize('name', ['mvpath_mean_1f', 'mvpath_concat_2f'])
def test_multiview_voxel_path_fp32_vs_reference_detector_and_neck(pkg, cfgs, name):
    """configs #4 ('mean', F = 1, OutdoorImVoxelNeck) and #5 ('concat', F = 2, DfMNeck, with scale /
    flip / crop augmentation) in fp32: lifted volume bit-exact, BEV map within CONV_TOL, eval and
    training-mode BatchNorm (batch statistics + the running-statistics update)"""
    z = np.load(os.path.join(util.GOLDEN, f'{name}.npz'))
    path, meta = _mv_path(pkg, cfgs, z)
    feats = _mv_feats(z).cuda()
    nv, nf = int(z['num_views']), int(z['num_frames'])
    vol = pkg.mv_feature_transformation(feats, [meta], nv, nf, path.voxel_range, path.n_voxels,
                                        path.temporal_aggregate)
    assert np.array_equal(vol.cpu().numpy(), z['ref_volume'])
    path.eval()
    with torch.no_grad():
        out = path(feats, [meta], nv, nf)
    assert out.shape == z['ref_out'].shape
    np.testing.assert_allclose(out.cpu().numpy(), z['ref_out'], **CONV_TOL)
    path.train()
    with torch.no_grad():
        out = path(feats, [meta], nv, nf)
    np.testing.assert_allclose(out.cpu().numpy(), z['ref_out_train'], **CONV_TOL)
    rm = path.neck_3d.state_dict()[str(z['train_running_mean_key'])]
    np.testing.assert_allclose(rm.cpu().numpy(), z['train_running_mean'], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('name,launches', [('mvpath_wide_mean_1f', 9), ('mvpath_wide_concat_2f', 18)])
def test_multiview_voxel_path_mfma_vs_reference_detector_and_neck(pkg, cv, cfgs, monkeypatch, name, launches):
    """the same two configs at the channel width the MFMA kernels take (C = 32 -> 64 -> 128): fp32
    through torch convolutions within CONV_TOL, and bf16 features -> channels-last volume -> every
    convolution of the neck one fused MFMA launch (eval-mode BatchNorm, residual and ReLU in the
    epilogue), against the reference's fp32 output"""
    z = np.load(os.path.join(util.GOLDEN, f'{name}.npz'))
    path, meta = _mv_path(pkg, cfgs, z)
    path.eval()
    feats = _mv_feats(z).cuda()
    nv, nf = int(z['num_views']), int(z['num_frames'])
    calls = {'g': 0}
    real = cv.conv3d_g

    def counted(*a, **k):
        calls['g'] += 1
        return real(*a, **k)
    monkeypatch.setattr(cv, 'conv3d_g', counted)
    with torch.no_grad():
        out = path(feats, [meta], nv, nf)
    assert calls['g'] == 0
    np.testing.assert_allclose(out.cpu().numpy(), z['ref_out'], **CONV_TOL)
    pb = path.to(torch.bfloat16)
    with torch.no_grad():
        out = pb(feats.bfloat16(), [meta], nv, nf)
    assert calls['g'] == launches, calls
    assert out.dtype == torch.bfloat16 and out.shape == z['ref_out'].shape
    ref = z['ref_out']
    e_max, e_rms = util.bf16_end_to_end_error(out.float().cpu().numpy(), ref, 'MultiViewVoxelPath ' + name)
    assert e_max <= BF16_PATH_MAX and e_rms <= BF16_PATH_RMS, (e_max, e_rms)


# ---------------------------------------------------------------------------------------------
# layer-by-layer replay
# ---------------------------------------------------------------------------------------------
class LayerReplay:
    """Wraps the package's launch points (``conv3d_g``, ``conv3d_k3_c32``, ``group_norm`` and the two
    weight packers).  Every call is executed by the HIP kernel AND replayed with torch in fp32 on the
    same device tensors; the relative error (max |got - ref| / max |ref|) is recorded per launch."""

    def __init__(self, monkeypatch):
        self.cv = importlib.import_module('depth-from-motion_amd.conv3d')
        self.gn = importlib.import_module('depth-from-motion_amd.group_norm')
        self.packs = {}     # data_ptr of a packed buffer -> how it was made (keeps the buffer alive)
        self.records = []   # (kind, detail, relative error)
        real = {n: getattr(self.cv, n) for n in ('conv3d_g', 'conv3d_k3_c32', 'pack_conv3d_g_weights',
                                                 'pack_conv3d_weights')}
        real_gn = self.gn.group_norm

        def pack_g(weight, cin, cout, swap=False, flip=0):
            pk = real['pack_conv3d_g_weights'](weight, cin, cout, swap=swap, flip=flip)
            w = weight.detach().float().clone()
            if w.dim() == 4:   # a 2-D weight, packed into the centre depth slice (dfm_conv3d_g_pack_weights_2d)
                w3 = w.new_zeros((*w.shape[:2], 3, 3, 3))
                w3[:, :, 1] = w
                w = w3
            self.packs[pk.data_ptr()] = (pk, w, swap, flip)
            return pk

        def pack_c32(weight, cin_offset=0, transposed=False):
            pk = real['pack_conv3d_weights'](weight, cin_offset, transposed)
            self.packs[pk.data_ptr()] = (pk, weight.detach().float().clone(), cin_offset, transposed)
            return pk

        def conv_g(x, packed, cout, stride=1, padding=1, transposed=False, relu=False, scale=None, shift=None,
                   residual=None, kernel1=False):
            got = real['conv3d_g'](x, packed, cout, stride=stride, padding=padding, transposed=transposed, relu=relu,
                                   scale=scale, shift=shift, residual=residual, kernel1=kernel1)
            _, w, swap, flip = self.packs[packed.data_ptr()]
            assert flip == 0, 'forward launches only'
            t3 = self.cv._triple
            stride, padding, transposed, kernel1 = t3(stride), t3(padding), t3(transposed), t3(kernel1)
            for ax, k1 in enumerate(kernel1):  # kernel extent 1 on this axis: the packed weight's centre slice
                if k1:
                    w = w.narrow(2 + ax, 1, 1)
            padding = tuple(0 if k1 else p for p, k1 in zip(padding, kernel1))
            xf = x.float()
            if any(transposed):
                assert swap
                axes = [t and not k1 for t, k1 in zip(transposed, kernel1)]
                ref = F.conv_transpose3d(xf, w, stride=tuple(2 if a else 1 for a in axes),
                                         padding=tuple(1 if a else 0 for a in axes),
                                         output_padding=tuple(1 if a else 0 for a in axes))
            else:
                assert not swap
                ref = F.conv3d(xf, w, stride=stride, padding=padding)
            if scale is not None:
                ref = ref * scale.float().view(1, -1, 1, 1, 1) + shift.float().view(1, -1, 1, 1, 1)
            if residual is not None:
                ref = ref + residual.float()
            if relu:
                ref = ref.relu()
            self._record('conv3d_g', f'{tuple(x.shape)}->{cout} s{stride} p{padding} T{transposed} k1{kernel1} '
                         f'fused={scale is not None} res={residual is not None} relu={relu}', got.float(), ref)
            return got

        def conv_c32(x, packed, relu=False, acc_in=None, out_f32=False, depth_chunk=0, stats=False):
            got = real['conv3d_k3_c32'](x, packed, relu=relu, acc_in=acc_in, out_f32=out_f32,
                                        depth_chunk=depth_chunk, stats=stats)
            _, w, off, tr = self.packs[packed.data_ptr()]
            assert not tr, 'forward launches only'
            ref = F.conv3d(x.float(), w[:, off:off + 32], padding=1)
            if acc_in is not None:
                ref = ref + acc_in.permute(0, 4, 1, 2, 3)
            if relu:
                ref = ref.relu()
            y = got[0] if stats else got
            yv = y.permute(0, 4, 1, 2, 3) if out_f32 else y.float()
            self._record('conv3d_k3_c32', f'{tuple(x.shape)} off={off} acc={acc_in is not None} f32={out_f32} '
                         f'stats={stats}', yv, ref)
            if stats:
                # the epilogue's (count, mean, M2) partials merge to the moments of the stored tensor
                p = got[1].double()                                   # (N, 32, splits, 3)
                cnt, mean, m2 = p[..., 0], p[..., 1], p[..., 2]
                n = cnt.sum(-1)
                mu = (cnt * mean).sum(-1) / n
                var = (m2.sum(-1) + (cnt * (mean - mu[..., None]) ** 2).sum(-1)) / n
                yd = y.double().flatten(2)
                assert float((n - yd.shape[2]).abs().max()) == 0
                self._record('conv3d_k3_c32.stats.mean', '', mu, yd.mean(-1), floor=float(yd.std()))
                self._record('conv3d_k3_c32.stats.var', '', var, yd.var(-1, unbiased=False))
            return got

        def gnorm(x, num_groups, weight, bias, eps=1e-5, relu=False, partials=None, residual=None):
            got = real_gn(x, num_groups, weight, bias, eps, relu, partials, residual)
            ref = F.group_norm(x.float(), int(num_groups), weight.float(), bias.float(), eps)
            if residual is not None:
                ref = ref + residual.float()
            if relu:
                ref = ref.relu()
            self._record('group_norm', f'{tuple(x.shape)} G={num_groups} partials={partials is not None} '
                         f'res={residual is not None} relu={relu}', got.float(), ref)
            return got

        monkeypatch.setattr(self.cv, 'conv3d_g', conv_g)
        monkeypatch.setattr(self.cv, 'conv3d_k3_c32', conv_c32)
        monkeypatch.setattr(self.cv, 'pack_conv3d_g_weights', pack_g)
        monkeypatch.setattr(self.cv, 'pack_conv3d_weights', pack_c32)
        monkeypatch.setattr(self.gn, 'group_norm', gnorm)

    def _record(self, kind, detail, got, ref, floor=0.0):
        assert got.shape == ref.shape, (kind, detail, got.shape, ref.shape)
        scale = max(float(ref.abs().max()), floor, 1e-30)
        self.records.append((kind, detail, float((got.double() - ref.double()).abs().max()) / scale))

    def count(self, kind):
        return sum(1 for r in self.records if r[0] == kind)

    def assert_all_within(self, bar):
        bad = [r for r in self.records if not r[2] <= bar]
        assert not bad, 'launches over the bf16 bar:\n' + '\n'.join(f'{k} {d}: {e:.3e}' for k, d, e in bad)


def test_every_launch_of_the_stereo_path_bf16_matches_its_fp32_replay(pkg, monkeypatch):
    """DfMStereoPath (config K's model dict) at inference in bf16 with the channels-last volume"""
    with open(os.path.join(util.GOLDEN, 'configs_dfm.json')) as f:
        model = json.load(f)['dfm_r34_1x8_kitti-3d-3class.py']['model']
    model = dict(model)
    model['depth_cfg'] = dict(model['depth_cfg'], num_bins=32)
    model['depth_head'] = dict(model['depth_head'], depth_cfg=dict(model['depth_head']['depth_cfg'], num_bins=32))
    model['voxel_cfg'] = dict(point_cloud_range=[2, -6.4, -3, 27.6, 6.4, 1], voxel_size=[0.2, 0.2, 0.2])
    torch.manual_seed(5)
    path = pkg.DfMStereoPath(model).cuda().eval().to(torch.bfloat16)
    path.backbone_stereo.volume_memory_format = torch.channels_last_3d
    # every launch replayed: dres0 / dres0_mono as their own launches on the materialised volume (the
    # fused plane-sweep + dres0 kernel has its own parity tests: tests/test_sweep_conv_gpu.py)
    path.backbone_stereo.fuse_sweep_dres0 = False
    H, W = 256, 512
    gen = torch.Generator().manual_seed(7)
    feats = [[torch.randn(1, c, H // s, W // s, generator=gen).cuda().bfloat16()
              for c, s in ((3, 1), (64, 2), (128, 4), (128, 4), (128, 4))] for _ in range(2)]
    K = util.KITTI_P2.copy()
    meta = dict(ori_cam2img=K, cam2img=K.tolist(), cur2prevs=util.pose(0.5, 0.02, 0.0, -0.8)[None],
                ori_shape=(H, W, 3), pad_shape=(H, W, 3), crop_offset=[0, 0], flip=False, scale_factor=[1.0])
    rp = LayerReplay(monkeypatch)
    with torch.no_grad():
        out = path(feats[0], feats[1], [meta])
    assert out['bev_feat'].shape == (1, 64, 64, 128)
    # 2 x 6 hourglass + 2 x 7 SPPUNetNeck + 7 BEVHourglass launches of the general kernel; the 32 -> 32
    # kernel: dres0 (2 halves + mono), dres1 x 2, pred.0 x 2, voxel_convs (2 halves)
    assert rp.count('conv3d_g') == 12 + 14 + 7 and rp.count('conv3d_k3_c32') == 9, \
        (rp.count('conv3d_g'), rp.count('conv3d_k3_c32'))
    assert rp.count('group_norm') >= 12 + 6 + 1 and rp.count('conv3d_k3_c32.stats.mean') >= 6
    rp.assert_all_within(BF16_BAR)


@pytest.mark.parametrize('neck,cin,seed', [('OutdoorImVoxelNeck', 64, 71), ('DfMNeck', 128, 72)])
def test_every_launch_of_the_voxel_necks_bf16_matches_its_fp32_replay(monkeypatch, neck, cin, seed):
    """the two voxel necks at their real widths (64 -> 128 -> 256 -> 256): fused conv + BN (+residual)
    (+ReLU) launches in eval mode; convolution launches + HipBatchNorm3d in training mode"""
    mods = importlib.import_module('depth-from-motion_amd.modules')
    m = getattr(mods, neck)(in_channels=64, out_channels=256, **({'num_frames': 2} if neck == 'DfMNeck' else {}))
    m.load_state_dict(util.synthetic_state_dict(m, seed), strict=True)
    m = m.cuda().to(torch.bfloat16).eval()
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(2, cin, 10, 14, 12, generator=gen).cuda().bfloat16().contiguous(memory_format=torch.channels_last_3d)
    rp = LayerReplay(monkeypatch)
    with torch.no_grad():
        y = m(x)[0]
    assert y.shape == (2, 256, 14, 10)
    n = 9 * (2 if neck == 'DfMNeck' else 1)
    assert rp.count('conv3d_g') == n
    rp.assert_all_within(BF16_BAR)
    rp.records.clear()
    m.train()
    with torch.no_grad():
        m(x)
    assert rp.count('conv3d_g') == n
    rp.assert_all_within(BF16_BAR)


# ---------------------------------------------------------------------------------------------
# the voxel necks' real extent
# ---------------------------------------------------------------------------------------------
NECK_LAYERS = [
    # cin, cout, Nz, stride, padding, fused epilogue
    (64, 64, 12, 1, 1, True),                 # ResModule(64)
    (64, 128, 12, (1, 1, 2), 1, False),       # 12 -> 6
    (128, 128, 12, 1, 1, False),              # DfMNeck stereo stack: ResModule(64 * 2 frames)
    (128, 128, 6, 1, 1, True),                # ResModule(128)
    (128, 256, 6, (1, 1, 2), 1, False),       # 6 -> 3
    (256, 256, 3, 1, 1, True),                # ResModule(256)
    (256, 256, 3, 1, (1, 1, 0), False),       # 3 -> 1
]


@pytest.mark.parametrize('cin,cout,nz,stride,padding,fused', NECK_LAYERS)
def test_neck_layers_at_waymo_extent_against_fp64_subset(cv, cin, cout, nz, stride, padding, fused):
    """every convolution shape of OutdoorImVoxelNeck / DfMNeck on the 220 x 300 x Nz Waymo grid
    (imvoxel_neck.py:26-55, dfm_neck.py:29-95): a strided voxel subset against fp64 on the GPU, with
    the folded-BatchNorm + residual + ReLU epilogue where the block has one"""
    dev = torch.device('cuda:0')
    size = (220, 300, nz)
    g = torch.Generator().manual_seed(cin + cout + nz)
    x = torch.randn(1, cin, *size, generator=g).bfloat16().to(dev).contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5).bfloat16().float().to(dev)
    stride, padding = cv._triple(stride), cv._triple(padding)
    kw = {}
    if fused:
        kw = dict(scale=(1 + 0.1 * torch.randn(cout, generator=g)).to(dev), shift=(0.1 * torch.randn(cout, generator=g)).to(dev),
                  residual=torch.randn(1, cout, *size, generator=g).bfloat16().to(dev).contiguous(
                      memory_format=torch.channels_last_3d), relu=True)
    out = cv.conv3d_g(x, cv.pack_conv3d_g_weights(w, cin, cout), cout, stride, padding, **kw)
    D, H, W = out.shape[2:]
    assert (D, H) == (220, 300) and W == (nz + 2 * padding[2] - 3) // stride[2] + 1
    xp = F.pad(x.double(), (padding[2], padding[2], 1, 1, 1, 1))
    wd = w.double()
    ref = torch.zeros(1, cout, len(range(0, D, 7)), len(range(0, H, 11)), W, dtype=torch.float64, device=dev)
    for kd in range(3):
        for kh in range(3):
            for kwi in range(3):
                patch = xp[:, :, kd:kd + D, kh:kh + H, kwi::stride[2]][..., :W][:, :, ::7, ::11]
                ref += torch.einsum('ncdhw,oc->nodhw', patch, wd[:, :, kd, kh, kwi])
    if fused:
        ref = (ref * kw['scale'].double().view(1, -1, 1, 1, 1) + kw['shift'].double().view(1, -1, 1, 1, 1) +
               kw['residual'][:, :, ::7, ::11].double()).relu()
    np.testing.assert_allclose(out[:, :, ::7, ::11].double().cpu().numpy(), ref.cpu().numpy(), rtol=2.0 ** -7,
                               atol=2e-3)
