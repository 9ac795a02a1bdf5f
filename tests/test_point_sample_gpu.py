"""GPU parity of the multi-view lifting kernel, through the C ABI.
Bar: fp32 bit-exact vs oracle and reference fixtures; bf16 storage bit-exact vs
bf16(oracle(bf16-rounded inputs))."""
import importlib
import os

import numpy as np
import pytest
import torch

from oracle import dfm_oracle as orc
from tests import util
from tests.test_point_sample_oracle import mv_cases, run_oracle_mv

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def pkg():
    assert torch.cuda.is_available()
    return importlib.import_module('depth-from-motion_amd')


def meta_from_fixture(z):
    nvf = int(z['num_views']) * int(z['num_frames'])
    meta = {'ori_lidar2img': [m for m in z['lidar2img']],
            'input_shape': tuple(int(v) for v in z['input_shape']),
            'img_shape': [tuple(int(v) for v in z['img_shape']) + (3,)] * nvf}
    if z['scale'].size:
        meta['scale_factor'] = z['scale']
    if bool(z['flip']):
        meta['flip'] = True
    if z['crop'].size:
        meta['img_crop_offset'] = z['crop']
    return meta


@pytest.mark.parametrize('path', mv_cases(), ids=lambda p: os.path.basename(p)[:-4])
def test_mv_fp32_bitexact_vs_reference_fixture(pkg, path):
    z = np.load(path)
    feats = torch.from_numpy(z['feats']).cuda()
    out = pkg.mv_feature_transformation(feats, [meta_from_fixture(z)], int(z['num_views']),
                                        int(z['num_frames']), z['voxel_range'], z['n_voxels'],
                                        str(z['aggregate']))
    torch.cuda.synchronize()
    assert out.shape == z['ref_out'].shape
    assert np.array_equal(util.bits(out.cpu().numpy()), util.bits(z['ref_out']))


@pytest.mark.parametrize('path', mv_cases(), ids=lambda p: os.path.basename(p)[:-4])
def test_mv_bf16_exact_vs_oracle(pkg, path):
    z = dict(np.load(path))
    z['feats'] = orc.bf16_round(z['feats'])
    ref = orc.bf16_round(run_oracle_mv(z))
    out = pkg.mv_feature_transformation(torch.from_numpy(z['feats']).cuda().bfloat16(),
                                        [meta_from_fixture(z)], int(z['num_views']),
                                        int(z['num_frames']), z['voxel_range'], z['n_voxels'],
                                        str(z['aggregate']))
    assert out.dtype == torch.bfloat16
    assert np.array_equal(util.bits(out[0].float().cpu().numpy()), util.bits(ref))


def test_point_sample_reference_golden_vector(pkg):
    z = np.load(os.path.join(util.GOLDEN, 'helpers.npz'))
    img = (torch.arange(370 * 1224, dtype=torch.float32).reshape(1, 1, 370, 1224) / (370 * 1224))
    out = pkg.point_sample({}, img.cuda(), torch.from_numpy(z['ps_points']),
                           torch.from_numpy(z['ps_lidar2img']), 'LIDAR', 1, 0, False, (370, 1224),
                           (370, 1224), aligned=True)
    got = out.cpu().numpy().reshape(-1)
    assert np.array_equal(util.bits(got), util.bits(z['ps_out'].reshape(-1)))
    np.testing.assert_allclose(got, z['ps_expected'], rtol=1e-4)


@pytest.mark.parametrize('aligned', [False, True])
def test_point_sample_valid_flag_vs_oracle(pkg, aligned):
    z = np.load(os.path.join(util.GOLDEN, 'mv_concat_2frames_aug.npz'))
    feat, proj = z['feats'][0, 3], z['lidar2img'][3]
    ref, ok = orc.point_sample(feat, z['points'], proj, (0.95, 1.05), (3.0, 2.0), True,
                               float(z['img_shape'][1]), z['input_shape'], aligned=aligned,
                               valid_flag=True)
    out, valid = pkg.point_sample({}, torch.from_numpy(feat[None]).cuda(),
                                  torch.from_numpy(z['points']), torch.from_numpy(proj), 'LIDAR',
                                  torch.tensor([0.95, 1.05]), torch.tensor([3.0, 2.0]), True,
                                  tuple(z['input_shape']), tuple(z['img_shape']), aligned=aligned,
                                  valid_flag=True)
    assert np.array_equal(valid.cpu().numpy(), ok)
    assert np.array_equal(util.bits(out.cpu().numpy()), util.bits(ref))
    assert 0.05 < ok.mean() < 0.95


def test_waymo_shape_sub_volume_vs_oracle(pkg):
    """config W geometry (5 views x 2 frames, 64 ch, 208x312 maps, 220x300x12 voxels):
    checks a strided subset of voxels against the oracle, bit-exact."""
    rng = np.random.RandomState(1)
    nv, nf, C, hf, wf = 5, 2, 64, 208, 312
    feats = rng.randn(nv * nf, C, hf, wf).astype(np.float32)
    from tests.golden.make_golden import waymo_like_cameras
    cams = waymo_like_cameras(nv, nf, 5)
    cams[:, 0, :] *= 1248 / 156.0   # intrinsics of the small fixture cameras -> 832x1248 input
    cams[:, 1, :] *= 832 / 104.0
    vr, nvox = [-35.0, -75.0, -2.0, 75.0, 75.0, 4.0], (220, 300, 12)
    meta = {'ori_lidar2img': [m for m in cams], 'input_shape': (832, 1248),
            'img_shape': [(832, 1248, 3)] * (nv * nf)}
    out = pkg.mv_feature_transformation(torch.from_numpy(feats)[None].cuda(), [meta], nv, nf, vr,
                                        nvox, 'concat')[0].cpu().numpy()
    assert out.shape == (2 * C, 220, 300, 12)
    pts = pkg.voxel_centers(vr, nvox).numpy().reshape(12, 300, 220, 3)
    sub = pts[::3, ::7, ::5].reshape(-1, 3)
    nz, ny, nx = pts[::3, ::7, ::5].shape[:3]
    ref = orc.mv_feature_transformation(feats, sub, cams, (nx, ny, nz), nv, nf, (832, 1248),
                                        (832, 1248), aggregate='concat')
    got = out[:, ::5, ::7, ::3]
    assert got.shape == ref.shape
    assert np.array_equal(util.bits(got), util.bits(ref))
    assert (ref != 0).mean() > 0.2


@pytest.mark.parametrize('aligned', [True, False])
@pytest.mark.parametrize('tag,kw', [('plain', dict(scale=(1.0, 1.0), crop=(0.0, 0.0), flip=False)),
                                    ('aug', dict(scale=(0.95, 1.05), crop=(3.0, 2.0), flip=True))])
def test_voxel_sample_bitexact_vs_reference_fixture(pkg, tag, kw, aligned):
    z = np.load(os.path.join(util.GOLDEN, 'voxel_sample.npz'))
    out = pkg.voxel_sample(torch.from_numpy(z['vox']).cuda(), z['voxel_range'], z['voxel_size'],
                           torch.from_numpy(z['depth_samples']), torch.from_numpy(z['proj']), 4,
                           torch.tensor(kw['scale']), torch.tensor(kw['crop']), kw['flip'],
                           (104, 156), (100, 150), aligned=aligned,
                           proj_inv=torch.from_numpy(z['proj_inv']))  # inverse taken where the fixture was made
    ref = z[f'out_{tag}_{"tri" if aligned else "near"}']
    assert np.array_equal(util.bits(out.cpu().numpy()), util.bits(ref))


def test_voxel_sample_with_host_inverse_is_close(pkg):
    """default path: the inverse is recomputed on this host (LAPACK may differ in the last
    bits from the machine that made the fixture) -> rtol 1e-4 like the reference's own tests"""
    z = np.load(os.path.join(util.GOLDEN, 'voxel_sample.npz'))
    out = pkg.voxel_sample(torch.from_numpy(z['vox']).cuda(), z['voxel_range'], z['voxel_size'],
                           torch.from_numpy(z['depth_samples']), torch.from_numpy(z['proj']), 4,
                           torch.tensor([1.0, 1.0]), torch.tensor([0.0, 0.0]), False, (104, 156),
                           (100, 150), aligned=True)
    np.testing.assert_allclose(out.cpu().numpy(), z['out_plain_tri'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('aligned', [True, False])
def test_voxel_sample_backward_vs_torch_grid_sample(pkg, aligned):
    """gradient w.r.t. the voxel features (dfm_voxel_sample_bwd) against torch autograd of a
    grid_sample whose sampling positions reproduce the forward output (linearity probe: the op
    is linear in the voxel features, so <out(v), g> must equal <v, grad> for random v, g, and the
    gradient must equal the forward applied to one-hot perturbations summed -- checked through
    the adjoint identity on two independent random pairs)."""
    z = np.load(os.path.join(util.GOLDEN, 'voxel_sample.npz'))
    dev = torch.device('cuda:0')
    args = (z['voxel_range'], z['voxel_size'], torch.from_numpy(z['depth_samples']),
            torch.from_numpy(z['proj']), 4, torch.tensor([0.95, 1.05]), torch.tensor([3.0, 2.0]), True,
            (104, 156), (100, 150))
    kw = dict(aligned=aligned, proj_inv=torch.from_numpy(z['proj_inv']))
    rng = np.random.RandomState(4)
    v = torch.from_numpy(rng.randn(*z['vox'].shape).astype(np.float32)).to(dev).requires_grad_(True)
    out = pkg.voxel_sample(v, *args, **kw)
    g = torch.from_numpy(rng.randn(*out.shape).astype(np.float32)).to(dev)
    out.backward(g)
    # adjoint identity <A v2, g> == <v2, A^T g> for an independent v2
    v2 = torch.from_numpy(rng.randn(*z['vox'].shape).astype(np.float32)).to(dev)
    lhs = float((pkg.voxel_sample(v2, *args, **kw).double() * g.double()).sum())
    rhs = float((v2.double() * v.grad.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs)), (lhs, rhs)
    assert float(v.grad.abs().sum()) > 0


@pytest.mark.parametrize('aligned', [True, False])
def test_voxel_sample_backward_vs_the_oracle_operator(pkg, aligned):
    """the gradient against the ORACLE's forward, not the library's own: voxel_sample is linear in the voxel
    features, out = A v; the CPU oracle (point_fusion.py:324-410 restated) applied to one-hot volumes -- one
    channel per voxel, a single call -- gives A column by column, and dfm_voxel_sample_bwd must return A^T g."""
    z = np.load(os.path.join(util.GOLDEN, 'voxel_sample.npz'))
    Nx, Ny, Nz = z['vox'].shape[2:]
    nvox = Nx * Ny * Nz
    kw = dict(scale=(0.95, 1.05), crop=(3.0, 2.0), flip=True)
    onehot = np.zeros((1, nvox, Nx, Ny, Nz), np.float32)
    onehot.reshape(nvox, nvox)[np.arange(nvox), np.arange(nvox)] = 1.0
    A = orc.voxel_sample(onehot, z['voxel_range'], z['voxel_size'], z['depth_samples'], z['proj_inv'], 4,
                         kw['scale'], kw['crop'], kw['flip'], (104, 156), (100, 150), aligned=aligned)[0]
    A = A.reshape(nvox, -1).astype(np.float64)          # row k: the output of a one at voxel k
    rng = np.random.RandomState(12)
    dev = torch.device('cuda:0')
    v = torch.from_numpy(rng.randn(1, 2, Nx, Ny, Nz).astype(np.float32)).to(dev).requires_grad_(True)
    out = pkg.voxel_sample(v, z['voxel_range'], z['voxel_size'], torch.from_numpy(z['depth_samples']),
                           torch.from_numpy(z['proj']), 4, torch.tensor(kw['scale']), torch.tensor(kw['crop']),
                           kw['flip'], (104, 156), (100, 150), aligned=aligned,
                           proj_inv=torch.from_numpy(z['proj_inv']))
    g = rng.randn(*out.shape).astype(np.float32)
    out.backward(torch.from_numpy(g).to(dev))
    got = v.grad.cpu().numpy()[0].reshape(2, nvox)
    want = (A @ g[0].reshape(2, -1).astype(np.float64).T).T   # (2, nvox)
    assert np.abs(want).max() > 0.5
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5 * np.abs(want).max())
    # and the forward of the same call equals the operator applied to v
    fwd = (v.detach().cpu().numpy()[0].reshape(2, nvox).astype(np.float64) @ A).reshape(out.shape[1:])
    np.testing.assert_allclose(out.detach().cpu().numpy()[0], fwd, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('path', mv_cases(), ids=lambda p: os.path.basename(p)[:-4])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_mv_channels_last_volume_is_the_same_tensor(pkg, path, dtype):
    """memory_format=channels_last_3d: the volume the NDHWC / MFMA neck convolutions read, written
    directly by the lifting kernel -- same shape, same values bit for bit, different strides;
    gradients flow through it like through the contiguous one"""
    z = np.load(path)
    feats = torch.from_numpy(z['feats']).cuda().to(dtype)
    args = ([meta_from_fixture(z)], int(z['num_views']), int(z['num_frames']), z['voxel_range'],
            z['n_voxels'], str(z['aggregate']))
    ref = pkg.mv_feature_transformation(feats, *args)
    f2 = feats.clone().requires_grad_(True)
    out = pkg.mv_feature_transformation(f2, *args, memory_format=torch.channels_last_3d)
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last_3d)
    assert torch.equal(out, ref)
    f1 = feats.clone().requires_grad_(True)
    g = torch.randn(ref.shape, device='cuda').to(dtype)
    pkg.mv_feature_transformation(f1, *args).backward(g)
    out.backward(g.contiguous(memory_format=torch.channels_last_3d))
    # fp32 atomics: the accumulation order differs from run to run
    torch.testing.assert_close(f1.grad.float(), f2.grad.float(), rtol=2e-2 if dtype == torch.bfloat16 else 1e-4,
                               atol=2e-2 if dtype == torch.bfloat16 else 1e-5)


FLOW_METAS = {  # the img_metas of tests/golden/make_golden_r02.py:FLOW_METAS that point_sample ran with
    'lidar_rsthf': dict(transformation_3d_flow=['R', 'S', 'T', 'HF'],
                        pcd_rotation=[[0.9801, -0.1987, 0.0], [0.1987, 0.9801, 0.0], [0.0, 0.0, 1.0]],
                        pcd_scale_factor=1.05, pcd_trans=[0.1, -0.2, 0.05], pcd_horizontal_flip=True),
    'lidar_hf_st': dict(transformation_3d_flow=['HF', 'S', 'T'], pcd_scale_factor=0.96,
                        pcd_trans=[-0.3, 0.4, 0.0], pcd_horizontal_flip=True),
}


@pytest.mark.parametrize('aligned', [True, False])
@pytest.mark.parametrize('name', sorted(FLOW_METAS))
def test_point_sample_with_3d_augmentation_flow_vs_reference(pkg, name, aligned):
    """img_meta['transformation_3d_flow'] is undone before the projection (point_fusion.py:57-58):
    output of the reference point_sample with the reference's own apply_3d_transformation.
    Without a rotation the transform is exact elementwise arithmetic -> bit-exact; with one the
    reference's N x 3 @ 3 x 3 BLAS product is restated as a multiply-add chain -> tolerance, and
    points whose nearest pixel / validity flips on the last bit are excluded."""
    z = np.load(os.path.join(util.GOLDEN, 'point_sample_flow.npz'))
    out, valid = pkg.point_sample(FLOW_METAS[name], torch.from_numpy(z['ps_img']).cuda(),
                                  torch.from_numpy(z['points']), torch.from_numpy(z['ps_lidar2img']), 'LIDAR',
                                  torch.tensor([0.125, 0.125]), torch.tensor([1.0, 0.5]), False, (46, 153),
                                  (46, 153), aligned=aligned, valid_flag=True)
    ref, ref_valid = z[f'ps__{name}__{int(aligned)}'], z[f'psvalid__{name}__{int(aligned)}']
    got, got_valid = out.cpu().numpy(), valid.cpu().numpy()
    assert (ref != 0).mean() > 0.5
    if 'R' not in FLOW_METAS[name]['transformation_3d_flow']:
        assert np.array_equal(util.bits(got), util.bits(ref))
        assert np.array_equal(got_valid, ref_valid)
    else:
        same = got_valid == ref_valid
        assert same.mean() > 0.99
        close = np.isclose(got, ref, rtol=1e-3, atol=1e-4).all(axis=1)
        assert close[same].mean() > (0.99 if aligned else 0.97)  # nearest: a pixel boundary may flip


def test_mv_feature_transformation_undoes_each_samples_flow(pkg):
    """a batch whose samples carry different 3-D flows == each sample run alone on its own
    pre-transformed points"""
    ps = importlib.import_module('depth-from-motion_amd.point_sample')
    z = np.load(mv_cases()[0])
    feats = torch.from_numpy(z['feats']).cuda()
    feats = torch.cat([feats, feats.flip(-1)], 0)
    metas = [dict(meta_from_fixture(z), **FLOW_METAS['lidar_hf_st']), meta_from_fixture(z)]
    nv, nf = int(z['num_views']), int(z['num_frames'])
    out = pkg.mv_feature_transformation(feats, metas, nv, nf, z['voxel_range'], z['n_voxels'], str(z['aggregate']))
    base = ps._device_voxel_centers(z['voxel_range'], z['n_voxels'], feats.device)
    for b in range(2):
        pts = ps._reverse_3d_flow(base, 'LIDAR', metas[b])
        alone = pkg.mv_feature_transformation(feats[b:b + 1], [meta_from_fixture(z)], nv, nf, z['voxel_range'],
                                              z['n_voxels'], str(z['aggregate']), points=pts)
        assert torch.equal(out[b], alone[0])
    assert not torch.equal(out[0], pkg.mv_feature_transformation(
        feats[:1], [meta_from_fixture(z)], nv, nf, z['voxel_range'], z['n_voxels'], str(z['aggregate']))[0])


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_mv_channels_last_view_features_are_sampled_in_place(pkg, dtype):
    """view features coming from a channels_last image backbone ((B*F*Nv, C, H, W) channels_last,
    viewed (B, F*Nv, C, H, W)) are the kernel's pixel-major layout: same volume, same gradient, no
    pixel-major copy (dfm_mv_desc.feats_channels_last)"""
    ps = importlib.import_module('depth-from-motion_amd.point_sample')
    z = np.load(mv_cases()[0])
    base = torch.from_numpy(z['feats']).cuda()                      # (1, F*Nv, C, H, W)
    B, V, C, H, W = base.shape
    reps = 16 // C + 1
    feats = base.repeat(2, 1, reps, 1, 1)[:, :, :16].to(dtype).contiguous()
    feats[1] = feats[1].flip(-1)
    cl = feats.reshape(2 * V, 16, H, W).contiguous(memory_format=torch.channels_last).view(2, V, 16, H, W)
    assert ps._views_channels_last(cl) and not ps._views_channels_last(feats)
    metas = [meta_from_fixture(z)] * 2
    args = (metas, int(z['num_views']), int(z['num_frames']), z['voxel_range'], z['n_voxels'], str(z['aggregate']))
    a = feats.clone().requires_grad_(True)
    b = cl.detach().requires_grad_(True)
    ya, yb = pkg.mv_feature_transformation(a, *args), pkg.mv_feature_transformation(b, *args)
    assert torch.equal(ya, yb) and float(ya.abs().sum()) > 0
    go = torch.randn_like(ya)
    ya.backward(go)
    yb.backward(go)
    torch.testing.assert_close(a.grad.float(), b.grad.float(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('C,dtype,nv,nf,agg', [(64, torch.bfloat16, 5, 2, 'concat'), (64, torch.bfloat16, 5, 1, 'mean'),
                                               (32, torch.bfloat16, 5, 3, 'mean'), (64, torch.float32, 5, 2, 'mean'),
                                               (16, torch.float32, 3, 2, 'concat')])
def test_batched_lanes_per_voxel_kernel_is_bit_identical_to_the_per_sample_kernel(pkg, C, dtype, nv, nf, agg):
    """channels-last views -> channels-last volume: ``dfm_point_sample_mv_fwd_batched`` (several lanes
    per voxel, the whole batch in one launch, per-sample image transforms from a by-value table)
    against the lane-per-voxel kernel launched per sample on NCHW views (the path the reference
    fixtures pin bit-exactly above)"""
    import sys
    sys.path.insert(0, util.GOLDEN)
    try:
        import make_golden as g1
    finally:
        sys.path.remove(util.GOLDEN)
    B, hf, wf = 3, 26, 39
    gen = torch.Generator().manual_seed(C + nv * nf)
    feats = torch.randn(B, nv * nf, C, hf, wf, generator=gen).cuda().to(dtype)
    feats_cl = feats.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)
    metas = []
    for b in range(B):
        meta = {'ori_lidar2img': [m for m in g1.waymo_like_cameras(nv, nf, 50 + b)], 'input_shape': (104, 156),
                'img_shape': [(100, 150, 3)] * (nv * nf)}
        if b == 1:
            meta.update(scale_factor=np.array([0.95, 1.05, 0.95, 1.05], np.float32), flip=True,
                        img_crop_offset=np.array([3.0, 2.0], np.float32))
        if b == 2:
            meta.update(img_crop_offset=np.array([1.0, 0.0], np.float32))
        metas.append(meta)
    vr, nvox = [-11.0, -15.0, -3.0, 11.0, 15.0, 3.0], [22, 30, 12]
    want = pkg.mv_feature_transformation(feats, metas, nv, nf, vr, nvox, agg)
    calls = []
    lib = pkg._capi.lib()
    real = lib.dfm_point_sample_mv_fwd_batched

    class Spy:
        def __call__(self, *a):
            rc = real(*a)
            calls.append(rc)
            return rc
    lib.dfm_point_sample_mv_fwd_batched = Spy()
    try:
        got = pkg.mv_feature_transformation(feats_cl, metas, nv, nf, vr, nvox, agg, memory_format=torch.channels_last_3d)
    finally:
        lib.dfm_point_sample_mv_fwd_batched = real
    assert calls == [0], 'the batched kernel took the call'
    assert got.is_contiguous(memory_format=torch.channels_last_3d) and got.shape == want.shape
    assert torch.equal(got.contiguous(), want)
