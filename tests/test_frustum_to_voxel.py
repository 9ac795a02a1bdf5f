"""FrustumToVoxel sampling stage: oracle vs fixtures produced by the reference
module (CPU, bit-exact) and HIP vs oracle / fixtures (GPU, bit-exact fp32;
bf16 storage exact vs bf16(oracle(bf16-rounded inputs)))."""
import glob
import importlib
import os

import numpy as np
import pytest
import torch

from oracle import dfm_oracle as orc
from tests import util


def cases():
    return sorted(glob.glob(os.path.join(util.GOLDEN, 'f2v_*.npz')))


def oracle_run(z, sem=True):
    return orc.frustum_to_voxel(z['stereo'], z['softmax'], z['sem'] if sem else None,
                                z['coordinates_3d'], z['cam2img'], z['pad_shape'],
                                float(z['depth_min']), float(z['depth_max']))


@pytest.mark.parametrize('path', cases(), ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_bitexact_vs_reference_module(path):
    z = np.load(path)
    out = oracle_run(z)
    assert np.array_equal(util.bits(out), util.bits(z['ref_out']))
    # the fixture covers both masks: some voxels project outside, some are beyond the depth range
    C = z['stereo'].shape[1]
    assert 0.05 < (z['ref_out'][:, :C] == 0).mean() < 0.95


VARIANTS = {  # tag in tests/golden/frustum_atten_variants.npz -> the module's switches
    'stereo_sem': dict(stereo_atten_feat=True, sem_atten_feat=True),
    'none': dict(stereo_atten_feat=False, sem_atten_feat=False),
    'stereo_only': dict(stereo_atten_feat=True, sem_atten_feat=False),
    'stereo_nocat': dict(stereo_atten_feat=True, sem_atten_feat=True),  # cat_img_feature=False
}


@pytest.mark.parametrize('tag', sorted(VARIANTS))
@pytest.mark.parametrize('name', ['f2v_small', 'f2v_batch2'])
def test_oracle_attention_switches_bitexact_vs_reference_module(name, tag):
    """stereo_atten_feat / sem_atten_feat (feature_transformation.py:141-142,154-155) against the
    sampled volume of the reference module built with those switches (make_golden_r02.py)."""
    z = np.load(os.path.join(util.GOLDEN, name + '.npz'))
    ref = np.load(os.path.join(util.GOLDEN, 'frustum_atten_variants.npz'))[f'{name}__{tag}']
    out = orc.frustum_to_voxel(z['stereo'], z['softmax'], None if tag == 'stereo_nocat' else z['sem'],
                               z['coordinates_3d'], z['cam2img'], z['pad_shape'],
                               float(z['depth_min']), float(z['depth_max']), **VARIANTS[tag])
    assert np.array_equal(util.bits(out), util.bits(ref))


def hip_run(z, dtype=torch.float32, sem=True, stereo_format=torch.contiguous_format, **kw):
    pkg = importlib.import_module('depth-from-motion_amd')
    dev = torch.device('cuda:0')
    metas = [{'cam2img': c.tolist(), 'pad_shape': tuple(int(v) for v in z['pad_shape']) + (3,)}
             for c in z['cam2img']]
    out = pkg.frustum_to_voxel_sample(
        torch.from_numpy(z['stereo']).to(dev).to(dtype).contiguous(memory_format=stereo_format),
        torch.from_numpy(z['softmax']).to(dev).to(dtype),
        metas, torch.from_numpy(z['sem']).to(dev).to(dtype) if sem else None,
        torch.from_numpy(z['coordinates_3d']),
        dict(depth_min=float(z['depth_min']), depth_max=float(z['depth_max'])), **kw)
    torch.cuda.synchronize()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize('tag', sorted(VARIANTS))
@pytest.mark.parametrize('name', ['f2v_small', 'f2v_batch2'])
def test_hip_attention_switches_bitexact_vs_reference_module(name, tag):
    z = np.load(os.path.join(util.GOLDEN, name + '.npz'))
    ref = np.load(os.path.join(util.GOLDEN, 'frustum_atten_variants.npz'))[f'{name}__{tag}']
    out = hip_run(z, sem=tag != 'stereo_nocat', **VARIANTS[tag]).cpu().numpy()
    assert np.array_equal(util.bits(out), util.bits(ref))
    # 16-byte channel blocks: the pixel-major kernel, channels-last in and out (bf16 vs the oracle)
    zz = {k: z[k] for k in z.files}
    reps = 8 // z['stereo'].shape[1] + 1
    zz['stereo'] = orc.bf16_round(np.tile(z['stereo'], (1, reps, 1, 1, 1))[:, :8])
    zz['sem'] = orc.bf16_round(np.tile(z['sem'], (1, reps, 1, 1))[:, :8])
    zz['softmax'] = orc.bf16_round(z['softmax'])
    sem = None if tag == 'stereo_nocat' else zz['sem']
    want = orc.bf16_round(orc.frustum_to_voxel(zz['stereo'], zz['softmax'], sem, z['coordinates_3d'],
                                               z['cam2img'], z['pad_shape'], float(z['depth_min']),
                                               float(z['depth_max']), **VARIANTS[tag]))
    for fmt in (torch.contiguous_format, torch.channels_last_3d):
        got = hip_run(zz, torch.bfloat16, sem=sem is not None, stereo_format=fmt, **VARIANTS[tag])
        assert np.array_equal(util.bits(got.float().cpu().numpy()), util.bits(want))


@pytest.mark.gpu
@pytest.mark.parametrize('path', cases(), ids=lambda p: os.path.basename(p)[:-4])
def test_hip_fp32_bitexact_vs_reference_fixture(path):
    z = np.load(path)
    out = hip_run(z).cpu().numpy()
    assert out.shape == z['ref_out'].shape
    assert np.array_equal(util.bits(out), util.bits(z['ref_out']))


@pytest.mark.gpu
@pytest.mark.parametrize('path', cases(), ids=lambda p: os.path.basename(p)[:-4])
def test_hip_bf16_exact_vs_oracle(path):
    z = dict(np.load(path))
    for k in ('stereo', 'softmax', 'sem'):
        z[k] = orc.bf16_round(z[k])
    ref = orc.bf16_round(oracle_run(z))
    out = hip_run(z, torch.bfloat16)
    assert np.array_equal(util.bits(out.float().cpu().numpy()), util.bits(ref))


@pytest.mark.gpu
def test_hip_without_semantic_branch():
    z = np.load(cases()[0])
    ref = oracle_run(z, sem=False)
    out = hip_run(z, sem=False).cpu().numpy()
    assert out.shape[1] == z['stereo'].shape[1]
    assert np.array_equal(util.bits(out), util.bits(ref))


@pytest.mark.gpu
def test_hip_kitti_config_shape_vs_oracle():
    """config K sizes: (1,32,72,80,320) cost volume, 288x320x1280 depth distribution,
    20x304x288 voxel grid; a z-slab of the grid is checked against the oracle."""
    rng = np.random.RandomState(0)
    C, D, H, W = 32, 72, 80, 320
    stereo = rng.randn(1, C, D, H, W).astype(np.float32)
    logits = torch.from_numpy(rng.randn(1, 1, 4 * D, 4 * H, 4 * W).astype(np.float32))
    soft = torch.softmax(logits, dim=2).numpy()
    sem = rng.randn(1, C, H, W).astype(np.float32)
    zs = torch.linspace(-3 + 0.1, 1 - 0.1, 20)
    ys = torch.linspace(-30.4 + 0.1, 30.4 - 0.1, 304)
    xs = torch.linspace(2 + 0.1, 59.6 - 0.1, 288)
    zz, yy, xx = torch.meshgrid(zs, ys, xs, indexing='ij')
    coords = torch.stack([xx, yy, zz], -1).numpy()
    K = util.KITTI_P2.copy()
    K[0, 2] -= 0.0
    K[1, 2] -= 55.0  # crop_offset (0, 55) folded into the augmented cam2img
    z = dict(stereo=stereo, softmax=soft, sem=sem, coordinates_3d=coords, cam2img=K[None],
             pad_shape=np.array([320, 1280]), depth_min=2.0, depth_max=59.6)
    out = hip_run(z).cpu().numpy()
    assert out.shape == (1, 64, 20, 304, 288)
    zsub = dict(z, coordinates_3d=coords[7:9])
    ref = oracle_run(zsub)
    assert np.array_equal(util.bits(out[:, :, 7:9]), util.bits(ref))
    assert 0.2 < (ref != 0).mean()


def test_geometry_tables_match_the_reference_generators():
    """prepare_coordinates_3d (dfm.py:174-211) bit-exact vs the grid stored by the reference
    run; prepare_depth (dfm.py:152-172) vs the closed form of SURVEY 8a a12."""
    import importlib
    pkg = importlib.import_module('depth-from-motion_amd')
    z = np.load(os.path.join(util.GOLDEN, 'f2v_small.npz'))
    c = pkg.prepare_coordinates_3d(dict(point_cloud_range=[2, -6.0, -3, 14.0, 6.0, 1],
                                        voxel_size=[0.5, 0.5, 0.5])).numpy()
    assert np.array_equal(util.bits(c), util.bits(z['coordinates_3d']))
    low, full = pkg.prepare_depth(dict(mode='UD', num_bins=288, depth_min=2, depth_max=59.6,
                                       downsample_factor=4))
    assert low.shape == (72,) and full.shape == (288,)
    np.testing.assert_allclose(low.numpy(), [2 + (i + 0.5) * 0.8 for i in range(72)], rtol=1e-6)
    zd = np.load(os.path.join(util.GOLDEN, 'depth_head_wide.npz'))
    _, full72 = pkg.prepare_depth(dict(num_bins=72, depth_min=2, depth_max=59.6, downsample_factor=4))
    assert np.array_equal(util.bits(full72.numpy()), util.bits(zd['depth_samples']))


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('C,Cs', [(8, 16), (40, 8), (16, 0)])
def test_hip_pixel_major_path_vs_oracle(dtype, C, Cs):
    """channel counts that are whole 16-byte blocks (fp32: 4, bf16: 8) take the
    pixel-major staged kernel, also with more than one 32-channel pass (C=40)"""
    rng = np.random.RandomState(C + Cs)
    D, H, W = 6, 10, 32
    base = np.load(os.path.join(util.GOLDEN, 'f2v_small.npz'))
    z = dict(stereo=rng.randn(2, C, D, H, W).astype(np.float32),
             softmax=torch.softmax(torch.from_numpy(rng.randn(2, 1, 4 * D, 4 * H, 4 * W).astype(np.float32)),
                                   dim=2).numpy(),
             sem=rng.randn(2, max(Cs, 1), H, W).astype(np.float32),
             coordinates_3d=base['coordinates_3d'], cam2img=np.repeat(base['cam2img'], 2, 0),
             pad_shape=base['pad_shape'], depth_min=base['depth_min'], depth_max=base['depth_max'])
    z['cam2img'][1, 0, 2] += 3.0
    if dtype == torch.bfloat16:
        for k in ('stereo', 'softmax', 'sem'):
            z[k] = orc.bf16_round(z[k])
    ref = oracle_run(z, sem=Cs > 0)
    out = hip_run(z, dtype, sem=Cs > 0).float().cpu().numpy()
    if dtype == torch.bfloat16:
        ref = orc.bf16_round(ref)
    assert out.shape[1] == C + Cs
    assert np.array_equal(util.bits(out), util.bits(ref))
    assert 0.05 < (ref != 0).mean()
    # a channels_last_3d cost volume is sampled in place (no pixel-major copy): same bits
    out_cl = hip_run(z, dtype, sem=Cs > 0, stereo_format=torch.channels_last_3d).float().cpu().numpy()
    assert np.array_equal(util.bits(out_cl), util.bits(ref))


@pytest.mark.gpu
@pytest.mark.parametrize('path', cases(), ids=lambda p: os.path.basename(p)[:-4])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_fused_depth_head_is_bit_identical_to_the_materialised_pipeline(path, dtype):
    """DepthHead -> FrustumToVoxel fusion (depth_head.py:205-207 +
    feature_transformation.py:130-158): sampling the distribution evaluated on the fly from the
    low-resolution cost and the column statistics == sampling the materialised
    softmax(Upsample_x4(cost)), bit for bit; depth_preds and materialize() agree as well."""
    pkg = importlib.import_module('depth-from-motion_amd')
    z = np.load(path)
    dev = torch.device('cuda:0')
    B, _, Ds, Hs, Ws = z['softmax'].shape
    gen = torch.Generator().manual_seed(Ds)
    cost = (torch.randn(B, 1, Ds // 4, Hs // 4, Ws // 4, generator=gen) * 4).to(dev).to(dtype)
    samples = torch.tensor([2 + (k + 0.5) * (57.6 / Ds) for k in range(Ds)])
    metas = [{'cam2img': c.tolist(), 'pad_shape': tuple(int(v) for v in z['pad_shape']) + (3,)}
             for c in z['cam2img']]
    stereo = torch.from_numpy(z['stereo']).to(dev).to(dtype)
    sem = torch.from_numpy(z['sem']).to(dev).to(dtype)
    coords = torch.from_numpy(z['coordinates_3d'])
    cfg = dict(depth_min=float(z['depth_min']), depth_max=float(z['depth_max']))
    vol, soft, pred = pkg.depth_head_forward(cost, samples, 4)
    with torch.no_grad():
        ref = pkg.frustum_to_voxel_sample(stereo, soft, metas, sem, coords, cfg)
        lazy, pred2 = pkg.depth_head_statistics(cost, samples, 4)
        assert lazy.shape == tuple(soft.shape)
        fused = pkg.frustum_to_voxel_sample(stereo, lazy, metas, sem, coords, cfg)
    assert torch.equal(fused, ref)
    C = stereo.shape[1]
    assert float(ref[:, C:].abs().sum()) > 0   # the distribution-weighted half is not trivially zero
    assert torch.equal(pred2, pred)
    assert torch.equal(lazy.materialize(), soft)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_channels_last_output_is_the_same_tensor(dtype):
    """memory_format=channels_last_3d (the default when stereo_feat is channels-last): the layout
    voxel_convs' MFMA convolution reads, written directly -- same values bit for bit"""
    pkg = importlib.import_module('depth-from-motion_amd')
    z = dict(np.load(os.path.join(util.GOLDEN, 'f2v_small.npz')))
    rng = np.random.RandomState(3)   # channel counts of whole 16-byte blocks
    z['stereo'] = rng.randn(1, 8, *z['stereo'].shape[2:]).astype(np.float32)
    z['sem'] = rng.randn(1, 8, *z['sem'].shape[2:]).astype(np.float32)
    ref = hip_run(z, dtype)
    assert ref.is_contiguous()
    cl = hip_run(z, dtype, stereo_format=torch.channels_last_3d)
    assert cl.shape == ref.shape and cl.is_contiguous(memory_format=torch.channels_last_3d)
    assert not cl.is_contiguous()
    assert torch.equal(cl, ref)
    # an NHWC semantic map (channels_last 2-D neck) is sampled in place: same tensor
    dev = torch.device('cuda:0')
    metas = [{'cam2img': c.tolist(), 'pad_shape': tuple(int(v) for v in z['pad_shape']) + (3,)}
             for c in z['cam2img']]
    sem = torch.from_numpy(z['sem']).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    assert not sem.is_contiguous()
    for fmt in (torch.contiguous_format, torch.channels_last_3d):
        got = pkg.frustum_to_voxel_sample(
            torch.from_numpy(z['stereo']).to(dev).to(dtype).contiguous(memory_format=fmt),
            torch.from_numpy(z['softmax']).to(dev).to(dtype), metas, sem, torch.from_numpy(z['coordinates_3d']),
            dict(depth_min=float(z['depth_min']), depth_max=float(z['depth_max'])))
        assert torch.equal(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_backward_reads_a_channels_last_gradient_in_place(dtype):
    """a channels_last_3d output (what the NDHWC voxel_convs reads) receives a channels_last_3d gradient from that
    convolution's backward: the backward kernel reads it where it lies (round 5; torch's strided re-layout to the
    planar form cost 2.1 ms per training step) -- same gradients as the planar output fed the same values"""
    pkg = importlib.import_module('depth-from-motion_amd')
    z = dict(np.load(os.path.join(util.GOLDEN, 'f2v_small.npz')))
    rng = np.random.RandomState(5)
    z['stereo'] = rng.randn(1, 8, *z['stereo'].shape[2:]).astype(np.float32)
    z['sem'] = rng.randn(1, 8, *z['sem'].shape[2:]).astype(np.float32)
    dev = torch.device('cuda:0')
    metas = [{'cam2img': c.tolist(), 'pad_shape': tuple(int(v) for v in z['pad_shape']) + (3,)} for c in z['cam2img']]
    cfg = dict(depth_min=float(z['depth_min']), depth_max=float(z['depth_max']))
    grads = {}
    gout = None
    for fmt in (torch.contiguous_format, torch.channels_last_3d):
        st = torch.from_numpy(z['stereo']).to(dev).to(dtype).contiguous(memory_format=fmt).requires_grad_(True)
        sem = torch.from_numpy(z['sem']).to(dev).to(dtype).requires_grad_(True)
        out = pkg.frustum_to_voxel_sample(st, torch.from_numpy(z['softmax']).to(dev).to(dtype), metas, sem,
                                          torch.from_numpy(z['coordinates_3d']), cfg)
        if gout is None:
            gout = torch.from_numpy(rng.randn(*out.shape).astype(np.float32)).to(dev).to(dtype)
        g = gout.contiguous(memory_format=fmt)
        assert g.is_contiguous(memory_format=fmt) and out.is_contiguous(memory_format=fmt)
        out.backward(g)
        torch.cuda.synchronize()
        grads[fmt] = (st.grad.float().contiguous(), sem.grad.float())
    for a, b in zip(grads[torch.contiguous_format], grads[torch.channels_last_3d]):
        assert float(a.abs().max()) > 0
        tol = 1e-5 if dtype == torch.float32 else 2e-2
        assert torch.allclose(a, b, rtol=tol, atol=tol * float(a.abs().max()))


# ---- backward as a gather (dfm_frustum_to_voxel_bwd_gather): a lane per cost-volume pixel x depth chunk -------

def _config_k_like(dtype, seed, Cs=32, yaw=0.0, ny=50, nz=6):
    """a small problem with config K's structure: 32 + 32 channels, the semantic map at the cost volume's
    resolution, a regular voxel grid (prepare_coordinates_3d's construction), KITTI-like intrinsics on the padded
    image, materialised and fused depth distributions"""
    pkg = importlib.import_module('depth-from-motion_amd')
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(seed)
    D, H, W = 18, 20, 80
    pad_h, pad_w = 4 * H, 4 * W
    stereo = torch.randn(1, 32, D, H, W, generator=gen)
    sem = torch.randn(1, Cs, H, W, generator=gen) if Cs else None
    cost = (torch.randn(1, 1, D, H, W, generator=gen) * 3).to(dev).to(dtype)
    dmin, dmax = 2.0, 16.4
    samples = torch.tensor([dmin + (k + 0.5) * ((dmax - dmin) / (4 * D)) for k in range(4 * D)])
    # voxel centres: x (depth) 2 .. 16.4 in 0.2 m steps, y -5 .. 5, z -1.5 .. 0.9 (pseudo-LiDAR frame), x fastest
    nx = 72
    xs = torch.linspace(dmin + 0.1, dmax - 0.1, nx)
    ys = torch.linspace(-5 + 0.1, 5 - 0.1, ny)
    zs = torch.linspace(-1.5 + 0.2, 0.9 - 0.2, nz)
    zz, yy, xx = torch.meshgrid(zs, ys, xs, indexing='ij')
    coords = torch.stack([xx, yy, zz], -1)
    f = 180.0
    K = np.array([[f, 0, pad_w / 2 - 3.3, 4.4], [0, f, pad_h / 2 + 1.7, 0.2], [0, 0, 1, 0.003], [0, 0, 0, 1]], np.float32)
    if yaw:
        K[0, 1] = yaw   # a skewed projection: the (y, z) solve is a genuine 2 x 2 system
    metas = [{'cam2img': K.tolist(), 'pad_shape': (pad_h, pad_w, 3)}]
    cfg = dict(depth_min=dmin, depth_max=dmax)
    _, soft, _ = pkg.depth_head_forward(cost, samples, 4)
    lazy, _ = pkg.depth_head_statistics(cost, samples, 4)
    return pkg, dev, stereo, sem, soft, lazy, metas, coords, cfg


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('fused', [False, True], ids=['materialised', 'fused_head'])
@pytest.mark.parametrize('fmt', [torch.contiguous_format, torch.channels_last_3d], ids=['planar', 'channels_last'])
@pytest.mark.parametrize('Cs,yaw', [(32, 0.0), (32, 9.0), (0, 0.0)])
def test_backward_gather_equals_the_scatter(dtype, fused, fmt, Cs, yaw):
    """the same (voxel, corner, weight) set summed in another order: gather (stores + a few atomics) against the
    pixel-major scatter; materialised and fused depth head, both gradient layouts, a skewed projection, no semantic
    branch"""
    f2v = importlib.import_module('depth-from-motion_amd.frustum_to_voxel')
    pkg, dev, stereo, sem, soft, lazy, metas, coords, cfg = _config_k_like(dtype, 40 + Cs + int(yaw), Cs, yaw)
    rng = torch.Generator().manual_seed(3)
    res = {}
    gout = None
    for gather in (True, False):
        st = stereo.to(dev).to(dtype).contiguous(memory_format=fmt).requires_grad_(True)
        sm = sem.to(dev).to(dtype).requires_grad_(True) if sem is not None else None
        calls = f2v._BWD_GATHER.get('calls', 0)
        with f2v.bwd_gather(gather):
            out = pkg.frustum_to_voxel_sample(st, lazy if fused else soft, metas, sm, coords, cfg)
            if gout is None:
                gout = torch.randn(out.shape, generator=rng).to(dev).to(dtype).contiguous(memory_format=fmt)
            out.backward(gout)
        torch.cuda.synchronize()
        assert f2v._BWD_GATHER.get('calls', 0) - calls == (1 if gather else 0), 'which form took the call'
        res[gather] = (st.grad.float().contiguous(), sm.grad.float() if sm is not None else None)
    for a, b in zip(res[True], res[False]):
        if a is None:
            continue
        assert float(b.abs().max()) > 0
        tol = 2e-5 if dtype == torch.float32 else 1.6e-2   # (bf16: the gradients themselves are rounded to bf16)
        diff = (a - b).abs()
        assert bool((diff <= tol * b.abs() + tol * float(b.abs().max())).all()), \
            f'max |diff| {float(diff.max()):.3e} of max |g| {float(b.abs().max()):.3e}, {int((diff > tol * float(b.abs().max())).sum())} cells off'


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('fused', [False, True], ids=['materialised', 'fused_head'])
@pytest.mark.parametrize('Cs', [32, 0])
def test_backward_gather_hands_a_channels_last_volume_its_gradient_in_place(dtype, fused, Cs):
    """a channels-last cost volume (the NDHWC stack) receives its gradient channels-last in its own type
    (dfm_frustum_to_voxel_bwd_gather_cl: the fp32 sums rounded once at the store) -- bit for bit the planar fp32
    gradient converted, and the layout the prediction convolution's backward produces for the same tensor"""
    f2v = importlib.import_module('depth-from-motion_amd.frustum_to_voxel')
    pkg, dev, stereo, sem, soft, lazy, metas, coords, cfg = _config_k_like(dtype, 90 + Cs, Cs)
    rng = torch.Generator().manual_seed(4)
    res, gout = {}, None
    for native in (True, False):
        st = stereo.to(dev).to(dtype).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
        sm = sem.to(dev).to(dtype).requires_grad_(True) if sem is not None else None
        calls = f2v._BWD_GATHER.get('calls', 0)
        with f2v.bwd_gather(True, native=native):
            out = pkg.frustum_to_voxel_sample(st, lazy if fused else soft, metas, sm, coords, cfg)
            if gout is None:
                gout = torch.randn(out.shape, generator=rng).to(dev).to(dtype).contiguous(
                    memory_format=torch.channels_last_3d)
            out.backward(gout)
        torch.cuda.synchronize()
        assert f2v._BWD_GATHER.get('calls', 0) - calls == 1, 'the gather form took the call'
        assert st.grad.dtype == dtype and st.grad.shape == st.shape
        if native:
            assert st.grad.is_contiguous(memory_format=torch.channels_last_3d) and not st.grad.is_contiguous()
        res[native] = (st.grad, sm.grad if sm is not None else None)
    assert float(res[False][0].abs().max()) > 0
    assert torch.equal(res[True][0], res[False][0])
    if sem is not None:  # (atomics: the order of the additions differs from run to run)
        a, b = res[True][1].float(), res[False][1].float()
        assert torch.allclose(a, b, rtol=2e-2, atol=2e-2 * float(b.abs().max()))
    assert f2v._BWD_GATHER['native'] and f2v._BWD_GATHER['on']


@pytest.mark.gpu
@pytest.mark.parametrize('fused', [False, True], ids=['materialised', 'fused_head'])
def test_backward_gather_walks_boxes_of_any_size(fused):
    """voxels much finer than a cost-volume pixel's footprint (0.06 m x 0.1 m against 0.36 m at the far planes): a
    pixel's candidate box is 6+ voxels wide.  The kernel used to cut boxes at 8 x 8 voxels and skip pixels whose
    half-width exceeded 4 -- contributions dropped without an error; it now walks the whole box."""
    f2v = importlib.import_module('depth-from-motion_amd.frustum_to_voxel')
    pkg, dev, stereo, sem, soft, lazy, metas, coords, cfg = _config_k_like(torch.float32, 77, 32, 0.0, ny=160, nz=24)
    rng = torch.Generator().manual_seed(6)
    res, gout = {}, None
    for gather in (True, False):
        st = stereo.to(dev).requires_grad_(True)
        sm = sem.to(dev).requires_grad_(True)
        calls = f2v._BWD_GATHER.get('calls', 0)
        with f2v.bwd_gather(gather):
            out = pkg.frustum_to_voxel_sample(st, lazy if fused else soft, metas, sm, coords, cfg)
            if gout is None:
                gout = torch.randn(out.shape, generator=rng).to(dev)
            out.backward(gout)
        torch.cuda.synchronize()
        assert f2v._BWD_GATHER.get('calls', 0) - calls == (1 if gather else 0), 'which form took the call'
        res[gather] = (st.grad.clone(), sm.grad.clone())
    for a, b in zip(res[True], res[False]):
        assert float(b.abs().max()) > 0
        diff = (a - b).abs()
        assert bool((diff <= 2e-5 * b.abs() + 2e-5 * float(b.abs().max())).all()), \
            f'max |diff| {float(diff.max()):.3e} of max |g| {float(b.abs().max()):.3e}'


@pytest.mark.gpu
def test_backward_gather_needs_a_regular_grid_and_says_so():
    """an irregular voxel grid (jittered centres) is detected on the device and takes the scatter form"""
    f2v = importlib.import_module('depth-from-motion_amd.frustum_to_voxel')
    pkg, dev, stereo, sem, soft, lazy, metas, coords, cfg = _config_k_like(torch.float32, 7)
    desc = type('D', (), dict(nz=coords.shape[0], ny=coords.shape[1], nx=coords.shape[2]))()
    flat = coords.reshape(-1, 3).to(dev)
    g = f2v._regular_grid(flat, desc)
    assert g is not None and abs(g[1] - 0.2) < 1e-4 and abs(g[0] - 2.1) < 1e-4
    jit = flat.clone()
    jit[12345, 1] += 0.05
    assert f2v._regular_grid(jit, desc) is None
