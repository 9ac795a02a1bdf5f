"""GPU: backward passes of the sampling ops against torch-CPU autograd of the
same expressions (grid_sample / Upsample / softmax).  Accumulation is by fp32
atomics in unspecified order: rtol 1e-4, atol 1e-5."""
import importlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import util

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-5)


@pytest.fixture(scope='module')
def pkg():
    assert torch.cuda.is_available()
    return importlib.import_module('depth-from-motion_amd')


def test_depth_head_backward(pkg):
    z = np.load(os.path.join(util.GOLDEN, 'depth_head_small.npz'))
    rng = np.random.RandomState(0)
    x = torch.from_numpy(z['x'])
    ds = torch.from_numpy(z['depth_samples'])
    gv = torch.from_numpy(rng.randn(*z['ref_vol'].shape).astype(np.float32))
    gs = torch.from_numpy(rng.randn(*z['ref_soft'].shape).astype(np.float32))
    gp = torch.from_numpy(rng.randn(*z['ref_pred'].shape).astype(np.float32))
    # reference: the torch ops DepthHead.forward is made of (depth_head.py:205-210)
    xr = x.clone().requires_grad_(True)
    vol = F.interpolate(xr, scale_factor=4, mode='trilinear', align_corners=True)
    soft = F.softmax(vol, dim=2)
    pred = torch.sum(soft * ds[None, None, :, None, None], 2)
    (vol * gv).sum().add((soft * gs).sum()).add((pred * gp).sum()).backward()
    xg = x.cuda().requires_grad_(True)
    v2, s2, p2 = pkg.depth_head_forward(xg, ds)
    ((v2 * gv.cuda()).sum() + (s2 * gs.cuda()).sum() + (p2 * gp.cuda()).sum()).backward()
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), **TOL)
    # only one of the three outputs used
    xg2 = x.cuda().requires_grad_(True)
    pkg.depth_head_forward(xg2, ds)[2].sum().backward()
    xr2 = x.clone().requires_grad_(True)
    torch.sum(F.softmax(F.interpolate(xr2, scale_factor=4, mode='trilinear', align_corners=True), 2)
              * ds[None, None, :, None, None], 2).sum().backward()
    np.testing.assert_allclose(xg2.grad.cpu().numpy(), xr2.grad.numpy(), **TOL)


@pytest.mark.parametrize('shape,scale', [((1, 1, 7, 9, 70), 4), ((2, 1, 5, 6, 40), 2)])
def test_depth_head_backward_tiles(pkg, shape, scale):
    """several 4 x 64 output tiles incl. partial ones; bf16 storage of the incoming gradients"""
    rng = np.random.RandomState(3)
    x = torch.from_numpy((rng.randn(*shape) * 2).astype(np.float32))
    D = shape[2]
    ds = torch.linspace(2.0, 59.6, scale * D)
    xr = x.clone().requires_grad_(True)
    vol = F.interpolate(xr, scale_factor=scale, mode='trilinear', align_corners=True)
    soft = F.softmax(vol, dim=2)
    pred = torch.sum(soft * ds[None, None, :, None, None], 2)
    gv, gs, gp = (torch.from_numpy(rng.randn(*t.shape).astype(np.float32)) for t in (vol, soft, pred))
    torch.autograd.backward([vol, soft, pred], [gv, gs, gp])
    xg = x.cuda().requires_grad_(True)
    v2, s2, p2 = pkg.depth_head_forward(xg, ds, scale)
    torch.autograd.backward([v2, s2, p2], [gv.cuda(), gs.cuda(), gp.cuda()])
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=2e-5)


def _f2v_torch(stereo, soft, sem, coords, cam2img, pad, dmin, dmax, sem_atten=True, stereo_atten=False):
    """feature_transformation.py:82-158 with torch ops (test-side restatement)."""
    outs = []
    for b in range(stereo.shape[0]):
        c3d = coords.reshape(-1, 3)
        rect = torch.stack([-c3d[:, 1], -c3d[:, 2], c3d[:, 0]], -1)
        p4 = torch.cat([rect, torch.ones(len(rect), 1)], 1) @ cam2img[b][:3].T
        uv = p4[:, :2] / p4[:, 2:3]
        ci = torch.cat([uv, rect[:, 2:]], -1).view(*coords.shape[:3], 3)
        v2d = (ci[..., 0] >= 0) & (ci[..., 0] <= pad[1]) & (ci[..., 1] >= 0) & (ci[..., 1] <= pad[0])
        norm = (ci - torch.tensor([0, 0, dmin])) / torch.tensor([pad[1] - 1, pad[0] - 1, dmax - dmin])
        norm = norm * 2. - 1.
        valid = (v2d & (norm[..., 2] >= -1) & (norm[..., 2] <= 1)).float()
        g = norm[None]
        vox = F.grid_sample(stereo[b:b + 1], g, align_corners=True) * valid[None, None]
        disp = F.grid_sample(soft[b:b + 1].detach(), g, align_corners=True) * valid[None, None]
        g2 = g.clone()
        g2[..., 2] = 0
        v2 = F.grid_sample(sem[b:b + 1].unsqueeze(2), g2, align_corners=True) * v2d.float()[None, None]
        if stereo_atten:
            vox = vox * disp
        outs.append(torch.cat([vox, v2 * disp if sem_atten else v2], 1))
    return torch.cat(outs)


@pytest.mark.parametrize('sem_atten,stereo_atten', [(True, True), (False, False), (False, True)])
@pytest.mark.parametrize('channels', [3, 8])
def test_frustum_to_voxel_backward_attention_switches(pkg, channels, sem_atten, stereo_atten):
    """gradients with stereo_atten_feat / sem_atten_feat (feature_transformation.py:141,154) through
    both backward kernels (scalar: 3 channels; pixel-major scratch: 8)"""
    z = np.load(os.path.join(util.GOLDEN, 'f2v_batch2.npz'))
    reps = channels // z['stereo'].shape[1] + 1
    rng = np.random.RandomState(7)
    stereo = torch.from_numpy(np.tile(z['stereo'], (1, reps, 1, 1, 1))[:, :channels] *
                              rng.rand(1, channels, 1, 1, 1).astype(np.float32)).contiguous()
    sem = torch.from_numpy(np.tile(z['sem'], (1, reps, 1, 1))[:, :channels] *
                           rng.rand(1, channels, 1, 1).astype(np.float32)).contiguous()
    soft = torch.from_numpy(z['softmax'])
    coords, cam = torch.from_numpy(z['coordinates_3d']), torch.from_numpy(z['cam2img'])
    pad = tuple(int(v) for v in z['pad_shape'])
    sr, mr = stereo.clone().requires_grad_(True), sem.clone().requires_grad_(True)
    ref = _f2v_torch(sr, soft, mr, coords, cam, pad, float(z['depth_min']), float(z['depth_max']),
                     sem_atten, stereo_atten)
    go = torch.from_numpy(rng.randn(*ref.shape).astype(np.float32))
    (ref * go).sum().backward()
    sg, mg = stereo.cuda().requires_grad_(True), sem.cuda().requires_grad_(True)
    metas = [{'cam2img': c.tolist(), 'pad_shape': pad + (3,)} for c in z['cam2img']]
    out = pkg.frustum_to_voxel_sample(sg, soft.cuda(), metas, mg, coords,
                                      dict(depth_min=float(z['depth_min']), depth_max=float(z['depth_max'])),
                                      sem_atten_feat=sem_atten, stereo_atten_feat=stereo_atten)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    (out * go.cuda()).sum().backward()
    np.testing.assert_allclose(sg.grad.cpu().numpy(), sr.grad.numpy(), **TOL)
    np.testing.assert_allclose(mg.grad.cpu().numpy(), mr.grad.numpy(), **TOL)


def test_frustum_to_voxel_backward(pkg):
    z = np.load(os.path.join(util.GOLDEN, 'f2v_batch2.npz'))
    stereo, soft, sem = (torch.from_numpy(z[k]) for k in ('stereo', 'softmax', 'sem'))
    coords, cam = torch.from_numpy(z['coordinates_3d']), torch.from_numpy(z['cam2img'])
    pad = tuple(int(v) for v in z['pad_shape'])
    go = torch.from_numpy(np.random.RandomState(1).randn(*z['ref_out'].shape).astype(np.float32))
    sr, mr = stereo.clone().requires_grad_(True), sem.clone().requires_grad_(True)
    ref = _f2v_torch(sr, soft, mr, coords, cam, pad, float(z['depth_min']), float(z['depth_max']))
    np.testing.assert_allclose(ref.detach().numpy(), z['ref_out'], rtol=1e-5, atol=1e-6)
    (ref * go).sum().backward()
    sg, mg = stereo.cuda().requires_grad_(True), sem.cuda().requires_grad_(True)
    metas = [{'cam2img': c.tolist(), 'pad_shape': pad + (3,)} for c in z['cam2img']]
    out = pkg.frustum_to_voxel_sample(sg, soft.cuda(), metas, mg, coords,
                                      dict(depth_min=float(z['depth_min']),
                                           depth_max=float(z['depth_max'])))
    (out * go.cuda()).sum().backward()
    np.testing.assert_allclose(sg.grad.cpu().numpy(), sr.grad.numpy(), **TOL)
    np.testing.assert_allclose(mg.grad.cpu().numpy(), mr.grad.numpy(), **TOL)


@pytest.mark.parametrize('channels', [1, 2, 3])
def test_frustum_to_voxel_backward_few_channels(pkg, channels):
    """1- and 2-channel maps: the pixel-major backward gives fewer than 4 lanes to a voxel; a
    wave must still scatter only its own 16 voxels (round-1 advisor finding: 2-4x gradients)."""
    z = np.load(os.path.join(util.GOLDEN, 'f2v_batch2.npz'))
    stereo = torch.from_numpy(z['stereo'])[:, :channels].contiguous()
    sem = torch.from_numpy(z['sem'])[:, :channels].contiguous()
    soft = torch.from_numpy(z['softmax'])
    coords, cam = torch.from_numpy(z['coordinates_3d']), torch.from_numpy(z['cam2img'])
    pad = tuple(int(v) for v in z['pad_shape'])
    sr, mr = stereo.clone().requires_grad_(True), sem.clone().requires_grad_(True)
    ref = _f2v_torch(sr, soft, mr, coords, cam, pad, float(z['depth_min']), float(z['depth_max']))
    go = torch.from_numpy(np.random.RandomState(3).randn(*ref.shape).astype(np.float32))
    (ref * go).sum().backward()
    sg, mg = stereo.cuda().requires_grad_(True), sem.cuda().requires_grad_(True)
    metas = [{'cam2img': c.tolist(), 'pad_shape': pad + (3,)} for c in z['cam2img']]
    out = pkg.frustum_to_voxel_sample(sg, soft.cuda(), metas, mg, coords,
                                      dict(depth_min=float(z['depth_min']),
                                           depth_max=float(z['depth_max'])))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    (out * go.cuda()).sum().backward()
    np.testing.assert_allclose(sg.grad.cpu().numpy(), sr.grad.numpy(), **TOL)
    np.testing.assert_allclose(mg.grad.cpu().numpy(), mr.grad.numpy(), **TOL)


@pytest.mark.parametrize('channels', [1, 2])
def test_mv_lifting_backward_few_channels(pkg, channels):
    """Multi-view lifting with 1 / 2 channels against the full-channel run: the gradient of
    channel c does not depend on how many channels ride along."""
    from tests.test_point_sample_gpu import meta_from_fixture
    z = np.load(os.path.join(util.GOLDEN, 'mv_mean_2frames.npz'))
    nv, nf = int(z['num_views']), int(z['num_frames'])
    feats = torch.from_numpy(z['feats'])
    go = torch.from_numpy(np.random.RandomState(2).randn(*z['ref_out'].shape).astype(np.float32))

    def grads(f, g):
        fg = f.cuda().requires_grad_(True)
        out = pkg.mv_feature_transformation(fg, [meta_from_fixture(z)], nv, nf, z['voxel_range'],
                                            z['n_voxels'], 'mean')
        (out * g.cuda()).sum().backward()
        return fg.grad.cpu().numpy()
    full = grads(feats, go)
    few = grads(feats[:, :, :channels].contiguous(), go[:, :channels].contiguous())
    np.testing.assert_allclose(few, full[:, :, :channels], **TOL)


@pytest.mark.parametrize('case', ['mv_mean_2frames', 'mv_concat_2frames_aug'])
def test_mv_lifting_backward(pkg, case):
    from oracle import dfm_oracle as orc
    from tests.test_point_sample_gpu import meta_from_fixture
    z = np.load(os.path.join(util.GOLDEN, case + '.npz'))
    nv, nf = int(z['num_views']), int(z['num_frames'])
    feats = torch.from_numpy(z['feats'])
    go = torch.from_numpy(np.random.RandomState(2).randn(*z['ref_out'].shape).astype(np.float32))
    fg = feats.cuda().requires_grad_(True)
    out = pkg.mv_feature_transformation(fg, [meta_from_fixture(z)], nv, nf, z['voxel_range'],
                                        z['n_voxels'], str(z['aggregate']))
    (out * go.cuda()).sum().backward()
    # reference: nearest sampling is a gather -> build it with index_select on the CPU from
    # the oracle's validity / pixel decisions, then autograd through the reduction
    fr = feats[0].clone().requires_grad_(True)
    C, hf, wf = fr.shape[1:]
    sc = (1.0, 1.0) if z['scale'].size == 0 else (z['scale'][0], z['scale'][1])
    cr = (0.0, 0.0) if z['crop'].size == 0 else (z['crop'][0], z['crop'][1])
    idx_img = torch.arange(hf * wf, dtype=torch.float32).reshape(1, hf, wf).numpy()
    vols, cnts = [], []
    for f in range(nf):
        vol, cnt = 0, 0
        for v in range(nv):
            i = f * nv + v
            pix, ok = orc.point_sample(idx_img, z['points'], z['lidar2img'][i], sc, cr, bool(z['flip']),
                                       float(z['img_shape'][1]), z['input_shape'], aligned=False,
                                       valid_flag=True)
            # pix == sampled linear pixel index (0 where outside; then masked by `inside`)
            plain = orc.point_sample(np.ones((1, hf, wf), np.float32), z['points'], z['lidar2img'][i],
                                     sc, cr, bool(z['flip']), float(z['img_shape'][1]),
                                     z['input_shape'], aligned=False, valid_flag=True)[0]
            inside = torch.from_numpy((plain[:, 0] > 0) & ok)
            gathered = fr[i].reshape(C, -1)[:, torch.from_numpy(pix[:, 0]).long()].T
            vol = vol + gathered * inside[:, None]
            cnt = cnt + torch.from_numpy(ok).long()
        vols.append(vol)
        cnts.append(cnt)
    if str(z['aggregate']) == 'mean':
        tot = sum(vols) / torch.clamp(sum(cnts), min=1)[:, None]
    else:
        tot = torch.cat([v / torch.clamp(c, min=1)[:, None] for v, c in zip(vols, cnts)], 1)
    nx, ny, nz = (int(v) for v in z['n_voxels'])
    ref = tot.reshape(nz, ny, nx, -1).permute(3, 2, 1, 0)
    np.testing.assert_allclose(ref.detach().numpy(), z['ref_out'][0], rtol=1e-6, atol=1e-6)
    (ref * go[0]).sum().backward()
    np.testing.assert_allclose(fg.grad[0].cpu().numpy(), fr.grad.numpy(), **TOL)


def test_backbone_two_stream_training_hooks_see_the_main_stream(pkg):
    """DfMBackbone trains its mono stack on a side HIP stream (modules.DfMBackbone._two_branches).  Gradient
    hooks -- DistributedDataParallel's reducer, parallel.GradientBucketReducer -- order their work against the
    stream that is current INSIDE the hook only, so every parameter's AccumulateGrad (and with it every hook)
    has to run on the main stream; and the gradients must be those of the one-stream run."""
    mods = importlib.import_module('depth-from-motion_amd.modules')
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    m = mods.DfMBackbone(in_channels=32, depth_cfg=dict(mode='UD', num_bins=32, depth_min=2, depth_max=59.6,
                                                        downsample_factor=4)).to(dev).to(torch.bfloat16).train()
    m.downsampled_depth = pkg.prepare_depth(dict(num_bins=32, depth_min=2, depth_max=59.6, downsample_factor=4))[0]
    m.volume_memory_format = torch.channels_last_3d
    P2 = np.array([[721.5377, 0, 609.5593, 44.85728], [0, 721.5377, 172.854, 0.2163791],
                   [0, 0, 1, 0.002745884], [0, 0, 0, 1]], dtype=np.float32)
    T = np.eye(4, dtype=np.float32)
    T[2, 3] = -0.8
    meta = dict(ori_cam2img=P2, cur2prevs=torch.from_numpy(T[None]), ori_shape=(375, 1242, 3),
                pad_shape=(64, 256, 3), crop_offset=[0, 55], flip=False, scale_factor=[1.0])
    cur = torch.randn(1, 32, 64, 256, device=dev).bfloat16().requires_grad_(True)
    prev = torch.randn(1, 32, 64, 256, device=dev).bfloat16().requires_grad_(True)
    main = torch.cuda.current_stream(dev)
    seen = {}
    handles = [p.register_post_accumulate_grad_hook(
        lambda p, n=n: seen.__setitem__(n, torch.cuda.current_stream(dev) == main))
        for n, p in m.named_parameters() if p.requires_grad]

    def run(two):
        m.two_streams = two
        m.zero_grad(set_to_none=True)
        cur.grad = prev.grad = None
        outs = m(cur, prev, [meta])
        g = torch.Generator(device=dev).manual_seed(5)
        torch.autograd.backward(list(outs), [torch.randn(o.shape, device=dev, generator=g).to(o.dtype) * 1e-2
                                             for o in outs])
        torch.cuda.synchronize()
        return {n: p.grad.float().clone() for n, p in m.named_parameters() if p.grad is not None}

    assert m.two_streams_training
    g2 = run(True)
    assert mods.DfMBackbone._side_streams.get(dev) is not None, 'the side stream was never used'
    off = [n for n, ok in seen.items() if not ok]
    assert seen and not off, f'hooks ran off the main stream for {off[:4]}'
    assert any(n.startswith('dres1_mono') for n in seen) and any(n.startswith('hg_mono') for n in seen)
    g1 = run(False)
    for h in handles:
        h.remove()
    assert g1.keys() == g2.keys()
    for n in g1:   # same kernels, same order per stack: bit-identical
        assert torch.equal(g1[n], g2[n]), n
