"""Training-mode DepthHead -> FrustumToVoxel fusion: DepthHead.loss evaluated from the
LOW-RESOLUTION cost (dfm_depth_loss_fused_fwd / _bwd) and FrustumToVoxel's backward with the depth head fused
(dfm_frustum_to_voxel_fused_bwd) against the materialised pipeline the reference runs
(dense_heads/depth_head.py:75-212, necks/feature_transformation.py:130-158): no (B, 1, 4D, 4H, 4W) tensor in
either direction.  The per-pixel loss is bit-identical; gradients agree to float-atomic summation order (and, in
bf16, to the rounding of the materialised gradient volume the fused path does not have)."""
import importlib
import os

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

LOSSES = ['ce', 'balanced_ce', 'focal', 'hard_ce', 'gaussian_0.5', 'laplacian_0.7']


@pytest.fixture(scope='module')
def pkg():
    assert torch.cuda.is_available(), 'GPU tests need a GPU (no fallback path exists)'
    return importlib.import_module('depth-from-motion_amd')


def _head(pkg, loss_type, nbins):
    mods = importlib.import_module('depth-from-motion_amd.modules')
    cfg = dict(type=loss_type, loss_weight=0.7)
    if 'balanced' in loss_type:
        cfg.update(fg_weight=5, bg_weight=1)
    if 'focal' in loss_type:
        cfg.update(alpha=0.75, gamma=2)
    m = mods.DepthHead(depth_cfg=dict(mode='UD', num_bins=nbins, min_depth=2, max_depth=59.6), with_convs=False,
                       depth_loss=cfg, downsample_factor=4, num_views=1)
    m.depth_samples = torch.tensor([2 + (k + 0.5) * (57.6 / nbins) for k in range(nbins)])
    return m


def _depth_img(B, H, W, seed, valid_frac=0.2):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, H, W, generator=g) * 70.0 - 5.0        # some below min / above max depth
    img[torch.rand(B, H, W, generator=g) > valid_frac] = 0.0   # LiDAR supervision: mostly empty
    return img


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('loss_type', LOSSES)
def test_loss_from_the_low_resolution_cost(pkg, loss_type, dtype):
    dev = torch.device('cuda:0')
    B, D, H, W = 2, 6, 9, 13   # odd sizes: the upsample's index / weight arithmetic off the easy cases
    gen = torch.Generator().manual_seed(3)
    cost0 = (torch.randn(B, 1, D, H, W, generator=gen) * 3).to(dev).to(dtype)
    m = _head(pkg, loss_type, 4 * D)
    img = _depth_img(B, 4 * H, 4 * W, 5).to(dev)
    fg = (torch.rand(B, 4 * H, 4 * W, generator=gen) > 0.5).to(dev)
    # materialised: DepthHead.forward -> loss on depth_volumes
    c1 = cost0.clone().requires_grad_(True)
    vol, soft, pred = m(c1)
    l1 = m.loss(pred.flatten(0, 1), vol.flatten(0, 1), img, depth_fgmask_img=fg)
    l1.backward()
    # fused: the lazy distribution in place of depth_volumes
    c2 = cost0.clone().requires_grad_(True)
    up, pred2, dist = m(c2, lazy=True)
    assert up is dist and dist.cost_with_grad is c2
    l2 = m.loss(pred2.flatten(0, 1), up, img, depth_fgmask_img=fg)
    l2.backward()
    assert float(l1) != 0.0
    assert torch.equal(l1.detach(), l2.detach()), (float(l1), float(l2))   # same logits, same arithmetic
    g1, g2 = c1.grad.float(), c2.grad.float()
    assert float(g1.abs().max()) > 0
    if dtype == torch.float32:
        assert torch.allclose(g2, g1, rtol=1e-4, atol=1e-6 * float(g1.abs().max()) + 1e-9)
    else:  # the unfused path rounds the (B, D, H, W) gradient volume and the result to bf16
        assert torch.allclose(g2, g1, rtol=3e-2, atol=2e-2 * float(g1.abs().max()))


def test_per_pixel_loss_is_bit_identical_at_config_k_columns(pkg):
    """288-bin columns (config K's depth axis), a narrow image: pixel_loss and the mask, bit for bit"""
    dh = importlib.import_module('depth-from-motion_amd.depth_head')
    dev = torch.device('cuda:0')
    B, D, H, W = 1, 72, 5, 16
    gen = torch.Generator().manual_seed(9)
    for dtype in (torch.float32, torch.bfloat16):
        cost = (torch.randn(B, 1, D, H, W, generator=gen) * 4).to(dev).to(dtype)
        samples = torch.tensor([2 + (k + 0.5) * 0.2 for k in range(4 * D)])
        img = _depth_img(B, 4 * H, 4 * W, 11, valid_frac=0.5).to(dev)
        vol, _, _ = dh.depth_head_forward(cost, samples, 4)
        p1, v1 = dh.depth_distribution_loss(vol.flatten(0, 1), img, samples, 'ce', 2, 59.6)
        dist, _ = dh.depth_head_statistics(cost, samples, 4)
        p2, v2 = dh.depth_distribution_loss(dist, img, samples, 'ce', 2, 59.6)
        assert bool(v1.any()) and torch.equal(v1, v2) and torch.equal(p1, p2)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_frustum_to_voxel_backward_with_the_depth_head_fused(pkg, dtype):
    z = np.load(os.path.join(util.GOLDEN, 'f2v_small.npz'))
    dev = torch.device('cuda:0')
    B, _, Ds, Hs, Ws = z['softmax'].shape
    gen = torch.Generator().manual_seed(Ds)
    cost = (torch.randn(B, 1, Ds // 4, Hs // 4, Ws // 4, generator=gen) * 4).to(dev).to(dtype)
    samples = torch.tensor([2 + (k + 0.5) * (57.6 / Ds) for k in range(Ds)])
    metas = [{'cam2img': c.tolist(), 'pad_shape': tuple(int(v) for v in z['pad_shape']) + (3,)} for c in z['cam2img']]
    coords = torch.from_numpy(z['coordinates_3d'])
    cfg = dict(depth_min=float(z['depth_min']), depth_max=float(z['depth_max']))
    _, soft, _ = pkg.depth_head_forward(cost, samples, 4)
    lazy, _ = pkg.depth_head_statistics(cost, samples, 4)
    for sem_att, st_att in ((True, False), (True, True)):
        outs = []
        for dist in (soft, lazy):
            stereo = torch.from_numpy(z['stereo']).to(dev).to(dtype).requires_grad_(True)
            sem = torch.from_numpy(z['sem']).to(dev).to(dtype).requires_grad_(True)
            out = pkg.frustum_to_voxel_sample(stereo, dist, metas, sem, coords, cfg, sem_atten_feat=sem_att,
                                              stereo_atten_feat=st_att)
            gsel = torch.Generator().manual_seed(1)
            go = torch.randn(out.shape, generator=gsel).to(dev).to(dtype)
            out.backward(go)
            outs.append((out.detach(), stereo.grad.float(), sem.grad.float()))
        (o1, gs1, gm1), (o2, gs2, gm2) = outs
        assert torch.equal(o1, o2)
        assert float(gm1.abs().max()) > 0 and float(gs1.abs().max()) > 0
        tol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
        assert torch.allclose(gs2, gs1, **tol) and torch.allclose(gm2, gm1, **tol)


def test_dfm_stereo_path_training_step_fused_vs_materialised(pkg):
    """the config's training path (neck -> backbone_stereo -> depth head -> FrustumToVoxel -> BEV + dense depth
    loss, dfm.py:288-356) with the depth head fused (no upsample_costs / softmax tensors, loss and both
    backwards from the low-resolution cost) against the same weights with ``fuse_depth_head = False``"""
    import json
    with open(os.path.join(util.GOLDEN, 'configs_dfm.json')) as f:
        model = json.load(f)['dfm_r34_1x8_kitti-3d-3class.py']['model']
    model = dict(model)
    model['depth_cfg'] = dict(model['depth_cfg'], num_bins=32)
    model['depth_head'] = dict(model['depth_head'], depth_cfg=dict(model['depth_head']['depth_cfg'], num_bins=32))
    model['voxel_cfg'] = dict(point_cloud_range=[2, -6.4, -3, 27.6, 6.4, 1], voxel_size=[0.2, 0.2, 0.2])
    H, W = 256, 512
    K = util.KITTI_P2.copy()
    results = []
    for fuse in (False, True):
        torch.manual_seed(11)
        path = pkg.DfMStereoPath(model).cuda()
        path.fuse_depth_head = fuse
        gen = torch.Generator().manual_seed(7)

        def pyramid():
            return [torch.randn(1, c, H // s, W // s, generator=gen).cuda()
                    for c, s in ((3, 1), (64, 2), (128, 4), (128, 4), (128, 4))]
        meta = dict(ori_cam2img=K, cam2img=K.tolist(), cur2prevs=util.pose(0.5, 0.02, 0.0, -0.8)[None],
                    ori_shape=(H, W, 3), pad_shape=(H, W, 3), crop_offset=[0, 0], flip=False, scale_factor=[1.0])
        out = path(pyramid(), pyramid(), [meta])
        lazy = not torch.is_tensor(out['upsample_costs'])
        assert lazy == fuse
        depth_img = (torch.rand(1, 1, H, W, generator=gen) * 60).cuda()
        depth_img[torch.rand(1, 1, H, W, generator=gen).cuda() < 0.8] = 0
        fg = (torch.rand(1, 1, H, W, generator=gen) < 0.3).float().cuda()
        dl = path.loss_dense_depth(out, depth_img, fg)
        loss = dl + out['bev_feat'].square().mean()
        loss.backward()
        results.append((float(dl), float(loss), out['volume_feat'].detach().clone(),
                        {n: p.grad.detach().clone() for n, p in path.named_parameters() if p.grad is not None}))
    (dl0, l0, v0, g0), (dl1, l1, v1, g1) = results
    assert dl0 == dl1                       # same logits, same arithmetic: bit for bit
    assert abs(l0 - l1) <= 1e-6 * abs(l0)
    # (torch's / MIOpen's convolutions behind the sampling are not run-to-run bit-stable)
    assert torch.allclose(v0, v1, rtol=1e-4, atol=1e-5 * float(v0.abs().max()))
    assert g0.keys() == g1.keys() and len(g0) > 50
    for n in g0:
        # float atomics and MIOpen's backward algorithms are not run-to-run bit-stable: a few 1e-3 of a
        # tensor's scale at single elements far upstream; the norm of the difference is the stable measure
        err, ref = float((g1[n] - g0[n]).norm()), float(g0[n].norm())
        assert err <= 5e-3 * ref + 1e-12, (n, err, ref)
