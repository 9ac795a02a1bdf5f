"""Fused GroupNorm(+ReLU): GPU forward/backward against torch's CPU GroupNorm
(the op the reference modules run).  Tolerance rtol 1e-4 / atol 1e-5 (different
reduction order; both fp32); bf16 storage: one bf16 ulp on top."""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F


@pytest.fixture(scope='module')
def pkg():
    return importlib.import_module('depth-from-motion_amd')


def test_cpu_tensors_raise_unless_the_test_opt_in_is_set(pkg):
    import importlib
    gn = importlib.import_module('depth-from-motion_amd.group_norm')
    m = pkg.HipGroupNorm(4, 8)
    x = torch.randn(2, 8, 3, 5, 7)
    with pytest.raises(RuntimeError, match='no CPU path'):
        m(x)
    prev = gn.allow_cpu_reference(True)  # test-only: module wiring on a CPU-only box
    try:
        assert torch.equal(m(x), F.group_norm(x, 4, m.weight, m.bias, m.eps))
        assert torch.equal(m(x, relu=True), F.relu(F.group_norm(x, 4, m.weight, m.bias, m.eps)))
    finally:
        gn.allow_cpu_reference(prev)
    assert list(m.state_dict()) == ['weight', 'bias']


@pytest.mark.gpu
def test_unsupported_gpu_inputs_raise(pkg):
    m = pkg.HipGroupNorm(4, 8).cuda()
    with pytest.raises(RuntimeError, match='unsupported GPU input'):
        m(torch.randn(2, 8, 3, 5, 7, device='cuda').half())
    with pytest.raises(RuntimeError, match='unsupported GPU input'):
        pkg.HipGroupNorm(4, 8, affine=False).cuda()(torch.randn(2, 8, 3, 5, 7, device='cuda'))


CASES = [(2, 32, (8, 12, 20), 32), (1, 32, (9, 7, 13), 32), (3, 16, (6, 10), 4), (1, 64, (18, 20, 40), 32),
         (2, 8, (5, 3, 3), 1)]


@pytest.mark.gpu
@pytest.mark.parametrize('relu', [False, True])
@pytest.mark.parametrize('n,c,sp,groups', CASES)
def test_forward_backward_vs_torch_cpu(pkg, n, c, sp, groups, relu):
    gen = torch.Generator().manual_seed(n * 100 + c)
    x = torch.randn(n, c, *sp, generator=gen) * 2 + 0.7
    w = 1 + 0.2 * torch.randn(c, generator=gen)
    b = 0.3 * torch.randn(c, generator=gen)
    gy = torch.randn(n, c, *sp, generator=gen)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = F.group_norm(xr, groups, wr, br, 1e-5)
    ref = F.relu(ref) if relu else ref
    (ref * gy).sum().backward()
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
    out = pkg.group_norm(xg, groups, wg, bg, 1e-5, relu)
    (out * gy.cuda()).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(wg.grad.cpu().numpy(), wr.grad.numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(bg.grad.cpu().numpy(), br.grad.numpy(), rtol=1e-3, atol=1e-3)


@pytest.mark.gpu
def test_config_k_volume_and_large_mean(pkg):
    """(1,32,72,80,320), per-channel groups, a large common offset (variance must not cancel)"""
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(1, 32, 72, 80, 320, generator=gen) + 100.0
    m = pkg.HipGroupNorm(32, 32).cuda()
    y = m(x.cuda(), relu=False)
    flat = y.float().reshape(32, -1)
    assert float(flat.mean(1).abs().max()) < 1e-3
    assert float((flat.var(1, unbiased=False) - 1).abs().max()) < 1e-3
    ref = F.group_norm(x[:, :2], 2, None, None, 1e-5)
    np.testing.assert_allclose(y[:, :2].detach().cpu().numpy(), ref.numpy(), rtol=1e-3, atol=2e-4)


@pytest.mark.gpu
def test_bf16_storage(pkg):
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(2, 32, 6, 8, 16, generator=gen).bfloat16()
    m = pkg.HipGroupNorm(32, 32).cuda()
    y = m(x.cuda(), relu=True)
    assert y.dtype == torch.bfloat16
    ref = F.relu(F.group_norm(x.float(), 32, m.weight.cpu(), m.bias.cpu(), 1e-5))
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), ref.detach().numpy(), rtol=1e-2, atol=1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('relu', [False, True])
@pytest.mark.parametrize('n,c,sp,groups', [(2, 32, (8, 12, 20), 32), (1, 64, (9, 7, 13), 32),
                                           (3, 16, (6, 10), 4), (1, 32, (18, 20, 41), 8)])
def test_channels_last_forward_backward_vs_torch_cpu(pkg, n, c, sp, groups, relu, dtype):
    """channels_last(_3d) input takes the channels-last kernels and stays channels-last"""
    gen = torch.Generator().manual_seed(n * 10 + c)
    x = torch.randn(n, c, *sp, generator=gen) * 2 + 0.7
    w = 1 + 0.2 * torch.randn(c, generator=gen)
    b = 0.3 * torch.randn(c, generator=gen)
    gy = torch.randn(n, c, *sp, generator=gen)
    if dtype == torch.bfloat16:
        x, gy = x.bfloat16().float(), gy.bfloat16().float()
    fmt = torch.channels_last_3d if len(sp) == 3 else torch.channels_last
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = F.group_norm(xr, groups, wr, br, 1e-5)
    ref = F.relu(ref) if relu else ref
    (ref * gy).sum().backward()
    xg = x.cuda().to(dtype).contiguous(memory_format=fmt).requires_grad_(True)
    wg, bg = (t.cuda().requires_grad_(True) for t in (w, b))
    out = pkg.group_norm(xg, groups, wg, bg, 1e-5, relu)
    assert out.is_contiguous(memory_format=fmt) and out.shape == ref.shape
    (out.float() * gy.cuda()).sum().backward()
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-2, atol=2e-2)
    np.testing.assert_allclose(out.detach().float().cpu().numpy(), ref.detach().numpy(), **tol)
    gtol = dict(rtol=1e-3, atol=2e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=3e-2)
    np.testing.assert_allclose(xg.grad.float().cpu().numpy(), xr.grad.numpy(), **gtol)
    np.testing.assert_allclose(wg.grad.cpu().numpy(), wr.grad.numpy(), rtol=2e-2, atol=2e-2 * float(wr.grad.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize('relu', [False, True])
@pytest.mark.parametrize('layout', ['channels_last', 'contiguous'])
@pytest.mark.parametrize('n,c,sp,groups', [(2, 32, (6, 10, 12), 32), (1, 64, (5, 9, 16), 32)])
def test_fused_residual_forward_backward(pkg, n, c, sp, groups, relu, layout):
    """y = relu?(GroupNorm(x) + residual): the residual connections of the aggregation stacks
    (dfm_backbone.py:176,183; conv_modules.py:124-139) in the normalisation pass itself
    (channels-last; other layouts take separate torch ops)"""
    gen = torch.Generator().manual_seed(n * 7 + c)
    x = torch.randn(n, c, *sp, generator=gen) * 2 + 0.7
    res = torch.randn(n, c, *sp, generator=gen)
    w = 1 + 0.2 * torch.randn(c, generator=gen)
    b = 0.3 * torch.randn(c, generator=gen)
    gy = torch.randn(n, c, *sp, generator=gen)
    xr, rr, wr, br = (t.clone().requires_grad_(True) for t in (x, res, w, b))
    ref = F.group_norm(xr, groups, wr, br, 1e-5) + rr
    ref = F.relu(ref) if relu else ref
    (ref * gy).sum().backward()
    fmt = torch.channels_last_3d if layout == 'channels_last' else torch.contiguous_format
    xg, rg = (t.cuda().contiguous(memory_format=fmt).requires_grad_(True) for t in (x, res))
    wg, bg = (t.cuda().requires_grad_(True) for t in (w, b))
    out = pkg.group_norm(xg, groups, wg, bg, 1e-5, relu, residual=rg)
    (out * gy.cuda()).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(rg.grad.cpu().numpy(), rr.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(wg.grad.cpu().numpy(), wr.grad.numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(bg.grad.cpu().numpy(), br.grad.numpy(), rtol=1e-3, atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('relu', [False, True])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('n,c,sp,groups', [(2, 32, (6, 10, 12), 32), (1, 64, (5, 9, 16), 32), (2, 16, (7, 5), 4)])
def test_channels_last_backward_matches_the_contiguous_one(pkg, n, c, sp, groups, relu, dtype):
    """dfm_group_norm_bwd_channels_last (no layout round trip) against the NC(D)HW backward kernels on
    the same values, and both against torch autograd in fp32"""
    gen = torch.Generator().manual_seed(n * 11 + c)
    x = (torch.randn(n, c, *sp, generator=gen) * 2 + 0.7).to(dtype)
    gy = torch.randn(n, c, *sp, generator=gen).to(dtype)
    w = 1 + 0.2 * torch.randn(c, generator=gen)
    b = 0.3 * torch.randn(c, generator=gen)
    fmt = torch.channels_last_3d if len(sp) == 3 else torch.channels_last
    outs = []
    for f in (torch.contiguous_format, fmt):
        xg = x.cuda().contiguous(memory_format=f).requires_grad_(True)
        wg, bg = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
        out = pkg.group_norm(xg, groups, wg, bg, 1e-5, relu)
        out.backward(gy.cuda().contiguous(memory_format=f))
        outs.append((out.detach().float().cpu(), xg.grad.float().cpu(), wg.grad.cpu(), bg.grad.cpu()))
    tol = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=2e-5)
    for a, bb in zip(outs[0], outs[1]):
        torch.testing.assert_close(a, bb, rtol=max(tol['rtol'], 1e-3), atol=max(tol['atol'], 1e-3) * float(bb.abs().max() + 1))
    xr, wr, br = x.float().clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.group_norm(xr, groups, wr, br, 1e-5)
    ref = F.relu(ref) if relu else ref
    ref.backward(gy.float())
    if dtype == torch.float32:
        torch.testing.assert_close(outs[1][1], xr.grad, rtol=1e-3, atol=2e-5)
        torch.testing.assert_close(outs[1][2], wr.grad, rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(outs[1][3], br.grad, rtol=1e-3, atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('relu,with_res', [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_hip_batch_norm3d_training_matches_torch(pkg, relu, with_res, dtype):
    """HipBatchNorm3d in training mode on a channels-last batch (the BN3d blocks of the voxel necks,
    imvoxel_neck.py:28-55): output, running statistics and all gradients against nn.BatchNorm3d"""
    import importlib
    gn = importlib.import_module('depth-from-motion_amd.group_norm')
    torch.manual_seed(4)
    N, C, sp = 2, 64, (5, 9, 12)
    ref = torch.nn.BatchNorm3d(C).cuda().train()
    with torch.no_grad():
        ref.weight.copy_(1 + 0.2 * torch.randn(C))
        ref.bias.copy_(0.3 * torch.randn(C))
    m = gn.HipBatchNorm3d(C).cuda().train()
    m.load_state_dict(ref.state_dict())
    assert list(m.state_dict()) == list(ref.state_dict())
    x = (torch.randn(N, C, *sp, device='cuda') * 2 + 0.5).to(dtype)
    res = torch.randn(N, C, *sp, device='cuda').to(dtype)
    gy = torch.randn(N, C, *sp, device='cuda').to(dtype)
    cl = torch.channels_last_3d
    xr, rr = x.float().clone().requires_grad_(True), res.float().clone().requires_grad_(True)
    yr = ref(xr)
    if with_res:
        yr = yr + rr
    if relu:
        yr = torch.relu(yr)
    yr.backward(gy.float())
    if dtype == torch.bfloat16:
        m = m.to(torch.bfloat16)
    xg = x.contiguous(memory_format=cl).requires_grad_(True)
    rg = res.contiguous(memory_format=cl).requires_grad_(True)
    y = m(xg, relu=relu, residual=rg if with_res else None)
    assert y.shape == x.shape and y.is_contiguous(memory_format=cl)
    y.backward(gy.contiguous(memory_format=cl))
    tol = dict(rtol=3e-2, atol=3e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)

    def close(a, b):
        if dtype == torch.float32:
            return torch.testing.assert_close(a, b, **tol)
        # bf16: a pre-activation that rounds to the other side of zero flips its gradient entirely
        bad = (a - b).abs() > tol['atol'] + tol['rtol'] * b.abs()
        assert float(bad.float().mean()) < 0.01, float(bad.float().mean())
    close(y.float(), yr)
    close(xg.grad.float(), xr.grad)
    if with_res:
        close(rg.grad.float(), rr.grad)
    ptol = 6e-2 if dtype == torch.bfloat16 else 1e-3   # sums over 1080 elements per channel, ReLU flips in bf16
    torch.testing.assert_close(m.weight.grad.float(), ref.weight.grad, rtol=ptol, atol=ptol * float(ref.weight.grad.abs().max()))
    torch.testing.assert_close(m.bias.grad.float(), ref.bias.grad, rtol=ptol, atol=ptol * float(ref.bias.grad.abs().max()))
    torch.testing.assert_close(m.running_mean.float(), ref.running_mean, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(m.running_var.float(), ref.running_var, rtol=1e-2, atol=1e-2)
    assert int(m.num_batches_tracked) == 1
    # eval mode is torch's BatchNorm (the necks fold it into the convolution epilogue instead)
    m.eval(); ref.eval()
    torch.testing.assert_close(m(xg.detach()).float(), ref(x.float()), **tol)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('n,c,sp,groups,partials', [(2, 32, (8, 12, 20), 32, False), (1, 64, (9, 7, 13), 32, False),
                                                    (1, 32, (6, 16, 40), 32, True), (3, 16, (6, 10), 4, False)])
def test_relu_mask_recomputed_from_x_is_the_mask_of_the_kept_output(pkg, monkeypatch, n, c, sp, groups, partials, dtype):
    """dfm_group_norm_bwd_channels_last_xmask: y = relu(GroupNorm(x)) without a residual keeps no y for its backward --
    the mask comes from x through the forward's own expression.  Same sums in the same order otherwise: the three
    gradients are BIT-IDENTICAL to the backward that reads the mask from the kept output (values spread around the
    ReLU's kink: a third of them within a few ulp of zero after the affine map would show a one-ulp disagreement)."""
    gn = importlib.import_module('depth-from-motion_amd.group_norm')
    if partials and dtype != torch.bfloat16:
        pytest.skip('the convolution epilogue that produces statistics partials is bf16')
    gen = torch.Generator().manual_seed(c + len(sp))
    fmt = torch.channels_last_3d if len(sp) == 3 else torch.channels_last
    x = (torch.randn(n, c, *sp, generator=gen) * 1.5 + 0.4).cuda().to(dtype).contiguous(memory_format=fmt)
    w = (1 + 0.2 * torch.randn(c, generator=gen)).cuda()
    b = torch.zeros(c).cuda()          # beta = 0: the kink sits at the group mean, where the values are dense
    gy = torch.randn(n, c, *sp, generator=gen).cuda().to(dtype).contiguous(memory_format=fmt)
    res = {}
    for xmask in (True, False):
        monkeypatch.setattr(gn, '_XMASK', xmask)
        xg = x.clone().requires_grad_(True)
        wg, bg = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        if partials:
            mods = importlib.import_module('depth-from-motion_amd.modules')
            conv = mods.MfmaConv3d(32, 32, 3, 1, 1, bias=False).cuda().to(torch.bfloat16).to(memory_format=fmt)
            torch.manual_seed(1)
            with torch.no_grad():
                conv.weight.copy_(torch.randn_like(conv.weight) * 0.05)
                yc, pt = conv.forward_with_stats(x)
            xg = yc.clone().requires_grad_(True)
            out = pkg.group_norm(xg, groups, wg, bg, 1e-5, True, partials=pt)
        else:
            out = pkg.group_norm(xg, groups, wg, bg, 1e-5, True)
        saved = [t for t in out.grad_fn.saved_tensors] if hasattr(out.grad_fn, 'saved_tensors') else []
        kept_y = any(t.data_ptr() == out.data_ptr() for t in saved)
        assert kept_y == (not xmask), 'the recomputing backward keeps no output tensor'
        out.backward(gy)
        res[xmask] = (out.detach(), xg.grad, wg.grad, bg.grad)
    frac_on = float((res[True][0] > 0).float().mean())
    assert 0.2 < frac_on < 0.8
    for a_, b_ in zip(res[True], res[False]):
        assert torch.equal(a_, b_)


@pytest.mark.gpu
@pytest.mark.parametrize('relu,with_res', [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('cls', [torch.nn.BatchNorm2d, torch.nn.SyncBatchNorm])
def test_training_batch_norm_of_an_nhwc_map_through_the_fused_kernels(pkg, relu, with_res, dtype, cls):
    """group_norm.batch_norm_train_channels_last on a 4-D channels-last batch (the BatchNorm blocks of the 2-D necks'
    up-convolutions, mmdet3d/models/necks/spp_unet_neck.py:83-91 -- nn.SyncBatchNorm in a single-process job): output,
    running statistics and every gradient against torch's own module on the same values"""
    gn = importlib.import_module('depth-from-motion_amd.group_norm')
    torch.manual_seed(6)
    N, C, sp = 2, 64, (18, 40)
    ref = torch.nn.BatchNorm2d(C).cuda().train()
    with torch.no_grad():
        ref.weight.copy_(1 + 0.2 * torch.randn(C))
        ref.bias.copy_(0.3 * torch.randn(C))
    m = cls(C).cuda().train()
    m.load_state_dict(ref.state_dict())
    x = (torch.randn(N, C, *sp, device='cuda') * 2 + 0.5).to(dtype)
    res = torch.randn(N, C, *sp, device='cuda').to(dtype)
    gy = torch.randn(N, C, *sp, device='cuda').to(dtype)
    cl = torch.channels_last
    xr, rr = x.float().clone().requires_grad_(True), res.float().clone().requires_grad_(True)
    yr = ref(xr)
    if with_res:
        yr = yr + rr
    if relu:
        yr = torch.relu(yr)
    yr.backward(gy.float())
    xg = x.contiguous(memory_format=cl).requires_grad_(True)
    rg = res.contiguous(memory_format=cl).requires_grad_(True)
    y = gn.batch_norm_train_channels_last(m, xg, relu=relu, residual=rg if with_res else None)
    assert y is not None and y.shape == x.shape and y.is_contiguous(memory_format=cl) and y.dtype == dtype
    y.backward(gy.contiguous(memory_format=cl))
    tol = dict(rtol=3e-2, atol=3e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)

    def close(a, b):
        if dtype == torch.float32:
            return torch.testing.assert_close(a, b, **tol)
        bad = (a - b).abs() > tol['atol'] + tol['rtol'] * b.abs()   # (bf16: a ReLU flip at a rounded zero)
        assert float(bad.float().mean()) < 0.01, float(bad.float().mean())
    close(y.float(), yr)
    close(xg.grad.float(), xr.grad)
    if with_res:
        close(rg.grad.float(), rr.grad)
    ptol = 6e-2 if dtype == torch.bfloat16 else 1e-3
    torch.testing.assert_close(m.weight.grad.float(), ref.weight.grad, rtol=ptol, atol=ptol * float(ref.weight.grad.abs().max()))
    torch.testing.assert_close(m.bias.grad.float(), ref.bias.grad, rtol=ptol, atol=ptol * float(ref.bias.grad.abs().max()))
    torch.testing.assert_close(m.running_mean.float(), ref.running_mean, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(m.running_var.float(), ref.running_var, rtol=1e-2, atol=1e-2)
    assert int(m.num_batches_tracked) == 1
    # what it does not take: eval mode, an NCHW tensor
    assert gn.batch_norm_train_channels_last(m, x.contiguous(), relu=relu) is None
    m.eval()
    assert gn.batch_norm_train_channels_last(m, xg.detach(), relu=relu) is None
