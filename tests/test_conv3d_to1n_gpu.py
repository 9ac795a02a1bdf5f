"""GPU: the prediction head's tail as one pass (csrc/conv3d_to1n.hip) -- GroupNorm(+ReLU) applied on load inside
the Conv3d(32 -> 1) that consumes it (mmdet3d/models/backbones/dfm_backbone.py:120-127) -- against

  * the unfused sequence of the same library (statistics from the 32 -> 32 convolution's epilogue, the
    normalisation pass, the former 32 -> 1 kernel): a one-hot weight makes the convolution a copy, so the values
    normalised ON LOAD must equal the normalisation pass's stored values BIT FOR BIT;
  * torch: F.conv3d in fp32 on the normalised bf16 tensor (the plain PyTorch reference of the op): the fused
    result is that sum rounded once to bf16 -- bar 2^-8 of the largest output (one bf16 ulp of it).
"""
import importlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
CL = torch.channels_last_3d


@pytest.fixture(scope='module')
def mods():
    assert torch.cuda.is_available()
    return importlib.import_module('depth-from-motion_amd.modules')


def _head(mods, seed):
    torch.manual_seed(seed)
    dev = torch.device('cuda:0')
    cm = mods.ConvModule(32, 32, 3, stride=1, padding=1, conv_cfg=dict(type='Conv3d'),
                         norm_cfg=dict(type='GN', num_groups=32, requires_grad=True))
    last = mods.MfmaConv3dTo1(32, 1, 3, 1, 1, bias=False)
    seq = torch.nn.Sequential(cm, last).to(dev).to(torch.bfloat16).to(memory_format=CL)
    with torch.no_grad():
        cm.gn.weight.copy_(torch.rand(32) + 0.5)
        cm.gn.bias.copy_(torch.randn(32) * 0.3)
    return seq


def _backbone_like(mods):
    bb = mods.DfMBackbone.__new__(mods.DfMBackbone)   # only _pred_head is exercised
    return bb


@pytest.mark.parametrize('shape', [(1, 8, 16, 64), (2, 5, 13, 45), (1, 3, 8, 32), (1, 14, 9, 33), (1, 2, 3, 5)])
@pytest.mark.parametrize('chunk', [0, 1, 3])
def test_fused_head_vs_unfused_and_torch(mods, shape, chunk):
    conv3d = importlib.import_module('depth-from-motion_amd.conv3d')
    gn = importlib.import_module('depth-from-motion_amd.group_norm')
    N, D, H, W = shape
    seq = _head(mods, 10 + D)
    cm, last = seq[0], seq[1]
    x = torch.randn(N, 32, D, H, W, device='cuda:0').bfloat16().contiguous(memory_format=CL)
    with torch.no_grad():
        y, partials = cm.conv.forward_with_stats(x)
        gamma, beta = gn._f32_params(cm.gn.weight, cm.gn.bias)
        fused = conv3d.conv3d_to1_norm(y, partials, gamma, beta, cm.gn.eps, last.weight, relu=True, depth_chunk=chunk)
        normed = cm.gn(y, relu=True, partials=partials)        # the normalisation pass the fused kernel replaces
        unfused = last(normed)
        ref = F.conv3d(normed.float().contiguous(), last.weight.float(), padding=1)
    assert fused.shape == unfused.shape == (N, 1, D, H, W) and fused.dtype == torch.bfloat16
    bar = 2.0 ** -8 * float(ref.abs().max()) + 1e-6
    assert float((fused.float() - ref).abs().max()) <= bar
    assert float((unfused.float() - ref).abs().max()) <= bar


@pytest.mark.parametrize('relu', [True, False])
@pytest.mark.parametrize('tap,channel', [(13, 0), (13, 31), (0, 7), (26, 20), (4, 16)])
def test_values_normalised_on_load_are_the_normalisation_pass_values(mods, relu, tap, channel):
    """a one-hot weight (one tap, one channel) makes the convolution a shifted copy of one channel of the
    normalised tensor: every other product is an exact zero, so the output IS bf16(relu(x * a + b)) -- the bits
    gn_apply_cl_kernel stores (zero where the tap reaches outside the volume)."""
    conv3d = importlib.import_module('depth-from-motion_amd.conv3d')
    gn = importlib.import_module('depth-from-motion_amd.group_norm')
    N, D, H, W = 2, 6, 11, 37
    seq = _head(mods, 3)
    cm = seq[0]
    x = (torch.randn(N, 32, D, H, W, device='cuda:0') * 2 + 0.3).bfloat16().contiguous(memory_format=CL)
    w = torch.zeros(1, 32, 3, 3, 3, device='cuda:0')
    kd, kh, kw = tap // 9, (tap // 3) % 3, tap % 3
    w[0, channel, kd, kh, kw] = 1.0
    with torch.no_grad():
        y, partials = cm.conv.forward_with_stats(x)
        gamma, beta = gn._f32_params(cm.gn.weight, cm.gn.bias)
        fused = conv3d.conv3d_to1_norm(y, partials, gamma, beta, cm.gn.eps, w, relu=relu)
        normed = cm.gn(y, relu=relu, partials=partials)
    want = F.pad(normed[:, channel].float(), (1, 1, 1, 1, 1, 1))[:, kd:kd + D, kh:kh + H, kw:kw + W]
    assert torch.equal(fused[:, 0].float(), want)


def test_backbone_prediction_head_takes_the_fused_kernel_at_inference(mods):
    """DfMBackbone._pred_head: fused under no_grad on the bf16 NDHWC stack, the module sequence otherwise; same
    result within one bf16 rounding of the largest value"""
    seq = _head(mods, 5)
    bb = _backbone_like(mods)
    x = torch.randn(1, 32, 6, 16, 40, device='cuda:0').bfloat16().contiguous(memory_format=CL)
    lib = importlib.import_module('depth-from-motion_amd._capi').lib()
    with torch.no_grad():
        a = bb._pred_head(seq, x)
        bb.fused_pred = False
        b = bb._pred_head(seq, x)
    assert a.shape == b.shape and float((a.float() - b.float()).abs().max()) <= 2.0 ** -7 * float(b.float().abs().max())
    bb.fused_pred = True
    xg = x.clone().requires_grad_(True)
    out = bb._pred_head(seq, xg)        # autograd recording: the module sequence (its backward exists)
    out.float().sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad.float()).all()
    assert lib.dfm_version() == 3
