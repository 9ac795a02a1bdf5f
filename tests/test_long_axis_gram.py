"""The weight gradient of 1x1 convolutions as a batched product over slices of the pixel axis
(conv3d.long_axis_gram, _PixelLinearFn; modules._GateLogitsFn): same gradients as torch's own autograd of
F.linear / matmul -- the reference's Conv2d(kernel_size=1) layers (spp_unet_neck.py lastconv, dfm_backbone.py:108-141)."""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F


@pytest.mark.parametrize('P', [12, 4096 * 3, 2048 * 7 + 5, 409600])
def test_long_axis_gram_equals_the_plain_product(P):
    cv = importlib.import_module('depth-from-motion_amd.conv3d')
    g = torch.Generator().manual_seed(P)
    a, b = torch.randn(P, 5, generator=g), torch.randn(P, 3, generator=g)
    ref = a.double().t() @ b.double()
    np.testing.assert_allclose(cv.long_axis_gram(a, b).double().numpy(), ref.numpy(), rtol=1e-4, atol=1e-3)
    # transposed (pixel-contiguous) operands: views, not copies
    at, bt = a.t().contiguous().t(), b.t().contiguous().t()
    np.testing.assert_allclose(cv.long_axis_gram(at, bt).double().numpy(), ref.numpy(), rtol=1e-4, atol=1e-3)
    ab, bb = a.bfloat16(), b.bfloat16()
    refb = ab.double().t() @ bb.double()
    got = cv.long_axis_gram(ab, bb)
    assert got.dtype == torch.float32
    scale = float(refb.abs().max()) + 1.0
    assert float((got.double() - refb).abs().max()) <= 2.0 ** -7 * scale


@pytest.mark.parametrize('bias', [False, True])
def test_pixel_linear_gradients_equal_autograd(bias):
    cv = importlib.import_module('depth-from-motion_amd.conv3d')
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 96, 8, generator=g).permute(0, 3, 1, 2)     # NCHW view of an NHWC tensor
    w = torch.randn(6, 8, generator=g)
    b = torch.randn(6, generator=g) if bias else None
    gy = torch.randn(2, 64, 96, 6, generator=g)
    outs = []
    for fn in (lambda xx, ww, bb: cv._PixelLinearFn.apply(xx, ww, bb), F.linear):
        xx = x.permute(0, 2, 3, 1).detach().requires_grad_(True)
        ww = w.detach().requires_grad_(True)
        bb = b.detach().requires_grad_(True) if bias else None
        y = fn(xx, ww, bb)
        y.backward(gy)
        outs.append((y.detach(), xx.grad, ww.grad, bb.grad if bias else None))
    for got, ref in zip(*outs):
        if ref is not None:
            np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=1e-3)


def test_gate_logits_gradients_equal_autograd():
    mods = importlib.import_module('depth-from-motion_amd.modules')
    g = torch.Generator().manual_seed(5)
    w2 = torch.randn(6, 12, generator=g)
    both = torch.randn(2, 12, 4096 * 2, generator=g)
    gy = torch.randn(2, 6, 4096 * 2, generator=g)
    res = []
    for fn in (mods._GateLogitsFn.apply, torch.matmul):
        ww, bb = w2.detach().requires_grad_(True), both.detach().requires_grad_(True)
        y = fn(ww, bb)
        y.backward(gy)
        res.append((y.detach(), ww.grad, bb.grad))
    for got, ref in zip(*res):
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=2e-3)


def test_fp32_product_capability_is_asked_per_backend():
    """torch.mm / torch.bmm (..., out_dtype=float32) exists on the GPU backend of this torch and not on its CPU backend:
    the answer is cached per (operator, device type) -- a CPU product after a GPU one must not inherit the GPU's."""
    cv = importlib.import_module('depth-from-motion_amd.conv3d')
    cv._OUT_DTYPE_OK[('mm', 'cuda')] = True
    cv._OUT_DTYPE_OK[('bmm', 'cuda')] = True
    a, b = torch.randn(6, 5).bfloat16(), torch.randn(5, 4).bfloat16()
    r = cv._mm_f32(a, b)
    assert r.dtype == torch.float32
    np.testing.assert_allclose(r.numpy(), (a.float() @ b.float()).numpy(), rtol=2e-2, atol=2e-2)
    r3 = cv._bmm_f32(a[None], b[None])
    assert r3.dtype == torch.float32 and r3.shape == (1, 6, 4)
