"""The gate of DfMBackbone.forward as one launch (csrc/cost_gate.hip) against the reference's sequence
(dfm_backbone.py:136-141: cat -> Conv2d(2D -> D, 1x1, bias=False) -> sigmoid -> blend) in fp64."""
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(s, m, w):
    s, m, w = s.double(), m.double(), w.double()
    both = torch.cat((s, m), dim=1).flatten(1, 2)                     # (B, 2D, H, W)
    gate = torch.einsum('dk,bkhw->bdhw', w, both).unsqueeze(1).sigmoid()
    return gate * s + (1 - gate) * m


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('wdtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(1, 72, 20, 80), (2, 7, 5, 13), (1, 96, 3, 70), (3, 1, 1, 1), (1, 18, 9, 64)])
def test_fused_gate_equals_the_reference_sequence(dtype, wdtype, shape):
    mods = importlib.import_module('depth-from-motion_amd.modules')
    B, D, H, W = shape
    g = torch.Generator().manual_seed(B * 1000 + D)
    s = (torch.randn(B, 1, D, H, W, generator=g) * 2).to(dtype).cuda()
    m = (torch.randn(B, 1, D, H, W, generator=g) * 2 + 0.5).to(dtype).cuda()
    bb = mods.DfMBackbone(in_channels=32, depth_cfg=dict(num_bins=D, downsample_factor=1)).cuda()
    with torch.no_grad():
        bb.aggregate_cost.weight.copy_(torch.randn(D, 2 * D, 1, 1, generator=g) * 0.3)
    bb.aggregate_cost.to(wdtype)
    with torch.no_grad():
        got = bb._gate_fused(s, m)
    assert got is not None and got.shape == s.shape and got.dtype == dtype
    ref = _reference(s, m, bb.aggregate_cost.weight.detach().flatten(1))
    if dtype == torch.float32:
        np.testing.assert_allclose(got.double().cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4)
    else:
        # one rounding of the fp32 result to bf16
        err = (got.double() - ref).abs()
        assert float((err - ref.abs() * 2.0 ** -8 - 1e-6).max()) <= 0.0


@pytest.mark.parametrize('shape', [(1, 72, 20, 80), (2, 7, 5, 13), (1, 96, 3, 70), (3, 1, 1, 1), (1, 33, 9, 67)])
def test_matrix_core_gate_is_taken_for_bf16_and_matches_the_valu_kernel(shape):
    """bf16 costs + a bf16 weight: the product runs on the matrix cores (round 6); every product is exact and the sums
    are fp32 in both kernels, so the two results differ by summation order only -- within one bf16 rounding of the
    fp64 reference each, and within one bf16 ulp of each other"""
    mods = importlib.import_module('depth-from-motion_amd.modules')
    B, D, H, W = shape
    g = torch.Generator().manual_seed(B * 77 + D)
    s = (torch.randn(B, 1, D, H, W, generator=g) * 2).bfloat16().cuda()
    m = (torch.randn(B, 1, D, H, W, generator=g) * 2 + 0.5).bfloat16().cuda()
    bb = mods.DfMBackbone(in_channels=32, depth_cfg=dict(num_bins=D, downsample_factor=1)).cuda()
    with torch.no_grad():
        bb.aggregate_cost.weight.copy_(torch.randn(D, 2 * D, 1, 1, generator=g) * 0.3)
    bb.aggregate_cost.to(torch.bfloat16)
    lib = importlib.import_module('depth-from-motion_amd._capi').lib()
    with torch.no_grad():
        assert bb.mfma_gate
        a = bb._gate_fused(s, m)
        assert bb.__dict__['_gate_pack'][1].numel() == lib.dfm_cost_gate_mfma_weight_bytes(D)
        bb.mfma_gate = False
        b = bb._gate_fused(s, m)
        assert bb.__dict__['_gate_pack'][1].numel() == lib.dfm_cost_gate_weight_bytes(D)
    ref = _reference(s, m, bb.aggregate_cost.weight.detach().flatten(1))
    for got in (a, b):
        err = (got.double() - ref).abs()
        assert float((err - ref.abs() * 2.0 ** -8 - 1e-6).max()) <= 0.0
    assert float(((a.double() - b.double()).abs() - b.double().abs() * 2.0 ** -7 - 1e-6).max()) <= 0.0


def test_gate_falls_back_with_autograd_or_too_many_planes():
    mods = importlib.import_module('depth-from-motion_amd.modules')
    bb = mods.DfMBackbone(in_channels=32, depth_cfg=dict(num_bins=8, downsample_factor=1)).cuda()
    s = torch.randn(1, 1, 8, 4, 4).cuda()
    assert bb._gate_fused(s, s) is None            # autograd is recording
    with torch.no_grad():
        assert bb._gate_fused(s, s) is not None
        assert bb._gate_fused(s, s.double()) is None
        bb.fused_gate = False
        assert bb._gate_fused(s, s) is None
    big = mods.DfMBackbone(in_channels=32, depth_cfg=dict(num_bins=100, downsample_factor=1)).cuda()
    with torch.no_grad():
        assert big._gate_fused(torch.randn(1, 1, 100, 2, 2).cuda(), torch.randn(1, 1, 100, 2, 2).cuda()) is None
