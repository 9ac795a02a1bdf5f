"""DepthHead.loss (reference: mmdet3d/models/dense_heads/depth_head.py:75-188) through the HIP
kernels dfm_depth_loss_fwd/bwd, against fixtures produced by the REFERENCE class
(tests/golden/make_golden_r02.py: loss value and torch-autograd gradients for every loss type).
Tolerance: the kernel's expf/logf vs torch's CPU kernels -> rtol 2e-5 on the loss, rtol 1e-4 /
atol 1e-6 on the gradients."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from tests import util

sys.path.insert(0, util.GOLDEN)
from make_golden_r02 import LOSS_TYPES  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def mods():
    return importlib.import_module('depth-from-motion_amd.modules')


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(util.GOLDEN, 'depth_loss.npz'))


def _head(mods, gold, cfg):
    m = mods.DepthHead(depth_cfg=dict(mode='UD', num_bins=len(gold['depth_samples']), min_depth=2,
                                      max_depth=59.6),
                       with_convs=False, depth_loss=cfg, downsample_factor=4, num_views=1)
    m.depth_samples = torch.from_numpy(gold['depth_samples'])
    return m


@pytest.mark.parametrize('loss_type', LOSS_TYPES)
def test_loss_and_gradients_vs_reference_class(mods, gold, loss_type):
    cfg = dict(type=loss_type, loss_weight=0.7)
    if 'balanced' in loss_type:
        cfg.update(fg_weight=5, bg_weight=1)
    if 'focal' in loss_type:
        cfg.update(alpha=0.75, gamma=2)
    m = _head(mods, gold, cfg)
    dev = torch.device('cuda:0')
    vol = torch.from_numpy(gold['volumes']).to(dev).requires_grad_(True)
    pred = torch.from_numpy(gold['preds']).to(dev).requires_grad_(True)
    img = torch.from_numpy(gold['depth_img']).to(dev)
    fg = torch.from_numpy(gold['fgmask']).to(dev)
    loss = m.loss(pred, vol, img, depth_fgmask_img=fg)
    loss.backward()
    key = loss_type.replace('.', 'p')
    np.testing.assert_allclose(float(loss), float(gold[f'{key}_loss']), rtol=2e-5)
    gv = vol.grad.cpu().numpy() if vol.grad is not None else np.zeros_like(gold['volumes'])
    gp = pred.grad.cpu().numpy() if pred.grad is not None else np.zeros_like(gold['preds'])
    np.testing.assert_allclose(gv, gold[f'{key}_gvol'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gp, gold[f'{key}_gpred'], rtol=1e-4, atol=1e-6)


def test_focal_generic_gamma(mods, gold):
    m = _head(mods, gold, dict(type='focal', loss_weight=1.0, alpha=1, gamma=3))
    dev = torch.device('cuda:0')
    vol = torch.from_numpy(gold['volumes']).to(dev).requires_grad_(True)
    loss = m.loss(torch.from_numpy(gold['preds']).to(dev), vol, torch.from_numpy(gold['depth_img']).to(dev))
    loss.backward()
    np.testing.assert_allclose(float(loss), float(gold['focal_g3_loss']), rtol=2e-5)
    np.testing.assert_allclose(vol.grad.cpu().numpy(), gold['focal_g3_gvol'], rtol=1e-4, atol=1e-6)


def test_no_valid_ground_truth_returns_a_zero_connected_to_the_graph(mods, gold, capsys):
    m = _head(mods, gold, dict(type='ce', loss_weight=1.0))
    dev = torch.device('cuda:0')
    vol = torch.from_numpy(gold['volumes']).to(dev)
    pred = torch.from_numpy(gold['preds']).to(dev).requires_grad_(True)
    loss = m.loss(pred, vol, torch.zeros_like(pred))  # depth_head.py:104-106
    assert float(loss) == 0.0 and loss.requires_grad
    assert 'no gt warning' in capsys.readouterr().out


def test_bf16_volume_and_config_k_size(mods):
    """config K size (B=1, D=288, 320x1280) in bf16 against the fp32 result of the same kernel on
    the bf16-rounded logits (loss identical; gradients rounded once to bf16)."""
    dev = torch.device('cuda:0')
    D, H, W = 288, 320, 1280
    m = mods.DepthHead(depth_cfg=dict(mode='UD', num_bins=D, min_depth=2, max_depth=59.6),
                       with_convs=False,
                       depth_loss=dict(type='balanced_focal', loss_weight=1.0, fg_weight=5, bg_weight=1,
                                       alpha=1, gamma=2), downsample_factor=4, num_views=1)
    m.depth_samples = torch.tensor([(k + 0.5) * (57.6 / D) + 2 for k in range(D)])
    gen = torch.Generator().manual_seed(3)
    vol16 = (torch.randn(1, D, H, W, generator=gen) * 2).to(dev).bfloat16()
    img = (torch.rand(1, H, W, generator=gen) * 60).to(dev)
    img[torch.rand(1, H, W, generator=gen).to(dev) < 0.9] = 0  # sparse LiDAR supervision
    fg = (torch.rand(1, H, W, generator=gen) < 0.3).to(dev).float()
    pred = torch.zeros(1, H, W, device=dev)
    a = vol16.clone().requires_grad_(True)
    b = vol16.float().requires_grad_(True)
    la, lb = m.loss(pred, a, img, fg), m.loss(pred, b, img, fg)
    la.backward()
    lb.backward()
    np.testing.assert_allclose(float(la), float(lb), rtol=1e-6)
    assert torch.equal(a.grad, b.grad.bfloat16())
    assert float((a.grad != 0).float().mean()) < 0.11  # only valid pixels carry gradient
