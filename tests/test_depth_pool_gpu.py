"""GPU: AvgPool3d((k, 1, 1)) on the channels-last stack (csrc/depth_pool.hip; FrustumToVoxel,
mmdet3d/models/necks/feature_transformation.py:167) against torch: nn.AvgPool3d on the contiguous fp32 tensor
(forward and autograd) -- fp32 exact up to the order of four additions (rtol 1e-6), bf16 the same sum rounded once
(bit-equal to torch's float() -> mean -> to(bf16) chain, the path rounds 3-5 took)."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu
CL = torch.channels_last_3d


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('shape,k', [((1, 32, 20, 19, 23), 4), ((2, 16, 6, 5, 7), 2), ((1, 8, 9, 4, 6), 3),
                                     ((1, 32, 20, 304, 288), 4)])
def test_depth_pool_forward_backward_vs_torch(dtype, shape, k):
    mods = importlib.import_module('depth-from-motion_amd.modules')
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).to(dev).to(dtype).contiguous(memory_format=CL).requires_grad_(True)
    pool = torch.nn.AvgPool3d((k, 1, 1), stride=(k, 1, 1))
    y = mods._depth_pool4(pool, x)
    assert y.shape == (shape[0], shape[1], shape[2] // k, shape[3], shape[4]) and y.dtype == dtype
    assert y.is_contiguous(memory_format=CL)
    gy = torch.randn(y.shape, generator=g).to(dev).to(dtype).contiguous(memory_format=CL)
    y.backward(gy)
    xr = x.detach().float().contiguous().requires_grad_(True)
    yr = pool(xr)
    yr.backward(gy.float().contiguous())
    if dtype == torch.float32:
        assert torch.allclose(y, yr, rtol=1e-6, atol=1e-7) and torch.allclose(x.grad, xr.grad, rtol=1e-6, atol=1e-7)
    else:
        # the chain the stack ran before: fp32 sum of the k bf16 values, / k, one rounding
        B, C, D, H, W = shape
        v = x.detach().permute(0, 2, 3, 4, 1).reshape(B, D // k, k, H, W, C)
        chain = v.float().mean(dim=2).to(dtype).permute(0, 4, 1, 2, 3)
        assert torch.equal(y, chain)
        assert torch.equal(x.grad, (xr.grad).to(dtype))
        assert torch.allclose(y.float(), yr, rtol=2 ** -8, atol=1e-3)
