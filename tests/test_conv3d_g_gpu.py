"""General MFMA Conv3d / ConvTranspose3d (csrc/conv3d_g.hip) against torch's fp32 convolution of
the same bf16-rounded inputs and weights (reference call sites: hourglass,
mmdet3d/models/utils/conv_modules.py:73-149; ResModule / OutdoorImVoxelNeck,
mmdet3d/models/necks/imvoxel_neck.py:26-55,85-117; DfMNeck, mmdet3d/models/necks/dfm_neck.py:29-95).

Tolerance: the kernel accumulates 27 * C_in bf16 products in fp32 (order differs from torch's) and
rounds ONCE to bf16: |got - ref| <= 2^-8 |ref| (half a bf16 ulp, doubled for the rounding of the
reference itself) + 2e-3 absolute (summation-order noise of O(1) sums)."""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
RTOL, ATOL = 2.0 ** -7, 2e-3


@pytest.fixture(scope='module')
def cv():
    assert torch.cuda.is_available()
    return importlib.import_module('depth-from-motion_amd.conv3d')


def _x(N, C, size, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(N, C, *size, generator=g).bfloat16()


def _w(d0, d1, fan_in, seed):
    g = torch.Generator().manual_seed(seed + 1000)
    return (torch.randn(d0, d1, 3, 3, 3, generator=g) * (2.0 / (27 * fan_in)) ** 0.5).bfloat16().float()


def _cl(t, dev):
    return t.to(dev).contiguous(memory_format=torch.channels_last_3d)


# (cin, cout, in_size, stride, padding): ragged tiles, every stride / padding combination of the path
CONV_CASES = [
    (32, 32, (5, 7, 9), 1, 1),
    (64, 64, (6, 10, 12), 1, 1),
    (32, 64, (8, 12, 16), 2, 1),          # hourglass conv1
    (64, 64, (8, 12, 20), 2, 1),          # hourglass conv3
    (64, 64, (7, 9, 11), 2, 1),           # odd extents under stride 2
    (64, 128, (5, 9, 12), (1, 1, 2), 1),  # neck down-sampling along z
    (128, 128, (4, 6, 6), 1, 1),
    (128, 256, (3, 5, 6), (1, 1, 2), 1),
    (256, 256, (3, 5, 3), 1, (1, 1, 0)),  # neck last conv: Nz 3 -> 1
    (64, 32, (4, 20, 40), 1, 1),
    (32, 32, (3, 4, 5), 1, (2, 2, 2)),    # padding 2 (backward-data of padding 0)
    (64, 64, (9, 17, 33), 1, 1),
]


@pytest.mark.parametrize('cin,cout,size,stride,padding', CONV_CASES)
@pytest.mark.parametrize('N', [1, 2])
def test_conv_matches_torch(cv, cin, cout, size, stride, padding, N):
    dev = torch.device('cuda:0')
    x, w = _x(N, cin, size, seed=cin + cout + size[2]), _w(cout, cin, cin, seed=cin * 3 + cout)
    ref = F.conv3d(x.float(), w, stride=stride, padding=padding)
    pk = cv.pack_conv3d_g_weights(w.to(dev), cin, cout)
    out = cv.conv3d_g(_cl(x, dev), pk, cout, stride, padding)
    assert out.shape == ref.shape and out.dtype == torch.bfloat16
    assert out.is_contiguous(memory_format=torch.channels_last_3d)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), rtol=RTOL, atol=ATOL)


TRANSPOSED_CASES = [(64, 64, (4, 5, 6)), (64, 32, (5, 6, 9)), (32, 32, (3, 3, 17)), (128, 64, (2, 9, 8))]


@pytest.mark.parametrize('cin,cout,size', TRANSPOSED_CASES)
def test_transposed_conv_matches_torch(cv, cin, cout, size):
    """hourglass conv5 / conv6: ConvTranspose3d(k 3, s 2, p 1, output_padding 1)"""
    dev = torch.device('cuda:0')
    x, w = _x(2, cin, size, seed=cin + size[0]), _w(cin, cout, cin, seed=cout)
    ref = F.conv_transpose3d(x.float(), w, stride=2, padding=1, output_padding=1)
    pk = cv.pack_conv3d_g_weights(w.to(dev), cin, cout, swap=True)
    out = cv.conv3d_g(_cl(x, dev), pk, cout, transposed=True)
    assert out.shape == ref.shape
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), rtol=RTOL, atol=ATOL)


def test_mixed_transposed_axis(cv):
    """backward-data of a stride-(1,1,2) convolution: z is a transposed axis, x / y are mirrored
    correlations -- checked against torch's conv_transpose3d with the same per-axis strides"""
    dev = torch.device('cuda:0')
    cin, cout, size = 64, 32, (4, 6, 5)   # gy channels 64 -> gx channels 32
    gy, w = _x(1, cin, size, seed=3), _w(cin, cout, cin, seed=4)   # w: conv weight (Cout=64, Cin=32)
    ref = F.conv_transpose3d(gy.float(), w, stride=(1, 1, 2), padding=1, output_padding=(0, 0, 1))
    pk = cv.pack_conv3d_g_weights(w.to(dev), cin, cout, swap=True, flip=4 | 2)
    out = cv.conv3d_g(_cl(gy, dev), pk, cout, stride=1, padding=1, transposed=(False, False, True))
    assert out.shape == ref.shape
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), rtol=RTOL, atol=ATOL)


def test_epilogue_scale_shift_residual_relu(cv):
    """folded BatchNorm3d (eval) + identity + ReLU of ResModule (imvoxel_neck.py:102-117)"""
    dev = torch.device('cuda:0')
    cin = cout = 64
    size = (5, 9, 12)
    x, w = _x(2, cin, size, seed=11), _w(cout, cin, cin, seed=12)
    g = torch.Generator().manual_seed(13)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    res = _x(2, cout, size, seed=14)
    conv = F.conv3d(x.float(), w, padding=1)
    sc, sh = scale.view(1, -1, 1, 1, 1), shift.view(1, -1, 1, 1, 1)
    pk = cv.pack_conv3d_g_weights(w.to(dev), cin, cout)
    xg, rg = _cl(x, dev), _cl(res, dev)
    for use_res in (False, True):
        for relu in (False, True):
            ref = conv * sc + sh
            if use_res:
                ref = ref + res.float()
            if relu:
                ref = torch.relu(ref)
            out = cv.conv3d_g(xg, pk, cout, relu=relu, scale=scale.to(dev), shift=shift.to(dev),
                              residual=rg if use_res else None)
            np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), rtol=RTOL, atol=3e-3)
    out = cv.conv3d_g(xg, pk, cout, relu=True, residual=rg)   # residual without scale
    np.testing.assert_allclose(out.float().cpu().numpy(), torch.relu(conv + res.float()).numpy(), rtol=RTOL,
                               atol=3e-3)


MODULE_CASES = [
    ('conv', 32, 64, (8, 12, 16), 2, 1),
    ('conv', 64, 64, (6, 10, 12), 1, 1),
    ('conv', 64, 128, (4, 8, 12), (1, 1, 2), 1),
    ('conv', 128, 64, (4, 6, 3), 1, (1, 1, 0)),
    ('convT', 64, 64, (4, 5, 6), 2, 1),
    ('convT', 64, 32, (3, 6, 8), 2, 1),
]


@pytest.mark.parametrize('kind,cin,cout,size,stride,padding', MODULE_CASES)
def test_modules_forward_backward_vs_torch_autograd(cv, kind, cin, cout, size, stride, padding):
    """MfmaConv3dG / MfmaConvTranspose3d: same state_dict as the torch module, forward and both
    gradients against torch fp32 autograd on the bf16-rounded operands"""
    dev = torch.device('cuda:0')
    torch.manual_seed(cin + cout)
    if kind == 'conv':
        m = cv.MfmaConv3dG(cin, cout, 3, stride=stride, padding=padding, bias=False).to(dev)
    else:
        m = cv.MfmaConvTranspose3d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=False).to(dev)
    assert list(m.state_dict()) == ['weight']
    with torch.no_grad():
        m.weight.copy_(m.weight.bfloat16().float())
    x = _x(2, cin, size, seed=7)
    xb = _cl(x, dev).requires_grad_(True)
    assert m.eligible(xb) and not m.eligible(x.float().to(dev))
    y = m(xb)
    xr = x.float().requires_grad_(True)
    wr = m.weight.detach().cpu().clone().requires_grad_(True)
    if kind == 'conv':
        yr = F.conv3d(xr, wr, stride=stride, padding=padding)
    else:
        yr = F.conv_transpose3d(xr, wr, stride=2, padding=1, output_padding=1)
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), yr.detach().numpy(), rtol=RTOL, atol=ATOL)
    gy = _x(2, cout, tuple(yr.shape[2:]), seed=8)
    y.backward(_cl(gy, dev))
    yr.backward(gy.float())
    np.testing.assert_allclose(xb.grad.float().cpu().numpy(), xr.grad.numpy(), rtol=RTOL, atol=5e-3)
    # weight gradient: torch's convolution backward in bf16 (MIOpen) -> loose
    np.testing.assert_allclose(m.weight.grad.cpu().numpy(), wr.grad.numpy(), rtol=5e-2,
                               atol=0.02 * float(wr.grad.abs().max()))
    # the fp32 NCDHW path is torch's convolution, unchanged
    y32 = m(x.float().to(dev))
    np.testing.assert_allclose(y32.detach().cpu().numpy(), yr.detach().numpy(), rtol=1e-3, atol=1e-3)


def test_hourglass_shapes_full_size_against_fp64_subset(cv):
    """conv2 of the hourglass at config-K size (64 -> 64 at 36 x 40 x 160) and conv1 (32 -> 64,
    stride 2 from 72 x 80 x 320): a strided voxel subset against an fp64 evaluation on the GPU"""
    dev = torch.device('cuda:0')
    for cin, cout, size, stride in ((64, 64, (36, 40, 160), 1), (32, 64, (72, 80, 320), 2)):
        x, w = _x(1, cin, size, seed=21), _w(cout, cin, cin, seed=22)
        xg = _cl(x, dev)
        out = cv.conv3d_g(xg, cv.pack_conv3d_g_weights(w.to(dev), cin, cout), cout, stride, 1)
        D, H, W = out.shape[2:]
        xp = F.pad(xg.double(), (1, 1, 1, 1, 1, 1))
        wd = w.to(dev).double()
        sd, sh, sw = range(0, D, 5), range(0, H, 7), range(0, W, 9)
        ref = torch.zeros(1, cout, len(sd), len(sh), len(sw), dtype=torch.float64, device=dev)
        for kd in range(3):
            for kh in range(3):
                for kw in range(3):
                    patch = xp[:, :, kd::stride, kh::stride, kw::stride][:, :, :D, :H, :W][:, :, ::5, ::7, ::9]
                    ref += torch.einsum('ncdhw,oc->nodhw', patch, wd[:, :, kd, kh, kw])
        np.testing.assert_allclose(out[:, :, ::5, ::7, ::9].double().cpu().numpy(), ref.cpu().numpy(), rtol=RTOL,
                                   atol=ATOL)


def test_channel_slice_of_a_wider_tensor_is_read_in_place(cv):
    """DfMNeck's mono stack convolves x[:, :C] of the (N, C*F, ...) volume (dfm_neck.py:108): the
    kernel reads the slice with the wide tensor's pixel stride, no copy"""
    dev = torch.device('cuda:0')
    size = (4, 6, 12)
    x, w = _x(2, 128, size, seed=31), _w(64, 64, 64, seed=32)
    xg = _cl(x, dev)
    pk = cv.pack_conv3d_g_weights(w.to(dev), 64, 64)
    for lo in (0, 64):
        xs = xg[:, lo:lo + 64]
        assert not xs.is_contiguous(memory_format=torch.channels_last_3d)
        out = cv.conv3d_g(xs, pk, 64)
        ref = F.conv3d(x[:, lo:lo + 64].float(), w, padding=1)
        np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), rtol=RTOL, atol=ATOL)
    m = cv.MfmaConv3dG(64, 64, 3, padding=1, bias=False).to(dev)
    assert m.eligible(xg[:, :64]) and not m.eligible(xg[:, 4:68])


# (a = g channels, b = x channels, x size, stride, padding): both contraction axes (the kernel picks the
# longer of H / W), ragged tiles, stride-2 rows (de-interleaved phases), padding 0 / 2
WGRAD_CASES = [
    (32, 32, (3, 5, 70), 1, 1),
    (32, 32, (4, 37, 9), 1, 1),            # contracts along H
    (64, 32, (5, 6, 40), 1, 1),
    (32, 64, (8, 12, 40), 2, 1),           # hourglass conv1: stride-2 rows
    (64, 64, (7, 9, 33), 2, 1),            # odd extents under stride 2
    (64, 128, (4, 20, 12), (1, 1, 2), 1),  # neck down-sampling along z (w stays short: contracts along H)
    (128, 64, (3, 24, 3), 1, (1, 1, 0)),   # neck last conv: Nz 3 -> 1
    (32, 32, (3, 4, 18), 1, (2, 2, 2)),
]


@pytest.mark.parametrize('a,b,size,stride,padding', WGRAD_CASES)
def test_weight_gradient_matches_torch_autograd(cv, a, b, size, stride, padding):
    """dfm_conv3d_wgrad (MFMA) against torch's fp32 autograd of the same bf16-rounded tensors"""
    dev = torch.device('cuda:0')
    x = _x(2, b, size, seed=a + b + size[1])
    w = torch.zeros(a, b, 3, 3, 3, requires_grad=True)
    y = F.conv3d(x.float(), w, stride=stride, padding=padding)
    gy = _x(2, a, tuple(y.shape[2:]), seed=77)
    y.backward(gy.float())
    got = cv.conv3d_weight_grad(_cl(x, dev), _cl(gy, dev), stride, padding)
    assert got.dtype == torch.float32 and got.shape == w.shape
    ref = w.grad.numpy()
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-4, atol=2e-4 * float(np.abs(ref).max()))


def test_weight_gradient_of_the_transposed_convolution_and_of_channel_slices(cv):
    dev = torch.device('cuda:0')
    x = _x(2, 64, (3, 5, 9), seed=1)
    w = torch.zeros(64, 32, 3, 3, 3, requires_grad=True)
    y = F.conv_transpose3d(x.float(), w, stride=2, padding=1, output_padding=1)
    gy = _x(2, 32, tuple(y.shape[2:]), seed=2)
    y.backward(gy.float())
    got = cv.conv3d_weight_grad(_cl(gy, dev), _cl(x, dev), 2, 1)     # x_in = grad_output, g_out = input
    ref = w.grad.numpy()
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-4, atol=2e-4 * float(np.abs(ref).max()))
    # a channel slice of a wider NDHWC tensor as the input (DfMNeck mono stack, dres0 halves)
    xw = _x(1, 128, (4, 6, 20), seed=3)
    w2 = torch.zeros(32, 64, 3, 3, 3, requires_grad=True)
    y2 = F.conv3d(xw[:, 64:].float(), w2, padding=1)
    g2 = _x(1, 32, tuple(y2.shape[2:]), seed=4)
    y2.backward(g2.float())
    got2 = cv.conv3d_weight_grad(_cl(xw, dev)[:, 64:], _cl(g2, dev), 1, 1)
    ref2 = w2.grad.numpy()
    np.testing.assert_allclose(got2.cpu().numpy(), ref2, rtol=2e-4, atol=2e-4 * float(np.abs(ref2).max()))
    # channel counts the MFMA kernel does not take: the implicit-im2col GEMM
    x3, g3 = _x(1, 8, (3, 4, 5), seed=5), _x(1, 4, (3, 4, 5), seed=6)
    w3 = torch.zeros(4, 8, 3, 3, 3, requires_grad=True)
    F.conv3d(x3.float(), w3, padding=1).backward(g3.float())
    got3 = cv.conv3d_weight_grad(_cl(x3, dev), _cl(g3, dev), 1, 1)
    np.testing.assert_allclose(got3.cpu().numpy(), w3.grad.numpy(), rtol=2e-2, atol=2e-2 * float(w3.grad.abs().max()))


# ---- 2-D 3x3 convolutions: the same kernel with extent 1 along depth (kernel (1, 3, 3)) on an NHWC
# tensor seen as a depth-1 volume -- SPPUNetNeck / BEVHourglass ---------------
def _w2(d0, d1, fan_in, seed):
    g = torch.Generator().manual_seed(seed + 2000)
    return (torch.randn(d0, d1, 3, 3, generator=g) * (2.0 / (9 * fan_in)) ** 0.5).bfloat16().float()


CONV2D_CASES = [(32, 32, (20, 37), 1), (64, 128, (22, 30), 2), (128, 128, (9, 11), 1), (512, 128, (12, 20), 1),
                (160, 64, (19, 18), 1), (128, 128, (15, 17), 2), (64, 32, (40, 72), 1)]


@pytest.mark.parametrize('cin,cout,size,stride', CONV2D_CASES)
def test_conv2d_through_the_depth1_kernel_matches_torch(cv, cin, cout, size, stride):
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(2, cin, *size, generator=g).bfloat16()
    w = _w2(cout, cin, cin, seed=cin + 7 * cout)
    ref = F.conv2d(x.float(), w, stride=stride, padding=1)
    pk = cv.pack_conv2d_g_weights(w.to(dev), cin, cout)
    xc = x.to(dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        assert cv.conv2d_g_eligible(xc, cin, cout)
        out = cv.conv2d_g(xc, pk, cout, stride=stride)
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), rtol=RTOL, atol=ATOL)
    # epilogue: folded BatchNorm (scale / shift) + residual + ReLU
    scale, shift = torch.rand(cout) + 0.5, torch.randn(cout)
    res = torch.randn(ref.shape, generator=g).bfloat16()
    ref2 = torch.relu(ref * scale[None, :, None, None] + shift[None, :, None, None] + res.float())
    with torch.no_grad():
        out2 = cv.conv2d_g(xc, pk, cout, stride=stride, relu=True, scale=scale.to(dev), shift=shift.to(dev),
                           residual=res.to(dev).contiguous(memory_format=torch.channels_last))
    np.testing.assert_allclose(out2.float().cpu().numpy(), ref2.numpy(), rtol=RTOL, atol=2 * ATOL)


@pytest.mark.parametrize('cin,cout,size', [(128, 128, (9, 10)), (128, 64, (12, 19)), (64, 32, (5, 33))])
def test_transposed_conv2d_through_the_depth1_kernel_matches_torch(cv, cin, cout, size):
    """hourglass2d conv5 / conv6: ConvTranspose2d(k 3, s 2, p 1, output_padding 1)"""
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(cin + size[1])
    x = torch.randn(2, cin, *size, generator=g).bfloat16()
    w = _w2(cin, cout, cin, seed=cout)
    ref = F.conv_transpose2d(x.float(), w, stride=2, padding=1, output_padding=1)
    pk = cv.pack_conv2d_g_weights(w.to(dev), cin, cout, swap=True)
    with torch.no_grad():
        out = cv.conv2d_g(x.to(dev).contiguous(memory_format=torch.channels_last), pk, cout, transposed=True)
    assert out.shape == ref.shape
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), rtol=RTOL, atol=ATOL)


def test_conv2d_modules_take_the_mfma_path_only_where_it_applies(cv, monkeypatch):
    dev = torch.device('cuda:0')
    calls = {'n': 0}
    real = cv.conv3d_g

    def counted(*a, **k):
        calls['n'] += 1
        return real(*a, **k)
    monkeypatch.setattr(cv, 'conv3d_g', counted)
    m = cv.MfmaConv2d(64, 96 + 32, 3, padding=1, bias=True).to(dev).bfloat16()
    x = torch.randn(1, 64, 17, 23, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        ref = F.conv2d(x.float(), m.weight.float(), m.bias.float(), padding=1)
        y = m(x)
    assert calls['n'] == 1
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.cpu().numpy(), rtol=RTOL, atol=ATOL)
    y2 = m(x)                                # autograd on: the training path, the same kernel
    assert calls['n'] == 2 and y2.requires_grad
    calls['n'] = 1
    with torch.no_grad():
        m(x.contiguous())                     # NCHW input: torch
        cv.MfmaConv2d(48, 32, 3, padding=1).to(dev).bfloat16()(x[:, :48].contiguous(memory_format=torch.channels_last))
    assert calls['n'] == 1                    # 48 input channels: not whole 32-channel chunks
    # a narrow input (the 3-channel image of upconv_module's last skip) is zero-padded to one chunk
    n3 = cv.MfmaConv2d(3, 32, 3, padding=1, bias=False).to(dev).bfloat16()
    x3 = torch.randn(2, 3, 21, 34, device=dev).bfloat16()
    with torch.no_grad():
        y3 = n3(x3)
        ref3 = F.conv2d(x3.float(), n3.weight.float(), padding=1)
    assert calls['n'] == 2
    np.testing.assert_allclose(y3.float().cpu().numpy(), ref3.cpu().numpy(), rtol=RTOL, atol=ATOL)
    calls['n'] = 1
    t = cv.MfmaConvTranspose2d(64, 32, 3, stride=2, padding=1, output_padding=1, bias=False).to(dev).bfloat16()
    with torch.no_grad():
        yt = t(x)
    assert calls['n'] == 2
    with torch.no_grad():
        reft = F.conv_transpose2d(x.float(), t.weight.float(), stride=2, padding=1, output_padding=1)
    np.testing.assert_allclose(yt.float().cpu().numpy(), reft.cpu().numpy(), rtol=RTOL, atol=ATOL)


MODULE2D_CASES = [
    ('conv', 64, 64, (20, 36), 1, False, 'nchw'),
    ('conv', 128, 64, (16, 24), 2, False, 'nchw'),
    ('conv', 32, 96, (9, 13), 1, True, 'nhwc'),
    ('conv', 3, 32, (12, 20), 1, False, 'nchw'),      # the image skip of upconv_module: padded to 32 channels
    ('convT', 64, 32, (7, 11), 2, False, 'nchw'),
    ('convT', 128, 128, (6, 10), 2, False, 'nhwc'),
]


@pytest.mark.parametrize('kind,cin,cout,size,stride,bias,layout', MODULE2D_CASES)
def test_conv2d_modules_train_through_the_mfma_kernels(cv, monkeypatch, kind, cin, cout, size, stride, bias, layout):
    """MfmaConv2d / MfmaConvTranspose2d with autograd recording (spp_unet_neck.py:93-119,
    bev_hourglass.py:36-137): forward, backward-data and backward-weight in the MFMA kernels, in the
    caller's layout, against torch fp32 autograd on the bf16-rounded operands"""
    dev = torch.device('cuda:0')
    launched = {'g': 0, 'w': 0}
    real_g, real_w = cv.conv3d_g, cv.conv3d_weight_grad
    monkeypatch.setattr(cv, 'conv3d_g', lambda *a, **k: (launched.__setitem__('g', launched['g'] + 1), real_g(*a, **k))[1])
    monkeypatch.setattr(cv, 'conv3d_weight_grad',
                        lambda *a, **k: (launched.__setitem__('w', launched['w'] + 1), real_w(*a, **k))[1])
    torch.manual_seed(cin + 3 * cout)
    if kind == 'conv':
        m = cv.MfmaConv2d(cin, cout, 3, stride=stride, padding=1, bias=bias).to(dev).bfloat16()
    else:
        m = cv.MfmaConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=False).to(dev).bfloat16()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, cin, *size, generator=g).bfloat16()
    xb = x.to(dev)
    if layout == 'nhwc':
        xb = xb.contiguous(memory_format=torch.channels_last)
    xb.requires_grad_(cin >= 32)
    assert m.train_why_not(xb) is None
    y = m(xb)
    assert launched['g'] == 1
    assert y.is_contiguous() if layout == 'nchw' else y.is_contiguous(memory_format=torch.channels_last)
    xr = x.float().requires_grad_(True)
    wr = m.weight.detach().float().cpu().requires_grad_(True)
    br = m.bias.detach().float().cpu().requires_grad_(True) if bias else None
    if kind == 'conv':
        yr = F.conv2d(xr, wr, br, stride=stride, padding=1)
    else:
        yr = F.conv_transpose2d(xr, wr, stride=2, padding=1, output_padding=1)
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), yr.detach().numpy(), rtol=RTOL, atol=ATOL)
    gy = torch.randn(yr.shape, generator=g).bfloat16()
    y.backward(gy.to(dev))
    yr.backward(gy.float())
    assert launched['w'] == 1 and launched['g'] == (2 if cin >= 32 else 1)
    if cin >= 32:
        np.testing.assert_allclose(xb.grad.float().cpu().numpy(), xr.grad.numpy(), rtol=RTOL, atol=5e-3)
        assert xb.grad.stride() == xb.stride()
    scale = float(wr.grad.abs().max())
    np.testing.assert_allclose(m.weight.grad.float().cpu().numpy(), wr.grad.numpy(), rtol=2.0 ** -7, atol=2.0 ** -8 * scale)
    if bias:
        np.testing.assert_allclose(m.bias.grad.float().cpu().numpy(), br.grad.numpy(), rtol=2.0 ** -7,
                                   atol=2.0 ** -8 * float(br.grad.abs().max()))


@pytest.mark.parametrize('hin,win,size,scale,ac', [(3, 7, (20, 33), None, True), (5, 8, None, 2.0, False),
                                                   (1, 4, (80, 320), None, True), (40, 160, None, 2.0, False)])
@pytest.mark.parametrize('nhwc', [False, True], ids=['nchw', 'nhwc'])
def test_bilinear_resize_backward_as_matrix_products(hin, win, size, scale, ac, nhwc):
    """modules.bilinear_resize: ATen's forward, backward gX = A_h^T gY A_w (spp_unet_neck.py:60-70, 83-91); an NHWC
    gradient (the necks train channels-last) takes the products on its memory as it lies and comes back NHWC"""
    import importlib
    mods = importlib.import_module('depth-from-motion_amd.modules')
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(hin + win)
    x = torch.randn(2, 32, hin, win, generator=g).bfloat16()
    xb = x.to(dev)
    if nhwc:
        xb = xb.contiguous(memory_format=torch.channels_last)
    xb = xb.requires_grad_(True)
    y = mods.bilinear_resize(xb, size=size, scale_factor=scale, align_corners=ac)
    xr = x.float().requires_grad_(True)
    kw = dict(scale_factor=scale) if scale is not None else dict(size=size)
    yr = F.interpolate(xr, mode='bilinear', align_corners=ac, **kw)
    assert torch.equal(y.detach().cpu(), F.interpolate(x.to(dev), mode='bilinear', align_corners=ac, **kw).cpu())
    gy = torch.randn(yr.shape, generator=g).bfloat16()
    gyd = gy.to(dev).contiguous(memory_format=torch.channels_last) if nhwc else gy.to(dev)
    y.backward(gyd)
    yr.backward(gy.float())
    ref = xr.grad.numpy()
    if nhwc and hin > 1 and win > 1:
        assert xb.grad.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_allclose(xb.grad.float().cpu().numpy(), ref, rtol=2.0 ** -7, atol=2.0 ** -8 * float(np.abs(ref).max()))


SPLIT_CASES = [
    ('conv', 'MfmaConv3dG', 32, 64, (8, 12, 16), 2, 1),
    ('conv', 'MfmaConv3dG', 64, 64, (6, 10, 12), 1, 1),
    ('conv', 'MfmaConv3d', 64, 32, (5, 9, 12), 1, 1),
    ('conv', 'MfmaConv3dG', 64, 128, (4, 8, 12), (1, 1, 2), 1),
    ('convT', 'MfmaConvTranspose3d', 64, 32, (3, 6, 8), 2, 1),
]


@pytest.mark.parametrize('mode,nl,tol', [('split', 6, 2e-6), ('split2', 3, 1e-4)])
@pytest.mark.parametrize('kind,cls,cin,cout,size,stride,padding', SPLIT_CASES)
def test_fp32_modules_run_the_mfma_kernels_in_split_precision(cv, monkeypatch, kind, cls, cin, cout, size, stride,
                                                              padding, mode, nl, tol):
    """the reference's default precision (dfm_backbone.py:175-201, conv_modules.py:73-149): an fp32 NCDHW
    call of an Mfma* module = six (three bf16 pieces per operand: fp32-equivalent) or three ('split2')
    bf16 launches accumulated in fp32, forward and both gradients; compared with torch fp32 autograd on the
    CPU (whose own summation order differs at the 1e-6 level)"""
    dev = torch.device('cuda:0')
    launched = {'f': 0, 'w': 0}
    real_f, real_w = cv.conv3d_g_f32, cv.conv3d_weight_grad
    monkeypatch.setattr(cv, 'conv3d_g_f32', lambda *a, **k: (launched.__setitem__('f', launched['f'] + 1), real_f(*a, **k))[1])
    monkeypatch.setattr(cv, 'conv3d_weight_grad',
                        lambda *a, **k: (launched.__setitem__('w', launched['w'] + 1), real_w(*a, **k))[1])
    prev = cv.set_fp32_mode(mode)
    try:
        torch.manual_seed(cin + cout)
        if kind == 'conv':
            m = getattr(cv, cls)(cin, cout, 3, stride=stride, padding=padding, bias=False).to(dev)
        else:
            m = cv.MfmaConvTranspose3d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=False).to(dev)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(2, cin, *size, generator=g)
        xb = x.to(dev).requires_grad_(True)
        y = m(xb)
        assert launched['f'] == nl and y.dtype == torch.float32 and y.is_contiguous()
        xr = x.clone().requires_grad_(True)
        wr = m.weight.detach().cpu().clone().requires_grad_(True)
        if kind == 'conv':
            yr = F.conv3d(xr, wr, stride=stride, padding=padding)
        else:
            yr = F.conv_transpose3d(xr, wr, stride=2, padding=1, output_padding=1)
        sc = float(yr.detach().abs().max())
        np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), rtol=tol, atol=tol * sc)
        gy = torch.randn(yr.shape, generator=g)
        y.backward(gy.to(dev))
        yr.backward(gy)
        assert launched['f'] == 2 * nl and launched['w'] == nl
        np.testing.assert_allclose(xb.grad.cpu().numpy(), xr.grad.numpy(), rtol=tol, atol=tol * float(xr.grad.abs().max()))
        np.testing.assert_allclose(m.weight.grad.cpu().numpy(), wr.grad.numpy(), rtol=tol,
                                   atol=4 * tol * float(wr.grad.abs().max()))
        cv.set_fp32_mode('torch')
        with torch.no_grad():
            y_t = m(x.to(dev))
        assert launched['f'] == 2 * nl
        np.testing.assert_allclose(y_t.cpu().numpy(), yr.detach().numpy(), rtol=1e-3, atol=1e-3 * sc)
    finally:
        cv.set_fp32_mode(prev)


def test_fp32_prediction_and_2d_convolutions_in_split_precision(cv, monkeypatch):
    """Conv3d(32 -> 1) (dfm_backbone.py:120-127), Conv2d 3x3 stride 1 | 2 (+bias, 3-channel input) and
    ConvTranspose2d x2 (conv_modules.py:152-214) of an fp32 model: forward + gradients vs torch fp32"""
    dev = torch.device('cuda:0')
    launched = {'f': 0}
    real_f = cv.conv3d_g_f32
    monkeypatch.setattr(cv, 'conv3d_g_f32', lambda *a, **k: (launched.__setitem__('f', launched['f'] + 1), real_f(*a, **k))[1])
    g = torch.Generator().manual_seed(3)
    cases = [
        (cv.MfmaConv3dTo1(32, 1, 3, 1, 1, bias=False), (2, 32, 5, 8, 12),
         lambda x, w, b: F.conv3d(x, w, padding=1)),
        (cv.MfmaConv2d(64, 96, 3, stride=1, padding=1, bias=True), (2, 64, 14, 22),
         lambda x, w, b: F.conv2d(x, w, b, padding=1)),
        (cv.MfmaConv2d(128, 64, 3, stride=2, padding=1, bias=False), (1, 128, 16, 24),
         lambda x, w, b: F.conv2d(x, w, stride=2, padding=1)),
        (cv.MfmaConv2d(3, 32, 3, stride=1, padding=1, bias=False), (2, 3, 12, 20),
         lambda x, w, b: F.conv2d(x, w, padding=1)),
        (cv.MfmaConvTranspose2d(64, 32, 3, stride=2, padding=1, output_padding=1, bias=False), (2, 64, 7, 11),
         lambda x, w, b: F.conv_transpose2d(x, w, stride=2, padding=1, output_padding=1)),
    ]
    for m, shape, ref_fn in cases:
        torch.manual_seed(shape[1])
        m = m.to(dev)
        x = torch.randn(*shape, generator=g)
        xb = x.to(dev).requires_grad_(True)
        before = launched['f']
        y = m(xb)
        assert launched['f'] == before + 6, type(m).__name__
        xr = x.clone().requires_grad_(True)
        wr = m.weight.detach().cpu().clone().requires_grad_(True)
        br = m.bias.detach().cpu().clone().requires_grad_(True) if m.bias is not None else None
        yr = ref_fn(xr, wr, br)
        assert y.shape == yr.shape and y.is_contiguous()
        tol = 2e-6
        np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), rtol=tol,
                                   atol=tol * float(yr.detach().abs().max()))
        gy = torch.randn(yr.shape, generator=g)
        y.backward(gy.to(dev))
        yr.backward(gy)
        np.testing.assert_allclose(xb.grad.cpu().numpy(), xr.grad.numpy(), rtol=tol, atol=tol * float(xr.grad.abs().max()))
        np.testing.assert_allclose(m.weight.grad.cpu().numpy(), wr.grad.numpy(), rtol=tol,
                                   atol=4 * tol * float(wr.grad.abs().max()))
        if br is not None:
            np.testing.assert_allclose(m.bias.grad.cpu().numpy(), br.grad.numpy(), rtol=1e-5,
                                       atol=1e-5 * float(br.grad.abs().max()))


def test_weight_gradient_in_the_parameters_type_is_the_fp32_gradient_rounded_once(cv):
    """dfm_conv3d_wgrad_to(DFM_BF16): the reduction kernel stores bf16 -- bit for bit the fp32 result converted"""
    dev = torch.device('cuda:0')
    for a, b, size, stride in ((32, 32, (5, 9, 12), 1), (64, 32, (6, 10, 12), 2), (64, 64, (4, 6, 9), 1)):
        x = _cl(_x(2, b, size, seed=a + b), dev)
        out_size = tuple((s + 2 - 3) // stride + 1 for s in size)
        g = _cl(_x(2, a, out_size, seed=a * 3), dev)
        w32 = cv.conv3d_weight_grad(x, g, stride, 1)
        w16 = cv.conv3d_weight_grad(x, g, stride, 1, out_dtype=torch.bfloat16)
        assert w32.dtype == torch.float32 and w16.dtype == torch.bfloat16 and w16.shape == w32.shape
        assert float(w32.abs().max()) > 0 and torch.equal(w16, w32.to(torch.bfloat16))


def test_channel_split_adds_the_slice_gradient_into_the_whole_tensors(cv):
    """conv3d.channel_split: (x, x[:, :32]) for the cost volume's two consumers -- the gradient of x equals the one
    autograd builds from two separate uses (bf16 additions of the same two values: bit-identical), and stays
    channels-last"""
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(1, 64, 6, 10, 16, generator=gen).bfloat16().to(dev).contiguous(memory_format=torch.channels_last_3d)
    w_all = torch.randn(1, 64, 6, 10, 16, generator=gen).bfloat16().to(dev).contiguous(memory_format=torch.channels_last_3d)
    w_cur = torch.randn(1, 32, 6, 10, 16, generator=gen).bfloat16().to(dev).contiguous(memory_format=torch.channels_last_3d)
    xa = x.clone().requires_grad_(True)
    a_all, a_cur = cv.channel_split(xa, 0, 32)
    assert a_all.shape == x.shape and a_cur.shape == (1, 32, 6, 10, 16) and a_cur.data_ptr() == xa.data_ptr()
    ((a_all * w_all).sum() + (a_cur * w_cur).sum()).backward()
    xb = x.clone().requires_grad_(True)
    ((xb * w_all).sum() + (xb[:, :32] * w_cur).sum()).backward()
    assert torch.equal(xa.grad, xb.grad)
    assert xa.grad.is_contiguous(memory_format=torch.channels_last_3d)
    # only one of the two outputs used
    xc = x.clone().requires_grad_(True)
    (cv.channel_split(xc, 0, 32)[1] * w_cur).sum().backward()
    ref = torch.zeros_like(x)
    ref[:, :32] = w_cur
    assert torch.equal(xc.grad, ref)
    xd = x.clone().requires_grad_(True)
    (cv.channel_split(xd, 0, 32)[0] * w_all).sum().backward()
    assert torch.equal(xd.grad, w_all)
    with torch.no_grad():
        p, q = cv.channel_split(x, 0, 32)
        assert p.data_ptr() == x.data_ptr() and q.shape[1] == 32
