"""Host-side logic of the MFMA convolution path that needs no GPU: the C-ABI planners / validators
(dfm_conv3d_g_plan, dfm_conv3d_wgrad_workspace_bytes), the layout helpers and the CPU behaviour of
the module classes (they fall through to torch, same state_dict keys)."""
import ctypes
import importlib

import pytest
import torch
import torch.nn.functional as F


@pytest.fixture(scope='module')
def cv():
    importlib.import_module('depth-from-motion_amd.build').build_hip()
    return importlib.import_module('depth-from-motion_amd.conv3d')


def test_planner_covers_every_layer_shape_of_the_path(cv):
    """hourglass (conv_modules.py:73-149) and voxel-neck (imvoxel_neck.py:26-55) layers: a tiling
    exists, fits the LDS, tiles the volume, and fills the chip where the layer is big enough"""
    cases = [(64, 64, (36, 40, 160), 1, 1, False), (32, 64, (72, 80, 320), 2, 1, False),
             (64, 64, (18, 20, 80), 1, 1, True), (64, 32, (36, 40, 160), 1, 1, True),
             (64, 64, (220, 300, 12), 1, 1, False), (64, 128, (220, 300, 12), (1, 1, 2), 1, False),
             (128, 128, (220, 300, 6), 1, 1, False), (256, 256, (220, 300, 3), 1, (1, 1, 0), False),
             (128, 256, (3, 5, 6), (1, 1, 2), 1, False)]
    for cin, cout, size, stride, padding, tr in cases:
        p = cv.conv3d_g_plan(1, cin, cout, size, stride, padding, tr)
        assert p['pfw'] in (1, 2, 3, 4) and p['cw'] == (2 if cout % 64 == 0 else 1)
        assert p['tile'][0] * p['tile'][1] * p['tile'][2] == 128 * p['pfw']
        assert p['block_px'] <= 1536 and p['lds'] <= 160 * 1024 and p['workgroups'] >= 1
        out = cv.conv3d_g_out_size(size, cv._triple(stride), cv._triple(padding), cv._triple(tr))
        space = size if tr else out
        tiles = 1
        for s, t in zip(space, p['tile']):
            tiles *= -(-s // t)
        assert p['workgroups'] % tiles == 0
    big = cv.conv3d_g_plan(1, 128, 128, (220, 300, 12))
    assert big['workgroups'] >= 512


def test_conv_descriptors_are_validated_without_a_gpu(cv):
    lib = importlib.import_module('depth-from-motion_amd._capi').lib()
    capi = importlib.import_module('depth-from-motion_amd._capi')
    d = cv._conv_desc(1, 48, 64, (4, 4, 4), (4, 4, 4), (1, 1, 1), (1, 1, 1), (False,) * 3, False)
    plan = (ctypes.c_int64 * 8)()
    assert lib.dfm_conv3d_g_plan(ctypes.byref(d), plan) == -2 and b'multiples of 32' in lib.dfm_last_error()
    d = cv._conv_desc(1, 32, 64, (4, 4, 4), (3, 4, 4), (1, 1, 1), (1, 1, 1), (False,) * 3, False)
    assert lib.dfm_conv3d_g_plan(ctypes.byref(d), plan) == -1 and b'out_size' in lib.dfm_last_error()
    d = cv._conv_desc(1, 32, 32, (4, 4, 4), (8, 8, 9), (1, 1, 1), (1, 1, 1), (True,) * 3, False)
    assert lib.dfm_conv3d_g_plan(ctypes.byref(d), plan) == -2 and b'transposed' in lib.dfm_last_error()
    assert lib.dfm_conv3d_g_fwd(ctypes.byref(d), None, None, None, None, None, None, None) != 0
    assert lib.dfm_conv3d_g_weight_bytes(64, 128) == 27 * 64 * 128 * 2 + 4096
    assert lib.dfm_conv3d_g_weight_bytes(48, 128) == 0
    w = capi.Conv3dWgradDesc()
    w.n, w.a, w.b = 1, 64, 32
    for i, (g, x) in enumerate(zip((36, 40, 160), (72, 80, 320))):
        w.g_size[i], w.x_size[i], w.stride[i], w.padding[i] = g, x, 2, 1
    for i, (gs, xs) in enumerate(zip((36 * 40 * 160 * 64, 40 * 160 * 64, 160 * 64, 64),
                                     (72 * 80 * 320 * 32, 80 * 320 * 32, 320 * 32, 32))):
        w.g_stride[i], w.x_stride[i] = gs, xs
    nbytes = lib.dfm_conv3d_wgrad_workspace_bytes(ctypes.byref(w))
    assert nbytes > 0 and nbytes % (27 * 1024 * 4) == 0      # whole 27 x 32 x 32 fp32 partials
    w.a = 40
    assert lib.dfm_conv3d_wgrad_workspace_bytes(ctypes.byref(w)) == 0
    w.a, w.x_stride[3] = 64, 36
    assert lib.dfm_conv3d_wgrad_workspace_bytes(ctypes.byref(w)) == 0   # strides: multiples of 8


def test_channel_stride_of_channels_last_views(cv):
    x = torch.zeros(2, 128, 3, 4, 5).contiguous(memory_format=torch.channels_last_3d)
    assert cv._ndhwc_channel_stride(x) == 128
    assert cv._ndhwc_channel_stride(x[:, :64]) == 128 and cv._ndhwc_channel_stride(x[:, 64:]) == 128
    assert cv._ndhwc_channel_stride(x[:, 4:68]) == 0          # not 16-byte aligned
    assert cv._ndhwc_channel_stride(x.contiguous()) == 0      # NCDHW
    assert cv._ndhwc_channel_stride(x[:, :, ::2]) == 0        # gaps between depth slices
    y = torch.zeros(1, 64, 3, 4, 1).contiguous(memory_format=torch.channels_last_3d)
    assert cv._ndhwc_channel_stride(y) == 64
    assert cv._ndhwc_strides(x[:, :64]) == (128 * 60, 128 * 20, 128 * 5, 128)


def test_weight_gradient_gemm_form_matches_autograd_on_the_cpu(cv):
    """the chunked implicit-im2col GEMM (the form channel counts outside the MFMA kernel take)"""
    torch.manual_seed(0)
    for cin, cout, size, stride, pad in [(4, 6, (5, 6, 7), 1, 1), (4, 6, (6, 8, 8), 2, 1),
                                         (3, 5, (4, 5, 6), (1, 1, 2), 1), (4, 4, (4, 5, 3), 1, (1, 1, 0))]:
        x = torch.randn(2, cin, *size).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
        w = torch.randn(cout, cin, 3, 3, 3, requires_grad=True)
        y = F.conv3d(x, w, stride=stride, padding=pad)
        gy = torch.randn_like(y)
        y.backward(gy)
        gw = cv.conv3d_weight_grad(x.detach(), gy.contiguous(memory_format=torch.channels_last_3d), stride, pad)
        torch.testing.assert_close(gw, w.grad, rtol=1e-4, atol=1e-4)
    x = torch.randn(2, 4, 3, 4, 5).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    w = torch.randn(4, 6, 3, 3, 3, requires_grad=True)
    y = F.conv_transpose3d(x, w, stride=2, padding=1, output_padding=1)
    gy = torch.randn_like(y)
    y.backward(gy)
    gw = cv.conv3d_weight_grad(gy.contiguous(memory_format=torch.channels_last_3d), x.detach(), 2, 1)
    torch.testing.assert_close(gw, w.grad, rtol=1e-4, atol=1e-4)


def test_module_classes_are_their_torch_parents_on_the_cpu(cv):
    """same parameters / state_dict keys; CPU tensors take torch's implementation"""
    gn = importlib.import_module('depth-from-motion_amd.group_norm')
    mods = importlib.import_module('depth-from-motion_amd.modules')
    torch.manual_seed(1)
    for m, ref in ((cv.MfmaConv3dG(32, 64, 3, stride=2, padding=1, bias=False), torch.nn.Conv3d),
                   (cv.MfmaConvTranspose3d(64, 32, 3, stride=2, padding=1, output_padding=1, bias=False),
                    torch.nn.ConvTranspose3d),
                   (cv.MfmaConv3dTo1(32, 1, 3, 1, 1, bias=False), torch.nn.Conv3d)):
        assert isinstance(m, ref) and list(m.state_dict()) == ['weight']
        x = torch.randn(1, m.in_channels, 4, 6, 6)
        assert not m.eligible(x)
        assert torch.equal(m(x), ref.forward(m, x))
    bn = gn.HipBatchNorm3d(8).train()
    ref = torch.nn.BatchNorm3d(8).train()
    ref.load_state_dict(bn.state_dict())
    x, r = torch.randn(2, 8, 3, 4, 5), torch.randn(2, 8, 3, 4, 5)
    torch.testing.assert_close(bn(x, relu=True, residual=r), torch.relu(ref(x) + r))
    torch.testing.assert_close(bn.running_var, ref.running_var)
    pool = mods._WindowMean2d(4, stride=4)
    xi = torch.randn(1, 3, 9, 13)
    assert torch.equal(pool(xi), F.avg_pool2d(xi, 4, stride=4))
    assert cv.channel_slice(torch.zeros(1, 8, 2, 2, 2), 0, 4).shape == (1, 4, 2, 2, 2)


def test_layout_predicates_and_graph_output_flattening_on_the_cpu():
    """host-side decisions of the NHWC edges, no GPU needed: which tensors count
    as channels-last for the in-place samplers, output sizes with a kernel-extent-1 axis, and the
    (nested) output rebuild of graphs.GraphedCallable"""
    import importlib
    import torch
    cv = importlib.import_module('depth-from-motion_amd.conv3d')
    ps = importlib.import_module('depth-from-motion_amd.point_sample')
    sw = importlib.import_module('depth-from-motion_amd.plane_sweep')
    graphs = importlib.import_module('depth-from-motion_amd.graphs')
    integ = importlib.import_module('depth-from-motion_amd.integration')
    # a 2-D convolution as a depth-1 volume: kernel (1, 3, 3), padding (0, 1, 1)
    assert cv.conv3d_g_out_size((1, 320, 1280), (1, 1, 1), (0, 1, 1), (False,) * 3, (True, False, False)) == (1, 320, 1280)
    assert cv.conv3d_g_out_size((1, 321, 1279), (1, 2, 2), (0, 1, 1), (False,) * 3, (True, False, False)) == (1, 161, 640)
    assert cv.conv3d_g_out_size((1, 40, 48), (1, 1, 1), (0, 1, 1), (False, True, True), (True, False, False)) == (1, 80, 96)
    # NHWC feature maps (plane sweep) / per-view NHWC features (multi-view lifting)
    x = torch.zeros(2, 16, 6, 10, dtype=torch.bfloat16)
    assert not sw._nhwc(x) and sw._nhwc(x.contiguous(memory_format=torch.channels_last))
    assert not sw._nhwc(torch.zeros(2, 12, 6, 10, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last))
    f = torch.zeros(2 * 3, 16, 6, 10, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert ps._views_channels_last(f.view(2, 3, 16, 6, 10)) and not ps._views_channels_last(torch.zeros(2, 3, 16, 6, 10))
    # GraphedCallable: nested outputs are flattened for the static buffers and rebuilt in shape
    a, b, c = torch.ones(1), torch.ones(2), torch.ones(3)
    flat, rebuild = graphs._flatten((a, [b, (c, None)]))
    assert [t.numel() for t in flat] == [1, 2, 3]
    out = rebuild(flat)
    assert isinstance(out, tuple) and isinstance(out[1], list) and out[1][1][1] is None and out[1][1][0] is c
    g = graphs.GraphedCallable(lambda ts: ts[0] + 1)
    assert torch.equal(g([a]), a + 1) and not g._graphs      # CPU tensors: plain call
    # the BEV view of a voxel volume is the reference's reshape
    vol = torch.arange(2 * 3 * 4 * 5 * 6, dtype=torch.float32).reshape(2, 3, 4, 5, 6)
    assert torch.equal(integ.bev_view(vol), vol.reshape(2, 12, 5, 6))


def test_depth_chunk_of_the_32_channel_convolution_fills_whole_rounds(cv):
    """dfm_conv3d_k3_c32_stats_splits = columns x chunks x 4 waves: the chunk of planes a workgroup walks is chosen
    by rounds x (planes + prologue) with one workgroup per CU -- config K (50 columns of 16 x 32, 72 planes) is ONE
    round of 250 workgroups (5 chunks of 15 planes), not 1.76 rounds of 8 planes (round 5)."""
    lib = cv._capi.lib()
    assert lib.dfm_conv3d_k3_c32_stats_splits(1, 72, 80, 320, 0) == 50 * 5 * 4
    for (n, d, h, w) in ((1, 72, 80, 320), (1, 36, 40, 160), (2, 9, 21, 70), (8, 72, 80, 320), (1, 4, 16, 32)):
        splits = lib.dfm_conv3d_k3_c32_stats_splits(n, d, h, w, 0)
        cols = -(-w // 32) * -(-h // 16)
        assert splits % (cols * 4) == 0
        chunks = splits // (cols * 4)
        dc = -(-d // chunks)
        # no other chunk size has a smaller rounds x (planes + 1.5)
        cost = lambda c: -(-(cols * n * -(-d // c)) // 256) * (c + 1.5)
        assert all(cost(dc) <= cost(c) + 1e-9 for c in range(min(d, 4), d + 1)), (n, d, h, w, dc)
    # an explicit chunk is taken as given
    assert lib.dfm_conv3d_k3_c32_stats_splits(1, 72, 80, 320, 8) == 50 * 9 * 4


def test_planner_choices_that_came_out_of_the_round_6_plan_sweeps(cv):
    """profiles/r06_c39_*: between two tilings of the same block size the one deeper along d wins (tw > 1), and a
    convolution transposed on all three axes takes two pixel fragments a wave at most (host logic: no GPU)"""
    p = cv.conv3d_g_plan(1, 256, 256, (220, 300, 3))                    # neck.res2: (8, 16, 3) and (16, 8, 3) tie
    assert p['pfw'] == 3 and p['tile'] == (16, 8, 3)
    p = cv.conv3d_g_plan(1, 64, 128, (220, 300, 12), (1, 1, 2), 1)      # neck.down0
    assert p['tile'][0] >= p['tile'][1]
    p = cv.conv3d_g_plan(1, 64, 32, (36, 40, 160), 1, 1, True)          # hourglass conv6 (x2 on d, h, w)
    assert p['pfw'] <= 2
    p = cv.conv3d_g_plan(1, 64, 64, (18, 20, 80), 1, 1, True)           # conv5
    assert p['pfw'] <= 2
    p = cv.conv3d_g_plan(1, 64, 32, (4, 6, 5), 1, 1, (False, False, True))  # one transposed axis: no cap
    assert p['pfw'] in (1, 2, 3, 4)


def test_workspace_sizes_of_the_round_6_entry_points(cv):
    """byte counts the Python host allocates from (no GPU): the strided backward's workspace holds the plane records
    AND the footprint table (12 bytes per sample, plane and lattice point); the 32 -> 1 weight gradient's partials"""
    capi = importlib.import_module('depth-from-motion_amd._capi')
    lib = capi.lib()
    assert lib.dfm_conv3d_to1_wgrad_workspace_bytes() == 1024 * 1024 * 4
    d = capi.SweepDesc()
    d.batch, d.channels, d.num_depths = 8, 32, 72
    d.h_in, d.w_in, d.h_out, d.w_out = 320, 1280, 80, 320
    d.feat_sample_factor, d.cost_sample_factor, d.dtype = 4.0, 4.0, capi.DFM_F32
    n = lib.dfm_plane_sweep_bwd_prev_gather_workspace_bytes(ctypes.byref(d))
    table = 8 * 72 * 80 * 320 * 12
    assert table <= n <= table + 8 * 72 * 12 * 4 + 1024
    d.batch, d.num_depths, d.h_out, d.w_out = 64, 288, 376, 1248      # a table beyond 2 GiB is not asked for
    n = lib.dfm_plane_sweep_bwd_prev_gather_workspace_bytes(ctypes.byref(d))
    assert n == ((64 * 288 * 12 * 4 + 255) // 256) * 256
