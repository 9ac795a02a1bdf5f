"""Registry modules of the plane-sweep path with the reference's constructor
arguments, forward signatures, attribute-injection points and ``state_dict``
keys, so checkpoints and ``configs/dfm/*`` carry over:

  DfMBackbone        mmdet3d/models/backbones/dfm_backbone.py:14-214
  hourglass          mmdet3d/models/utils/conv_modules.py:73-149
  DepthHead          mmdet3d/models/dense_heads/depth_head.py:13-73,190-212 (forward)
  FrustumToVoxel     mmdet3d/models/necks/feature_transformation.py:12-173
  ResModule / OutdoorImVoxelNeck   mmdet3d/models/necks/imvoxel_neck.py
  DfMNeck            mmdet3d/models/necks/dfm_neck.py

The sampling stages call the HIP kernels (plane sweep, frustum-to-voxel,
depth head), GroupNorm(+ReLU) is the fused HIP kernel, and the full-resolution
3x3x3 convolutions with 32 output channels (dres0 / dres1 / pred / voxel_convs)
are ``MfmaConv3d``: the hand-written MFMA kernel of csrc/conv3d.hip when the
stack runs bf16 channels_last_3d.  The stride-2 / 64-channel / transposed
convolutions of the hourglass and the Conv3d+BN3d(+ReLU) blocks of the voxel necks
(64 .. 256 channels, stride (1,1,2), padding (1,1,0)) are ``MfmaConv3dG`` /
``MfmaConvTranspose3d``: the general MFMA kernel of csrc/conv3d_g.hip (in eval mode the
BatchNorm folds into its epilogue together with the residual add and the ReLU).
The 3x3 2-D convolutions of SPPUNetNeck / BEVHourglass run in the same MFMA kernel (kernel extent 1
along depth); 1x1 convolutions and bilinear up-sampling outside the fused SPP tail are torch ops.
"""
import ctypes
import os
import weakref

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .conv3d import (DerivedStateMixin, conv3d_to1_norm, long_axis_gram, note_derived_build, MfmaConv2d, MfmaConv3d, MfmaConv3dG, MfmaConv3dTo1, MfmaConvTranspose2d,
                     MfmaConvTranspose3d, channel_slice, channel_split)
from .depth_head import depth_distribution_loss, depth_head_forward, depth_head_statistics
from .frustum_to_voxel import frustum_to_voxel_sample
from .geometry import stack_meta
from .group_norm import HipBatchNorm3d, HipGroupNorm, _f32_params, batch_norm_train_channels_last
from . import _capi
from .plane_sweep import _DTYPES, _Workspace, _ptr, _stream_ptr, build_dfm_cost
from .sweep_conv import pack_sweep_conv_weights, sweep_conv_supported, sweep_dres0
from .registry import register_module


def _on_device(owner, name, device):
    """device copy of a small tensor the detector injected on the host (depth planes, depth
    samples): uploaded once per (tensor, device) instead of a blocking H2D copy every forward"""
    t = getattr(owner, name)
    if not torch.is_tensor(t) or t.device == device:
        return t
    # the entry holds a weak reference to the host tensor: a re-injected tensor that reuses a freed
    # id() (version 0 again) must not hit a stale device copy
    key = (t._version, str(device))
    cache = owner.__dict__.setdefault('_dev_cache', {})
    hit = cache.get(name)
    if hit is None or hit[0]() is not t or hit[1] != key:
        hit = cache[name] = (weakref.ref(t), key, t.to(device=device, dtype=torch.float32).contiguous())
        note_derived_build()
    return hit[2]


# --------------------------------------------------------------------------
# conv -> norm -> act block with mmcv.cnn.ConvModule's sub-module names
# (`conv`, `gn` / `bn`, `activate`) and bias rule (no conv bias under a norm)
# --------------------------------------------------------------------------
def _make_norm(norm_cfg, channels):
    cfg = dict(norm_cfg)
    kind = cfg.pop('type')
    trainable = cfg.pop('requires_grad', True)
    if kind == 'GN':
        name, layer = 'gn', HipGroupNorm(num_channels=channels, **cfg)
    elif kind == 'BN3d':
        name, layer = 'bn', HipBatchNorm3d(channels, **cfg)
    elif kind in ('BN', 'BN2d'):
        name, layer = 'bn', nn.BatchNorm2d(channels, **cfg)
    else:
        raise NotImplementedError(f'norm type {kind}')
    for p in layer.parameters():
        p.requires_grad = trainable
    return name, layer


class ConvModule(DerivedStateMixin, nn.Module):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU')):
        super().__init__()
        conv_type = 'Conv2d' if conv_cfg is None else conv_cfg['type']
        conv_cls = {'Conv2d': nn.Conv2d, 'Conv3d': nn.Conv3d}[conv_type]
        if (conv_type == 'Conv3d' and out_channels == 32 and in_channels % 32 == 0 and kernel_size == 3
                and stride == 1 and padding == 1 and norm_cfg is not None and norm_cfg.get('type') == 'GN'):
            # the full-resolution 3x3x3 convolutions of the aggregation stacks (GroupNorm follows: its
            # statistics come out of the epilogue): nn.Conv3d whose bf16 / NDHWC forward is the
            # hand-written MFMA kernel (csrc/conv3d.hip).  Under BatchNorm3d (a voxel neck narrowed to
            # 32 channels) the general kernel below folds the norm into its epilogue instead.
            conv_cls = MfmaConv3d
        elif (conv_type == 'Conv3d' and out_channels % 32 == 0 and in_channels % 32 == 0 and kernel_size == 3
              and norm_cfg is not None):
            # every other 3x3x3 convolution of the path (64 .. 256 channels, stride (1,1,2), padding
            # (1,1,0): the BN3d stacks of the voxel necks): the general MFMA kernel (csrc/conv3d_g.hip)
            conv_cls = MfmaConv3dG
        elif conv_type == 'Conv2d' and MfmaConv2d.covers(in_channels, out_channels, kernel_size, stride, padding):
            # the convolutions of the 2-D necks either side of the path (SPPUNetNeck, BEVHourglass): 3x3 /
            # padding 1 / stride 1 | 2 with whole 32-channel chunks run the same MFMA kernel with a (1, 3, 3)
            # kernel on the NHWC tensor as a depth-1 volume; a 1x1 convolution of an NHWC tensor is a matrix
            # product.  A configuration outside that coverage (48 channels, padding 2, stride 3 ...) stays a
            # plain nn.Conv2d: it never had a kernel here, so strict mode has nothing to complain about
            conv_cls = MfmaConv2d
        self.conv = conv_cls(in_channels, out_channels, kernel_size, stride=stride,
                             padding=padding, bias=norm_cfg is None)
        self.norm_name = None
        if norm_cfg is not None:
            self.norm_name, norm = _make_norm(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        self.activate = None
        if act_cfg is not None:
            if act_cfg['type'] != 'ReLU':
                raise NotImplementedError(act_cfg['type'])
            self.activate = nn.ReLU(inplace=act_cfg.get('inplace', True))

    def forward(self, x, residual=None, relu=None):
        """``residual`` (extension): added after the norm, before the activation -- fused into the
        normalisation pass of the HIP GroupNorm / BatchNorm.  ``relu``: overrides the block's own
        activation (ResModule applies its ReLU after the identity add)."""
        norm = getattr(self, self.norm_name) if self.norm_name is not None else None
        act = (self.activate is not None) if relu is None else bool(relu)
        if (isinstance(self.conv, MfmaConv3d) and isinstance(norm, HipGroupNorm) and
                norm.num_groups == self.conv.out_channels and self.conv.eligible(x)):
            # MFMA conv whose epilogue already produced the per-channel GroupNorm statistics:
            # the normalisation (+residual, +ReLU) is one read and one write of the tensor
            y, partials = self.conv.forward_with_stats(x)
            return norm(y, relu=act, partials=partials, residual=residual)
        if residual is None and relu is None and self.fusable(x):
            return self.forward_fused(x)
        x = self.conv(x)
        if self.norm_name is not None:
            if isinstance(norm, (HipGroupNorm, HipBatchNorm3d)):
                # norm (+residual) and ReLU in one pass
                return norm(x, relu=act, residual=residual)
            x = norm(x)
        if residual is not None:
            x = x + residual
        if act:
            x = F.relu(x)
        return x

    # -- inference path of the Conv3d + BatchNorm3d (+ReLU) blocks of the voxel necks: the running
    # statistics fold into a per-channel scale / shift applied to the fp32 accumulator in the
    # convolution's epilogue, together with the residual add and the ReLU (one kernel per block)
    def fusable(self, x):
        norm = getattr(self, self.norm_name) if self.norm_name is not None else None
        return (isinstance(self.conv, MfmaConv3dG) and isinstance(norm, nn.BatchNorm3d) and
                not norm.training and norm.track_running_stats and not torch.is_grad_enabled() and
                self.conv.eligible(x))

    def _folded_norm(self):
        """(scale, shift) of the eval-mode BatchNorm, cached until one of its tensors changes (eight
        tiny kernels per block and forward otherwise: launch-bound on the 9-block necks)"""
        norm = getattr(self, self.norm_name)
        ts = [norm.running_mean, norm.running_var] + ([norm.weight, norm.bias] if norm.affine else [])
        key = tuple((t._version, t.data_ptr()) for t in ts) + (norm.eps,)
        if getattr(self, '_fold_key', None) != key:
            scale = torch.rsqrt(norm.running_var.float() + norm.eps)
            if norm.affine:
                scale = scale * norm.weight.float()
            shift = -norm.running_mean.float() * scale
            if norm.affine:
                shift = shift + norm.bias.float()
            self._fold, self._fold_key = (scale.contiguous(), shift.contiguous()), key
            note_derived_build()
        return self._fold

    def forward_fused(self, x, residual=None, relu=None):
        scale, shift = self._folded_norm()
        relu = (self.activate is not None) if relu is None else relu
        return self.conv.forward_fused(x, scale, shift, residual, relu)


class _WindowMean2d(nn.AvgPool2d):
    """nn.AvgPool2d(k, stride=k) (SPP branches, spp_unet_neck.py:60-70) as a reshape + mean on the GPU:
    torch's avg_pool2d kernel walks the 64 x 64 / 32 x 32 windows serially per output (0.44 ms per
    branch at config K, 3.5 ms per forward of the two necks); the windows do not overlap, so the
    pooled map is a mean over two reshaped axes (floor mode: trailing rows / columns dropped)."""

    def forward(self, x):
        kh, kw = self.kernel_size if isinstance(self.kernel_size, tuple) else (self.kernel_size,) * 2
        st = self.stride if isinstance(self.stride, tuple) else (self.stride,) * 2
        if not x.is_cuda or (kh, kw) != tuple(st) or self.padding not in (0, (0, 0)) or self.ceil_mode:
            return super().forward(x)
        B, C, H, W = x.shape
        ho, wo = H // kh, W // kw
        if not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last):
            v = x.permute(0, 2, 3, 1)[:, :ho * kh, :wo * kw].reshape(B, ho, kh, wo, kw, C)
            return v.float().mean(dim=(2, 4)).to(x.dtype).permute(0, 3, 1, 2)  # stays channels_last
        v = x[:, :, :ho * kh, :wo * kw].reshape(B, C, ho, kh, wo, kw)
        return v.float().mean(dim=(3, 5)).to(x.dtype)


class _DepthPoolFn(torch.autograd.Function):
    """mean over groups of ``k`` consecutive depth planes of a channels_last_3d (N, C, D, H, W) tensor, forward and
    backward one HIP pass each (csrc/depth_pool.hip)"""

    @staticmethod
    def forward(ctx, x, k):
        N, C, D, H, W = x.shape
        ctx.k, ctx.shape = k, tuple(x.shape)
        y = torch.empty((N, D // k, H, W, C), dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _capi.check(_capi.lib().dfm_depth_pool_fwd(N * (D // k), k, H * W * C, _DTYPES[x.dtype], _ptr(x), _ptr(y),
                                                       _stream_ptr(x.device)))
        return y.permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, gy):
        N, C, D, H, W = ctx.shape
        k = ctx.k
        gy = gy.contiguous(memory_format=torch.channels_last_3d)
        gx = torch.empty((N, D, H, W, C), dtype=gy.dtype, device=gy.device)
        with torch.cuda.device(gy.device):
            _capi.check(_capi.lib().dfm_depth_pool_bwd(N * (D // k), k, H * W * C, _DTYPES[gy.dtype], _ptr(gy), _ptr(gx),
                                                       _stream_ptr(gy.device)))
        return gx.permute(0, 4, 1, 2, 3), None


def _depth_pool4(pool, x):
    """AvgPool3d((4,1,1)) of FrustumToVoxel (feature_transformation.py:167).  torch's kernel makes a
    channels_last_3d input contiguous first (a 224 MB copy at config K); on the NDHWC path the pooled depth axis is
    the slowest axis of a sample: one pass of csrc/depth_pool.hip each way (round 6; rounds 3-5 ran
    float() -> mean -> cast, three kernels and their autograd twins), and the result stays channels-last."""
    k = pool.kernel_size if isinstance(pool.kernel_size, tuple) else (pool.kernel_size,) * 3
    if (x.is_cuda and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last_3d) and
            k[1:] == (1, 1) and tuple(pool.stride) == tuple(k) and x.shape[2] % k[0] == 0):
        B, C, D, H, W = x.shape
        if x.dtype in _DTYPES and (H * W * C * x.element_size()) % 16 == 0 and x.data_ptr() % 16 == 0:
            return _DepthPoolFn.apply(x, k[0])
        v = x.permute(0, 2, 3, 4, 1).reshape(B, D // k[0], k[0], H, W, C)
        return v.float().mean(dim=2).to(x.dtype).permute(0, 4, 1, 2, 3)
    return pool(x)


def _conv3(cin, cout, norm_cfg, act=True, stride=1, padding=1):
    return ConvModule(cin, cout, 3, stride=stride, padding=padding, conv_cfg=dict(type='Conv3d'),
                      norm_cfg=norm_cfg, act_cfg=dict(type='ReLU', inplace=True) if act else None)


# --------------------------------------------------------------------------
# hourglass (conv_modules.py:73-149): keys conv1.0.0, conv2.0, ..., conv5.0/1
# --------------------------------------------------------------------------
def _norm3d(channels, gn):
    # conv_modules.py:42-43,113-127: GroupNorm(32, C) or (Sync)BatchNorm3d -- same state_dict keys
    return HipGroupNorm(32, channels) if gn else HipBatchNorm3d(channels)


def _convgn3d(cin, cout, stride, gn=True):
    return nn.Sequential(MfmaConv3dG(cin, cout, 3, stride=stride, padding=1, bias=False), _norm3d(cout, gn))


class hourglass(nn.Module):  # noqa: N801  (reference class name)

    def __init__(self, inplanes, gn=True):
        super().__init__()
        c = inplanes
        self.conv1 = nn.Sequential(_convgn3d(c, 2 * c, 2, gn), nn.ReLU(inplace=True))
        self.conv2 = _convgn3d(2 * c, 2 * c, 1, gn)
        self.conv3 = nn.Sequential(_convgn3d(2 * c, 2 * c, 2, gn), nn.ReLU(inplace=True))
        self.conv4 = nn.Sequential(_convgn3d(2 * c, 2 * c, 1, gn), nn.ReLU(inplace=True))
        self.conv5 = nn.Sequential(
            MfmaConvTranspose3d(2 * c, 2 * c, 3, padding=1, output_padding=1, stride=2, bias=False),
            _norm3d(2 * c, gn))
        self.conv6 = nn.Sequential(
            MfmaConvTranspose3d(2 * c, c, 3, padding=1, output_padding=1, stride=2, bias=False),
            _norm3d(c, gn))

    def forward(self, x, presqu, postsqu):
        # GroupNorm and the ReLU that follows it are one pass of the fused kernel
        return self.forward_add(x, presqu, postsqu, None)

    def forward_add(self, x, presqu, postsqu, out_residual):
        """forward with ``out_residual`` added to the first output (DfMBackbone's
        ``cost = cost + hourglass(cost)[0]``, dfm_backbone.py:180-183).  GroupNorm, the residual adds
        and the ReLUs that follow them are one pass of the fused kernel each."""
        steps = self.forward_add_steps(x, presqu, postsqu, out_residual)
        while True:
            try:
                next(steps)
            except StopIteration as done:
                return done.value

    def forward_add_steps(self, x, presqu, postsqu, out_residual):
        """``forward_add`` as a generator that yields after every layer (conv + norm: four launches) and returns the
        result: DfMBackbone issues the layers of its two stacks alternately, each on its own HIP stream"""
        down1 = _gn_relu(self.conv1[0], x, True)
        yield
        pre = _gn_relu(self.conv2, down1, True, postsqu)
        yield
        mid = _gn_relu(self.conv3[0], pre, True)
        yield
        bottom = _gn_relu(self.conv4[0], mid, True)
        yield
        post = _gn_relu(self.conv5, bottom, True, pre if presqu is None else presqu)
        yield
        return _gn_relu(self.conv6, post, False, out_residual), pre, post


# --------------------------------------------------------------------------
# DfMBackbone
# --------------------------------------------------------------------------
@register_module
class DfMBackbone(DerivedStateMixin, nn.Module):

    def __init__(self,
                 in_channels,
                 num_hg=1,
                 cost_sample_factor=4,
                 feat_sample_factor=1,
                 cv_channels=32,
                 depth_cfg=dict(mode='UD', num_bins=288, depth_min=2, depth_max=59.6,
                                downsample_factor=4),
                 norm_cfg=dict(type='GN', num_groups=32, requires_grad=True),
                 init_cfg=None):
        super().__init__()
        self.norm_cfg = norm_cfg
        self.GN = True
        self.cost_sample_factor = cost_sample_factor
        self.feat_sample_factor = feat_sample_factor
        self.num_hg = num_hg
        self.cv_channels = cv_channels
        self.in_channels = in_channels
        cv = cv_channels

        def branch(cin):
            return (_conv3(cin, cv, norm_cfg), _conv3(cv, cv, norm_cfg, act=False),
                    nn.ModuleList(hourglass(cv, gn=True) for _ in range(num_hg)),
                    nn.ModuleList(
                        nn.Sequential(_conv3(cv, cv, norm_cfg),
                                      (MfmaConv3dTo1 if cv == 32 else nn.Conv3d)(cv, 1, 3, 1, 1, bias=False))
                        for _ in range(num_hg)))

        self.dres0, self.dres1, self.hg_stereo, self.pred_stereo = branch(2 * in_channels)
        self.dres0_mono, self.dres1_mono, self.hg_mono, self.pred_mono = branch(in_channels)
        planes = round(depth_cfg['num_bins'] // depth_cfg['downsample_factor'])
        self.aggregate_cost = nn.Conv2d(2 * planes, planes, kernel_size=1, bias=False)
        # injected by the detector (dfm.py:88-89); a tensor of plane depths
        self.downsampled_depth = None
        # extension: torch.channels_last_3d builds the cost volume (B, D, H, W, 2C) so the
        # Conv3d / GroupNorm stack runs NDHWC end to end (convert the module as well:
        # backbone.to(memory_format=torch.channels_last_3d)); values are unchanged
        self.volume_memory_format = torch.contiguous_format
        # extension (inference, bf16 NDHWC, 32-channel maps): build_dfm_cost, dres0.conv and
        # dres0_mono.conv run as ONE kernel and the (B, 2C, D, H, W) volume is never written
        # (csrc/sweep_conv.hip); False keeps the materialised volume
        self.fuse_sweep_dres0 = True
        self._sweep_conv_pack = (None, None)

    def init_weights(self):
        pass

    @staticmethod
    def _aggregate(first, second, hgs, x):
        return DfMBackbone._aggregate_rest(second, hgs, first(x))

    @staticmethod
    def _aggregate_rest(second, hgs, cost):
        cost = second(cost, residual=cost)
        outs = []
        for hg in hgs:
            cost, _, _ = hg.forward_add(cost, None, None, cost)  # cost + hourglass(cost)[0]
            outs.append(cost)
        return outs if outs else [cost]

    # The stereo and the mono aggregation stacks (dfm_backbone.py:175-183 / 189-198) share nothing until
    # `_predict`: at inference they run on TWO HIP streams (round 5).  Each stack is a chain of ~30 launches --
    # full-resolution convolutions that leave a quarter of the CUs idle in their second round of workgroups,
    # hourglass levels of 250 workgroups (half a chip), GroupNorm passes -- and the other stack's kernels fill
    # what one leaves free.  The side stream waits for what the main stream has produced so far, the main stream
    # joins it before the prediction heads; scratch is per (device, stream) (plane_sweep._Workspace) and the
    # side stream's results are handed to the main stream with record_stream.  two_streams = False pins one stream.
    two_streams = True
    # With autograd recording: every backward node runs on the stream its forward ran on (the engine inserts the
    # cross-stream waits), so the two stacks overlap in the backward pass as well (backbone_train 12.5 -> 10.7 ms,
    # profiles/archive/r05_c16_*).  The mono stack's PARAMETER gradients, however, must not be produced on the side stream:
    # gradient hooks (DistributedDataParallel's reducer, GradientBucketReducer) run under the stream of the
    # AccumulateGrad node and order their bucket copies / collectives against that stream only.  An AccumulateGrad
    # node takes the stream that is current when it is CREATED, so `_two_branches` creates the mono parameters'
    # nodes on the main stream before it forks (`_pin_accumulators`): the weight-gradient kernels still run on the
    # side stream, the accumulation -- and every hook -- on the main stream behind the engine's event wait
    # (tests/test_backward_gpu.py checks the stream the hooks see).  DFM_TRAIN_ONE_STREAM=1 pins one stream.
    two_streams_training = os.environ.get('DFM_TRAIN_ONE_STREAM') != '1'
    _side_streams = {}

    # DFM_BACKBONE_SEQUENTIAL_ISSUE=1: the round-5 issue order (all of the mono stack, then all of the stereo stack)
    interleaved_issue = os.environ.get('DFM_BACKBONE_SEQUENTIAL_ISSUE') != '1'

    def _stack_steps(self, first, second, hgs, pred):
        """one aggregation stack + its prediction head as a generator: ``first()`` produces the stack's input (the
        first block, or its normalisation on the fused-sweep path), then dres1, the hourglasses and the head; yields
        after every layer, returns (features, cost)"""
        cost = first()
        yield
        cost = second(cost, residual=cost)
        yield
        outs = []
        for hg in hgs:
            cost, _, _ = yield from hg.forward_add_steps(cost, None, None, cost)  # cost + hourglass(cost)[0]
            outs.append(cost)
            yield
        outs = outs if outs else [cost]
        assert len(outs) == 1, 'Only support num_hg=1 for now.'
        return outs, self._pred_head(pred, outs[0])

    def _two_stacks_interleaved(self, stereo_first, mono_first, device):
        """Inference: the two stacks on two HIP streams with their layers ISSUED ALTERNATELY (round 6).  A layer costs
        the host 20-27 us to enqueue (tools/host_overhead_probe.py) against 15-60 us on the device, and round 5
        enqueued the whole mono stack (~0.5 ms of host time) before the first stereo launch: the main stream sat idle
        behind the fused sweep for as long, and the forward took sweep + host(mono) + stereo.  Alternating the two
        generators keeps both streams fed from the start."""
        main = torch.cuda.current_stream(device)
        side = DfMBackbone._side_streams.get(device)
        if side is None:
            side = DfMBackbone._side_streams[device] = torch.cuda.Stream(device=device)
        side.wait_stream(main)
        gens = [(side, self._stack_steps(mono_first, self.dres1_mono, self.hg_mono, self.pred_mono[0])),
                (main, self._stack_steps(stereo_first, self.dres1, self.hg_stereo, self.pred_stereo[0]))]
        results = {}
        try:
            while gens:
                for item in list(gens):
                    st, gen = item
                    torch.cuda.set_stream(st)
                    try:
                        next(gen)
                    except StopIteration as done:
                        results[st is side] = done.value
                        gens.remove(item)
        finally:
            torch.cuda.set_stream(main)
        (mono, m_cost), (stereo, s_cost) = results[True], results[False]
        main.wait_stream(side)
        for t in list(mono) + [m_cost]:
            t.record_stream(main)
        return (stereo, s_cost), (mono, m_cost)

    def _two_branches(self, stereo_fn, mono_fn, device):
        # each branch ends with its own prediction head (dfm_backbone.py:120-127): -> (features, cost)
        def stereo_all():
            st = stereo_fn()
            assert len(st) == 1, 'Only support num_hg=1 for now.'
            return st, self._pred_head(self.pred_stereo[0], st[0])

        def mono_all():
            mo = mono_fn()
            assert len(mo) == 1, 'Only support num_hg=1 for now.'
            return mo, self._pred_head(self.pred_mono[0], mo[0])
        if not (self.two_streams and device.type == 'cuda' and
                (self.two_streams_training or not torch.is_grad_enabled()) and
                not torch.cuda.is_current_stream_capturing()):
            return stereo_all(), mono_all()
        main = torch.cuda.current_stream(device)
        side = DfMBackbone._side_streams.get(device)
        if side is None:
            side = DfMBackbone._side_streams[device] = torch.cuda.Stream(device=device)
        pins = self._pin_accumulators() if torch.is_grad_enabled() else None
        side.wait_stream(main)
        with torch.cuda.stream(side):
            mono, m_cost = mono_all()
        del pins  # (the recorded graph holds the nodes from here on)
        stereo, s_cost = stereo_all()
        main.wait_stream(side)
        for t in list(mono) + [m_cost]:
            t.record_stream(main)
        return (stereo, s_cost), (mono, m_cost)

    # DFM_PRED_UNFUSED=1 keeps the three-launch prediction head at inference too (A/B runs)
    fused_pred = os.environ.get('DFM_PRED_UNFUSED') != '1'

    def _pred_head(self, seq, x):
        """ConvModule(32 -> 32, GN, ReLU) -> Conv3d(32 -> 1) (dfm_backbone.py:120-127).  Inference on the bf16 NDHWC
        stack: the 32 -> 32 MFMA convolution emits its GroupNorm statistics, and the 32 -> 1 convolution normalises
        (+ReLU) the raw volume ON LOAD (csrc/conv3d_to1n.hip) -- the normalisation pass (a read and a write of the
        118 MB volume per stack at config K) and the 31 zero weight rows of the former 32 -> 1 kernel are gone."""
        cm, last = seq[0], seq[1]
        norm = getattr(cm, cm.norm_name or '', None)
        if (self.fused_pred and not torch.is_grad_enabled() and isinstance(cm.conv, MfmaConv3d) and
                isinstance(last, MfmaConv3dTo1) and isinstance(norm, HipGroupNorm) and norm.num_groups == 32 and
                norm.affine and cm.conv.out_channels == 32 and cm.activate is not None and cm.conv.eligible(x) and
                last.weight.shape == (1, 32, 3, 3, 3) and last.bias is None):
            y, partials = cm.conv.forward_with_stats(x)
            gamma, beta = _f32_params(norm.weight, norm.bias)
            return conv3d_to1_norm(y, partials, gamma, beta, norm.eps, last.weight, relu=True)
        return seq(x)

    def _interleave_ok(self, device):
        return (self.interleaved_issue and self.two_streams and device.type == 'cuda' and
                not torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing())

    def _pin_accumulators(self):
        """views of the mono stack's trainable parameters taken on the CURRENT (main) stream: each creates the
        parameter's AccumulateGrad node, which keeps the stream it was created under; holding the views keeps the
        nodes alive until the side-stream forward has linked them into the graph"""
        # (torch >= 2.10 notes once per process that these nodes' stream differs from their producers': intended --
        #  the engine's event wait orders them; the process-wide switch for that note is left to the application)
        return [p.view_as(p) for m in (self.dres0_mono, self.dres1_mono, self.hg_mono, self.pred_mono)
                for p in m.parameters() if p.requires_grad]


    def _sweep_dres0_fusable(self, cur, prev=None):
        """the fused plane sweep + dres0 / dres0_mono kernel takes this call: inference, bf16 32-channel
        maps, the NDHWC stack, both first blocks Conv3d(-> 32) + GroupNorm(one channel per group) + ReLU"""
        def block_ok(cm, cin):
            norm = getattr(cm, cm.norm_name or '', None)
            return (isinstance(cm.conv, MfmaConv3d) and cm.conv.in_channels == cin and cm.conv.out_channels == 32 and
                    cm.conv.weight.dtype == torch.bfloat16 and isinstance(norm, HipGroupNorm) and
                    norm.num_groups == 32 and norm.affine and cm.activate is not None)
        return (self.fuse_sweep_dres0 and not torch.is_grad_enabled() and sweep_conv_supported(cur) and
                (prev is None or (prev.dtype == cur.dtype and prev.shape == cur.shape and prev.device == cur.device)) and
                self.in_channels == 32 and self.volume_memory_format == torch.channels_last_3d and
                block_ok(self.dres0, 64) and block_ok(self.dres0_mono, 32))

    def _sweep_conv_packed(self):
        ws, wm = self.dres0.conv.weight, self.dres0_mono.conv.weight
        key = (ws._version, ws.data_ptr(), wm._version, wm.data_ptr(), str(ws.device))
        if self._sweep_conv_pack[0] != key:
            self._sweep_conv_pack = (key, pack_sweep_conv_weights(ws, wm))
            note_derived_build()
        return self._sweep_conv_pack[1]

    def forward(self, cur_stereo_feats, prev_stereo_feats, img_metas, cur_sem_feats=None):
        # (matrices staged on the device by data_geometry.stage_geometry are read where they lie)
        ori_cam2imgs = stack_meta(img_metas, 'ori_cam2img')
        cur2prevs = stack_meta(img_metas, 'cur2prevs')
        meta0 = img_metas[0]
        if self._sweep_dres0_fusable(cur_stereo_feats, prev_stereo_feats):
            # plane sweep + dres0.conv + dres0_mono.conv: one kernel, no cost volume in HBM; GroupNorm
            # (+ReLU) of both branches from the kernel's statistics partials
            ys, ps, ym, pm = sweep_dres0(
                cur_stereo_feats, prev_stereo_feats, _on_device(self, 'downsampled_depth', cur_stereo_feats.device),
                self.feat_sample_factor, self.cost_sample_factor, ori_cam2imgs, cur2prevs[:, 0],
                meta0['ori_shape'][:2], self._sweep_conv_packed(), meta0.get('flip', False), meta0['crop_offset'],
                img_scale_factor=meta0.get('scale_factor', [1.0])[0])
            if self._interleave_ok(ys.device):
                stereo, mono = self._two_stacks_interleaved(
                    lambda: self.dres0.gn(ys, relu=True, partials=ps),
                    lambda: self.dres0_mono.gn(ym, relu=True, partials=pm), ys.device)
                return self._predict(*stereo, *mono)
            stereo, mono = self._two_branches(
                lambda: self._aggregate_rest(self.dres1, self.hg_stereo, self.dres0.gn(ys, relu=True, partials=ps)),
                lambda: self._aggregate_rest(self.dres1_mono, self.hg_mono,
                                             self.dres0_mono.gn(ym, relu=True, partials=pm)), ys.device)
            return self._predict(*stereo, *mono)
        # plane sweep: HIP kernel (reference: build_dfm_cost, batch semantics per sample)
        cost_raw = build_dfm_cost(
            cur_stereo_feats, prev_stereo_feats,
            _on_device(self, 'downsampled_depth', cur_stereo_feats.device), self.feat_sample_factor,
            self.cost_sample_factor, ori_cam2imgs, cur2prevs[:, 0], meta0['ori_shape'][:2],
            meta0.get('flip', False), meta0['crop_offset'],
            img_scale_factor=meta0.get('scale_factor', [1.0])[0],
            memory_format=self.volume_memory_format)
        # the volume's two consumers: dres0 (whole) and dres0_mono (the cur half) -- one autograd node whose backward adds
        # the half's gradient into the whole volume's in place (conv3d.channel_split)
        cost_all, cost_cur = channel_split(cost_raw, 0, self.in_channels)
        if self._interleave_ok(cost_raw.device):
            stereo, mono = self._two_stacks_interleaved(
                lambda: self.dres0(cost_all), lambda: self.dres0_mono(cost_cur), cost_raw.device)
            return self._predict(*stereo, *mono)
        stereo, mono = self._two_branches(
            lambda: self._aggregate(self.dres0, self.dres1, self.hg_stereo, cost_all),
            lambda: self._aggregate(self.dres0_mono, self.dres1_mono, self.hg_mono, cost_cur), cost_raw.device)
        return self._predict(*stereo, *mono)

    # DFM_GATE_TORCH=1 keeps the torch sequence of the gate at inference too (A/B runs)
    fused_gate = os.environ.get('DFM_GATE_TORCH') != '1'
    mfma_gate = os.environ.get('DFM_GATE_VALU') != '1'

    def _gate_fused(self, s_cost, m_cost):
        """cat + Conv2d(2D -> D, 1x1) + sigmoid + blend (dfm_backbone.py:136-141) as one launch
        (csrc/cost_gate.hip): inference, both costs (B, 1, D, H, W) contiguous on the GPU; None otherwise."""
        w = self.aggregate_cost.weight
        if not (self.fused_gate and s_cost.is_cuda and not torch.is_grad_enabled() and s_cost.dtype in _DTYPES
                and w.dtype in _DTYPES and m_cost.dtype == s_cost.dtype and m_cost.shape == s_cost.shape
                and s_cost.dim() == 5 and s_cost.shape[1] == 1 and s_cost.shape[2] <= 96
                and s_cost.is_contiguous() and m_cost.is_contiguous() and w.is_contiguous()
                and w.shape[0] == s_cost.shape[2] and w.shape[1] == 2 * s_cost.shape[2]):
            return None
        B, _, D, H, W = s_cost.shape
        lib = _capi.lib()
        out = torch.empty_like(s_cost)
        # bf16 costs and a bf16 weight (the fast path's model): the product on the matrix cores (round 6; every
        # bf16 x bf16 product is exact, sums in fp32) -- DFM_GATE_VALU=1 keeps the VALU kernel (A/B runs)
        mfma = (self.mfma_gate and s_cost.dtype == torch.bfloat16 and w.dtype == torch.bfloat16)
        key = (w._version, w.data_ptr(), str(w.device), w.dtype, mfma)
        with torch.cuda.device(s_cost.device):
            if self.__dict__.get('_gate_pack', (None, None))[0] != key:  # packed once per weight version
                nb = lib.dfm_cost_gate_mfma_weight_bytes(D) if mfma else lib.dfm_cost_gate_weight_bytes(D)
                packed = torch.empty(nb, dtype=torch.uint8, device=w.device)
                pack = lib.dfm_cost_gate_mfma_pack_weights if mfma else lib.dfm_cost_gate_pack_weights
                _capi.check(pack(_ptr(w.detach()), _DTYPES[w.dtype], D, _ptr(packed), _stream_ptr(s_cost.device)))
                self.__dict__['_gate_pack'] = (key, packed)
                note_derived_build()
            if mfma:
                _capi.check(lib.dfm_cost_gate_mfma_fwd(B, D, H * W, _ptr(s_cost), _ptr(m_cost),
                                                       _ptr(self.__dict__['_gate_pack'][1]), _ptr(out),
                                                       _stream_ptr(s_cost.device)))
            else:
                _capi.check(lib.dfm_cost_gate_fwd(B, D, H * W, _DTYPES[s_cost.dtype], _ptr(s_cost), _ptr(m_cost),
                                                  _ptr(self.__dict__['_gate_pack'][1]), _ptr(out),
                                                  _stream_ptr(s_cost.device)))
        return out

    def _predict(self, stereo, s_cost, mono, m_cost):
        fused = self._gate_fused(s_cost, m_cost)
        if fused is not None:
            return fused, stereo[0], mono[0]
        both = torch.cat((s_cost, m_cost), dim=1).flatten(start_dim=1, end_dim=2)
        if both.is_cuda:
            # the 1x1 Conv2d(2D -> D) as a GEMM over the flattened image (hipBLASLt forward and
            # backward): MIOpen's kernels for this shape are naive fallbacks in bf16 (1.2 ms backward-weight)
            w2 = self.aggregate_cost.weight.flatten(1)
            if torch.is_grad_enabled() and w2.requires_grad:
                gate = _GateLogitsFn.apply(w2, both.flatten(2)).view(both.shape[0], -1, *both.shape[2:])
            else:
                gate = torch.matmul(w2, both.flatten(2)).view(both.shape[0], -1, *both.shape[2:])
            gate = gate.unsqueeze(dim=1).sigmoid()
        else:
            gate = self.aggregate_cost(both).unsqueeze(dim=1).sigmoid()
        return gate * s_cost + (1 - gate) * m_cost, stereo[0], mono[0]


class _GateLogitsFn(torch.autograd.Function):
    """``w2 @ both`` (the 1x1 Conv2d(2D -> D) of the gate as a GEMM over the flattened image) whose weight gradient is a
    batched product over slices of the pixel axis (``conv3d.long_axis_gram``: D x 2D output tiles that each walk all
    H * W pixels are a few workgroups for the whole chip)."""

    @staticmethod
    def forward(ctx, w2, both):
        ctx.save_for_backward(w2, both)
        return torch.matmul(w2, both)

    @staticmethod
    def backward(ctx, g):
        w2, both = ctx.saved_tensors
        gw = gb = None
        if ctx.needs_input_grad[1]:
            gb = torch.matmul(w2.t(), g)
        if ctx.needs_input_grad[0]:
            gw = sum(long_axis_gram(g[i].t(), both[i].t()) for i in range(g.shape[0])).to(w2.dtype)
        return gw, gb


# --------------------------------------------------------------------------
# DepthHead (forward path; the loss is outside SURVEY.md 8a)
# --------------------------------------------------------------------------
@register_module
class DepthHead(DerivedStateMixin, nn.Module):

    def __init__(self, depth_cfg, in_channels=32, with_convs=True,
                 depth_loss=dict(type='ce', loss_weight=1.0), downsample_factor=4, num_views=5,
                 norm_cfg=dict(type='GN', num_groups=32, requires_grad=True)):
        super().__init__()
        self.in_channels = in_channels
        self.depth_cfg = depth_cfg
        self.with_convs = with_convs
        self.depth_loss = depth_loss
        self.downsample_factor = downsample_factor
        self.num_views = num_views
        self.norm_cfg = norm_cfg
        self.depth_loss_type = depth_loss['type']
        if self.depth_loss_type in ('balanced_ce', 'balanced_focal'):
            self.fg_weight = depth_loss['fg_weight']
            self.bg_weight = depth_loss['bg_weight']
        if self.depth_loss_type in ('focal', 'balanced_focal'):
            self.alpha = depth_loss['alpha']
            self.gamma = depth_loss['gamma']
        self.loss_weight = depth_loss['loss_weight']
        self.min_depth = depth_cfg['min_depth']
        self.max_depth = depth_cfg['max_depth']
        if with_convs:
            self.conv_depth = nn.Conv3d(in_channels, 1, 3, 1, 1, bias=False)
        self.depth_samples = None  # injected by the detector (dfm.py:90)

    def forward(self, stereo_features, lazy=False):
        """``lazy=True`` (extension, single view): returns (None, depth_preds, LazyDepthDistribution) --
        the distribution is handed to FrustumToVoxel unmaterialised and evaluated inside its sampling
        kernel (the DepthHead -> FrustumToVoxel fusion).  While autograd records (training) the first
        result is the distribution too: ``loss`` accepts it in place of ``depth_volumes`` and
        evaluates the valid pixels' logits from the low-resolution cost, forward and backward."""
        _, _, D, H, W = stereo_features.shape
        x = stereo_features
        if lazy and not self.with_convs and x.shape[1] == 1:
            # with autograd recording (training) the distribution keeps the cost's graph node: DepthHead.loss
            # takes it in place of depth_volumes (returned FIRST for that); depth_preds carries no gradient
            train = torch.is_grad_enabled() and x.requires_grad
            dist, pred = depth_head_statistics(x, _on_device(self, 'depth_samples', x.device),
                                               self.downsample_factor, keep_graph=train)
            # (grad mode with a cost that records nothing -- a frozen backbone, a validation loss -- still
            # hands the distribution over: loss() evaluates it, there is just nothing to differentiate)
            return (dist if torch.is_grad_enabled() else None), pred, dist
        if self.with_convs:
            x = self.conv_depth(x).view(-1, self.num_views, D, H, W)
        if x.shape[1] != 1:
            # several views share one launch: fold them into the batch and back
            B, V = x.shape[:2]
            vol, soft, pred = depth_head_forward(x.reshape(B * V, 1, D, H, W), self.depth_samples,
                                                 self.downsample_factor)
            s = self.downsample_factor
            return (vol.view(B, V, s * D, s * H, s * W), soft.view(B, V, s * D, s * H, s * W),
                    pred.view(B, V, s * H, s * W))
        return depth_head_forward(x, _on_device(self, 'depth_samples', x.device), self.downsample_factor)

    def loss(self, depth_preds, depth_volumes, depth_img, depth_fgmask_img=None):
        """depth_head.py:75-188.  depth_preds [B*N,H,W], depth_volumes [B*N,D,H,W],
        depth_img [B*N,H,W]; depth_fgmask_img: fg mask with box ids (balanced_* losses).

        The distribution losses run in one HIP kernel over the volume in place (no
        (n_valid, D) gather copy, no log_softmax tensor); only the (B*N,H,W)-sized reductions
        are torch ops.  Quirk kept: the result is scaled by loss_weight TWICE (:186)."""
        loss_type = self.depth_loss_type
        mask = (depth_img > self.min_depth) & (depth_img < self.max_depth)
        n_valid = mask.sum()
        if loss_type in ('l1', 'purel1'):
            if int(n_valid) == 0:
                print('no gt warning')
                return depth_preds.mean() * 0.0
            fn = F.smooth_l1_loss if loss_type == 'l1' else F.l1_loss
            loss = fn(depth_preds[mask], depth_img[mask], reduction='none').mean()
            return self.loss_weight * self.loss_weight * loss
        pix, valid = depth_distribution_loss(
            depth_volumes, depth_img, self.depth_samples, loss_type, self.min_depth, self.max_depth,
            getattr(self, 'alpha', 1.0), getattr(self, 'gamma', 2.0))
        n = valid.sum()
        if int(n) == 0:
            print('no gt warning')
            return depth_preds.mean() * 0.0
        if loss_type in ('balanced_ce', 'balanced_focal'):
            fg = depth_fgmask_img.bool() & valid
            bg = valid & ~fg
            loss = (self.fg_weight * (pix * fg).sum() + self.bg_weight * (pix * bg).sum()) / n
        else:
            loss = pix.sum() / n
        return self.loss_weight * self.loss_weight * loss


# --------------------------------------------------------------------------
# FrustumToVoxel
# --------------------------------------------------------------------------
@register_module
class FrustumToVoxel(DerivedStateMixin, nn.Module):

    def __init__(self, num_3dconvs=1, cv_channels=32, out_channels=32, in_sem_channels=32,
                 sem_atten_feat=True, stereo_atten_feat=False, cat_img_feature=True,
                 norm_cfg=dict(type='GN', num_groups=32, requires_grad=True), init_cfg=None):
        super().__init__()
        self.GN = True
        self.num_3dconvs = num_3dconvs
        self.cv_channels = cv_channels
        self.out_channels = out_channels
        self.in_sem_channels = in_sem_channels
        self.sem_atten_feat = sem_atten_feat
        self.stereo_atten_feat = stereo_atten_feat
        self.cat_img_feature = bool(cat_img_feature)
        cin = cv_channels + (in_sem_channels if self.cat_img_feature else 0)
        self.voxel_convs = nn.Sequential(*[
            nn.Sequential(_conv3(cin if i == 0 else out_channels, out_channels, norm_cfg))
            for i in range(num_3dconvs)
        ])
        self.voxel_pool = nn.AvgPool3d((4, 1, 1), stride=(4, 1, 1))
        self.coordinates_3d = None  # injected by the detector (dfm.py:99-100)
        self.depth_cfg = None       # injected by the detector (dfm.py:86)
        # extension: layout of the returned voxel volume.  None = follow the input (channels_last_3d
        # for an NDHWC cost volume).  The reference detector does ``volume_feat.view(-1, Cv * Nz, Ny,
        # Nx)`` (dfm.py:325-326), which needs the contiguous layout: enable_fast_path() sets
        # torch.contiguous_format here for models that are not DfMStereoPath (a 14 MB copy at config K).
        self.output_memory_format = None

    def init_weights(self):
        pass

    def _coords_on(self, device):
        # the detector injects a host tensor (dfm.py:99-100) and the reference uploads it every
        # forward (.cuda(), feature_transformation.py:82): 21 MB at config K; uploaded once here
        c = self.coordinates_3d
        key = (c._version, str(device))
        ref = self.__dict__.get('_coords_ref')
        if ref is None or ref() is not c or self.__dict__.get('_coords_key') != key:
            self.__dict__['_coords_dev'] = c.to(device=device, dtype=torch.float32).contiguous()
            self.__dict__['_coords_ref'], self.__dict__['_coords_key'] = weakref.ref(c), key
            note_derived_build()
        return self._coords_dev

    def forward(self, stereo_feat, stereo_feat_softmax, img_metas, cur_sem_feats=None):
        voxel = frustum_to_voxel_sample(stereo_feat, stereo_feat_softmax, img_metas,
                                        cur_sem_feats if self.cat_img_feature else None,
                                        self._coords_on(stereo_feat.device), self.depth_cfg,
                                        sem_atten_feat=self.sem_atten_feat,
                                        stereo_atten_feat=self.stereo_atten_feat)
        out = _depth_pool4(self.voxel_pool, self.voxel_convs(voxel))
        if self.output_memory_format is not None:
            out = out.contiguous(memory_format=self.output_memory_format)
        return out


# --------------------------------------------------------------------------
# voxel necks
# --------------------------------------------------------------------------
class ResModule(nn.Module):

    def __init__(self, n_channels, norm_cfg=dict(type='BN3d')):
        super().__init__()
        self.conv0 = _conv3(n_channels, n_channels, norm_cfg)
        self.conv1 = _conv3(n_channels, n_channels, norm_cfg, act=False)
        self.activation = nn.ReLU(inplace=True)

    def forward(self, x):
        if self.conv0.fusable(x) and self.conv1.fusable(x):
            # inference: conv-bn-relu and conv-bn + identity + relu are one MFMA kernel each
            res = x if x.is_contiguous(memory_format=torch.channels_last_3d) else \
                x.contiguous(memory_format=torch.channels_last_3d)  # a channel slice (DfMNeck mono stack)
            return self.conv1.forward_fused(self.conv0.forward_fused(x), residual=res, relu=True)
        # training: BatchNorm (batch statistics) + identity + ReLU are one fused pass of HipBatchNorm3d
        return self.conv1(self.conv0(x), residual=x, relu=True)


def _bev_stack(c_in, widths, out_channels, norm_cfg):
    """Res(c_in) -> conv s(1,1,2) -> Res -> conv s(1,1,2) -> Res -> conv p(1,1,0):
    collapses Nz 12 -> 6 -> 3 -> 1 (imvoxel_neck.py:26-55)."""
    c0, c1, c2 = widths
    return nn.Sequential(
        ResModule(c_in, norm_cfg=norm_cfg),
        _conv3(c_in, c1, norm_cfg, stride=(1, 1, 2)),
        ResModule(c1, norm_cfg=norm_cfg),
        _conv3(c1, c2, norm_cfg, stride=(1, 1, 2)),
        ResModule(c2, norm_cfg=norm_cfg),
        _conv3(c2, out_channels, norm_cfg, padding=(1, 1, 0)))


def _to_bev(x):
    assert x.shape[-1] == 1
    return x[..., 0].transpose(-1, -2)  # Anchor3DHead axis order (y, x)


@register_module
class OutdoorImVoxelNeck(DerivedStateMixin, nn.Module):

    def __init__(self, in_channels, out_channels, norm_cfg=dict(type='BN3d'), output_bev=True):
        super().__init__()
        self.output_bev = output_bev
        widths = in_channels if isinstance(in_channels, list) else \
            [in_channels, in_channels * 2, in_channels * 4]
        self.model = _bev_stack(widths[0], widths, out_channels, norm_cfg)

    def forward(self, x):
        assert self.output_bev
        return [_to_bev(self.model(x))]

    def init_weights(self):
        pass


@register_module
class DfMNeck(DerivedStateMixin, nn.Module):

    def __init__(self, in_channels, out_channels, norm_cfg=dict(type='BN3d'), num_frames=2):
        super().__init__()
        widths = in_channels if isinstance(in_channels, list) else \
            [in_channels, in_channels * 2, in_channels * 4]
        self.in_channels = widths
        self.num_frames = num_frames
        self.mono_layers = _bev_stack(widths[0], widths, out_channels, norm_cfg)
        self.stereo_layers = _bev_stack(widths[0] * num_frames, widths, out_channels, norm_cfg)
        self.aggregate_layer = nn.Conv2d(2 * out_channels, 1, kernel_size=1, bias=False)

    two_streams = True

    def forward(self, x):
        assert x.shape[1] == self.in_channels[0] * self.num_frames
        # the mono and the stereo stacks (dfm_neck.py:108-111) are independent: two HIP streams at inference, like
        # DfMBackbone's branches (the stacks' launches are 4.2 rounds of workgroups each: one fills the other's
        # last round); two_streams = False pins one stream
        if (self.two_streams and x.is_cuda and not torch.is_grad_enabled() and
                not torch.cuda.is_current_stream_capturing()):
            main = torch.cuda.current_stream(x.device)
            side = DfMBackbone._side_streams.get(x.device)
            if side is None:
                side = DfMBackbone._side_streams[x.device] = torch.cuda.Stream(device=x.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                mono = _to_bev(self.mono_layers(channel_slice(x, 0, self.in_channels[0])))
            stereo = _to_bev(self.stereo_layers(x))
            main.wait_stream(side)
            mono.record_stream(main)
        else:
            x_all, x_mono = channel_split(x, 0, self.in_channels[0])   # (one backward node for the two uses of x)
            mono = _to_bev(self.mono_layers(x_mono))
            stereo = _to_bev(self.stereo_layers(x_all))
        # 1x1 Conv2d(2 C_out -> 1): MIOpen's kernel for this shape is a 58 ms naive convolution in
        # bf16 (profiles/archive/r02_c26_*); it is a weighted channel sum of the two maps
        w = self.aggregate_layer.weight.view(2, -1, 1, 1).to(mono.dtype)
        gate = ((mono * w[0]).sum(1, keepdim=True) + (stereo * w[1]).sum(1, keepdim=True)).sigmoid() \
            if mono.is_cuda and mono.dtype == torch.bfloat16 else \
            self.aggregate_layer(torch.cat([mono, stereo], dim=1)).sigmoid()
        return [gate * mono + (1 - gate) * stereo]

    def init_weights(self):
        pass


# --------------------------------------------------------------------------
# producers / consumers either side of the path: ordinary 2-D conv
# stacks with the reference's constructor arguments and state_dict keys.  GroupNorm runs in
# the fused HIP kernel; the 2-D convolutions are MIOpen through torch.
#   convbn / upconv_module   mmdet3d/models/utils/conv_modules.py:6-24,46-70
#   hourglass2d, BEVHourglass mmdet3d/models/backbones/bev_hourglass.py
#   SPPUNetNeck              mmdet3d/models/necks/spp_unet_neck.py
# --------------------------------------------------------------------------
def convbn(in_planes, out_planes, kernel_size, stride, pad, dilation=1, gn=False, groups=32):
    # (MfmaConv2d is an nn.Conv2d: same parameters and keys; its MFMA path covers kernel 3 / padding 1)
    return nn.Sequential(
        MfmaConv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride,
                   padding=dilation if dilation > 1 else pad, dilation=dilation, bias=False),
        nn.SyncBatchNorm(out_planes) if not gn else HipGroupNorm(groups, out_planes))


def _conv_norm_2d(seq, x, residual=None, relu=False):
    """Sequential(conv, norm) of the 2-D necks (+ residual) (+ ReLU): an eval-mode BatchNorm, the
    residual add and the ReLU fold into the MFMA convolution's epilogue (one launch); otherwise conv,
    norm and plain torch ops"""
    conv, norm = seq[0], seq[1]
    if (isinstance(conv, MfmaConv2d) and isinstance(norm, nn.modules.batchnorm._BatchNorm) and
            not norm.training and norm.track_running_stats and norm.affine and conv.eligible(x) and
            (residual is None or (residual.dtype == x.dtype and
                                  residual.is_contiguous(memory_format=torch.channels_last)))):
        key = tuple((t._version, t.data_ptr()) for t in (norm.weight, norm.bias, norm.running_mean, norm.running_var))
        if seq.__dict__.get('_fold_key') != key:
            scale = norm.weight.float() / torch.sqrt(norm.running_var.float() + norm.eps)
            seq.__dict__['_fold'] = (scale, norm.bias.float() - norm.running_mean.float() * scale)
            seq.__dict__['_fold_key'] = key
            note_derived_build()
        scale, shift = seq.__dict__['_fold']
        return conv.forward_fused(x, scale, shift, residual=residual, relu=relu)
    y = conv(x)
    if isinstance(norm, nn.modules.batchnorm._BatchNorm) and _BN2D_FUSED:
        # training: the batch statistics, the normalisation, the residual and the ReLU in the fused GroupNorm kernels on
        # the NHWC map as it lies (round 6; MIOpen's training BatchNorm + separate add / ReLU passes were 0.9 ms of
        # the DfMStereoPath step) -- None when it does not apply (eval mode, NCHW, SyncBatchNorm across ranks)
        out = batch_norm_train_channels_last(norm, y, relu=relu, residual=residual)
        if out is not None:
            return out
    y = norm(y)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


_BN2D_FUSED = os.environ.get('DFM_BN2D_TORCH') != '1'   # (A/B runs: torch's training BatchNorm in the 2-D necks)
_INTERP_MATRICES = {}
_BILINEAR_GATHER = os.environ.get('DFM_BILINEAR_MATMUL') != '1'   # (A/B runs: the matrix-product backward)


def _interp_matrix(n_in, n_out, align_corners, scale, device):
    """(n_out, n_in) fp32 matrix of 1-D bilinear interpolation, taken from ATen itself (its forward on
    the identity), so it carries exactly the weights F.interpolate applies"""
    key = (n_in, n_out, bool(align_corners), scale, str(device))
    m = _INTERP_MATRICES.get(key)
    if m is None:
        eye = torch.eye(n_in, dtype=torch.float32, device=device).view(1, n_in, 1, n_in)
        kw = dict(scale_factor=(1.0, scale)) if scale is not None else dict(size=(1, n_out))
        m = F.interpolate(eye, mode='bilinear', align_corners=align_corners, **kw).view(n_in, n_out).t().contiguous()
        if len(_INTERP_MATRICES) > 256:
            _INTERP_MATRICES.clear()
        _INTERP_MATRICES[key] = m
    return m


_INTERP_TABLES = {}


def _interp_table(n_in, n_out, align_corners, scale, device):
    """the non-zeros of the transposed interpolation matrix, padded: (idx (n_in, K) int32, w (n_in, K) fp32, K) --
    for input index i the outputs that interpolate from it and their weights (dfm_bilinear_resize_bwd_nhwc)"""
    key = (n_in, n_out, bool(align_corners), scale, str(device))
    t = _INTERP_TABLES.get(key)
    if t is None:
        m = _interp_matrix(n_in, n_out, align_corners, scale, device).t().cpu().numpy()   # (n_in, n_out)
        nz = m != 0
        K = max(1, int(nz.sum(1).max()))
        idx = np.zeros((n_in, K), np.int32)
        w = np.zeros((n_in, K), np.float32)
        for i in range(n_in):
            j = np.nonzero(nz[i])[0]
            idx[i, :len(j)] = j
            w[i, :len(j)] = m[i, j]
        t = (torch.from_numpy(idx).to(device), torch.from_numpy(w).to(device), K)
        if len(_INTERP_TABLES) > 256:
            _INTERP_TABLES.clear()
        _INTERP_TABLES[key] = t
        note_derived_build()
    return t


class _BilinearResizeFn(torch.autograd.Function):
    """F.interpolate(mode='bilinear') whose BACKWARD is two small matrix products, gX = A_h^T gY A_w
    (bilinear resampling is separable and linear).  ATen's upsample_bilinear2d_backward scatters with
    atomics: 0.67 ms per call on the necks' maps, 12 calls = 8 of the 41 ms of a DfMStereoPath training
    step at config K (profiles/archive/r04_c8_*); the SPP branches' few-pixel maps serialise on a handful of
    addresses.  The forward is ATen's own (bit-identical to the reference's)."""

    @staticmethod
    def forward(ctx, x, size, scale, align_corners):
        ctx.cfg = (tuple(x.shape[2:]), scale, align_corners)
        kw = dict(scale_factor=scale) if scale is not None else dict(size=size)
        return F.interpolate(x, mode='bilinear', align_corners=align_corners, **kw)

    @staticmethod
    def backward(ctx, gy):
        (h_in, w_in), scale, ac = ctx.cfg
        B, C, h_out, w_out = gy.shape
        a_w = _interp_matrix(w_in, w_out, ac, scale, gy.device)            # (w_out, w_in)
        a_h = _interp_matrix(h_in, h_out, ac, scale, gy.device)            # (h_out, h_in)
        vec = 16 // gy.element_size()
        if (gy.stride(1) == 1 and C % vec == 0 and gy.is_contiguous(memory_format=torch.channels_last) and
                gy.dtype in (torch.float32, torch.bfloat16) and _BILINEAR_GATHER):
            # NHWC: the gather kernel (csrc/bilinear_bwd.hip) -- a lane per input pixel and 16-byte channel vector walks
            # the few output pixels that interpolate from it (an up-sampling by 2 has at most 4 x 4)
            ri, rw, kh = _interp_table(h_in, h_out, ac, scale, gy.device)
            ci, cw, kw = _interp_table(w_in, w_out, ac, scale, gy.device)
            if kh * kw <= 64 and gy.data_ptr() % 16 == 0:
                gx = torch.empty((B, h_in, w_in, C), dtype=gy.dtype, device=gy.device)
                with torch.cuda.device(gy.device):
                    _capi.check(_capi.lib().dfm_bilinear_resize_bwd_nhwc(
                        B, C, h_in, w_in, h_out, w_out, _capi.DFM_BF16 if gy.dtype == torch.bfloat16 else _capi.DFM_F32,
                        gy.data_ptr(), ri.data_ptr(), rw.data_ptr(), kh, ci.data_ptr(), cw.data_ptr(), kw,
                        gx.data_ptr(), torch.cuda.current_stream(gy.device).cuda_stream))
                return gx.permute(0, 3, 1, 2), None, None, None
        if gy.stride(1) == 1 and C > 1 and gy.is_contiguous(memory_format=torch.channels_last):
            # an NHWC gradient (the 2-D necks train channels-last): the same two products on the memory as it lies --
            # rows of C channels ride along as the matrices' columns, gX[b, hi, wi, :] = sum A_h[ho, hi] A_w[wo, wi]
            # gY[b, ho, wo, :] -- and an NHWC result.  (Round 5 reshaped to (B C, h, w): a transposing copy of the
            # gradient, a second one inside bmm and a third of the result: 0.47 ms of copies per training step.)
            g = gy.permute(0, 2, 3, 1).float()                                          # (B, h_out, w_out, C), no move
            g = torch.matmul(a_w.t(), g.view(B * h_out, w_out, C))                      # (B h_out, w_in, C)
            g = torch.matmul(a_h.t(), g.view(B, h_out, w_in * C))                       # (B, h_in, w_in C)
            return g.view(B, h_in, w_in, C).to(gy.dtype).permute(0, 3, 1, 2), None, None, None
        g = gy.reshape(B * C, h_out, w_out).float()
        g = torch.matmul(g, a_w)                                           # (BC, h_out, w_in)
        g = torch.matmul(a_h.t(), g)                                       # (BC, h_in, w_in)
        return g.view(B, C, h_in, w_in).to(gy.dtype), None, None, None


def bilinear_resize(x, size=None, scale_factor=None, align_corners=False):
    """``F.interpolate(x, size / scale_factor, mode='bilinear', align_corners=...)``; on the GPU with autograd
    recording the backward runs as matrix products (``_BilinearResizeFn``)"""
    if x.is_cuda and torch.is_grad_enabled() and x.requires_grad:
        return _BilinearResizeFn.apply(x, tuple(size) if size is not None else None, scale_factor, bool(align_corners))
    kw = dict(scale_factor=scale_factor) if scale_factor is not None else dict(size=size)
    return F.interpolate(x, mode='bilinear', align_corners=align_corners, **kw)


class upconv_module(nn.Module):  # noqa: N801  (reference class name)

    def __init__(self, in_channels, up_channels):
        super().__init__()
        self.num_stage = len(in_channels) - 1
        self.conv = nn.ModuleList(
            convbn(in_channels[0] if i == 0 else up_channels[i - 1], up_channels[i], 3, 1, 1, 1)
            for i in range(self.num_stage))
        self.redir = nn.ModuleList(
            convbn(in_channels[i + 1], up_channels[i], 3, 1, 1, 1) for i in range(self.num_stage))
        self.up = nn.Upsample(scale_factor=2, mode='bilinear')

    def forward(self, feats):
        x = feats[0]
        for i in range(self.num_stage):
            # relu(up(conv(x)) + redir(skip)): the add and the ReLU ride in redir's convolution epilogue
            up = bilinear_resize(_conv_norm_2d(self.conv[i], x), scale_factor=self.up.scale_factor,
                                 align_corners=bool(self.up.align_corners))
            x = _conv_norm_2d(self.redir[i], feats[i + 1], residual=up, relu=True)
        return x


def _channels_last_2d(module, feats):
    """The 2-D producers / consumers either side of the path (SPPUNetNeck,
    BEVHourglass) run NHWC on the GPU at INFERENCE: the MFMA convolution kernel and the fused SPP tail
    read NHWC, MIOpen's own NHWC kernels wrap an NCHW call in two transposes each (profiles/archive/r02_c29: 588
    batched_transpose launches per DfMStereoPath forward), and the stereo / semantic maps they emit are
    then already in the pixel-major layout the plane sweep and FrustumToVoxel sample (no pack pass).
    The 4-D weights are re-laid once; shapes, values and state_dict keys are untouched.

    Training (autograd recording) runs NHWC as well since round 4: the 3x3 convolutions train through the
    MFMA kernels (conv3d._Conv2dGFn: forward, backward-data, backward-weight), the 1x1 convolutions of an
    NHWC tensor are matrix products, GroupNorm is the HIP kernel in either layout and the bilinear resizes
    have a matrix-product backward (bilinear_resize) -- the three things that made torch's NHWC training
    path 2.1 s per DfMStereoPath step in round 3 (MIOpen's naive NHWC convolutions, ATen's channels-last
    bilinear backward; profiles/archive/r03_c43_*).  ``DFM_TRAIN_NCHW=1`` (or ``module.train_nhwc = False``) keeps the
    round-3 behaviour: NCHW between the layers, each MFMA convolution converting its operands."""
    if not feats[0].is_cuda:
        return feats
    train = torch.is_grad_enabled() and (module.training or any(f.requires_grad for f in feats))
    # (an fp32 model's 3x3 convolutions are torch's: keep them away from MIOpen's NHWC kernels)
    nhwc = not train or (feats[0].dtype == torch.bfloat16 and
                         module.__dict__.get('train_nhwc', os.environ.get('DFM_TRAIN_NCHW') != '1'))
    want = torch.channels_last if nhwc else torch.contiguous_format
    if module.__dict__.get('_weights_format') is not want:
        module.to(memory_format=want)
        if nhwc:
            # ... except the weights of the convolutions that run in the MFMA kernels: those are packed from the
            # parameter's logical (cout, cin, kh, kw) view, and a channels-last PARAMETER costs a strided copy wherever
            # it is made contiguous (each pack, forward and backward) and another when autograd lays the incoming
            # gradient out like the parameter (80 small launches per training step of the stereo path, round 6)
            for m in module.modules():
                if isinstance(m, (MfmaConv2d, MfmaConvTranspose2d)) and m.weight.dim() == 4 and \
                        MfmaConv2d.covers(m.weight.shape[1] if isinstance(m, MfmaConv2d) else m.weight.shape[0],
                                          m.weight.shape[0] if isinstance(m, MfmaConv2d) else m.weight.shape[1],
                                          tuple(m.weight.shape[2:]), m.stride, m.padding):
                    m.weight.data = m.weight.data.contiguous()
        module.__dict__['_weights_format'] = want
        module.__dict__['_weights_channels_last'] = nhwc
    if not nhwc:
        return [f.contiguous() for f in feats]
    return [f.contiguous(memory_format=torch.channels_last) for f in feats]


def _gn_relu(seq, x, relu, residual=None):
    """conv -> norm (+residual) (+ReLU), one pass when the norm is the HIP GroupNorm / BatchNorm3d"""
    if isinstance(seq[1], (HipGroupNorm, HipBatchNorm3d)):
        return seq[1](seq[0](x), relu=relu, residual=residual)
    x = _conv_norm_2d(seq, x) if isinstance(seq[0], MfmaConv2d) else seq[1](seq[0](x))
    if residual is not None:
        x = x + residual
    return F.relu(x) if relu else x


class hourglass2d(nn.Module):  # noqa: N801  (reference class name)

    def __init__(self, inplanes, gn=False):
        super().__init__()
        c = inplanes
        self.conv1 = nn.Sequential(convbn(c, 2 * c, 3, 2, 1, 1, gn=gn), nn.ReLU(inplace=True))
        self.conv2 = convbn(2 * c, 2 * c, 3, 1, 1, 1, gn=gn)
        self.conv3 = nn.Sequential(convbn(2 * c, 2 * c, 3, 2, 1, 1, gn=gn), nn.ReLU(inplace=True))
        self.conv4 = nn.Sequential(convbn(2 * c, 2 * c, 3, 1, 1, 1, gn=gn), nn.ReLU(inplace=True))

        def up(cin, cout):
            return nn.Sequential(
                MfmaConvTranspose2d(cin, cout, 3, padding=1, output_padding=1, stride=2, bias=False),
                nn.SyncBatchNorm(cout) if not gn else HipGroupNorm(32, cout))
        self.conv5 = up(2 * c, 2 * c)
        self.conv6 = up(2 * c, c)

    def forward(self, x, presqu, postsqu):
        out = _gn_relu(self.conv1[0], x, True)
        pre = _gn_relu(self.conv2, out, False)
        pre = F.relu(pre if postsqu is None else pre + postsqu)
        out = _gn_relu(self.conv4[0], _gn_relu(self.conv3[0], pre, True), True)
        post = _gn_relu(self.conv5, out, True, pre if presqu is None else presqu)
        return _gn_relu(self.conv6, post, False), pre, post


@register_module
class BEVHourglass(DerivedStateMixin, nn.Module):

    def __init__(self, in_channels, out_channels, norm_cfg=None, output_prehg_feat=True,
                 init_cfg=None):
        super().__init__()
        self.out_channels = out_channels
        self.norm_cfg = norm_cfg
        self.output_prehg_feat = output_prehg_feat
        self.compress_conv = ConvModule(in_channels, out_channels, 3, stride=1, padding=1,
                                        norm_cfg=norm_cfg)
        self.bev_hourglass = hourglass2d(out_channels, gn=(norm_cfg['type'] == 'GN'))
        self.num_bev_features = out_channels

    def init_weights(self):
        pass

    def forward(self, spatial_features):
        spatial_features, = _channels_last_2d(self, [spatial_features])
        prehg = self.compress_conv(spatial_features)
        x = self.bev_hourglass(prehg, None, None)[0]
        return (prehg, x) if self.output_prehg_feat else x


@register_module
class SPPUNetNeck(DerivedStateMixin, nn.Module):

    def __init__(self, in_channels, start_level, sem_channels=[128, 32], stereo_channels=[32, 32],
                 spp_channel=32, with_upconv=True, cat_img_feature=True, norm_cfg=None,
                 init_cfg=None):
        super().__init__()
        self.in_channels = in_channels
        self.start_level = start_level
        self.sem_channels = sem_channels
        self.stereo_channels = stereo_channels
        self.spp_channel = spp_channel
        self.with_upconv = with_upconv
        self.cat_img_feature = cat_img_feature
        self.spp_branches = nn.ModuleList(
            nn.Sequential(_WindowMean2d(s, stride=s),
                          ConvModule(in_channels[-1], spp_channel, 1, stride=1, padding=0,
                                     norm_cfg=norm_cfg))
            for s in [(64, 64), (32, 32), (16, 16), (8, 8)])
        concat_channel = spp_channel * len(self.spp_branches) + sum(in_channels[start_level:])
        if with_upconv:
            assert start_level == 2
            self.upconv_module = upconv_module([concat_channel, in_channels[1], in_channels[0]],
                                               [64, 32])
            stereo_channel = 32
        else:
            stereo_channel = concat_channel
            assert start_level >= 1
        self.lastconv = nn.Sequential(
            ConvModule(stereo_channel, stereo_channels[0], 3, stride=1, padding=1, norm_cfg=norm_cfg),
            MfmaConv2d(stereo_channels[0], stereo_channels[1], kernel_size=1, padding=0, stride=1,
                       bias=False))
        if cat_img_feature:
            self.rpnconv = nn.Sequential(
                ConvModule(concat_channel, sem_channels[0], 3, stride=1, padding=1, norm_cfg=norm_cfg),
                ConvModule(sem_channels[0], sem_channels[1], 3, stride=1, padding=1,
                           norm_cfg=norm_cfg))

    def init_weights(self):
        pass

    def _spp_pool(self, x):
        """the four non-overlapping window means of the SPP branches (spp_unet_neck.py:60-70).  On the GPU
        the map is read ONCE: the finest windows (8 x 8) are averaged in fp32 and the coarser means
        (16, 32, 64: whole multiples, floor mode) are means of those -- instead of four passes over the
        128-channel map, each through an fp32 copy."""
        pools = [b[0] for b in self.spp_branches]
        ks = [p.kernel_size if isinstance(p.kernel_size, tuple) else (p.kernel_size,) * 2 for p in pools]
        fine = min(ks)
        ok = (x.is_cuda and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last) and
              all(isinstance(p, _WindowMean2d) and tuple(p.stride if isinstance(p.stride, tuple) else (p.stride,) * 2) == k
                  and p.padding in (0, (0, 0)) and not p.ceil_mode and k[0] % fine[0] == 0 and k[1] % fine[1] == 0
                  for p, k in zip(pools, ks)))
        if not ok:
            return [p(x) for p in pools]
        B, C, H, W = x.shape
        hf, wf = H // fine[0], W // fine[1]
        base = x.permute(0, 2, 3, 1)[:, :hf * fine[0], :wf * fine[1]].reshape(B, hf, fine[0], wf, fine[1], C)
        base = base.mean(dim=(2, 4), dtype=torch.float32)                       # (B, hf, wf, C) fp32
        out = []
        for k in ks:
            rh, rw = k[0] // fine[0], k[1] // fine[1]
            ho, wo = H // k[0], W // k[1]
            v = base if (rh, rw) == (1, 1) else \
                base[:, :ho * rh, :wo * rw].reshape(B, ho, rh, wo, rw, C).mean(dim=(2, 4))
            out.append(v.to(x.dtype).permute(0, 3, 1, 2))                       # channels_last (B, C, ho, wo)
        return out

    def _spp_tail_fused(self, feats):
        """inference, bf16 NHWC on the GPU: the branches' 1x1 ConvModule (GroupNorm with one channel per
        group, ReLU), their bilinear up-sampling and the concatenation in two launches
        (``dfm_spp_tail_fwd``, csrc/spp_tail.hip) instead of ~25; None when the call does not qualify."""
        srcs = feats[self.start_level:]
        x = feats[-1]
        cms = [b[1] for b in self.spp_branches]
        ok = (x.is_cuda and x.dtype == torch.bfloat16 and not torch.is_grad_enabled() and
              1 <= len(cms) <= 4 and 1 <= len(srcs) <= 4 and
              all(t.dtype == x.dtype and t.shape[1] % 8 == 0 and t.shape[2:] == srcs[0].shape[2:] and
                  not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last) for t in srcs) and
              all(isinstance(m.conv, nn.Conv2d) and m.conv.kernel_size == (1, 1) and m.conv.bias is None and
                  m.conv.stride == (1, 1) and m.conv.padding == (0, 0) and m.conv.groups == 1 and
                  isinstance(getattr(m, m.norm_name or '', None), HipGroupNorm) and
                  getattr(m, m.norm_name).num_groups == m.conv.out_channels and m.activate is not None and
                  m.conv.out_channels == self.spp_channel and m.conv.out_channels % 8 == 0 and
                  m.conv.out_channels <= 64 for m in cms))
        if not ok:
            return None
        pooled = [p.permute(0, 2, 3, 1).contiguous() for p in self._spp_pool(x)]     # (B, ho, wo, C) tiny
        if ((x.shape[1] * self.spp_channel + 32 * x.shape[1]) * 4 +
                max(p.shape[1] * p.shape[2] for p in pooled) * self.spp_channel * 2 > 62 * 1024 or
                256 % self.spp_channel):
            return None
        key = tuple((m.conv.weight._version, m.conv.weight.data_ptr()) for m in cms) + \
            tuple((getattr(m, m.norm_name).weight._version, getattr(m, m.norm_name).bias._version) for m in cms)
        if self.__dict__.get('_spp_key') != key:
            self.__dict__['_spp_params'] = [
                (m.conv.weight.detach().float().reshape(m.conv.out_channels, -1).contiguous(),
                 getattr(m, m.norm_name).weight.detach().float().contiguous(),
                 getattr(m, m.norm_name).bias.detach().float().contiguous()) for m in cms]
            self.__dict__['_spp_key'] = key
            note_derived_build()
        params = self.__dict__['_spp_params']
        B, _, H, W = srcs[0].shape
        d = _capi.SppDesc()
        d.batch, d.h, d.w = B, H, W
        d.num_sources, d.num_branches = len(srcs), len(cms)
        d.in_channels, d.spp_channels = x.shape[1], self.spp_channel
        d.eps = float(getattr(cms[0], cms[0].norm_name).eps)
        for i, t in enumerate(srcs):
            d.source_channels[i] = t.shape[1]
        for i, p in enumerate(pooled):
            d.pooled_h[i], d.pooled_w[i] = p.shape[1], p.shape[2]
        ctot = sum(t.shape[1] for t in srcs) + len(cms) * self.spp_channel
        out = torch.empty((B, H, W, ctot), dtype=x.dtype, device=x.device)
        lib = _capi.lib()
        nbytes = lib.dfm_spp_tail_workspace_bytes(ctypes.byref(d))
        ws = _Workspace.get(x.device, nbytes)

        def arr(ts):
            return (ctypes.c_void_p * 4)(*[t.data_ptr() for t in ts], *([None] * (4 - len(ts))))
        with torch.cuda.device(x.device):
            _capi.check(lib.dfm_spp_tail_fwd(
                ctypes.byref(d), arr(pooled), arr([p[0] for p in params]), arr([p[1] for p in params]),
                arr([p[2] for p in params]), arr(list(srcs)), out.data_ptr(), ws.data_ptr(), nbytes,
                torch.cuda.current_stream(x.device).cuda_stream))
        return out.permute(0, 3, 1, 2)

    def forward(self, feats):
        feat_shape = tuple(feats[self.start_level].shape[2:])
        assert len(feats) == len(self.in_channels)
        feats = _channels_last_2d(self, list(feats))
        concat_feature = self._spp_tail_fused(feats)
        if concat_feature is None:
            spp = [bilinear_resize(branch[1](pooled), size=feat_shape, align_corners=True)
                   for branch, pooled in zip(self.spp_branches, self._spp_pool(feats[-1]))]
            concat_feature = torch.cat((*feats[self.start_level:], *spp), 1)
        stereo_feature = concat_feature
        if self.with_upconv:
            stereo_feature = self.upconv_module([stereo_feature, feats[1], feats[0]])
        stereo_feature = self.lastconv(stereo_feature)
        sem_feature = self.rpnconv(concat_feature) if self.cat_img_feature else None
        return stereo_feature, sem_feature
