"""Detector-level drop-in of the hot path.

The reference's detectors wire the path's modules together and inject attributes into them:

  DfM.__init__                mmdet3d/models/detectors/dfm.py:30-112 (attribute injection :82-100)
  DfM.extract_feat / forward  dfm.py:268-356
  MultiViewDfM.feature_transformation   mmdet3d/models/detectors/multiview_dfm.py:119-268

Three ways to use this package from there, all routed to the HIP kernels:

1. ``patch_reference()`` -- inside a real mmdet3d installation: re-registers the module
   classes (``DfMBackbone``, ``FrustumToVoxel``, ``DepthHead``, ``OutdoorImVoxelNeck``,
   ``DfMNeck``, ``BEVHourglass``, ``SPPUNetNeck``) under the same ``type`` names (force=True) so
   ``configs/dfm/*`` build them unchanged, rebinds the functions the reference modules call
   (``build_dfm_cost``, ``point_sample``, ``voxel_sample``) and replaces
   ``MultiViewDfM.feature_transformation`` by ``MultiViewDfMMixin.feature_transformation``.
2. ``DfMStereoPath`` -- the KITTI student's path (neck -> backbone_stereo -> depth_head ->
   feature_transformation -> height compression -> backbone_3d) built from the ``model`` dict of
   ``configs/dfm/dfm_r34_1x8_kitti-3d-3class.py`` with the detector's attribute injection;
   usable without mmdet3d (this is what the tests and tools run).
3. ``MultiViewDfMMixin`` -- for a subclass ``class MultiViewDfM(MultiViewDfMMixin, RefMultiViewDfM)``.
"""
import importlib

import os

import numpy as np
import torch
from torch import nn

from . import registry
from .geometry import prepare_coordinates_3d, prepare_depth
from .graphs import GraphedCallable
from .plane_sweep import build_dfm_cost
from .point_sample import mv_feature_transformation, point_sample, voxel_centers, voxel_sample


def inject_detector_attributes(detector, depth_cfg=None, voxel_cfg=None):
    """What DfM.__init__ writes onto its sub-modules after building them (dfm.py:82-100):
    ``backbone_stereo.downsampled_depth``, ``depth_head.depth_samples`` / ``.downsample_factor``,
    ``feature_transformation.depth_cfg`` / ``.coordinates_3d``.  ``detector`` is any object with
    those sub-modules as attributes (missing ones are skipped)."""
    if depth_cfg is not None:
        ds_factor = depth_cfg['downsample_factor']
        downsampled, depth = prepare_depth(depth_cfg)
        detector.downsampled_depth, detector.depth = downsampled, depth
        detector.depth_downsample_factor = ds_factor
        ft = getattr(detector, 'feature_transformation', None)
        if ft is not None:
            ft.depth_cfg = depth_cfg
        if getattr(detector, 'depth_head', None) is not None:
            detector.backbone_stereo.downsampled_depth = downsampled
            detector.depth_head.depth_samples = depth
            detector.depth_head.downsample_factor = ds_factor
    if voxel_cfg is not None:
        coords = prepare_coordinates_3d(voxel_cfg)
        detector.coordinates_3d = coords
        ft = getattr(detector, 'feature_transformation', None)
        if ft is not None:
            ft.coordinates_3d = coords
    return detector


def bev_view(vol):
    """``volume_feat.reshape(-1, C * Nz, Ny, Nx)`` (detectors/dfm.py: the voxel volume seen from above,
    height folded into the channels).  On the GPU the copy this reshape needs anyway (the volume is
    channels_last_3d) lands directly in NHWC, the layout BEVHourglass's 2-D convolutions run in."""
    B, C, nz, ny, nx = vol.shape
    if not vol.is_cuda:
        return vol.reshape(B, C * nz, ny, nx)
    return vol.permute(0, 3, 4, 1, 2).reshape(B, ny, nx, C * nz).permute(0, 3, 1, 2)


class DfMStereoPath(nn.Module):
    """The plane-sweep path of the ``DfM`` detector, built from its config ``model`` dict.

    ``forward(cur_feats, prev_feats, img_metas)`` takes the two image pyramids
    ``[img, *backbone(img)]`` (what dfm.py:281-284 hands to the neck) and returns what
    ``DfM.forward_train`` computes up to the detection head (dfm.py:268-322):
    ``dict(mono_stereo_costs, stereo_feats, mono_feats, upsample_costs, upsample_costs_softmax,
    depth_preds, volume_feat, bev_feat_prehg, bev_feat)``.  ``loss_dense_depth(out, depth_img,
    depth_fgmask_img)`` is dfm.py:348-356.
    """

    def __init__(self, model_cfg):
        super().__init__()
        cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in model_cfg.items()}
        depth_cfg, voxel_cfg = cfg.get('depth_cfg'), cfg.get('voxel_cfg')
        self.neck = registry.build_neck(cfg['neck'])
        bs = cfg['backbone_stereo']
        if depth_cfg is not None:
            bs.update(depth_cfg=depth_cfg)                       # dfm.py:45-47
        self.backbone_stereo = registry.build_backbone(bs)
        ft = cfg.get('feature_transformation')
        if ft is not None:
            ft.update(cat_img_feature=self.neck.cat_img_feature,     # dfm.py:56-64
                      in_sem_channels=self.neck.sem_channels[-1])
            self.feature_transformation = registry.build_neck(ft)
        self.depth_head = registry.build_head(cfg['depth_head']) if cfg.get('depth_head') else None
        self.backbone_3d = registry.build_backbone(cfg['backbone_3d']) if cfg.get('backbone_3d') else None
        inject_detector_attributes(self, depth_cfg, voxel_cfg)
        # inference-time DepthHead -> FrustumToVoxel fusion; set False to get
        # the materialised upsample_costs / upsample_costs_softmax in eval mode as well
        self.fuse_depth_head = True
        # inference: replay the launch-bound 2-D necks (SPPUNetNeck, BEVHourglass) as hipGraphs
        # (graphs.py); opt-in -- the graphs keep static input / output buffers per input signature
        self.hip_graphs = False
        self._graphed = {}

    two_streams = os.environ.get('DFM_PATH_ONE_STREAM') != '1'

    def _run_2d(self, name, module, tensors):
        """module(tensors) -- through a captured hipGraph when ``hip_graphs`` is on and nothing records
        gradients.  Each call site has its own graph (the outputs are static buffers: the neck's
        results for the current frame must survive its call on the previous frame)."""
        if not (self.hip_graphs and not torch.is_grad_enabled() and not self.training and tensors[0].is_cuda):
            return module(tensors) if name.startswith('neck') else module(tensors[0])
        g = self._graphed.get(name)
        if g is None:
            fn = (lambda ts: module(ts)) if name.startswith('neck') else (lambda ts: module(ts[0]))
            g = self._graphed[name] = GraphedCallable(fn)
        return g(tensors)

    def forward(self, cur_feats, prev_feats, img_metas):
        # the two frames' 2-D necks are independent calls of the same module: at inference (eval mode: no running
        # statistics to update, nothing recorded) the previous frame's runs on a side HIP stream
        if (self.two_streams and cur_feats[0].is_cuda and not torch.is_grad_enabled() and not self.training and
                not torch.cuda.is_current_stream_capturing()):
            from .modules import DfMBackbone
            from .conv3d import derived_builds
            d0 = cur_feats[0].device
            main = torch.cuda.current_stream(d0)
            side = DfMBackbone._side_streams.get(d0)
            if side is None:
                side = DfMBackbone._side_streams[d0] = torch.cuda.Stream(device=d0)
            side.wait_stream(main)
            built = derived_builds()
            with torch.cuda.stream(side):
                prev_stereo, _ = self._run_2d('neck_prev', self.neck, list(prev_feats))
            if derived_builds() != built:
                # the first call after a weight load / change built the neck's derived state (packed weights,
                # folded norms) with kernels on the SIDE stream: the main-stream call finds the caches filled
                # and would read them unordered.  That one forward runs the two frames back to back.
                main.wait_stream(side)
            cur_stereo, cur_sem = self._run_2d('neck_cur', self.neck, list(cur_feats))
            main.wait_stream(side)
            prev_stereo.record_stream(main)
        else:
            cur_stereo, cur_sem = self._run_2d('neck_cur', self.neck, list(cur_feats))
            prev_stereo, _ = self._run_2d('neck_prev', self.neck, list(prev_feats))
        dev = cur_stereo.device
        for meta in img_metas:  # dfm.py:288-293: (N-1,4,4) tensors on the device
            meta['cur2prevs'] = torch.as_tensor(np.asarray(meta['cur2prevs'], dtype=np.float32)
                                                if not torch.is_tensor(meta['cur2prevs'])
                                                else meta['cur2prevs'], dtype=torch.float32).to(dev)
        costs, stereo_feats, mono_feats = self.backbone_stereo(cur_stereo, prev_stereo, img_metas)
        out = dict(mono_stereo_costs=costs, stereo_feats=stereo_feats, mono_feats=mono_feats,
                   cur_sem_feat=cur_sem)
        if self.depth_head is not None:
            # the depth head is fused into FrustumToVoxel's sampling kernel: the distribution is never
            # materialised.  Inference: upsample_costs is None.  Training: upsample_costs is the lazy
            # distribution, which DepthHead.loss (dfm.py:348-356) evaluates from the low-resolution cost;
            # the l1 losses differentiate depth_preds and keep the materialised head.
            fuse = (self.fuse_depth_head and hasattr(self, 'feature_transformation') and
                    not self.depth_head.with_convs and
                    (not torch.is_grad_enabled() or self.depth_head.depth_loss_type not in ('l1', 'purel1')))
            up, preds, soft = self.depth_head(costs, lazy=True) if fuse else \
                (lambda v, s, p: (v, p, s))(*self.depth_head(costs))
            out.update(upsample_costs=up, upsample_costs_softmax=soft, depth_preds=preds)
            if hasattr(self, 'feature_transformation'):
                vol = self.feature_transformation(stereo_feats, soft, img_metas, cur_sem)
                out['volume_feat'] = vol
                if self.backbone_3d is not None:
                    _, cv, nz, ny, nx = vol.shape
                    out['bev_feat_prehg'], out['bev_feat'] = self._run_2d('bev', self.backbone_3d, [bev_view(vol)])
        return out

    def loss_dense_depth(self, out, depth_img, depth_fgmask_img=None):
        if depth_fgmask_img is not None:
            depth_fgmask_img = depth_fgmask_img.flatten(start_dim=0, end_dim=1)
        return self.depth_head.loss(out['depth_preds'].flatten(start_dim=0, end_dim=1),
                                    out['upsample_costs'].flatten(start_dim=0, end_dim=1),
                                    depth_img.flatten(start_dim=0, end_dim=1),
                                    depth_fgmask_img=depth_fgmask_img)


class MultiViewDfMMixin:
    """``feature_transformation`` of ``MultiViewDfM`` (multiview_dfm.py:119-268) on the HIP path:
    one launch per batch lifts all (frame, view) feature maps into the voxel volume (sampling,
    valid-count reduction over views and frames, permute), then ``backbone_3d`` / ``neck_3d`` /
    the optional ``voxel_sample`` for the depth head run as in the reference.  The host class
    provides n_voxels, voxel_range, voxel_size, valid_sample, temporal_aggregate,
    transform_depth and the usual with_* properties."""

    def feature_transformation(self, batch_feats, img_metas, num_views, num_frames):
        # bf16 features: the volume is written channels-last, the layout the MFMA convolutions of
        # neck_3d read (no conversion copy); ``volume_memory_format`` on the host class overrides
        fast = getattr(self, 'fast_dtype', None)   # set by enable_fast_path(): lift in bf16 / NDHWC
        out_dtype = batch_feats.dtype
        if fast is not None and batch_feats.is_floating_point() and batch_feats.dtype != fast:
            batch_feats = batch_feats.to(fast)     # the view features, not the volume: Nv*F small maps
        fmt = getattr(self, 'volume_memory_format', None)
        if fmt is None:
            fmt = torch.channels_last_3d if batch_feats.dtype == torch.bfloat16 else torch.contiguous_format
        volume_feat = mv_feature_transformation(batch_feats, img_metas, num_views, num_frames,
                                                self.voxel_range, self.n_voxels,
                                                self.temporal_aggregate,
                                                valid_sample=getattr(self, 'valid_sample', True),
                                                memory_format=fmt)
        if getattr(self, 'with_backbone_3d', False):
            outputs = self.backbone_3d(volume_feat)
            volume_feat = outputs[0]
            if self.backbone_3d.output_bev:
                bev_feat = outputs[-1]
        batch_stereo_feats = None
        if getattr(self, 'with_depth_head', False):
            feats = []
            for b, meta in enumerate(img_metas):
                for v in range(num_views):
                    td = self.transform_depth
                    sf = meta.get('scale_factor', 1.0) if td else 1.0
                    feats.append(voxel_sample(
                        volume_feat[b][None], voxel_range=self.voxel_range, voxel_size=self.voxel_size,
                        depth_samples=torch.as_tensor(self.depth_samples, dtype=torch.float32),
                        proj_mat=torch.as_tensor(np.asarray(meta['ori_lidar2img'][v], np.float32)),
                        downsample_factor=self.depth_head.downsample_factor, img_scale_factor=sf,
                        img_crop_offset=meta.get('img_crop_offset', 0) if td else 0,
                        img_flip=meta.get('flip', False) if td else False,
                        img_pad_shape=meta['input_shape'] if td else meta['ori_shape'][:2],
                        img_shape=meta['img_shape'][v][:2], aligned=True))
            batch_stereo_feats = torch.cat(feats)
        if getattr(self, 'with_neck_3d', False):
            if getattr(self, 'with_backbone_3d', False) and self.backbone_3d.output_bev:
                volume_feat = self.neck_3d(bev_feat)[1]
            else:
                volume_feat = self.neck_3d(volume_feat)[0]
        if fast is not None and volume_feat.dtype != out_dtype:
            # the caller's dtype at the path's exit (a (B, C_out, Ny, Nx) BEV map: small)
            volume_feat = volume_feat.to(out_dtype)
            if batch_stereo_feats is not None:
                batch_stereo_feats = batch_stereo_feats.to(out_dtype)
        out = (volume_feat, )
        if batch_stereo_feats is not None:
            out += (batch_stereo_feats, )
        return out


class MultiViewVoxelPath(MultiViewDfMMixin, nn.Module):
    """The multi-view path of ``MultiViewDfM`` from its config ``model`` dict (neck_3d,
    voxel_size, anchor_generator.ranges, temporal_aggregate), without mmdet3d: voxel lifting +
    the 3-D neck.  ``forward(batch_feats (B, Nv*F, C, Hf, Wf), img_metas, num_views, num_frames)``
    -> BEV feature (B, C_out, Ny, Nx)."""

    def __init__(self, model_cfg):
        super().__init__()
        self.voxel_size = list(model_cfg['voxel_size'])
        self.voxel_range = list(model_cfg['anchor_generator']['ranges'][0])
        self.n_voxels = [round((self.voxel_range[3 + i] - self.voxel_range[i]) / self.voxel_size[i])
                         for i in range(3)]                      # multiview_dfm.py:54-61
        self.valid_sample = model_cfg.get('valid_sample', True)
        self.temporal_aggregate = model_cfg.get('temporal_aggregate', 'mean')
        self.transform_depth = model_cfg.get('transform_depth', True)
        self.neck_3d = registry.build_neck(dict(model_cfg['neck_3d']))
        self.with_neck_3d, self.with_backbone_3d, self.with_depth_head = True, False, False

    def forward(self, batch_feats, img_metas, num_views, num_frames):
        return self.feature_transformation(batch_feats, img_metas, num_views, num_frames)[0]


# the functions the reference's modules look up by name, and the files that define them
_FUNCTION_PATCHES = (
    ('mmdet3d.models.backbones.dfm_backbone', 'build_dfm_cost', build_dfm_cost),
    ('mmdet3d.models.fusion_layers.point_fusion', 'point_sample', point_sample),
    ('mmdet3d.models.fusion_layers.point_fusion', 'voxel_sample', voxel_sample),
    ('mmdet3d.models.fusion_layers', 'point_sample', point_sample),
    ('mmdet3d.models.fusion_layers', 'voxel_sample', voxel_sample),
    ('mmdet3d.models.detectors.multiview_dfm', 'point_sample', point_sample),
    ('mmdet3d.models.detectors.multiview_dfm', 'voxel_sample', voxel_sample),
    ('mmdet3d.models.detectors.imvoxelnet', 'point_sample', point_sample),
)


def patch_reference(precision=None, strict=False):
    """Route a real mmdet3d (the reference fork) to the HIP path.  Call once after
    ``import mmdet3d`` and before building the model from ``configs/dfm/*``.  Returns a report
    dict {'modules': [...], 'functions': [...], 'methods': [...]}.  Raises ImportError when
    mmdet3d is not importable (this package never needs it otherwise).

    ``precision='bf16'``: every ``DfM`` / ``MultiViewDfM`` detector built afterwards is passed through
    ``enable_fast_path(detector, torch.bfloat16, strict=strict)`` at the end of its constructor, so
    the reference's fp32 pipeline reaches the MFMA kernels without another line of user code (the
    sampling / norm kernels run either way; without this switch the Mfma* convolutions of an fp32
    model run torch's convolution and say so once).  ``precision=None`` leaves models as built."""
    if precision not in (None, 'fp32', 'bf16'):
        raise ValueError(f'precision must be None, "fp32" or "bf16", got {precision!r}')
    report = {'modules': registry.register_into_mmdet3d(), 'functions': [], 'methods': []}
    for mod_name, attr, fn in _FUNCTION_PATCHES:
        try:
            mod = importlib.import_module(mod_name)
        except ImportError:
            continue
        if hasattr(mod, attr):
            setattr(mod, attr, fn)
            report['functions'].append(f'{mod_name}.{attr}')
    try:
        det = importlib.import_module('mmdet3d.models.detectors.multiview_dfm')
        det.MultiViewDfM.feature_transformation = MultiViewDfMMixin.feature_transformation
        report['methods'].append('MultiViewDfM.feature_transformation')
    except (ImportError, AttributeError):
        pass
    try:
        # dfm.py:288-291 builds ``torch.tensor([img_meta['cur2prevs'] ...])``, which cannot take the device
        # tensors of a batch staged by data_geometry.stage_geometry: hand it host arrays (one 64-byte-per-
        # frame copy back; DfMStereoPath and the registered DfMBackbone read staged matrices in place)
        cls = importlib.import_module('mmdet3d.models.detectors.dfm').DfM
        if not getattr(cls.extract_feat, '_dfm_staged_metas', False):
            cls.extract_feat = _extract_feat_with_staged_metas(cls.extract_feat)
            report['methods'].append('DfM.extract_feat (accepts device-staged cur2prevs)')
    except (ImportError, AttributeError):
        pass
    if precision == 'bf16':
        # every module of a detector is built inside DfM.__init__ (MultiViewDfM.__init__ calls it and
        # adds plain attributes only, multiview_dfm.py:36-65): converting at its end covers both
        try:
            cls = importlib.import_module('mmdet3d.models.detectors.dfm').DfM
        except (ImportError, AttributeError):
            cls = None
        if cls is not None and not getattr(cls.__init__, '_dfm_fast', False):
            cls.__init__ = _init_then_fast_path(cls.__init__, strict)
            report['methods'].append('DfM.__init__ -> enable_fast_path(bf16)')
        elif cls is not None:  # patched before: the wrapper reads the strictness of the latest call
            _patch_state['strict'] = bool(strict)
            report['methods'].append(f'DfM.__init__ already patched (strict={bool(strict)} from now on)')
    return report


def _extract_feat_with_staged_metas(extract_feat):
    def wrapped(self, img, img_metas, *args, **kwargs):
        for meta in img_metas:
            v = meta.get('cur2prevs')
            if torch.is_tensor(v):
                meta['cur2prevs'] = v.detach().cpu().numpy()
        return extract_feat(self, img, img_metas, *args, **kwargs)
    wrapped._dfm_staged_metas = True
    wrapped.__wrapped__ = extract_feat
    return wrapped


_patch_state = {'strict': False}  # the latest patch_reference(precision='bf16', strict=...) call decides


def _init_then_fast_path(init, strict):
    _patch_state['strict'] = bool(strict)

    def __init__(self, *args, **kwargs):
        init(self, *args, **kwargs)
        self.fast_path_report = enable_fast_path(self, torch.bfloat16, strict=_patch_state['strict'])
    __init__._dfm_fast = True
    __init__.__wrapped__ = init
    return __init__


# ---------------------------------------------------------------------------------------------
# the fast path, explicitly: bf16 / channels-last conversion of the path's modules inside a model
# ---------------------------------------------------------------------------------------------
def _map_tensors(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, (list, tuple)):
        out = [_map_tensors(o, fn) for o in obj]
        return type(obj)(out) if not hasattr(obj, '_fields') else type(obj)(*out)
    return obj   # dicts are NOT entered: img_metas carry fp32 camera matrices / poses that must stay fp32


def _cast_hook(src, dst):
    """forward pre-hook: floating tensors of dtype ``src`` (or any floating dtype when src is None)
    among the positional / keyword arguments become ``dst``"""
    def cast(t):
        # feature maps and volumes only (>= 4-D): small fp32 geometry tensors pass through untouched
        if t.dim() >= 4 and t.is_floating_point() and t.dtype != dst and (src is None or t.dtype == src):
            return t.to(dst)
        return t

    def hook(module, args, kwargs):
        return _map_tensors(args, cast), {k: _map_tensors(v, cast) for k, v in kwargs.items()}
    return hook


def _is_norm(m):
    return isinstance(m, (nn.modules.batchnorm._BatchNorm, nn.GroupNorm))


def enable_fast_path(model, dtype=torch.bfloat16, strict=True, boundary_casts=True):
    """Route every convolution of the path inside ``model`` to the hand-written MFMA kernels.

    ``model``: a reference ``DfM`` / ``MultiViewDfM`` detector built after ``patch_reference()``, a
    ``DfMStereoPath`` / ``MultiViewVoxelPath``, or any ``nn.Module`` that contains this package's
    registered classes (the *path roots*: SPPUNetNeck, DfMBackbone, DepthHead, FrustumToVoxel,
    BEVHourglass, OutdoorImVoxelNeck, DfMNeck).  The reference pipeline runs fp32 / NCHW; the MFMA
    kernels take bf16 / channels-last.  This call

    * converts the convolution (and every other non-norm) parameter of the path roots to ``dtype``;
      GroupNorm / BatchNorm parameters and running statistics stay fp32 (the fused norm kernels and
      the folded epilogues read fp32);
    * sets ``DfMBackbone.volume_memory_format = channels_last_3d`` (the cost volume is built NDHWC)
      and marks a multi-view detector so its lifted voxel volume is written bf16 / NDHWC;
    * ``boundary_casts``: installs forward pre-hooks -- a path root casts floating inputs to ``dtype``;
      every OTHER child of a module that owns a path root (the 2-D backbone, the detection heads)
      casts ``dtype`` inputs back to the model's original floating dtype -- so the caller keeps feeding
      and receiving the tensors it did before;
    * ``strict`` (default True): every Mfma* module INSIDE ``model`` gets ``fallback_policy = 'raise'`` -- an
      input its kernel does not take (wrong dtype / layout, a shape no tiling fits) is an error there instead
      of a one-time warning and torch's convolution.  Per model: other models in the process keep the
      process-wide mode (conv3d.set_fallback_policy; 'warn' by default).  ``strict=False`` leaves the
      modules' policy as it is.

    Returns a report dict(roots=[names], converted_parameters=n, cast_back=[names], dtype=...).
    Idempotent.  ``state_dict`` keys are unchanged; ``load_state_dict`` of an fp32 checkpoint casts
    on copy as usual."""
    from . import modules as _m
    path_classes = tuple(registry.registered().values())
    names = dict((m, n) for n, m in model.named_modules())
    roots, inside = [], set()
    for n, m in model.named_modules():
        if m in inside:
            continue
        if isinstance(m, path_classes):
            roots.append(m)
            inside.update(m.modules())
    orig = next((p.dtype for p in model.parameters() if p.is_floating_point() and p.dtype != dtype),
                torch.float32)
    converted = 0
    for root in roots:
        for sub in root.modules():
            if _is_norm(sub):
                continue
            for p in sub.parameters(recurse=False):
                if p.is_floating_point() and p.dtype != dtype:
                    p.data = p.data.to(dtype)
                    if p.grad is not None:
                        p.grad = None
                    converted += 1
            for k, b in sub.named_buffers(recurse=False):
                if b is not None and b.is_floating_point() and b.dtype != dtype:
                    setattr(sub, k, b.to(dtype))
        if isinstance(root, _m.DfMBackbone):
            root.volume_memory_format = torch.channels_last_3d
        if isinstance(root, _m.FrustumToVoxel) and not isinstance(model, DfMStereoPath):
            root.output_memory_format = torch.contiguous_format   # dfm.py:325-326 views the volume
    for m in model.modules():
        if isinstance(m, MultiViewDfMMixin) or \
                getattr(type(m), 'feature_transformation', None) is MultiViewDfMMixin.feature_transformation:
            m.fast_dtype = dtype
    cast_back = []
    if boundary_casts:
        for root in roots:
            if not root.__dict__.get('_dfm_fast_hook'):
                root.register_forward_pre_hook(_cast_hook(None, dtype), with_kwargs=True)
                root.__dict__['_dfm_fast_hook'] = True
        owners = [m for m in model.modules() if m not in inside and any(c in roots for c in m.children())]
        for owner in owners:
            for child in owner.children():
                if child in inside or any(sub in inside for sub in child.modules()):
                    continue
                if not child.__dict__.get('_dfm_fast_hook'):
                    child.register_forward_pre_hook(_cast_hook(dtype, orig), with_kwargs=True)
                    child.__dict__['_dfm_fast_hook'] = True
                cast_back.append(names.get(child, type(child).__name__))
    report = dict(roots=[names.get(r, type(r).__name__) for r in roots], converted_parameters=converted,
                  cast_back=cast_back, dtype=dtype, fallback_policy=None)
    if strict:
        from . import conv3d as _c
        kinds = (_c.MfmaConv3d, _c.MfmaConv3dG, _c.MfmaConvTranspose3d, _c.MfmaConv2d, _c.MfmaConvTranspose2d)
        n_strict = 0
        for root in roots:
            for sub in root.modules():
                if isinstance(sub, kinds):
                    sub.fallback_policy = 'raise'
                    n_strict += 1
        report['fallback_policy'] = f"'raise' on the {n_strict} Mfma* modules of this model"
    return report


__all__ = ['inject_detector_attributes', 'DfMStereoPath', 'MultiViewDfMMixin', 'MultiViewVoxelPath',
           'patch_reference', 'enable_fast_path', 'voxel_centers']
