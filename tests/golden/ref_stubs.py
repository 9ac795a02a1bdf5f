"""Minimal stand-ins for the un-vendored mmcv / mmdet symbols the reference's
hot-path module files import, so those files can be executed UNMODIFIED from
/root/reference on PyTorch-CPU (build container only).

This is our own code: `ConvModule` restates mmcv 1.6's documented composition
(conv -> norm -> activation, `bias='auto'` = no conv bias when a norm follows,
norm sub-module named after its type: gn / bn, ReLU default).  Weights are
always copied between the reference module and ours through `state_dict`, so
mmcv's initialisation scheme is irrelevant to parity.
"""
import importlib.util
import sys
import types

import torch
from torch import nn

REF_ROOT = '/root/reference'


class Registry:

    def __init__(self, name='models'):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):

        def _reg(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls

        return _reg(module) if module is not None else _reg

    def build(self, cfg):
        cfg = dict(cfg)
        return self.module_dict[cfg.pop('type')](**cfg)


class BaseModule(nn.Module):

    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg

    def init_weights(self):
        pass


_CONV = {'Conv2d': nn.Conv2d, 'Conv3d': nn.Conv3d, None: nn.Conv2d}


def build_norm(norm_cfg, num_features):
    cfg = dict(norm_cfg)
    t = cfg.pop('type')
    requires_grad = cfg.pop('requires_grad', True)
    if t == 'GN':
        layer, name = nn.GroupNorm(num_channels=num_features, **cfg), 'gn'
    elif t in ('BN', 'BN2d'):
        layer, name = nn.BatchNorm2d(num_features, **cfg), 'bn'
    elif t == 'BN3d':
        layer, name = nn.BatchNorm3d(num_features, **cfg), 'bn'
    else:
        raise KeyError(t)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return name, layer


class ConvModule(nn.Module):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'),
                 inplace=True, order=('conv', 'norm', 'act')):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        conv = _CONV[None if conv_cfg is None else conv_cfg['type']]
        self.conv = conv(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                         dilation=dilation, groups=groups, bias=bias)
        if self.with_norm:
            self.norm_name, norm = build_norm(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            assert act_cfg['type'] == 'ReLU'
            self.activate = nn.ReLU(inplace=act_cfg.get('inplace', inplace))

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = getattr(self, self.norm_name)(x)
        if self.with_activation:
            x = self.activate(x)
        return x


def install():
    """Put the stubs where the reference files' import statements look."""
    reg = Registry()

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__path__ = []
        sys.modules[name] = m
        return m

    mod('mmcv')
    mod('mmcv.cnn', ConvModule=ConvModule)
    mod('mmcv.runner', BaseModule=BaseModule, force_fp32=lambda *a, **k: (lambda f: f))
    mod('mmdet')
    mod('mmdet.models', BACKBONES=reg, NECKS=reg, HEADS=reg, DETECTORS=reg)
    mod('mmdet.models.builder', BACKBONES=reg, NECKS=reg, HEADS=reg, DETECTORS=reg)
    # package skeleton so the reference files' absolute / relative imports resolve
    mod('mmdet3d')
    mod('mmdet3d.models')
    mod('mmdet3d.models.builder', BACKBONES=reg, NECKS=reg, HEADS=reg, DETECTORS=reg, MODELS=reg)
    mod('mmdet3d.models.necks')
    mod('mmdet3d.models.backbones')
    mod('mmdet3d.core')
    geo = {'torch': torch}
    import ast
    tree = ast.parse(open(f'{REF_ROOT}/mmdet3d/core/bbox/structures/utils.py').read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ('points_cam2img', 'points_img2cam'):
            node.decorator_list = []
            exec(compile(ast.Module(body=[node], type_ignores=[]), 'utils.py', 'exec'), geo)
    mod('mmdet3d.core.bbox', points_cam2img=geo['points_cam2img'],
        points_img2cam=geo['points_img2cam'])
    load_file('mmdet3d/models/utils/conv_modules.py', 'mmdet3d.models.utils')
    return reg


def load_hot_path_modules():
    """The reference's hot-path module files, executed unmodified."""
    install()
    out = {}
    out['dfm_backbone'] = load_file('mmdet3d/models/backbones/dfm_backbone.py',
                                    'mmdet3d.models.backbones.dfm_backbone')
    out['imvoxel_neck'] = load_file('mmdet3d/models/necks/imvoxel_neck.py',
                                    'mmdet3d.models.necks.imvoxel_neck')
    out['dfm_neck'] = load_file('mmdet3d/models/necks/dfm_neck.py', 'mmdet3d.models.necks.dfm_neck')
    out['feature_transformation'] = load_file('mmdet3d/models/necks/feature_transformation.py',
                                              'mmdet3d.models.necks.feature_transformation')
    out['depth_head'] = load_file('mmdet3d/models/dense_heads/depth_head.py', 'ref_depth_head')
    return out


def load_file(relpath, modname, extra_modules=None):
    """exec a reference source file as module `modname` (no package import)"""
    for k, v in (extra_modules or {}).items():
        sys.modules[k] = v
    spec = importlib.util.spec_from_file_location(modname, f'{REF_ROOT}/{relpath}')
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


# --------------------------------------------------------------------------------------------
# detector level (round 3): the reference's own DfM / MultiViewDfM classes, executed unmodified,
# with stand-ins for what lies OUTSIDE the path (2-D backbone, detection heads, anchor generator,
# box utilities).  Used by tests/test_reference_detectors.py to run patch_reference() against the
# real module files and build configs/dfm/* through the reference's own __init__ / build_* calls.
# --------------------------------------------------------------------------------------------
class Placeholder(nn.Module):
    """what an out-of-path ``type`` builds to (LIGAResNet, ResNet+DCN, FPN, the detection heads):
    keeps its config, has no parameters, must never be called on the path"""

    def __init__(self, **cfg):
        super().__init__()
        self.cfg = cfg

    def init_weights(self):
        pass

    def forward(self, *a, **k):
        raise AssertionError(f'out-of-path module {self.cfg.get("type")} was called')


class FallbackRegistry(Registry):
    """Registry.build that resolves unknown type names to ``Placeholder`` (everything SURVEY.md 8
    marks out of scope) and records them"""

    def __init__(self, name='models'):
        super().__init__(name)
        self.placeholders = []

    def build(self, cfg, default_args=None):
        cfg = dict(cfg)
        for k, v in (default_args or {}).items():
            cfg.setdefault(k, v)
        kind = cfg.pop('type')
        if kind not in self.module_dict:
            self.placeholders.append(kind)
            return Placeholder(type=kind, **cfg)
        return self.module_dict[kind](**cfg)


def load_detectors():
    """installs the stubs and executes the reference's detector files (dfm.py, multiview_dfm.py),
    the path's module files and point_fusion.py from /root/reference.  Returns (registry, modules)."""
    hot = load_hot_path_modules()          # also installs the mmcv / mmdet / mmdet3d skeleton
    old = sys.modules['mmdet3d.models.builder'].MODELS
    reg = FallbackRegistry()
    reg.module_dict.update(old.module_dict)  # the reference classes the hot-path files registered

    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        m.__dict__.update(attrs)
        if not hasattr(m, '__path__'):
            m.__path__ = []
        sys.modules[name] = m
        return m

    def build(cfg, train_cfg=None, test_cfg=None):
        return reg.build(cfg)

    for name in ('mmdet.models', 'mmdet.models.builder', 'mmdet3d.models.builder'):
        mod(name, BACKBONES=reg, NECKS=reg, HEADS=reg, DETECTORS=reg, MODELS=reg, FUSION_LAYERS=reg,
            build_backbone=build, build_neck=build, build_head=build, build_detector=build)
    mod('mmcv.ops')
    mod('mmcv.ops.points_in_boxes', points_in_boxes_part=None)

    class BaseDetector(BaseModule):
        pass
    mod('mmdet.models.detectors', BaseDetector=BaseDetector)
    core = sys.modules['mmdet3d.core.bbox']
    mod('mmdet3d.core', bbox3d2result=None,
        build_prior_generator=lambda cfg: types.SimpleNamespace(cfg=dict(cfg)))
    mod('mmdet3d.core.bbox.structures', points_cam2img=core.points_cam2img, points_img2cam=core.points_img2cam,
        get_proj_mat_by_coord_type=lambda img_meta, coord_type: img_meta['lidar2img'])
    mod('mmdet3d.core.points', get_points_type=None)
    mod('mmdet3d.models.dense_heads', LIGAATSSHead=type('LIGAATSSHead', (), {}),
        CenterHead=type('CenterHead', (), {}))
    mod('mmdet3d.models.utils.common_utils', dist_reduce_mean=lambda x: x)
    mod('mmdet3d.models.detectors')
    mod('mmdet3d.models.detectors.imitation_utils', NormalizeLayer=nn.Identity, WeightedL2WithSigmaLoss=nn.Identity)
    out = dict(hot)
    # fusion_layers: the real coord_transform.py and point_fusion.py (what multiview_dfm.py imports from)
    fl = mod('mmdet3d.models.fusion_layers')
    ct = load_file('mmdet3d/models/fusion_layers/coord_transform.py', 'mmdet3d.models.fusion_layers.coord_transform')
    fl.apply_3d_transformation = ct.apply_3d_transformation
    out['point_fusion'] = load_file('mmdet3d/models/fusion_layers/point_fusion.py',
                                    'mmdet3d.models.fusion_layers.point_fusion')
    fl.point_sample, fl.voxel_sample = out['point_fusion'].point_sample, out['point_fusion'].voxel_sample
    # the hot-path files executed above registered into the old registry object; re-run them so the
    # reference classes are what `reg` holds before patch_reference() overrides them
    for key, rel, name in (('dfm_backbone', 'mmdet3d/models/backbones/dfm_backbone.py',
                            'mmdet3d.models.backbones.dfm_backbone'),
                           ('imvoxel_neck', 'mmdet3d/models/necks/imvoxel_neck.py', 'mmdet3d.models.necks.imvoxel_neck'),
                           ('dfm_neck', 'mmdet3d/models/necks/dfm_neck.py', 'mmdet3d.models.necks.dfm_neck'),
                           ('feature_transformation', 'mmdet3d/models/necks/feature_transformation.py',
                            'mmdet3d.models.necks.feature_transformation'),
                           ('depth_head', 'mmdet3d/models/dense_heads/depth_head.py', 'mmdet3d.models.dense_heads.depth_head'),
                           ('bev_hourglass', 'mmdet3d/models/backbones/bev_hourglass.py',
                            'mmdet3d.models.backbones.bev_hourglass'),
                           ('spp_unet_neck', 'mmdet3d/models/necks/spp_unet_neck.py', 'mmdet3d.models.necks.spp_unet_neck')):
        out[key] = load_file(rel, name)
    out['dfm'] = load_file('mmdet3d/models/detectors/dfm.py', 'mmdet3d.models.detectors.dfm')
    out['multiview_dfm'] = load_file('mmdet3d/models/detectors/multiview_dfm.py',
                                     'mmdet3d.models.detectors.multiview_dfm')
    return reg, out
