"""A minimal `type=`-string registry with the mmcv build convention
(`dict(type='Name', **kwargs)`), so the hot-path modules build from the
reference's config dicts without mmcv installed.  When mmdet3d IS importable,
`register_into_mmdet3d()` re-registers the same classes into its MODELS
registry (force=True) and `configs/dfm/*` resolve to these implementations."""

_MODULES = {}


def register_module(cls=None, *, name=None):

    def _do(c):
        _MODULES[name or c.__name__] = c
        return c

    return _do(cls) if cls is not None else _do


def build(cfg, **default_args):
    """build_backbone / build_neck / build_head equivalent."""
    if not isinstance(cfg, dict) or 'type' not in cfg:
        raise TypeError(f'cfg must be a dict with a "type" key, got {cfg!r}')
    args = dict(cfg)
    kind = args.pop('type')
    if kind not in _MODULES:
        raise KeyError(f'{kind} is not registered in depth-from-motion_amd '
                       f'(known: {sorted(_MODULES)})')
    for k, v in default_args.items():
        args.setdefault(k, v)
    return _MODULES[kind](**args)


build_backbone = build_neck = build_head = build


def registered():
    return dict(_MODULES)


def register_into_mmdet3d():
    """Override the reference classes inside a real mmdet3d installation."""
    from mmdet3d.models.builder import MODELS  # noqa: raises ImportError without mmdet3d
    for name, cls in _MODULES.items():
        MODELS.register_module(name=name, force=True, module=cls)
    return sorted(_MODULES)
