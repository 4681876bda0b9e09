import inspect


class _Cfg:
    def __init__(self, d):
        self._d = dict(d)

    def to_dict(self):
        return {k: v for k, v in self._d.items() if v is not None}


def resolve_pretrained_cfg(variant, pretrained_cfg=None, pretrained_cfg_overlay=None):
    # look the variant up in the caller module's default_cfgs (what timm's registry would hold)
    frame = inspect.currentframe().f_back
    cfgs = frame.f_globals.get('default_cfgs', {})
    return _Cfg(cfgs.get(variant, {}))


def _update_default_model_kwargs(pretrained_cfg, kwargs, kwargs_filter):
    default_kwarg_names = ('num_classes', 'global_pool', 'in_chans')
    if pretrained_cfg.get('fixed_input_size', False):
        default_kwarg_names += ('img_size',)
    for n in default_kwarg_names:
        if n == 'img_size':
            input_size = pretrained_cfg.get('input_size', None)
            if input_size is not None:
                kwargs.setdefault(n, input_size[-2:])
        elif n == 'in_chans':
            input_size = pretrained_cfg.get('input_size', None)
            if input_size is not None:
                kwargs.setdefault(n, input_size[0])
        else:
            default_val = pretrained_cfg.get(n, None)
            if default_val is not None:
                kwargs.setdefault(n, pretrained_cfg[n])
    if kwargs_filter:
        for k in kwargs_filter:
            kwargs.pop(k, None)
