"""Shim of the timm.models names fastervit/validate.py imports (validate.py:28)."""
import torch

from .registry import _entrypoints


def is_model(name):
    return name in _entrypoints


def list_models(filter="", **kwargs):
    import fnmatch
    names = sorted(_entrypoints)
    return fnmatch.filter(names, filter) if filter else names


def create_model(model_name, pretrained=False, pretrained_cfg=None, pretrained_cfg_overlay=None, checkpoint_path="",
                 scriptable=None, exportable=None, no_jit=None, **kwargs):
    # timm 0.9.6 drops None-valued kwargs and forwards the pretrained_cfg pair to the entrypoint
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    model = _entrypoints[model_name](pretrained=pretrained, pretrained_cfg=pretrained_cfg,
                                     pretrained_cfg_overlay=pretrained_cfg_overlay, **kwargs)
    if checkpoint_path:
        load_checkpoint(model, checkpoint_path)
    return model


def load_checkpoint(model, checkpoint_path, use_ema=True, strict=True):
    ckpt = torch.load(checkpoint_path, map_location="cpu")
    sd = ckpt.get("state_dict_ema" if (use_ema and "state_dict_ema" in ckpt) else "state_dict", ckpt)
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    return model.load_state_dict(sd, strict=strict)
