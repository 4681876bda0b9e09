"""Model registry and factory with the reference's pip-side API (fastervit/models/registry.py):
``register_pip_model`` (:30-55), ``list_models`` (:60-98), ``is_model`` / ``model_entrypoint``
(:101-112), ``load_state_dict`` / ``load_checkpoint`` (:161-193), ``create_model`` (:195-205)."""
from __future__ import annotations

import fnmatch
import os
import re
import sys
from collections import OrderedDict, defaultdict
from copy import deepcopy

import torch

__all__ = ["list_models", "is_model", "model_entrypoint", "list_modules", "is_model_in_modules",
           "is_model_default_key", "has_model_default_key", "get_model_default_value", "is_model_pretrained",
           "register_pip_model", "load_state_dict", "load_checkpoint", "create_model"]

_module_to_models = defaultdict(set)
_model_to_module = {}
_model_entrypoints = {}
_model_has_pretrained = set()
_model_default_cfgs = {}


def register_pip_model(fn):
    mod = sys.modules[fn.__module__]
    module_name = fn.__module__.split(".")[-1]
    name = fn.__name__
    if hasattr(mod, "__all__"):
        mod.__all__.append(name)
    else:
        mod.__all__ = [name]
    _model_entrypoints[name] = fn
    _model_to_module[name] = module_name
    _module_to_models[module_name].add(name)
    cfgs = getattr(mod, "default_cfgs", {})
    if name in cfgs:
        if "http" in cfgs[name].get("url", ""):
            _model_has_pretrained.add(name)
        _model_default_cfgs[name] = deepcopy(cfgs[name])
    return fn


def _natural_key(s):
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s.lower())]


def list_models(filter="", module="", pretrained=False, exclude_filters="", name_matches_cfg=False):
    names = set(_module_to_models[module]) if module else set(_model_entrypoints)
    if filter:
        pats = filter if isinstance(filter, (tuple, list)) else [filter]
        names = {n for p in pats for n in fnmatch.filter(names, p)}
    if exclude_filters:
        pats = exclude_filters if isinstance(exclude_filters, (tuple, list)) else [exclude_filters]
        for p in pats:
            names -= set(fnmatch.filter(names, p))
    if pretrained:
        names &= _model_has_pretrained
    if name_matches_cfg:
        names &= set(_model_default_cfgs)
    return sorted(names, key=_natural_key)


def is_model(model_name):
    return model_name in _model_entrypoints


def model_entrypoint(model_name):
    return _model_entrypoints[model_name]


def list_modules():
    return sorted(_module_to_models)


def is_model_in_modules(model_name, module_names):
    assert isinstance(module_names, (tuple, list, set))
    return any(model_name in _module_to_models[n] for n in module_names)


def has_model_default_key(model_name, cfg_key):
    return model_name in _model_default_cfgs and cfg_key in _model_default_cfgs[model_name]


def is_model_default_key(model_name, cfg_key):
    return bool(model_name in _model_default_cfgs and _model_default_cfgs[model_name].get(cfg_key, False))


def get_model_default_value(model_name, cfg_key):
    return _model_default_cfgs[model_name].get(cfg_key, None) if model_name in _model_default_cfgs else None


def is_model_pretrained(model_name):
    return model_name in _model_has_pretrained


def load_state_dict(checkpoint_path, use_ema=False):
    if not (checkpoint_path and os.path.isfile(checkpoint_path)):
        print("No checkpoint found at '{}'".format(checkpoint_path))
        raise FileNotFoundError()
    ckpt = torch.load(checkpoint_path, map_location="cpu")
    key = "state_dict"
    if isinstance(ckpt, dict) and use_ema and "state_dict_ema" in ckpt:
        key = "state_dict_ema"
    if isinstance(ckpt, dict) and key in ckpt:
        sd = OrderedDict((k[7:] if k.startswith("module") else k, v) for k, v in ckpt[key].items())
    else:
        sd = ckpt
    print("Loaded {} from checkpoint '{}'".format(key, checkpoint_path))
    return sd


def load_checkpoint(model, checkpoint_path, use_ema=False, strict=True):
    if os.path.splitext(checkpoint_path)[-1].lower() in (".npz", ".npy"):
        if hasattr(model, "load_pretrained"):
            model.load_pretrained(checkpoint_path)
            return
        raise NotImplementedError("Model cannot load numpy checkpoint")
    model.load_state_dict(load_state_dict(checkpoint_path, use_ema), strict=strict)


def create_model(model_name, pretrained=False, checkpoint_path="", **kwargs):
    """Same signature as the reference's ``fastervit.create_model`` (registry.py:195-205)."""
    model = model_entrypoint(model_name)(pretrained=pretrained, **kwargs)
    if checkpoint_path:
        load_checkpoint(model, checkpoint_path)
    return model
