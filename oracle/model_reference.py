"""CPU oracle for the whole FasterViT forward (conv side + HAT stages), functional over a state_dict.

TEST INFRASTRUCTURE ONLY (see hat_reference.py header).  The conv side stays PyTorch in the product
(north_star); here it is restated with torch.nn.functional calls so that the oracle is independent
of the product's nn.Module tree.  Citations: ``AR:`` = reference faster_vit_any_res.py,
``FV:`` = reference faster_vit.py.

Known unpinned constant: timm's LayerNorm2d eps (1e-6) is from memory of timm 0.9.6 -- timm is not
installed in the build container -- and is shared with the golden generator's shim.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

from . import hat_reference as hr

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def _bn(x: Tensor, sd: SD, prefix: str, eps: float) -> Tensor:
    d = x.dtype
    return F.batch_norm(x, sd[prefix + "running_mean"].to(d), sd[prefix + "running_var"].to(d),
                        sd[prefix + "weight"].to(d), sd[prefix + "bias"].to(d), False, 0.0, eps)


def patch_embed(x: Tensor, sd: SD, prefix: str = "patch_embed.") -> Tensor:
    """PatchEmbed (AR:444-470 / FV:443-469): conv3x3 s2 -> BN(1e-4) -> ReLU, twice."""
    d = x.dtype
    x = F.conv2d(x, sd[prefix + "conv_down.0.weight"].to(d), None, stride=2, padding=1)
    x = torch.relu(_bn(x, sd, prefix + "conv_down.1.", 1e-4))
    x = F.conv2d(x, sd[prefix + "conv_down.3.weight"].to(d), None, stride=2, padding=1)
    x = torch.relu(_bn(x, sd, prefix + "conv_down.4.", 1e-4))
    return x


def conv_block(x: Tensor, sd: SD, prefix: str) -> Tensor:
    """ConvBlock.forward (AR:503-513 / FV:502-512)."""
    d = x.dtype
    y = F.conv2d(x, sd[prefix + "conv1.weight"].to(d), sd[prefix + "conv1.bias"].to(d), padding=1)
    y = F.gelu(_bn(y, sd, prefix + "norm1.", 1e-5))
    y = F.conv2d(y, sd[prefix + "conv2.weight"].to(d), sd[prefix + "conv2.bias"].to(d), padding=1)
    y = _bn(y, sd, prefix + "norm2.", 1e-5)
    if prefix + "gamma" in sd:
        y = y * sd[prefix + "gamma"].to(d).view(1, -1, 1, 1)
    return x + y


def downsample(x: Tensor, sd: SD, prefix: str) -> Tensor:
    """Downsample.forward (AR:438-441 / FV:437-440): LayerNorm2d(eps 1e-6) + conv3x3 s2, no bias."""
    d = x.dtype
    C = x.shape[1]
    y = F.layer_norm(x.permute(0, 2, 3, 1), (C,), sd[prefix + "norm.weight"].to(d), sd[prefix + "norm.bias"].to(d), 1e-6)
    y = y.permute(0, 3, 1, 2)
    return F.conv2d(y, sd[prefix + "reduction.0.weight"].to(d), None, stride=2, padding=1)


def model_forward(sd: SD, x: Tensor, arch: dict, dtype=torch.float32,
                  capture: Optional[dict] = None) -> Tensor:
    """FasterViT.forward (AR:979-995 / FV:949-965).

    ``arch`` keys: depths, num_heads, window_size, ct_size, dim, resolution (int or [H, W]),
    hat (list of bool), do_propagation, layer_norm_last, any_res (bool: which file's tokenizer rule).
    ``capture`` (optional dict) receives 'level{i}' feature maps and 'blocks{i}' per-block (x, ct).
    """
    res = arch["resolution"]
    if not isinstance(res, (list, tuple)):
        res = [res, res]
    depths = arch["depths"]
    x = x.to(dtype)
    x = patch_embed(x, sd)
    for i, depth in enumerate(depths):
        prefix = f"levels.{i}."
        if i < 2:
            for j in range(depth):
                x = conv_block(x, sd, f"{prefix}blocks.{j}.")
        else:
            cap = [] if capture is not None else None
            x = hr.hat_stage(
                x, sd, prefix, depth=depth, heads=arch["num_heads"][i], ws=arch["window_size"][i],
                cw=arch["ct_size"],
                input_resolution=[int(2 ** (-2 - i) * res[0]), int(2 ** (-2 - i) * res[1])],
                only_local=not arch["hat"][i], do_propagation=arch.get("do_propagation", False),
                any_res=arch.get("any_res", False), capture=cap, qk_scale=arch.get("qk_scale"))
            if capture is not None:
                capture[f"blocks{i}"] = cap
        if capture is not None:
            capture[f"level{i}"] = x.clone()
        if i < 3:
            x = downsample(x, sd, prefix + "downsample.")
    if arch.get("layer_norm_last", False):
        C = x.shape[1]
        x = F.layer_norm(x.permute(0, 2, 3, 1), (C,), sd["norm.weight"].to(dtype), sd["norm.bias"].to(dtype), 1e-6).permute(0, 3, 1, 2)
    else:
        x = _bn(x, sd, "norm.", 1e-5)
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    return F.linear(x, sd["head.weight"].to(dtype), sd["head.bias"].to(dtype))
