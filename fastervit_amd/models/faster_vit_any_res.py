"""Any-resolution entrypoints (reference: fastervit/models/faster_vit_any_res.py:1005-1440).

The network classes are shared with ``faster_vit.py`` (built with ``any_res=True``): list-valued
``resolution``, zero-padding of transformer stages to a window multiple (AR:851-857), rectangular
carrier-token pooling (AR:729-741), ``hat_pos_embed`` only on square carrier grids (AR:658)."""
from .faster_vit import (HAT, ConvBlock, Downsample, FasterViT, FasterViTLayer, LayerNorm2d, Mlp, PatchEmbed,  # noqa: F401
                         PosEmbMLPSwinv1D, PosEmbMLPSwinv2D, TokenInitializer, WindowAttention, _ARCH, _make_entrypoint,
                         _variant_cfg)

default_cfgs = {f"faster_vit_{v}_any_res": _variant_cfg(v, True) for v in _ARCH}

for _v in _ARCH:
    _name = f"faster_vit_{_v}_any_res"
    globals()[_name] = _make_entrypoint(_v, True, _name, __name__)
del _v, _name
