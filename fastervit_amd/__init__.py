"""fastervit_amd -- MI355X-native FasterViT forward path (drop-in for ``fastervit``'s model API).

>>> from fastervit_amd import create_model
>>> model = create_model('faster_vit_0_224').cuda().eval()

The Hierarchical-Attention stages run in hand-written gfx950 HIP kernels (``csrc/``) behind the C
ABI of ``include/fvit_hip.h``; the conv side is PyTorch-ROCm.  See DESIGN.md / INTEGRATION.md.
"""
from .models import create_model, list_models, is_model, model_entrypoint  # noqa: F401
from .models.registry import load_checkpoint, load_state_dict, register_pip_model  # noqa: F401

__version__ = "0.1.0"
