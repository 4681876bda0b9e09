from .faster_vit import *  # noqa: F401,F403  (entrypoints, registered on import)
from .faster_vit_any_res import *  # noqa: F401,F403
from .faster_vit import FasterViT, FasterViTLayer, HAT, WindowAttention, Mlp  # noqa: F401
from .registry import create_model, list_models, is_model, model_entrypoint  # noqa: F401
