from .models.layers import DropPath, LayerNorm2d, trunc_normal_  # noqa: F401


def apply_test_time_pool(model, config, use_test_size=False):
    return model, False


def set_fast_norm(enable=True):
    pass
