"""Parity cases shared by the golden generator, the oracle tests and the GPU parity tests.

``entry`` + ``kwargs`` is how the model is built through the create_model API (reference and product
take the same arguments); ``arch`` is the explicit architecture dict the functional oracle takes.
Hyper-parameters of the named variants follow the reference entrypoints (FV:975-1160, AR:1159-1198).
"""

_H = [False, False, True, False]


def _arch(depths, heads, ws, dim, res, hat=_H, prop=False, any_res=False, ct=2, qk_scale=None):
    return dict(depths=depths, num_heads=heads, window_size=ws, ct_size=ct, dim=dim, resolution=res, hat=hat,
                do_propagation=prop, layer_norm_last=False, any_res=any_res, qk_scale=qk_scale)


CASES = {
    # ---- BASELINE.json configs (full size) ----
    "fvit0_224": dict(entry="faster_vit_0_224", kwargs={}, batch=8, hw=(224, 224), family="init", per_block=False, stage_maps=True,
                      arch=_arch([2, 3, 6, 5], [2, 4, 8, 16], [7, 7, 7, 7], 64, 224)),
    "fvit0_224_stress": dict(entry="faster_vit_0_224", kwargs={}, batch=2, hw=(224, 224), family="stress", per_block=False,
                             stage_maps=True,
                             arch=_arch([2, 3, 6, 5], [2, 4, 8, 16], [7, 7, 7, 7], 64, 224)),
    "fvit4_224": dict(entry="faster_vit_4_224", kwargs={}, batch=2, hw=(224, 224), family="init", per_block=False,
                      arch=_arch([3, 3, 12, 5], [4, 8, 16, 32], [7, 7, 7, 7], 196, 224, prop=True)),
    "fvit4_anyres_576x960": dict(entry="faster_vit_4_any_res",
                                 kwargs=dict(resolution=[576, 960], window_size=[7, 7, 12, 6], ct_size=2), batch=1,
                                 hw=(576, 960), family="init", per_block=False,
                                 arch=_arch([3, 3, 12, 5], [4, 8, 16, 32], [7, 7, 12, 6], 196, [576, 960], prop=True,
                                            any_res=True)),
    # ---- ImageNet-21k fine-tune geometries with ONE long window per image in stage 2 (576 / 2304 tokens): logits only ----
    "fvit4_21k_384": dict(entry="faster_vit_4_21k_384", kwargs={}, batch=1, hw=(384, 384), family="init", per_block=False,
                          arch=_arch([3, 3, 12, 5], [4, 8, 16, 32], [7, 7, 24, 12], 196, 384, hat=[False] * 4, prop=True)),
    "fvit4_21k_768": dict(entry="faster_vit_4_21k_768", kwargs={}, batch=1, hw=(768, 768), family="init", per_block=False,
                          arch=_arch([3, 3, 12, 5], [4, 8, 16, 32], [7, 7, 48, 24], 196, 768, hat=[False] * 4, prop=True)),
    # ---- small configs with per-block goldens ("stress" weights so every sub-path matters) ----
    # square hierarchical stage, layer scale + propagation, head_dim 24 (padded to 32)
    "tiny_hier": dict(entry="faster_vit_4_224",
                      kwargs=dict(depths=[1, 1, 3, 2], num_heads=[1, 2, 4, 8], dim=24, in_dim=16), batch=2, hw=(224, 224),
                      family="stress", per_block=True,
                      arch=_arch([1, 1, 3, 2], [1, 2, 4, 8], [7, 7, 7, 7], 24, 224, prop=True)),
    # head_dim 40 (padded to 64), no layer scale (gamma = 1), no propagation
    "tiny_d40": dict(entry="faster_vit_0_224",
                     kwargs=dict(depths=[1, 1, 2, 1], num_heads=[1, 1, 2, 4], dim=20, in_dim=16), batch=2, hw=(224, 224),
                     family="stress", per_block=True,
                     arch=_arch([1, 1, 2, 1], [1, 1, 2, 4], [7, 7, 7, 7], 20, 224)),
    # non-square any-res: stage 2 is 6x10 -> padded 6x12 with ws 3 (sr [2,4], G = 32, carrier bias 25 in 32),
    # stage 3 is 3x5 -> padded 3x6; exercises F.pad / crop and the ct_window scramble
    "tiny_anyres": dict(entry="faster_vit_4_any_res",
                        kwargs=dict(depths=[1, 1, 2, 2], num_heads=[1, 1, 2, 4], dim=16, in_dim=16, resolution=[96, 160],
                                    window_size=[7, 7, 3, 3], ct_size=2), batch=2, hw=(96, 160), family="stress",
                        per_block=True,
                        arch=_arch([1, 1, 2, 2], [1, 1, 2, 4], [7, 7, 3, 3], 16, [96, 160], prop=True, any_res=True)),
    # local-only stages with a 14x14 window (S = 196), as in faster_vit_4_21k_224
    "tiny_w14": dict(entry="faster_vit_4_21k_224",
                     kwargs=dict(depths=[1, 1, 2, 1], num_heads=[1, 1, 2, 4], dim=16, in_dim=16), batch=1, hw=(224, 224),
                     family="stress", per_block=True,
                     arch=_arch([1, 1, 2, 1], [1, 1, 2, 4], [7, 7, 14, 7], 16, 224, hat=[False] * 4, prop=True)),
    # head_dim 80 (padded to 96) in both transformer stages, as in FasterViT-5 / -6; hierarchical, layer scale, propagation
    "tiny_d80": dict(entry="faster_vit_5_224",
                     kwargs=dict(depths=[1, 1, 2, 1], num_heads=[1, 1, 1, 2], dim=20, in_dim=16), batch=2, hw=(224, 224),
                     family="stress", per_block=True,
                     arch=_arch([1, 1, 2, 1], [1, 1, 1, 2], [7, 7, 7, 7], 20, 224, prop=True)),
    # qk_scale override (pass-through kwarg of every entrypoint, FV:538), head_dim 32
    "tiny_qk": dict(entry="faster_vit_0_224",
                    kwargs=dict(depths=[1, 1, 2, 1], num_heads=[1, 1, 2, 4], dim=16, in_dim=16, qk_scale=0.31), batch=2, hw=(224, 224),
                    family="stress", per_block=True,
                    arch=_arch([1, 1, 2, 1], [1, 1, 2, 4], [7, 7, 7, 7], 16, 224, qk_scale=0.31)),
    # ---- long windows (> 208 tokens): the online-softmax attention kernel with the compact relative-bias table ----
    # one 24x24 window (576 tokens) in stage 2, 12x12 (144, dense path) in stage 3: the geometry of faster_vit_4_21k_384
    "tiny_21k_384": dict(entry="faster_vit_4_21k_384",
                         kwargs=dict(depths=[1, 1, 2, 1], num_heads=[1, 1, 2, 4], dim=16, in_dim=16), batch=1, hw=(384, 384),
                         family="stress", per_block=True,
                         arch=_arch([1, 1, 2, 1], [1, 1, 2, 4], [7, 7, 24, 12], 16, 384, hat=[False] * 4, prop=True)),
    # hierarchical any-res stage with 16x16 windows: S = 256 + 4 carrier tokens (n_g = 4 bias-free rows/columns), 2 windows
    "tiny_anyres_w16": dict(entry="faster_vit_4_any_res",
                            kwargs=dict(depths=[1, 1, 2, 1], num_heads=[1, 1, 2, 4], dim=16, in_dim=16, resolution=[256, 512],
                                        window_size=[7, 7, 16, 8], ct_size=2), batch=1, hw=(256, 512), family="stress",
                            per_block=True,
                            arch=_arch([1, 1, 2, 1], [1, 1, 2, 4], [7, 7, 16, 8], 16, [256, 512], prop=True, any_res=True)),
}

SEED = 1234
