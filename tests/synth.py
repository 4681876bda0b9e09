"""Deterministic synthetic weights keyed by state_dict name (no checkpoint files travel).

The same function is applied to the real reference model (when generating goldens in the build
container) and to the product model / oracle (in tests and bench), so weights are identical on
both sides without committing 31-365 M parameters.  torch's CPU generator is deterministic across
machines for a given torch version.

Families (SURVEY.md §4 trap, §8d):
  * "init"   -- what a freshly constructed reference model holds (trunc-normal std 0.02 Linear
               weights, kaiming-uniform conv weights) plus seeded perturbation of gamma~U(0.5,1.5),
               BN running stats, biases ~N(0,0.02).  Headline logits tolerance is quoted on this.
  * "stress" -- variance-preserving weights (std 1/sqrt(fan_in)), norm weights ~U(0.5,1.5), biases
               ~N(0,0.1): makes every HAT sub-path (position MLPs, attention, carrier tokens,
               propagation) contribute O(1) so per-block parity tests can see bugs.
"""
import hashlib
import math

import torch

_KEEP = ("relative_coords_table", "relative_position_index", "relative_bias", "num_batches_tracked")


def _gen(key: str, seed: int) -> torch.Generator:
    h = hashlib.sha256(f"{seed}:{key}".encode()).digest()
    g = torch.Generator(device="cpu")
    g.manual_seed(int.from_bytes(h[:7], "little"))
    return g


def synth_tensor(key: str, ref: torch.Tensor, seed: int, family: str) -> torch.Tensor:
    leaf = key.rsplit(".", 1)[-1]
    if leaf in _KEEP or not ref.is_floating_point():
        return ref.clone()
    g = _gen(key, seed)
    shape = tuple(ref.shape)

    def normal(std, mean=0.0):
        return torch.randn(shape, generator=g, dtype=torch.float32) * std + mean

    def uniform(lo, hi):
        return torch.rand(shape, generator=g, dtype=torch.float32) * (hi - lo) + lo

    stress = family == "stress"
    if leaf == "running_mean":
        return normal(0.1)
    if leaf == "running_var":
        return uniform(0.5, 1.5)
    if leaf.startswith("gamma"):
        return uniform(0.5, 1.5)
    if ref.ndim == 1:
        is_norm = ".norm" in key or key.startswith("norm") or "hat_norm" in key or ".conv_down.1." in key or ".conv_down.4." in key
        if leaf == "weight":           # norm scale (LayerNorm / BatchNorm / LayerNorm2d)
            return uniform(0.5, 1.5) if stress else torch.ones(shape)
        if leaf == "bias":
            return normal(0.1 if stress else 0.02)
        del is_norm
        return normal(0.02)
    # matrices / conv kernels
    fan_in = ref[0].numel()
    if stress:
        return normal(1.0 / math.sqrt(fan_in))
    if ref.ndim == 2:                  # nn.Linear: trunc_normal_(std=.02) (FV:930-934)
        return normal(0.02).clamp_(-2.0, 2.0)
    bound = 1.0 / math.sqrt(fan_in)    # nn.Conv2d default kaiming_uniform_(a=sqrt(5))
    return uniform(-bound, bound)


def synth_state_dict(template: dict, seed: int = 0, family: str = "init") -> dict:
    """Return a new state_dict with the same keys/shapes as ``template`` and synthetic values."""
    out = {}
    alias = {}
    for k, v in template.items():
        # TokenInitializer registers ONE conv under two names (AR:737-739); keep them identical.
        canon = k.replace("global_tokenizer.to_global_feature.pos.", "global_tokenizer.pos_embed.")
        if canon in alias:
            out[k] = out[alias[canon]].clone()
            continue
        alias[canon] = k
        out[k] = synth_tensor(canon, v, seed, family).to(v.dtype)
    return out


def synth_input(batch: int, h: int, w: int, seed: int = 0, in_chans: int = 3) -> torch.Tensor:
    g = _gen(f"input:{batch}x{in_chans}x{h}x{w}", seed)
    return torch.randn((batch, in_chans, h, w), generator=g, dtype=torch.float32)
