"""CPU simulation of the operand-rounding modes of the HAT kernels (test tooling; uses the oracle).

Which 16-bit roundings dominate the logits error?  Replays the fp32 oracle with the operands of chosen matmuls rounded the way the
kernels round them (fp32 accumulate everywhere), for the operand modes of hat_runtime:

  f16 / bf16        both operands of every HAT matmul rounded once
  *x2 (split-A)     activations hi + lo (two 16-bit terms), weights rounded once
  *x3               both operands hi + lo, the lo*lo term dropped

  python tests/tools/precision_sim.py [case] [batch]
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import hat_reference as hr          # noqa: E402
from oracle import model_reference as mr        # noqa: E402
from tests.cases import CASES, SEED             # noqa: E402
from tests.synth import synth_input, synth_state_dict   # noqa: E402


def rnd(t, dt, terms):
    if dt is None:
        return t
    hi = t.to(dt).float()
    if terms == 1:
        return hi
    return hi + (t - hi).to(dt).float()


class Mode:
    def __init__(self, dt=None, a_terms=1, w_terms=1, attn=True, conv=None, conv_w=0):
        self.dt, self.a, self.w, self.attn, self.conv = dt, a_terms, w_terms, attn, conv
        self.conv_w = conv_w    # 0: conv weights rounded like the maps; 2: hi + lo; -1: exact


MODE = Mode()


def lin(x, w, b):
    m = MODE
    if m.dt is None:
        return F.linear(x, w, b)
    xa, wa = rnd(x, m.dt, m.a), rnd(w, m.dt, m.w)
    if m.a > 1 and m.w > 1:   # hi*hi + lo*hi + hi*lo (lo*lo dropped)
        xh, wh = rnd(x, m.dt, 1), rnd(w, m.dt, 1)
        return F.linear(xh, wh, b) + F.linear(xa - xh, wh) + F.linear(xh, wa - wh)
    return F.linear(xa, wa, b)


def window_attention(x, sd, prefix, heads, res, qk_scale=None):
    m = MODE
    dtype = x.dtype
    Bw, S, C = x.shape
    d = C // heads
    qkv = lin(x, sd[prefix + "qkv.weight"].to(dtype), sd[prefix + "qkv.bias"].to(dtype))
    qkv = qkv.reshape(Bw, S, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    if m.dt is not None and m.attn:
        ta = m.a if m.w > 1 else 1   # q/k/v are all activations: split them only in the x3 mode
        q, k, v = rnd(q, m.dt, ta), rnd(k, m.dt, ta), rnd(v, m.dt, ta)
    attn = (q @ k.transpose(-2, -1)) * (qk_scale or d ** -0.5)
    attn = attn + hr.attn_bias(sd, prefix + "pos_emb_funct.", res, heads, S, dtype)
    attn = attn.softmax(dim=-1)
    if m.dt is not None and m.attn:
        attn = rnd(attn, m.dt, 1)    # P is rounded once in every mode (stays in registers)
    out = (attn @ v).transpose(1, 2).reshape(Bw, -1, C)
    return lin(out, sd[prefix + "proj.weight"].to(dtype), sd[prefix + "proj.bias"].to(dtype))


def mlp(x, sd, prefix):
    dtype = x.dtype
    h = lin(x, sd[prefix + "fc1.weight"].to(dtype), sd[prefix + "fc1.bias"].to(dtype))
    h = F.gelu(h)
    return lin(h, sd[prefix + "fc2.weight"].to(dtype), sd[prefix + "fc2.bias"].to(dtype))


_conv2d = F.conv2d


def conv2d(x, w, b=None, **kw):
    c = MODE.conv
    if c is None or x.shape[1] == w.shape[1] and w.shape[1] == x.shape[1] and kw.get("groups", 1) != 1:
        return _conv2d(x, w, b, **kw)
    wr = w if MODE.conv_w < 0 else rnd(w, c, 2 if MODE.conv_w == 2 else 1)
    return _conv2d(x.to(c).float(), wr, b, **kw)


def main():
    global MODE
    case = sys.argv[1] if len(sys.argv) > 1 else "fvit0_224"
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    c = CASES[case]
    sys.path.insert(0, os.path.join(ROOT))
    import fastervit_amd
    model = fastervit_amd.create_model(c["entry"], **c["kwargs"])
    sd = synth_state_dict(model.state_dict(), SEED, c["family"])
    x = synth_input(nb, *c["hw"], seed=SEED)
    hr.window_attention, hr.mlp = window_attention, mlp

    class FF:   # F with a rounding conv2d for the conv side
        def __getattr__(self, k):
            return conv2d if k == "conv2d" else getattr(F, k)
    mr.F = FF()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref = mr.model_forward(sd, x, c["arch"])
    print(f"{case}: |logits| max {ref.abs().max():.3f}")
    H, B = torch.float16, torch.bfloat16
    if len(sys.argv) > 3 and sys.argv[3] == "short":
        modes = [("conv f16 only", Mode(None, conv=H)), ("conv f16 maps, exact weights", Mode(None, conv=H, conv_w=-1)),
                 ("conv f16 maps, weights hi+lo", Mode(None, conv=H, conv_w=2)),
                 ("f16", Mode(H)), ("f16 split-W", Mode(H, 1, 2)), ("f16 + conv f16", Mode(H, conv=H)),
                 ("f16 split-W + conv f16", Mode(H, 1, 2, conv=H)), ("f16 split-W + conv f16 split-W", Mode(H, 1, 2, conv=H, conv_w=2)),
                 ("bf16", Mode(B)), ("bf16 split-W", Mode(B, 1, 2)), ("bf16 split-W + conv f16", Mode(B, 1, 2, conv=H)),
                 ("bf16 split-W + conv f16 split-W", Mode(B, 1, 2, conv=H, conv_w=2))]
    else:
      modes = [("conv f16 only", Mode(None, conv=H)),
             ("f16", Mode(H)), ("f16 + conv f16", Mode(H, conv=H)),
             ("bf16", Mode(B)), ("bf16 + conv f16", Mode(B, conv=H)),
             ("bf16x2 (split-A)", Mode(B, 2, 1)), ("bf16x2 + conv f16", Mode(B, 2, 1, conv=H)),
             ("bf16 split-W only", Mode(B, 1, 2)),
             ("bf16x3", Mode(B, 2, 2)), ("bf16x3 + conv f16", Mode(B, 2, 2, conv=H)),
             ("bf16x3, attention core single bf16", Mode(B, 2, 2, attn=True)),
             ("f16x2", Mode(H, 2, 1)), ("f16x3", Mode(H, 2, 2)), ("f16x3 + conv f16", Mode(H, 2, 2, conv=H))]
    for name, m in modes:  # noqa: E111
        MODE = m
        y = mr.model_forward(sd, x, c["arch"])
        e = (y - ref).abs().max().item()
        print(f"  {name:38s} max-abs {e:.3e}   rel {e / ref.abs().max().item():.3e}")


if __name__ == "__main__":
    main()
