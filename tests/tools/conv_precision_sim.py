"""CPU replay of the deploy plan's CONV-SIDE storage roundings on the fp32 oracle (test tooling; uses the oracle).

The HAT stages run exact (fp32): what is measured is the logits error caused by WHERE the conv side of conv_runtime.DeployPlan keeps a
16-bit value.  Rounding points (each 'r' = one 16-bit rounding, 'd' = two 16-bit terms hi + lo, 'f' = fp32):

  img    the input image as the stem reads it            stem   PatchEmbed output map
  mid    ConvBlock conv1 + BN + GELU output              res    ConvBlock output = the residual stream of levels 0 / 1
  ln     LayerNorm2d output (Downsample conv input)      down   Downsample conv output (next level's input map)
  hat    a transformer level's output map                w      conv weights ('r' once, 'd' hi + lo, 'f' exact); wstem: first stem conv

  python tests/tools/conv_precision_sim.py [case] [batch]
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

DT = torch.float16


def rnd(t, how):
    if how == "f":
        return t
    hi = t.to(DT).float()
    if how == "r":
        return hi
    return hi + (t - hi).to(DT).float()


class Cfg(dict):
    __getattr__ = dict.__getitem__


BASE = dict(img="r", stem="r", mid="r", res="r", ln="r", down="r", hat="r", w="d", wstem="r")


def forward(sd, x, arch, c):
    c = Cfg(c)
    res = arch["resolution"]
    if not isinstance(res, (list, tuple)):
        res = [res, res]
    bn = mr._bn
    x = rnd(x, c.img)
    p = "patch_embed."
    x = F.conv2d(x, rnd(sd[p + "conv_down.0.weight"], c.wstem), None, stride=2, padding=1)
    x = rnd(torch.relu(bn(x, sd, p + "conv_down.1.", 1e-4)), c.get("stem0", "f"))   # inside the fused stem / 16-bit between the two kernels
    x = F.conv2d(x, rnd(sd[p + "conv_down.3.weight"], c.w), None, stride=2, padding=1)
    x = rnd(torch.relu(bn(x, sd, p + "conv_down.4.", 1e-4)), c.stem)
    for i, depth in enumerate(arch["depths"]):
        prefix = f"levels.{i}."
        if i < 2:
            for j in range(depth):
                q = f"{prefix}blocks.{j}."
                # the conv reads a 16-bit A operand unless the stream point says two-term
                y = F.conv2d(x, rnd(sd[q + "conv1.weight"], c.w), sd[q + "conv1.bias"], padding=1)
                y = rnd(F.gelu(bn(y, sd, q + "norm1.", 1e-5)), c.mid)
                y = F.conv2d(y, rnd(sd[q + "conv2.weight"], c.w), sd[q + "conv2.bias"], padding=1)
                y = bn(y, sd, q + "norm2.", 1e-5)
                if q + "gamma" in sd:
                    y = y * sd[q + "gamma"].view(1, -1, 1, 1)
                x = rnd(x + y, c.res)
        else:
            x = hr.hat_stage(x, sd, prefix, depth=depth, heads=arch["num_heads"][i], ws=arch["window_size"][i], cw=arch["ct_size"],
                             input_resolution=[int(2 ** (-2 - i) * res[0]), int(2 ** (-2 - i) * res[1])], only_local=not arch["hat"][i],
                             do_propagation=arch.get("do_propagation", False), any_res=arch.get("any_res", False), capture=None,
                             qk_scale=arch.get("qk_scale"))
            x = rnd(x, c.hat)
        if i < 3:
            q = prefix + "downsample."
            C = x.shape[1]
            y = F.layer_norm(x.permute(0, 2, 3, 1), (C,), sd[q + "norm.weight"], sd[q + "norm.bias"], 1e-6).permute(0, 3, 1, 2)
            y = rnd(y, c.ln)
            wk = c.get("wdown", c.w)
            x = rnd(F.conv2d(y, rnd(sd[q + "reduction.0.weight"], wk), None, stride=2, padding=1), c.down)
    x = bn(x, sd, "norm.", 1e-5)
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    return F.linear(x, sd["head.weight"], sd["head.bias"])


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "fvit4_224"
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    c = CASES[case]
    import fastervit_amd
    model = fastervit_amd.create_model(c["entry"], **c["kwargs"])
    sd = synth_state_dict(model.state_dict(), SEED, c["family"])
    x = synth_input(nb, *c["hw"], seed=SEED)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    exact = {k: "f" for k in BASE}
    ref = forward(sd, x, c["arch"], exact)
    ref2 = mr.model_forward(sd, x, c["arch"])
    print(f"{case}: |logits| max {ref.abs().max():.3f}   (restatement vs oracle {float((ref - ref2).abs().max()):.1e})")

    def run(name, **over):
        cfg = dict(BASE)
        cfg.update(over)
        e = (forward(sd, x, c["arch"], cfg) - ref).abs().max().item()
        print(f"  {name:64s} {e:.3e}", flush=True)

    run("deploy plan as built (16-bit maps, weights hi+lo, stem conv0 once)")
    run("... all weights exact", w="f", wstem="f")
    run("... single-term weights everywhere", w="r")
    for k in ("img", "stem", "mid", "res", "ln", "down", "hat", "wstem"):
        run(f"ONLY {k} rounded (everything else exact)", **{**{kk: "f" for kk in BASE}, k: "r"})
    for k in ("img", "stem", "mid", "res", "ln", "down", "hat", "wstem"):
        run(f"plan with {k} exact", **{k: "f"})
    run("plan with res + ln + down + hat exact", res="f", ln="f", down="f", hat="f")
    run("plan with res + ln + down + hat + stem exact", res="f", ln="f", down="f", hat="f", stem="f")
    run("plan with ln + down + hat exact", ln="f", down="f", hat="f")
    run("plan with ln + down exact", ln="f", down="f")
    run("plan with res two-term, ln/down/hat two-term", res="d", ln="d", down="d", hat="d")


if __name__ == "__main__":
    main()
