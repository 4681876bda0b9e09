"""Localise the 'first eager 3-stream forward differs' effect (race_hunt.py): record every intermediate of every shard for the first
calls of a fresh deploy plan and report the first tensor that differs between call k and call k + 1."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastervit_amd  # noqa: E402
from fastervit_amd import conv_runtime, hat_runtime  # noqa: E402

torch.manual_seed(0)
model = fastervit_amd.create_model("faster_vit_0_224").eval().cuda().to(memory_format=torch.channels_last)
x = torch.randn(256, 3, 224, 224, generator=torch.Generator().manual_seed(1000)).cuda().contiguous(memory_format=torch.channels_last)

# prime the HAT state / graphs exactly as race_hunt.py does before the failing configuration
r = model.compile_inference(x, dtype=torch.float16, streams=3, graph=True)
for _ in range(3):
    r(x)
torch.cuda.synchronize()
del r

rec = {}
cur_call = [0]
orig_conv = conv_runtime.DeployPlan._conv
orig_ln2d = conv_runtime.DeployPlan._ln2d
orig_stage = hat_runtime.stage_forward
orig_linear = torch.nn.functional.linear


def tag(name, t):
    key = (cur_call[0], hat_runtime._slot())
    rec.setdefault(key, []).append((name, t.detach().clone()))


def conv(self, x_, w, bias, stride, act, residual=None):
    out = orig_conv(self, x_, w, bias, stride, act, residual)
    tag(f"conv s{stride} a{act} r{int(residual is not None)} {tuple(out.shape)}", out)
    return out


def ln2d(self, x_, w, b, eps, c_valid=None):
    out = orig_ln2d(self, x_, w, b, eps, c_valid)
    tag(f"ln2d {tuple(out.shape)}", out)
    return out


def stage(layer, x_, tokenizer=None, out=None):
    o = orig_stage(layer, x_, tokenizer, out)
    tag(f"hat stage C={x_.shape[1]}", o)
    return o


def linear(inp, w, b=None):
    tag(f"pooled {tuple(inp.shape)}", inp)
    o = orig_linear(inp, w, b)
    tag(f"logits {tuple(o.shape)}", o)
    return o


conv_runtime.DeployPlan._conv = conv
conv_runtime.DeployPlan._ln2d = ln2d
hat_runtime.stage_forward = stage
conv_runtime.hat_runtime.stage_forward = stage
conv_runtime.F.linear = linear

from fastervit_amd.conv_runtime import DeployPlan  # noqa: E402
plan = DeployPlan(model, torch.float16)
plan.streams = 3
outs = []
with torch.no_grad():
    for k in range(5):
        cur_call[0] = k
        outs.append(plan.forward(x).clone())
        torch.cuda.synchronize()
for k in range(4):
    same = torch.equal(outs[k], outs[k + 1])
    print(f"call {k} vs {k + 1}: logits equal = {same}")
    if same:
        continue
    for slot in range(3):
        a, b = rec.get((k, slot), []), rec.get((k + 1, slot), [])
        for (na, ta), (nb, tb) in zip(a, b):
            if not torch.equal(ta, tb):
                d = (ta.float() - tb.float()).abs()
                imgs = (d.flatten(1).max(dim=1).values > 0).nonzero().flatten().tolist()
                print(f"   shard {slot}: first difference at '{na}': max {d.max().item():.3e}, images {imgs[:10]} ({len(imgs)} total)")
                break
        else:
            print(f"   shard {slot}: all {len(a)} recorded tensors equal")
