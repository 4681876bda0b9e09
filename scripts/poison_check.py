"""Do any kernels read registers or LDS they never wrote?  One stream, a poison kernel in front of every launch (all 512 vector registers
of every SIMD and all LDS set to a pattern): the logits must not depend on the pattern.  Then per kernel family (knobs) to narrow it."""
import os
os.environ.setdefault("FVIT_DIAG", "1")   # diagnosis build of the library (fvit_debug_* entry points, ablation knobs)
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastervit_amd  # noqa: E402
from fastervit_amd import _lib  # noqa: E402
from fastervit_amd.conv_runtime import DeployPlan  # noqa: E402

torch.manual_seed(0)
name = sys.argv[1] if len(sys.argv) > 1 else "faster_vit_0_224"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 86
model = fastervit_amd.create_model(name).eval().cuda().to(memory_format=torch.channels_last)
x = torch.randn(bs, 3, 224, 224, generator=torch.Generator().manual_seed(1000)).cuda().contiguous(memory_format=torch.channels_last)
lib = _lib.lib()
sink = torch.zeros(16, dtype=torch.int32, device="cuda")
PATTERNS = [("none", None), ("NaN", 0x7fc07fc0), ("zero", 0), ("1.0h/2.0f", 0x40004000), ("-big", 0xfbfffbff)]


def run(plan, pattern):
    lib.fvit_debug_poison_launches(sink.data_ptr() if pattern is not None else None, pattern or 0)
    with torch.no_grad():
        y = plan.forward(x).float().clone()
    torch.cuda.synchronize()
    lib.fvit_debug_poison_launches(None, 0)
    return y


for knobs in [{}, {"mlp_fused": 0}, {"attn_fused": 0}, {"mlp_fused": 0, "attn_fused": 0}]:
    for k, v in {"mlp_fused": 1, "attn_fused": 1}.items():
        _lib.tune(k, knobs.get(k, v))
    plan = DeployPlan(model, torch.float16)
    plan.streams = 1
    with torch.no_grad():
        for _ in range(2):
            plan.forward(x)
    base = run(plan, None)
    print(f"{name} batch {bs} knobs {knobs or 'defaults'}:", flush=True)
    for label, pat in PATTERNS:
        y = run(plan, pat)
        d = (y - base).abs()
        bad = (d.max(dim=1).values > 0).sum().item() if torch.isfinite(d).all() else -1
        print(f"   poison {label:>10}: equal to the unpoisoned run: {torch.equal(y, base)}; max |diff| {d.max().item():.3e}; images differing {bad}; "
              f"finite {bool(torch.isfinite(y).all())}", flush=True)
