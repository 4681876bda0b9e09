"""GPU diagnostic: run faster_vit_0_224 under one conv-side configuration (own process: a GPU fault kills it)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastervit_amd  # noqa: E402
from fastervit_amd import hat_runtime  # noqa: E402

variant = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.manual_seed(0)
model = fastervit_amd.create_model("faster_vit_0_224").eval().cuda()
x = torch.randn(B, 3, 224, 224, device="cuda")
cl = "cl" in variant
if "half" in variant:
    model = model.half()
    x = x.half()
if cl:
    model = model.to(memory_format=torch.channels_last)
    x = x.contiguous(memory_format=torch.channels_last)
if "nohat" in variant:
    hat_runtime.stage_forward = lambda layer, t: t
    import fastervit_amd.models.faster_vit as fv
ac = torch.autocast("cuda", dtype=torch.float16) if "amp" in variant else torch.autocast("cuda", enabled=False)


def fwd():
    with torch.no_grad(), ac:
        return model(x)


y = fwd()
torch.cuda.synchronize()
print(variant, "first forward ok", tuple(y.shape), y.dtype, float(y.float().abs().max()), flush=True)
for _ in range(3):
    y = fwd()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    y = fwd()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"{variant} B={B}: {dt * 1e3:.2f} ms/forward = {B / dt:.0f} img/s", flush=True)
