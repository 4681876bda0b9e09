"""Uninitialised-LDS hypothesis: single-stream forwards (repeatable by themselves) with an LDS-poisoning kernel running beside them
on another stream.  If any kernel reads LDS it did not write, the logits change / turn NaN."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastervit_amd  # noqa: E402
from fastervit_amd import _lib  # noqa: E402
from fastervit_amd.conv_runtime import DeployPlan  # noqa: E402

lib = _lib.lib()
torch.manual_seed(0)
model = fastervit_amd.create_model("faster_vit_0_224").eval().cuda().to(memory_format=torch.channels_last)
x = torch.randn(86, 3, 224, 224, generator=torch.Generator().manual_seed(1000)).cuda().contiguous(memory_format=torch.channels_last)
plan = DeployPlan(model, torch.float16)
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
noise = torch.cuda.Stream()
with torch.no_grad():
    ref = plan.forward(x).clone()
    ref2 = plan.forward(x).clone()
    torch.cuda.synchronize()
    print("single stream repeat equal:", torch.equal(ref, ref2))
    for blocks, spin in ((256, 200), (1024, 50), (4096, 10), (512, 2000)):
        bad = 0
        nan = 0
        for rep in range(6):
            with torch.cuda.stream(noise):
                for _ in range(40):
                    _lib.check(lib.fvit_debug_lds_poison(sink.data_ptr(), blocks, spin, noise.cuda_stream), "poison")
            y = plan.forward(x).clone()
            torch.cuda.synchronize()
            bad += 0 if torch.equal(y, ref) else 1
            nan += 0 if torch.isfinite(y).all() else 1
        print(f"LDS poison beside the forward ({blocks} blocks x spin {spin}): {bad} of 6 forwards differ from the reference, {nan} non-finite", flush=True)
    # module-mode HAT stages too
    for li, C, R in ((2, 256, 14), (3, 512, 7)):
        from fastervit_amd import hat_runtime
        lvl = model.levels[li]
        xs = torch.randn(86, C, R, R, generator=torch.Generator().manual_seed(5)).cuda().half().contiguous(memory_format=torch.channels_last)
        r0 = hat_runtime.stage_forward(lvl, xs).clone()
        bad = 0
        for rep in range(8):
            with torch.cuda.stream(noise):
                for _ in range(30):
                    _lib.check(lib.fvit_debug_lds_poison(sink.data_ptr(), 1024, 50, noise.cuda_stream), "poison")
            y = hat_runtime.stage_forward(lvl, xs).clone()
            torch.cuda.synchronize()
            bad += 0 if torch.equal(y, r0) else 1
        print(f"HAT level {li} with LDS poison beside it: {bad} of 8 differ; finite {bool(torch.isfinite(y).all())}", flush=True)
