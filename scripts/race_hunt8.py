"""Which overlap makes the shards non-repeatable?  Eager 3-stream forwards with an event chain that serialises selected stages
across the shards (the rest still overlaps):  none / stage 2 / stage 3 / both HAT stages / everything-but-HAT."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastervit_amd  # noqa: E402
from fastervit_amd import conv_runtime, hat_runtime  # noqa: E402
from fastervit_amd.conv_runtime import DeployPlan  # noqa: E402

torch.manual_seed(0)
model = fastervit_amd.create_model("faster_vit_0_224").eval().cuda().to(memory_format=torch.channels_last)
x = torch.randn(256, 3, 224, 224, generator=torch.Generator().manual_seed(1000)).cuda().contiguous(memory_format=torch.channels_last)
orig_stage = hat_runtime.stage_forward
MODE = ["none"]
chain = {}


def stage(layer, x_, tokenizer=None, out=None):
    C = x_.shape[1]
    ser = (MODE[0] == "stage2" and C == 256) or (MODE[0] == "stage3" and C == 512) or MODE[0] == "hat"
    s = torch.cuda.current_stream()
    if ser:
        ev = chain.get("last")
        if ev is not None:
            s.wait_event(ev)
    o = orig_stage(layer, x_, tokenizer, out)
    if ser:
        ev = torch.cuda.Event()
        ev.record(s)
        chain["last"] = ev
    return o


hat_runtime.stage_forward = stage
conv_runtime.hat_runtime.stage_forward = stage
n = 14
for trial in range(3):
    for mode in ("none", "stage2", "stage3", "hat"):
        MODE[0] = mode
        chain.clear()
        plan = DeployPlan(model, torch.float16)
        plan.streams = 3
        outs = []
        with torch.no_grad():
            for _ in range(n):
                outs.append(plan.forward(x).clone())
                torch.cuda.synchronize()
        bad = sum(0 if torch.equal(a, b) else 1 for a, b in zip(outs[1:-1], outs[2:]))
        print(f"trial {trial} serialise={mode}: {bad} of {n - 2} consecutive pairs differ", flush=True)
