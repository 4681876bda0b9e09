"""Is the run-to-run difference an uninitialised read of a torch.empty buffer?  Poison the caching allocator's free blocks before
every call (allocate + free a tensor filled with NaN / 1e4 on the call's stream) and compare."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastervit_amd  # noqa: E402
from fastervit_amd import hat_runtime  # noqa: E402
from fastervit_amd.conv_runtime import DeployPlan  # noqa: E402

torch.manual_seed(0)
model = fastervit_amd.create_model("faster_vit_0_224").eval().cuda().to(memory_format=torch.channels_last)
x = torch.randn(86, 3, 224, 224, generator=torch.Generator().manual_seed(1000)).cuda().contiguous(memory_format=torch.channels_last)
plan = DeployPlan(model, torch.float16)
side = torch.cuda.Stream()


def poison(val, mb=256):
    t = torch.full((mb * 1024 * 1024 // 4,), val, device="cuda")
    del t


def run(tag, stream, val):
    outs = []
    for k in range(6):
        with torch.cuda.stream(stream), torch.no_grad():
            if val is not None:
                poison(val)
            outs.append(plan.forward(x).clone())
        torch.cuda.synchronize()
    bad = [k for k in range(1, 6) if not torch.equal(outs[k], outs[0])]
    nan = [k for k in range(6) if not torch.isfinite(outs[k]).all()]
    md = max((outs[k].float() - outs[0].float()).abs().max().item() for k in range(1, 6))
    print(f"{tag}: calls differing from call 0: {bad}; calls with non-finite logits: {nan}; max diff {md:.3e}", flush=True)


cur = torch.cuda.current_stream()
run("main stream, no poison", cur, None)
run("side stream, no poison", side, None)
run("side stream, free blocks poisoned with 1e4", side, 1.0e4)
run("side stream, free blocks poisoned with NaN", side, float("nan"))
run("main stream, free blocks poisoned with NaN", cur, float("nan"))

# stage 2 alone with poisoned allocator: which torch.empty buffer is read before it is written?
lvl = model.levels[2]
xs = torch.randn(86, 256, 14, 14, generator=torch.Generator().manual_seed(3)).cuda().half().contiguous(memory_format=torch.channels_last)
outs = []
for k in range(5):
    with torch.cuda.stream(side), torch.no_grad():
        poison(float("nan"))
        outs.append(hat_runtime.stage_forward(lvl, xs).clone())
    torch.cuda.synchronize()
print("stage 2 alone, NaN-poisoned allocator: finite =", [bool(torch.isfinite(o).all()) for o in outs], "equal to call 0 =",
      [bool(torch.equal(o, outs[0])) for o in outs])
# token_init alone
tok = lvl.global_tokenizer
for k in range(3):
    with torch.cuda.stream(side), torch.no_grad():
        poison(float("nan"))
        ct = hat_runtime.token_init(tok, xs)
    torch.cuda.synchronize()
    print("token_init output finite:", bool(torch.isfinite(ct).all()), tuple(ct.shape))
