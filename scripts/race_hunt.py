"""Stress test for run-to-run bit repeatability of the deploy-plan forward (r02: one replay mismatch seen at batch 256, 3 stream shards).
Replays / re-runs the same input many times per configuration and reports which images differ and by how much, then localises by
running the conv side and the HAT stages separately on concurrent streams.   usage: python scripts/race_hunt.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastervit_amd  # noqa: E402
from fastervit_amd import hat_runtime  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.manual_seed(0)
model = fastervit_amd.create_model("faster_vit_0_224").eval().cuda().to(memory_format=torch.channels_last)
x = torch.randn(256, 3, 224, 224, generator=torch.Generator().manual_seed(1000)).cuda().contiguous(memory_format=torch.channels_last)


def report(tag, outs):
    ref = outs[0]
    bad = 0
    for i, o in enumerate(outs[1:], 1):
        if not torch.equal(o, ref):
            bad += 1
            d = (o.float() - ref.float()).abs()
            rows = (d.flatten(1).max(dim=1).values > 0).nonzero().flatten().tolist() if d.dim() > 1 else []
            print(f"  {tag}: rep {i} differs: max {d.max().item():.3e}, {len(rows)} images/rows differ, first {rows[:8]}")
    print(f"{tag}: {bad} of {len(outs) - 1} repeats differ", flush=True)


for streams, graph in ((3, True), (3, False), (1, True), (1, False)):
    runner = model.compile_inference(x, dtype=torch.float16, streams=streams, graph=graph)
    outs = [runner(x).clone() for _ in range(reps)]
    torch.cuda.synchronize()
    report(f"deploy streams={streams} graph={graph}", outs)
    del runner

# ---- conv side only (levels 0, 1 + stem) on 3 concurrent streams, eager ----
from fastervit_amd.conv_runtime import DeployPlan  # noqa: E402
plan = DeployPlan(model, torch.float16)
plan._enter(x)
plan._refresh()
parts = x.chunk(3)
side = [torch.cuda.Stream() for _ in range(3)]


def conv_side(xi):
    t = plan.t
    with torch.no_grad():
        w0, b0, w1, b1 = t["stem"]
        B, _, Hi, Wi = xi.shape
        H1, W1 = (Hi - 1) // 2 + 1, (Wi - 1) // 2 + 1
        y = torch.empty((B, 64, (H1 - 1) // 2 + 1, (W1 - 1) // 2 + 1), dtype=torch.float16, device=xi.device, memory_format=torch.channels_last)
        from fastervit_amd import _lib
        view = hat_runtime._map_view(xi)
        _lib.check(_lib.lib().fvit_stem_fused(plan.code, view, t["stem_k"].data_ptr(), b0.data_ptr(), w1[1].data_ptr(), b1.data_ptr(), y.data_ptr(), B, Hi, Wi,
                                              torch.cuda.current_stream().cuda_stream), "stem")
        xx = y
        outs = [xx.clone()]
        for lvl, e in list(zip(model.levels, t["levels"]))[:2]:
            for wa, ba, wb, bb in e["blocks"]:
                yy = plan._conv(xx, wa, ba, 1, 2)
                xx = plan._conv(yy, wb, bb, 1, 0, residual=xx)
            outs.append(xx.clone())
            lw, lb, eps, wd, cin = e["down"]
            xx = plan._conv(plan._ln2d(xx, lw, lb, eps, cin), wd, None, 2, 0)
            outs.append(xx.clone())
        return outs


plan.dev = x.device
res = []
for rep in range(reps):
    cur = []
    for s, p in zip(side, parts):
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            cur.append(conv_side(p))
    torch.cuda.synchronize()
    res.append(cur)
names = ["stem", "level0", "down0", "level1", "down1"]
for k, nm in enumerate(names):
    for sh in range(3):
        report(f"conv side, 3 concurrent streams, shard {sh}, after {nm}", [r[sh][k] for r in res])

# ---- HAT stages only on 3 concurrent streams ----
for li, C, R in ((2, 256, 14), (3, 512, 7)):
    lvl = model.levels[li]
    xs = [torch.randn(n, C, R, R, generator=torch.Generator().manual_seed(5 + i)).cuda().half().contiguous(memory_format=torch.channels_last)
          for i, n in enumerate((86, 86, 84))]
    res = []
    for rep in range(reps):
        cur = []
        for i, (s, xi) in enumerate(zip(side, xs)):
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s), torch.no_grad(), hat_runtime.workspace_slot(i):
                cur.append(hat_runtime.stage_forward(lvl, xi).clone())
        torch.cuda.synchronize()
        res.append(cur)
    for sh in range(3):
        report(f"HAT level {li}, 3 concurrent streams, shard {sh}", [r[sh] for r in res])
    one = [hat_runtime.stage_forward(lvl, xs[0]).clone() for _ in range(reps)]
    report(f"HAT level {li}, single stream", one)
