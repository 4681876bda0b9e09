"""Inside the fused MLP: which intermediate quantity of a wave is the first to differ between two identical calls under 3 concurrent
stream shards?  (fvit_debug_mlp_trace_begin / _end: per-lane hashes of the LN fragments and, per hidden chunk, of the W1 fragments as
read from LDS, the pre-GELU accumulators, the GELU output, the W2 fragments as read and the output accumulators.)"""
import ctypes as C
import os
import sys
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastervit_amd  # noqa: E402
from fastervit_amd import _lib  # noqa: E402
from fastervit_amd.conv_runtime import DeployPlan  # noqa: E402

torch.manual_seed(0)
model = fastervit_amd.create_model("faster_vit_0_224").eval().cuda().to(memory_format=torch.channels_last)
x = torch.randn(256, 3, 224, 224, generator=torch.Generator().manual_seed(1000)).cuda().contiguous(memory_format=torch.channels_last)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
lib = _lib.lib()
CAP = 230_000_000
bufs = [torch.zeros(CAP, dtype=torch.int32, device="cuda") for _ in range(2)]
NAMES = ["W1 fragments read", "pre-GELU accumulators", "GELU output fragment", "W2 fragments read", "output accumulators"]


def traced(plan, k):
    torch.cuda.synchronize()
    lib.fvit_debug_mlp_trace_begin(bufs[k].data_ptr(), CAP)
    y = plan.forward(x).clone()
    offs = (C.c_int64 * 64)()
    rows = (C.c_int32 * 64)()
    nl = lib.fvit_debug_mlp_trace_end(offs, rows, 64)
    torch.cuda.synchronize()
    return y, [(offs[i], rows[i]) for i in range(nl)]


for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    plan = DeployPlan(model, torch.float16)
    plan.streams = 3
    with torch.no_grad():
        for _ in range(3):
            plan.forward(x)
        torch.cuda.synchronize()
        prev = None
        for it in range(n):
            k = it & 1
            y, launches = traced(plan, k)
            if prev is not None:
                same = torch.equal(y, prev)
                print(f"trial {trial} call {it}: logits equal={same}; {len(launches)} traced launches", flush=True)
                for li, (off, rows) in enumerate(launches):
                    wg = (rows + 63) // 64
                    Q = 4 + 5 * 32
                    a = bufs[k][off:off + wg * 4 * Q * 64].view(wg * 4, Q, 64)
                    b = bufs[1 - k][off:off + wg * 4 * Q * 64].view(wg * 4, Q, 64)
                    dq = (a != b).any(dim=2)                      # [waves, Q]
                    bad = dq.any(dim=1).nonzero().flatten().tolist()
                    if not bad:
                        continue
                    firsts = Counter()
                    detail = []
                    for w in bad:
                        q = int(dq[w].nonzero().flatten()[0])
                        what = ["LN fragments", "raw input rows", "LN weight/bias as loaded", "mean/rstd"][q] if q < 4 else f"{NAMES[(q - 4) % 5]}"
                        firsts[what] += 1
                        if len(detail) < 6:
                            nl = int((a[w, q] != b[w, q]).sum())
                            qs = dq[w].nonzero().flatten().tolist()
                            detail.append(f"wg {w // 4} wave {w % 4}: first at q={q} ({what}, chunk iteration {(q - 4) // 5 if q >= 4 else '-'}), {nl} lanes; "
                                          f"{len(qs)} of {Q} slots differ; slots {qs[:8]}")
                    print(f"   launch {li} (rows {rows}): {len(bad)} waves differ; first differing quantity: {dict(firsts)}")
                    for dline in detail:
                        print("      " + dline)
                    break
            prev = y
