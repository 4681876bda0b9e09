"""Does a wave of the fused MLP see stale input rows under 3 concurrent stream shards?  The kernel stores the rows exactly as its loads
returned them (fvit_debug_mlp_inputs_begin); the row-hash trace dumps the residual stream after every kernel.  Rows the kernel saw that
differ from the stream content in front of it are compared with the OLDER states of the stream."""
import ctypes as C
import os
import sys

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
CAP = 16 << 20
hbuf = torch.zeros(CAP, dtype=torch.int32, device="cuda")
seen = torch.zeros(90_000_000, dtype=torch.float32, device="cuda")
TCAP = 230_000_000
tbufs = [torch.zeros(TCAP, dtype=torch.int32, device="cuda") for _ in range(2)]
seen_prev = torch.zeros_like(seen)
STATES = []    # per shard: list of (record index, tag) of the window-stream states of stage 2, in order
dumps = {}


def traced(plan, k=0):
    lib.fvit_debug_rowhash_dump(-1, None, 0)
    for rec, t in dumps.items():
        lib.fvit_debug_rowhash_dump(rec, t.data_ptr(), t.numel() * 4)
    torch.cuda.synchronize()
    lib.fvit_debug_rowhash_begin(hbuf.data_ptr(), CAP)
    lib.fvit_debug_mlp_inputs_begin(seen.data_ptr(), seen.numel())
    lib.fvit_debug_mlp_trace_begin(tbufs[k].data_ptr(), TCAP)
    y = plan.forward(x).clone()
    toffs = (C.c_int64 * 64)()
    lib.fvit_debug_mlp_trace_end(toffs, None, 64)
    traced.toffs = list(toffs)
    offs = (C.c_int64 * 64)()
    rows = (C.c_int32 * 64)()
    nl = lib.fvit_debug_mlp_inputs_end(offs, rows, 64)
    recs = (_lib.FvitDebugRowhashRecord * 512)()
    nrec = lib.fvit_debug_rowhash_end(recs, 512)
    torch.cuda.synchronize()
    return y, [(offs[i], rows[i]) for i in range(nl)], [(recs[i].tag.decode(), recs[i].offset, recs[i].rows) for i in range(nrec)]


for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    plan = DeployPlan(model, torch.float16)
    plan.streams = 3
    with torch.no_grad():
        for _ in range(3):
            plan.forward(x)
        torch.cuda.synchronize()
        if not STATES:
            _, launches, recs = traced(plan)
            parts = [i for i, r in enumerate(recs) if r[0] == "partition"]
            for sh in range(3):
                lo, hi = parts[2 * sh], parts[2 * sh + 1]
                STATES.append([(i, recs[i][0]) for i in range(lo, hi) if recs[i][0] in ("partition", "win.attnblk", "mlpf.out")])
                for i, _ in STATES[-1]:
                    dumps[i] = torch.zeros(recs[i][2] * 256, dtype=torch.float32, device="cuda")
            print("states per shard:", [len(s) for s in STATES], "; fused-MLP launches:", len(launches), flush=True)
        prev = None
        for it in range(n):
            seen_prev.copy_(seen)
            y, launches, recs = traced(plan, it & 1)
            if prev is not None:
                print(f"trial {trial} call {it}: logits equal to the previous call: {torch.equal(y, prev)}", flush=True)
                for li, (off, rows) in enumerate(launches):
                    wg = (rows + 63) // 64
                    Q = 164
                    to = traced.toffs[li]
                    a = tbufs[it & 1][to:to + wg * 4 * Q * 64].view(wg * 4, Q, 64)
                    b = tbufs[1 - (it & 1)][to:to + wg * 4 * Q * 64].view(wg * 4, Q, 64)
                    badw = (a[:, 0] != b[:, 0]).any(dim=1).nonzero().flatten().tolist()
                    if badw:
                        w = badw[0]
                        r0 = w * 16
                        s_now = seen[off:off + rows * 256].view(rows, 256)[r0:r0 + 16]
                        s_old = seen_prev[off:off + rows * 256].view(rows, 256)[r0:r0 + 16]
                        print(f"   launch {li}: LN fragments of {len(badw)} waves differ from the previous call (first: wave {w}, rows {r0}..{r0 + 15}); "
                              f"rows as loaded by the kernel equal in both calls: {torch.equal(s_now, s_old)}; lanes differing {int((a[w, 0] != b[w, 0]).sum())}")
                        break
            prev = y
            nbad = 0
            for li, (off, rows) in enumerate(launches):
                sh, b = divmod(li, 6)
                states = STATES[sh]
                # the stream in front of MLP b is the state after attention block b: states = [partition, attn0, mlp0, attn1, mlp1, ...]
                cur = dumps[states[1 + 2 * b][0]].view(rows, 256)
                saw = seen[off:off + rows * 256].view(rows, 256)
                badrows = (saw != cur).any(dim=1).nonzero().flatten()
                if badrows.numel() == 0:
                    continue
                nbad += 1
                br = badrows.tolist()
                msg = f"   call {it} shard {sh} block {b}: the kernel saw {len(br)} rows that differ from its input stream; rows {br[:8]}{'...' if len(br) > 8 else ''} (waves {sorted({r // 16 for r in br})[:8]})"
                for back in range(2 * b, -1, -1):
                    old = dumps[states[back][0]].view(rows, 256)
                    eq = (saw[badrows] == old[badrows]).all(dim=1)
                    if bool(eq.any()):
                        msg += f"\n      {int(eq.sum())} of them are bitwise the state '{states[back][1]}' #{back} ({2 * b + 1 - back} kernel(s) older)"
                mx = (saw[badrows] - cur[badrows]).abs().max().item()
                msg += f"\n      max |saw - current| {mx:.3e}; elements differing per row: {(saw[badrows] != cur[badrows]).sum(dim=1).tolist()[:8]}"
                print(msg, flush=True)
            print(f"trial {trial} call {it}: {nbad} launches saw rows that were not the current stream", flush=True)
