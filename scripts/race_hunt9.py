"""Row-hash trace of every kernel of the HAT stages under 3 concurrent stream shards: which launch is the first whose output rows differ
between two identical calls?  (fvit_debug_rowhash_begin / _end; profiles/r02_repeatability_hunt.log)"""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib = _lib.lib()
CAP = 16 << 20
bufs = [torch.zeros(CAP, dtype=torch.int32, device="cuda") for _ in range(2)]
DUMP_REC = int(os.environ.get("DUMP_REC", "117"))
dumps = [torch.zeros(20000 * 256, dtype=torch.float32, device="cuda") for _ in range(2)]


def traced(plan, buf, dump):
    buf.zero_()
    lib.fvit_debug_rowhash_dump(DUMP_REC, dump.data_ptr(), dump.numel() * 4)
    torch.cuda.synchronize()
    lib.fvit_debug_rowhash_begin(buf.data_ptr(), CAP)
    y = plan.forward(x).clone()
    recs = (_lib.FvitDebugRowhashRecord * 512)()
    nrec = lib.fvit_debug_rowhash_end(recs, 512)
    torch.cuda.synchronize()
    return y, [(recs[i].tag.decode(), recs[i].offset, recs[i].rows) for i in range(nrec)]


for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    plan = DeployPlan(model, torch.float16)
    plan.streams = 3
    with torch.no_grad():
        for _ in range(3):
            plan.forward(x)
        torch.cuda.synchronize()
        prev = None
        for it in range(n):
            y, recs = traced(plan, bufs[it & 1], dumps[it & 1])
            if prev is not None:
                py, precs = prev
                assert [r[0] for r in recs] == [r[0] for r in precs] and [r[2] for r in recs] == [r[2] for r in precs]
                a, b = bufs[it & 1], bufs[(it - 1) & 1]
                first = None
                ndiff = 0
                lines = []
                for k, (tag, off, rows) in enumerate(recs):
                    d = (a[off:off + rows] != b[off:off + rows]).nonzero().flatten()
                    if d.numel():
                        ndiff += 1
                        if len(lines) < 6:
                            dl = d.tolist()
                            lines.append(f"      rec {k} {tag} rows={rows}: {len(dl)} rows differ: {dl[:24]}{' ...' if len(dl) > 24 else ''}")
                print(f"trial {trial} call {it}: logits equal={torch.equal(y, py)}; {len(recs)} records, {ndiff} differ")
                for ln in lines:
                    print(ln)
                tag, off, rows = recs[DUMP_REC]
                d = (a[off:off + rows] != b[off:off + rows]).nonzero().flatten().tolist()
                if d:
                    A = dumps[it & 1][:rows * 256].view(rows, 256)
                    B = dumps[(it - 1) & 1][:rows * 256].view(rows, 256)
                    for r in d[:3] + d[16:18] + d[-1:]:
                        df = (A[r] - B[r])
                        nz = df.nonzero().flatten().tolist()
                        print(f"      dump rec {DUMP_REC} row {r} (wg {r // 64} wave {(r // 16) % 4} s {r % 16}): {len(nz)} channels differ, max abs {df.abs().max().item():.3e}, "
                              f"|x| max {A[r].abs().max().item():.2f}; channels {nz[:40]}")
                        if nz:
                            c = nz[0]
                            print(f"         ch {c}: {A[r, c].item():.8f} vs {B[r, c].item():.8f}")
                if lines:
                    k0 = int(lines[0].split()[1])
                    print("      context: " + " | ".join(f"{k}:{recs[k][0]}({recs[k][2]})" for k in range(max(0, k0 - 4), min(len(recs), k0 + 2))))
            prev = (y, recs)
