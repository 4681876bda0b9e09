"""Stage-level A/B of fvit_tune knob sets: HAT stage 2 / 3 of FasterViT-0 at shard size (86 images), eager launches on one stream,
interleaved rounds.  usage: bench_stage.py "<knob>=<v>,<knob>=<v>;<knob>=<v>;..."   (empty set = defaults)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastervit_amd  # noqa: E402
from fastervit_amd import _lib, hat_runtime  # noqa: E402

sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in part.split(",") if kv) for part in (sys.argv[1] if len(sys.argv) > 1 else ";ct_fused=1").split(";")]
DEF = {"ct_touch": 1}
keys = sorted({k for st in sets for k in st})
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 86
torch.manual_seed(0)
model = fastervit_amd.create_model("faster_vit_0_224").eval().cuda()
g = torch.Generator(device="cpu").manual_seed(5)
for li, R, C in ((2, 14, 256), (3, 7, 512)):
    lvl = model.levels[li]
    x = torch.randn(bs, C, R, R, generator=g).cuda().half().contiguous(memory_format=torch.channels_last)
    ref = None
    for rnd in range(2):
        for st in sets:
            for k in keys:
                _lib.tune(k, st.get(k, DEF.get(k, 0)))
            for _ in range(3):
                y = hat_runtime.stage_forward(lvl, x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                y = hat_runtime.stage_forward(lvl, x)
            e1.record()
            torch.cuda.synchronize()
            if ref is None:
                ref = y.float().clone()
            err = (y.float() - ref).abs().max().item()
            if rnd == 0:
                _lib.prof_enable(True)
                for _ in range(4):
                    hat_runtime.stage_forward(lvl, x)
                torch.cuda.synchronize()
                recs = _lib.prof_records()
                _lib.prof_enable(False)
                agg = {}
                for r in recs:
                    a = agg.setdefault((r["name"], r["grid"]), [0, 0.0])
                    a[0] += 1
                    a[1] += r["ms"]
                print("   per launch shape (4 forwards): " + "; ".join(f"{n} x{g}: {c // 4}/fwd {1000 * ms / c:.1f} us" for (n, g), (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:9]), flush=True)
            print(f"round {rnd} level {li} batch {bs} {st or 'defaults'}: {e0.elapsed_time(e1) / 20 * 1000:8.1f} us per stage forward; max |diff| vs first set {err:.3e} (|y| max {ref.abs().max().item():.2f})", flush=True)
    for k in keys:
        _lib.tune(k, 0)
