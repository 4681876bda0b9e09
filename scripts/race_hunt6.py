"""fp32 view of the run-to-run difference: after each eager 3-stream forward read the stage-2 workspace's residual stream X (fp32,
first buffer of the workspace) of every slot and compare with the previous call's."""
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
for k, v in (dict(a.split("=") for a in sys.argv[1:])).items():
    _lib.tune(k, int(v))
plan = DeployPlan(model, torch.float16)
plan.streams = 3
lvl = model.levels[2]
snaps = []
with torch.no_grad():
    for call in range(6):
        plan.forward(x)
        torch.cuda.synchronize()
        st = lvl.__dict__["_fvit_state"][("cuda", 0)]
        cur = {}
        for key, w in st.workspaces.items():
            B, slot = key[0], key[-1]
            rows = B * 4 * 53
            cur[slot] = w.buf[: rows * 256 * 4].view(torch.float32).view(rows, 256).clone()
        snaps.append(cur)
for a, b, k in zip(snaps[1:-1], snaps[2:], range(1, 5)):
    for slot in sorted(a):
        d = (a[slot] - b[slot]).abs()
        nz = (d > 0)
        if nz.any():
            rows = nz.any(dim=1).nonzero().flatten()
            rel = (d / a[slot].abs().clamp_min(1e-6))[nz]
            print(f"call {k} vs {k + 1}, slot {slot}: {int(nz.sum())} elements in {rows.numel()} rows differ; max abs {d.max().item():.3e}, median rel {rel.median().item():.2e}, "
                  f"rows {rows[:6].tolist()} .. {rows[-3:].tolist()}; row%53 of first rows {[int(r) % 53 for r in rows[:8]]}; blocks64 {sorted(set((rows // 64).tolist()))[:10]}")
        else:
            print(f"call {k} vs {k + 1}, slot {slot}: identical")
