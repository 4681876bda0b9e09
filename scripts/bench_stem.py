"""Micro-benchmark: fused two-conv stem vs stem kernel + stride-2 conv kernel (FasterViT-0 shapes)."""
import os
os.environ.setdefault("FVIT_DIAG", "1")   # diagnosis build of the library (fvit_debug_* entry points, ablation knobs)
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastervit_amd import _lib, hat_runtime  # noqa: E402

lib = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 86
st = torch.cuda.current_stream().cuda_stream
dt, code = torch.float16, 1
g = torch.Generator(device="cpu").manual_seed(0)
xs = [torch.randn(B, 3, 224, 224, generator=g).cuda() for _ in range(4)]
wk1 = torch.zeros(64, 32)
wk1[:, :27] = torch.randn(64, 27, generator=g) / 5
wk1 = wk1.to(dt).cuda()
wk2 = (torch.randn(64, 3, 3, 64, generator=g) / 24).to(dt).cuda()
b1, b2 = torch.randn(64, generator=g).cuda(), torch.randn(64, generator=g).cuda()
mid = torch.empty(B, 112, 112, 64, dtype=dt, device="cuda")
out = torch.empty(B, 56, 56, 64, dtype=dt, device="cuda")
zeros = torch.zeros(256, dtype=dt, device="cuda")


def fused(i):
    v = hat_runtime._map_view(xs[i % 4])
    _lib.check(lib.fvit_stem_fused(code, C.byref(v), wk1.data_ptr(), b1.data_ptr(), wk2.data_ptr(), b2.data_ptr(), out.data_ptr(), B, 224, 224, st), "f")


def two(i):
    v = hat_runtime._map_view(xs[i % 4])
    _lib.check(lib.fvit_stem_conv3x3s2(code, C.byref(v), wk1.data_ptr(), b1.data_ptr(), mid.data_ptr(), B, 224, 224, st), "s")
    _lib.check(lib.fvit_conv3x3_nhwc(code, mid.data_ptr(), wk2.data_ptr(), b2.data_ptr(), None, out.data_ptr(), B, 112, 112, 64, 64, 2, 1,
                                     zeros.data_ptr(), st), "c")


for rnd in range(2):
    for name, fn in (("fused", fused), ("two kernels", two)):
        for i in range(3):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        print(f"B={B} {name:12s}: {e0.elapsed_time(e1) * 1e3 / 20:7.1f} us", flush=True)

# phase accounting of the fused kernel (fvit_debug_stem_timeline): ticks per wave accumulated over its tiles
ts = torch.zeros(512 * 4 * 8, dtype=torch.int64, device="cuda")
v = hat_runtime._map_view(xs[0])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
_lib.check(lib.fvit_debug_stem_timeline(C.byref(v), wk1.data_ptr(), b1.data_ptr(), wk2.data_ptr(), b2.data_ptr(), out.data_ptr(), B, 224, 224,
                                        ts.data_ptr(), st), "timeline")
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
t = ts.view(512, 4, 8).double()
t = t[t[..., 5] > 0]
tot = t[:, 6]
ghz = tot.max().item() / us / 1e3
print(f"timeline launch {us:.1f} us; {t.shape[0]} waves, tiles per wave mean {t[:, 5].mean().item():.2f}; longest wave {tot.max().item():.0f} ticks -> {ghz:.2f} ticks/ns")
for i, name in enumerate(["phase A (gathers + conv1 + LDS writes)", "barrier after A", "phase B (conv2)", "epilogue (bias, ReLU, stores)", "barrier before A"]):
    per_tile = (t[:, i] / t[:, 5]) / ghz / 1e3
    print(f"  {name:40s} {100 * (t[:, i].sum() / tot.sum()).item():5.1f} % of wave time   {per_tile.mean().item():6.2f} us per tile (max {per_tile.max().item():.2f})")
print(f"  wave total per tile {((tot / t[:, 5]) / ghz / 1e3).mean().item():.2f} us")
