"""Micro-benchmark of the 3x3 conv kernels: halo-tiled (Cin = Cout = 64, stride 1) vs the implicit-GEMM kernel, A/B in one process.

usage: python scripts/bench_conv.py [B] [H] [W] [variants]      variants: comma list of halo[:grid] | gemm
"""
import os
os.environ.setdefault("FVIT_DIAG", "1")   # diagnosis build of the library (fvit_debug_* entry points, ablation knobs)
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastervit_amd import _lib  # noqa: E402

lib = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 85
H = int(sys.argv[2]) if len(sys.argv) > 2 else 56
W = int(sys.argv[3]) if len(sys.argv) > 3 else 56
variants = sys.argv[4].split(",") if len(sys.argv) > 4 else ["gemm", "halo", "halo:256", "halo:768", "halo:1024"]
C = int(os.environ.get("CONV_C", "64"))
dt, code = torch.float16, 1
g = torch.Generator(device="cpu").manual_seed(0)
# several input/output buffers so that consecutive launches do not find their map in the 256-MiB infinity cache
NBUF = max(2, int(1.5e9 // (B * H * W * C * 2 * 3)))
xs = [torch.randn(B, H, W, C, generator=g).to(dt).cuda() for _ in range(min(NBUF, 4))]
while len(xs) < NBUF:
    xs.append(xs[len(xs) % 4].clone())
rs = [x.clone() for x in xs]
outs = [torch.empty_like(x) for x in xs]
w = (torch.randn(C, 3, 3, C, generator=g) / 24).to(dt).cuda()
bias = torch.randn(C, generator=g).cuda()
zeros = torch.zeros(256, dtype=dt, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def conv(i, res, act):
    k = i % NBUF
    _lib.check(lib.fvit_conv3x3_nhwc(code, xs[k].data_ptr(), w.data_ptr(), bias.data_ptr(), rs[k].data_ptr() if res else None,
                                     outs[k].data_ptr(), B, H, W, C, C, 1, act, zeros.data_ptr(), st), "conv")


def run(name, res, act, n=24):
    if name.startswith("halo"):
        _lib.tune("conv_halo", 1)
        _lib.tune("conv_halo_grid", int(name.split(":")[1]) if ":" in name else 512)
        _lib.tune("conv_halo_ablate", int(name.split("a")[2]) if name.count("a") > 1 else 0)   # e.g. haloa3 = ablate bits 1|2
    else:
        _lib.tune("conv_halo", 0)
        _lib.tune("conv_ablate", int(name[5:]) if name.startswith("gemma") else 0)   # e.g. gemma3 = conv_ablate bits 1|2
        _lib.tune("conv128_narrow", 1 if name == "gemmn" else 0)      # "gemmn": 128 x 64 tiles; everything else the library default (128 x 128).  (r01-r06 passed -1 for "auto",
        #                                                                 which the library reads as non-zero = narrow: calls 29 / 30 of r06 measured the 128 x 64 form under the name "gemm")
    for i in range(3):
        conv(i, res, act)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        conv(i, res, act)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    flops = 2.0 * B * H * W * C * C * 9
    byts = B * H * W * C * 2 * (3 if res else 2)
    print(f"{name:10s} res={int(res)} act={act}: {us:8.1f} us  {flops / us * 1e-6:7.1f} TFLOP/s  {byts / us * 1e-3:7.1f} GB/s (algorithmic)", flush=True)
    return us


print(f"conv3x3 {C}->{C} s1, B={B} {H}x{W}, {NBUF} rotating buffers")
for rnd in range(2):
    print("round", rnd + 1)
    for res, act in ((False, 2), (True, 0)):
        for v in variants:
            run(v, res, act)
if C != 64:
    sys.exit(0)
# cross-check the two kernels against each other on the last buffers
_lib.tune("conv_halo", 0)
conv(0, True, 2)
torch.cuda.synchronize()
ref = outs[0].clone()
_lib.tune("conv_halo", 1)
_lib.tune("conv_halo_grid", 512)
_lib.tune("conv_halo_ablate", 0)
_lib.tune("conv_halo_debug", 1)
conv(0, True, 2)
torch.cuda.synchronize()
print("max |halo - gemm| =", (outs[0].float() - ref.float()).abs().max().item(), " max |ref| =", ref.float().abs().max().item())
