"""Micro-benchmark of the fused MLP kernel vs the unfused LN + 2 GEMM path (interleaved A/B in one process)."""
import os
os.environ.setdefault("FVIT_DIAG", "1")   # diagnosis build of the library (fvit_debug_* entry points, ablation knobs)
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastervit_amd import _lib, hat_runtime  # noqa: E402

lib = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 54272
variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["v0", "v1", "v2", "v3", "v4", "unfused"]
C, hid = 256, 1024
dt, code = torch.float16, 1
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(M, C, generator=g).cuda()
lnw, lnb = torch.ones(C).cuda(), torch.zeros(C).cuda()
w1 = (torch.randn(hid, C, generator=g) / 16).to(dt).cuda()
w2 = (torch.randn(C, hid, generator=g) / 32).to(dt).cuda()
b1, b2 = torch.zeros(hid).cuda(), torch.zeros(C).cuda()
gamma = torch.full((C,), 0.01).cuda()
w1f = hat_runtime.frag_pack_fc1(w1.float()).to(dt).contiguous()
w2c = hat_runtime.frag_pack_fc2(w2.float()).to(dt).contiguous()
Mp = (M + 127) // 128 * 128
xn = torch.zeros(Mp, C, dtype=dt, device="cuda")
h = torch.zeros(Mp, hid, dtype=dt, device="cuda")
st = torch.cuda.current_stream().cuda_stream
eps = ctypes.c_float(1e-5)


def fused():
    _lib.check(lib.fvit_mlp_fused(code, x.data_ptr(), M, C, hid, lnw.data_ptr(), lnb.data_ptr(), eps, w1f.data_ptr(), b1.data_ptr(),
                                  w2c.data_ptr(), b2.data_ptr(), gamma.data_ptr(), st), "fused")


def unfused():
    _lib.check(lib.fvit_gather_layernorm(code, x.data_ptr(), 0, None, 0, None, None, None, None, xn.data_ptr(), C, lnw.data_ptr(),
                                         lnb.data_ptr(), eps, M, 1, C, st), "ln")
    _lib.check(lib.fvit_gemm_bias_act(code, xn.data_ptr(), C, w1.data_ptr(), C, b1.data_ptr(), h.data_ptr(), hid, M, hid, C, 1, st), "fc1")
    _lib.check(lib.fvit_gemm_residual(code, h.data_ptr(), hid, w2.data_ptr(), hid, b2.data_ptr(), gamma.data_ptr(), x.data_ptr(), C,
                                      M, C, hid, st), "fc2")


def run(name, n=20):
    if name == "unfused":
        fn = unfused
    else:
        _lib.tune("mlp_variant", int(name[1:2]))
        _lib.tune("mlp_ablate", int(name.split("a")[1]) if "a" in name else 0)
        _lib.tune("mlp_stagger", 0 if name.endswith("s0") else (1 if name.endswith("s1") else 2))
        fn = fused
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


flops = 4.0 * M * C * hid
for rnd in range(3):
    for v in variants:
        us = run(v)
        print(f"round {rnd} {v:10s} M={M}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
