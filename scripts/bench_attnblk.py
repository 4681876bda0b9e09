"""Micro-benchmark of the fused attention block kernel (with timing-only ablations) vs the unfused 4-kernel path."""
import os
os.environ.setdefault("FVIT_DIAG", "1")   # diagnosis build of the library (fvit_debug_* entry points, ablation knobs)
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastervit_amd import _lib, hat_runtime  # noqa: E402

lib = _lib.lib()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 53
nwin = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
variants = sys.argv[3].split(",") if len(sys.argv) > 3 else ["a0", "a0v1", "a15", "a15v1", "unfused"]
C, heads, d = 256, 8, 32
dt, code = torch.float16, 1
g = torch.Generator(device="cpu").manual_seed(0)
rows = nwin * S
X = torch.randn(rows, C, generator=g).cuda()
lnw, lnb = torch.ones(C).cuda(), torch.zeros(C).cuda()
wqkv = (torch.randn(3 * C, C, generator=g) / 16).to(dt).cuda()
bqkv = torch.zeros(3 * C).cuda()
wproj = (torch.randn(C, C, generator=g) / 16).to(dt).cuda()
bproj = torch.zeros(C).cuda()
gamma = torch.full((C,), 0.01).cuda()
spad = lib.fvit_attention_spad(S)
bp = torch.zeros(heads, spad, spad, device="cuda")
bp[:, :, S:] = _lib.FVIT_MASK_BIAS
wqf = hat_runtime.frag_pack_qkv(wqkv.float(), heads).to(dt).contiguous()
bqh = bqkv.view(3, heads, 32).permute(1, 0, 2).reshape(heads, 96).contiguous()
wpf = hat_runtime.frag_pack_fc2(wproj.float()).to(dt).contiguous()
Mp = (rows + 127) // 128 * 128
xn = torch.zeros(Mp, C, dtype=dt, device="cuda")
qkv = torch.zeros(Mp, 3 * C, dtype=dt, device="cuda")
ao = torch.zeros(Mp, C, dtype=dt, device="cuda")
wq_p = torch.zeros(768, C, dtype=dt, device="cuda"); wq_p[:] = wqkv
wp_p = torch.zeros(256, C, dtype=dt, device="cuda"); wp_p[:] = wproj
st = torch.cuda.current_stream().cuda_stream
eps, scale = ctypes.c_float(1e-5), ctypes.c_float(d ** -0.5)


def fused():
    _lib.check(lib.fvit_attn_block_fused(code, X.data_ptr(), S, None, 0, None, None, None, lnw.data_ptr(), lnb.data_ptr(), eps, S,
                                         wqf.data_ptr(), bqh.data_ptr(), wpf.data_ptr(), bproj.data_ptr(), gamma.data_ptr(), bp.data_ptr(),
                                         X.data_ptr(), nwin, S, heads, C, scale, st), "fused")


def unfused():
    _lib.check(lib.fvit_gather_layernorm(code, X.data_ptr(), 0, None, 0, None, None, None, X.data_ptr(), xn.data_ptr(), C, lnw.data_ptr(),
                                         lnb.data_ptr(), eps, rows, 1, C, st), "ln")
    _lib.check(lib.fvit_gemm_bias_act(code, xn.data_ptr(), C, wq_p.data_ptr(), C, bqkv.data_ptr(), qkv.data_ptr(), 3 * C, rows, 3 * C, C, 0, st), "qkv")
    _lib.check(lib.fvit_window_attention(code, qkv.data_ptr(), 3 * C, ao.data_ptr(), C, bp.data_ptr(), nwin, S, heads, 32, scale, st), "attn")
    _lib.check(lib.fvit_gemm_residual(code, ao.data_ptr(), C, wp_p.data_ptr(), C, bproj.data_ptr(), gamma.data_ptr(), X.data_ptr(), C, rows, C, C, st), "proj")


def run(name, n=20):
    if name == "unfused":
        fn = unfused
    else:
        _lib.tune("ab_stagger", 1 if name.endswith("g") else 0)
        name = name.rstrip("g")
        _lib.tune("ab_ablate", int(name[1:].split("v")[0]))
        _lib.tune("ab_variant", int(name.split("v")[1]) if "v" in name else 0)
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


flops = 2.0 * rows * C * 4 * C + 4.0 * nwin * heads * S * S * 32
for rnd in range(2):
    for v in variants:
        us = run(v)
        print(f"round {rnd} {v:8s} S={S} nwin={nwin}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
