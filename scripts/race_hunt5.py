"""Fused MLP kernel alone on 3 concurrent streams (separate X buffers, shared weights): run-to-run repeatability per knob setting."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastervit_amd import _lib, hat_runtime  # noqa: E402

lib = _lib.lib()
C, hid = 256, 1024
dt, code = torch.float16, 1
g = torch.Generator(device="cpu").manual_seed(0)
Ms = (18020, 18020, 17808)
x0 = [torch.randn(M, C, generator=g).cuda() for M in Ms]
lnw, lnb = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
w1 = (torch.randn(hid, C, generator=g) / 16).to(dt).cuda()
w2 = (torch.randn(C, hid, generator=g) / 32).to(dt).cuda()
b1, b2 = (torch.randn(hid, generator=g) * 0.1).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
gamma = torch.full((C,), 0.5).cuda()
w1f = hat_runtime.frag_pack_fc1(w1.float()).to(dt).contiguous()
w2f = hat_runtime.frag_pack_fc2(w2.float()).to(dt).contiguous()
eps = ctypes.c_float(1e-5)
streams = [torch.cuda.Stream() for _ in range(3)]
# some concurrent "noise" traffic on a 4th stream, like the other shards' conv kernels
noise_s = torch.cuda.Stream()
big = torch.randn(64 * 1024 * 1024, device="cuda")


def run(tag, knobs, reps=12, noise=False):
    for k, v in knobs.items():
        _lib.tune(k, v)
    outs = []
    for r in range(reps):
        xs = [t.clone() for t in x0]
        torch.cuda.synchronize()
        if noise:
            with torch.cuda.stream(noise_s):
                for _ in range(4):
                    big.mul_(1.0001)
        for s, x, M in zip(streams, xs, Ms):
            with torch.cuda.stream(s):
                _lib.check(lib.fvit_mlp_fused(code, x.data_ptr(), M, C, hid, lnw.data_ptr(), lnb.data_ptr(), eps, w1f.data_ptr(), b1.data_ptr(),
                                              w2f.data_ptr(), b2.data_ptr(), gamma.data_ptr(), s.cuda_stream), "fused")
        torch.cuda.synchronize()
        outs.append(xs)
    bad = [sum(0 if torch.equal(outs[r][i], outs[0][i]) else 1 for r in range(1, reps)) for i in range(3)]
    md = max((outs[r][i] - outs[0][i]).abs().max().item() for r in range(1, reps) for i in range(3))
    print(f"{tag}: mismatching repeats per stream {bad}, max diff {md:.3e}", flush=True)
    for k in knobs:
        _lib.tune(k, {"mlp_stagger": 2, "mlp_ablate": 0, "mlp_variant": -1}[k])


run("stagger 2", {"mlp_stagger": 2})
run("stagger 2 + noise", {"mlp_stagger": 2}, noise=True)
run("stagger 0", {"mlp_stagger": 0})
run("stagger 0 + noise", {"mlp_stagger": 0}, noise=True)
run("stagger 1", {"mlp_stagger": 1})
run("stagger 2, extra end barrier", {"mlp_stagger": 2, "mlp_ablate": 16})
run("stagger 2, synchronous DMA", {"mlp_stagger": 2, "mlp_ablate": 32})
run("stagger 2, variant 4 (no KEEPX)", {"mlp_stagger": 2, "mlp_variant": 4})
# single stream
outs = []
for r in range(8):
    x = x0[0].clone()
    _lib.check(lib.fvit_mlp_fused(code, x.data_ptr(), Ms[0], C, hid, lnw.data_ptr(), lnb.data_ptr(), eps, w1f.data_ptr(), b1.data_ptr(), w2f.data_ptr(),
                                  b2.data_ptr(), gamma.data_ptr(), torch.cuda.current_stream().cuda_stream), "fused")
    torch.cuda.synchronize()
    outs.append(x)
print("single stream, stagger 2: mismatching repeats", sum(0 if torch.equal(o, outs[0]) else 1 for o in outs[1:]))
