"""Micro-benchmark of the carrier-branch kernel (fvit_ct_block_fused): warm weights (same launch repeated) vs cold (L2 thrashed
between launches), per variant."""
import os
os.environ.setdefault("FVIT_DIAG", "1")   # diagnosis build of the library (fvit_debug_* entry points, ablation knobs)
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastervit_amd import _lib, hat_runtime  # noqa: E402

lib = _lib.lib()
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 86
C, heads, hid, G, dt, code = 256, 8, 1024, 16, torch.float16, 1
g = torch.Generator(device="cpu").manual_seed(0)
rowsA = 4 * 53
X = torch.randn(batch * rowsA, C, generator=g).cuda()
src_idx = torch.randperm(rowsA, generator=g)[:G].int().cuda()
add = torch.randn(G, C, generator=g).cuda()
ones, zeros = torch.ones(C).cuda(), torch.zeros(C).cuda()
wqkv = (torch.randn(3 * C, C, generator=g) / 16).to(dt).cuda()
wproj = (torch.randn(C, C, generator=g) / 16).to(dt).cuda()
w1 = (torch.randn(hid, C, generator=g) / 16).to(dt).cuda()
w2 = (torch.randn(C, hid, generator=g) / 32).to(dt).cuda()
bq, bp, b1, b2 = torch.zeros(heads * 96).cuda(), torch.zeros(C).cuda(), torch.zeros(hid).cuda(), torch.zeros(C).cuda()
bias = torch.zeros(heads, 16, 16).cuda()
wqf = hat_runtime.frag_pack_qkv(wqkv.float(), heads).to(dt).contiguous()
wpf = hat_runtime.frag_pack_fc2(wproj.float()).to(dt).contiguous()
w1f = hat_runtime.frag_pack_fc1(w1.float()).to(dt).contiguous()
w2f = hat_runtime.frag_pack_fc2(w2.float()).to(dt).contiguous()
R = torch.zeros(batch * G, C).cuda()
st = torch.cuda.current_stream().cuda_stream
thrash = torch.zeros(256 << 20, dtype=torch.uint8, device="cuda")


def launch():
    _lib.check(lib.fvit_ct_block_fused(code, X.data_ptr(), rowsA, src_idx.data_ptr(), add.data_ptr(), R.data_ptr(), batch, G, heads, C, hid,
                                       ones.data_ptr(), zeros.data_ptr(), wqf.data_ptr(), bq.data_ptr(), wpf.data_ptr(), bp.data_ptr(), None,
                                       bias.data_ptr(), ctypes.c_float(32 ** -0.5), ones.data_ptr(), zeros.data_ptr(), w1f.data_ptr(), b1.data_ptr(),
                                       w2f.data_ptr(), b2.data_ptr(), None, ctypes.c_float(1e-5), st), "ct")


for variant, touch in ((0, 0), (2, 0), (3, 2), (3, 3), (3, 4)):   # 3: the 8-wave form; "touch" there = ring depth (ct8_depth)
    _lib.tune("ct_variant", variant)
    _lib.tune("ct_touch", touch if variant != 3 else 0)
    _lib.tune("ct8_depth", touch if variant == 3 else 3)
    for cold in (False, True):
        ts = []
        for _ in range(12):
            if cold:
                thrash.add_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1000)
        ts = sorted(ts[2:])
        print(f"batch {batch} variant {variant} touch {touch} {'cold (L2 thrashed)' if cold else 'warm (repeated)   '}: median {ts[len(ts) // 2]:6.1f} us  min {ts[0]:6.1f}", flush=True)
_lib.tune("ct_variant", 3)
_lib.tune("ct_touch", 0)

# phase timeline of the 8-wave form (fvit_debug_ct_block_timeline); s_memtime differences inside a wave only
ts = torch.zeros(batch * 8 * 16, dtype=torch.int64, device="cuda")
for _ in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(lib.fvit_debug_ct_block_timeline(X.data_ptr(), rowsA, src_idx.data_ptr(), add.data_ptr(), R.data_ptr(), batch, G, heads, C, hid,
                                                ones.data_ptr(), zeros.data_ptr(), wqf.data_ptr(), bq.data_ptr(), wpf.data_ptr(), bp.data_ptr(), None,
                                                bias.data_ptr(), ctypes.c_float(32 ** -0.5), ones.data_ptr(), zeros.data_ptr(), w1f.data_ptr(), b1.data_ptr(),
                                                w2f.data_ptr(), b2.data_ptr(), None, ctypes.c_float(1e-5), ts.data_ptr(), st), "timeline")
    e1.record()
    torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
t = ts.view(batch, 8, 16).double()
tot = t[..., 12] - t[..., 0]
rate = tot.max().item() / us / 1e3
print(f"timeline (8-wave form, warm): launch {us:.1f} us, {rate:.3f} ticks/ns; wave total mean {tot.mean().item() / rate / 1e3:.2f} us, max {tot.max().item() / rate / 1e3:.2f}")
pairs = [("rows + constants landed, constants -> LDS, ring started", 0, 1), ("barrier", 1, 3), ("LayerNorm 1", 3, 2), ("qkv (6 steps) + attention", 2, 4),
         ("barrier", 4, 5), ("proj (2 steps) + residual", 5, 6), ("barrier", 6, 7), ("LayerNorm 2 (rows from LDS)", 7, 8), ("fc1 (8 steps) + GELU", 8, 9),
         ("barrier", 9, 10), ("fc2 (8 steps)", 10, 11), ("residual + stores + drain", 11, 12)]
for name, a, b in pairs:
    v = (t[..., b] - t[..., a]) / rate / 1e3
    print(f"   {name:60s} mean {v.mean().item():6.2f} us   min {v.min().item():6.2f}   max {v.max().item():6.2f}")
