"""Phase timeline of attnblk_kernel<256> (stage-2 window attention of FasterViT-0) at shard size: fvit_debug_attn_block_timeline.
s_memtime is per XCD and unsynchronized: only differences inside one wave are used; the tick rate comes from the longest wave against
the launch's event time."""
import os
os.environ.setdefault("FVIT_DIAG", "1")   # diagnosis build of the library (fvit_debug_* entry points, ablation knobs)
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastervit_amd import _lib, hat_runtime  # noqa: E402

lib = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 344
C, heads, S, dt = 256, 8, 50, torch.float16
g = torch.Generator(device="cpu").manual_seed(0)
X = (torch.randn(nwin * S, C, generator=g) * 1.3).cuda()
lnw, lnb = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
wqkv = (torch.randn(3 * C, C, generator=g) / C ** 0.5).cuda()
bqkv = (torch.randn(3 * C, generator=g) * 0.3).cuda()
wproj = (torch.randn(C, C, generator=g) / C ** 0.5).cuda()
bproj = (torch.randn(C, generator=g) * 0.3).cuda()
gamma = (torch.rand(C, generator=g) + 0.5).cuda()
bp = torch.zeros(heads, 64, 64, device="cuda")
bp[:, :S, :S] = (torch.randn(heads, S, S, generator=g) * 2).cuda()
bp[:, :, S:] = _lib.FVIT_MASK_BIAS
wqf = hat_runtime.frag_pack_qkv(wqkv, heads).to(dt).contiguous()
bqh = bqkv.view(3, heads, 32).permute(1, 0, 2).reshape(heads, 96).contiguous()
wpf = hat_runtime.frag_pack_fc2(wproj).to(dt).contiguous()
out = torch.empty_like(X)
args = (X.data_ptr(), S, None, 0, None, None, None, lnw.data_ptr(), lnb.data_ptr(), ctypes.c_float(1e-5), S, wqf.data_ptr(), bqh.data_ptr(), wpf.data_ptr(),
        bproj.data_ptr(), gamma.data_ptr(), bp.data_ptr(), out.data_ptr(), nwin, S, heads, C, ctypes.c_float(32 ** -0.5))
for _ in range(3):
    _lib.check(lib.fvit_attn_block_fused(1, *args, st), "attn_block")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    _lib.check(lib.fvit_attn_block_fused(1, *args, st), "attn_block")
e1.record()
torch.cuda.synchronize()
print(f"attnblk_kernel<256,S64> x {nwin} windows: {e0.elapsed_time(e1) * 1e3 / 20:.1f} us per launch (production instance)")
ts = torch.zeros(nwin * 4 * 16, dtype=torch.int64, device="cuda")
for _ in range(2):
    e0.record()
    _lib.check(lib.fvit_debug_attn_block_timeline(*args, ts.data_ptr(), st), "timeline")
    e1.record()
    torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
t = ts.view(nwin, 4, 16).double()
tot = t[..., 15] - t[..., 0]
rate = tot.max().item() / us / 1e3
print(f"timeline launch {us:.1f} us, {rate:.3f} ticks/ns; wave total mean {tot.mean().item() / rate / 1e3:.2f} us, max {tot.max().item() / rate / 1e3:.2f}")


def row(name, a, b):
    v = (t[..., b] - t[..., a]) / rate / 1e3
    print(f"   {name:44s} mean {v.mean().item():6.2f} us   min {v.min().item():6.2f}   max {v.max().item():6.2f}")


row("entry -> rows gathered (first slice requested)", 0, 1)
row("LayerNorm -> fragments", 1, 2)
row("head 0 (incl. waiting for its weight slice)", 2, 3)
for h in range(1, 8):
    row(f"head {h}", 2 + h, 3 + h)
row("epilogue (residual, stores, drain)", 14, 15)
wg = tot.max(dim=1).values / rate / 1e3
print(f"   workgroup totals: 10 % {wg.quantile(0.1).item():.2f} us, median {wg.median().item():.2f}, 90 % {wg.quantile(0.9).item():.2f}")
