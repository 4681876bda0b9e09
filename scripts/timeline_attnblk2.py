"""Phase timeline of attnblk2_kernel (wave per (window, head); stage-2 window attention of FasterViT-0) at shard size, through
fvit_debug_attn_block_timeline.  s_memtime is per XCD and unsynchronized: only differences inside one wave are used."""
import os
os.environ.setdefault("FVIT_DIAG", "1")
import ctypes
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastervit_amd import _lib, hat_runtime  # noqa: E402

lib = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 512
C, heads, S, dt = 256, 8, 53, torch.float16
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
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for v in (0, 3):
    _lib.tune("ab_variant", v)
    for _ in range(3):
        _lib.check(lib.fvit_attn_block_fused(1, *args, st), "attn_block")
    e0.record()
    for _ in range(20):
        _lib.check(lib.fvit_attn_block_fused(1, *args, st), "attn_block")
    e1.record()
    torch.cuda.synchronize()
    print(f"ab_variant {v} x {nwin} windows: {e0.elapsed_time(e1) * 1e3 / 20:.1f} us per launch (production instance)")
nwg = (nwin + 1) // 2
ts = torch.zeros(nwin * 4 * 16, dtype=torch.int64, device="cuda")
for _ in range(2):
    e0.record()
    _lib.check(lib.fvit_debug_attn_block_timeline(*args, ts.data_ptr(), st), "timeline")
    e1.record()
    torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
t = ts[: nwg * 64].view(nwg, 4, 16).double()
tot = t[..., 10] - t[..., 0]
rate = tot.max().item() / us / 1e3
print(f"timeline launch {us:.1f} us, {rate:.3f} ticks/ns; wave total mean {tot.mean().item() / rate / 1e3:.2f} us, max {tot.max().item() / rate / 1e3:.2f}")
names = ["row table + small tables (barrier)", "rows gathered", "LayerNorm, fragments published (2 barriers + 1)", "q k v of head 0 (8 k steps, 384 MFMAs)",
         "attention of head 0 (2 windows)", "q k v of head 1", "attention of head 1", "barrier: O of every wave visible (+ proj weights)",
         "residual rows re-read + proj (256 MFMAs)", "epilogue stores drained"]
for i, nm in enumerate(names):
    v = (t[..., i + 1] - t[..., i]) / rate / 1e3
    print(f"   {nm:52s} mean {v.mean().item():6.2f} us   min {v.min().item():6.2f}   max {v.max().item():6.2f}")
