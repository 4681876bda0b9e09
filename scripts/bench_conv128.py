"""Micro-benchmark: row-band 128 -> 128 conv (fvit_conv3x3_c128_band) vs the implicit-GEMM kernel at level-1 shapes of FasterViT-0."""
import os
os.environ.setdefault("FVIT_DIAG", "1")   # diagnosis build of the library (fvit_debug_* entry points, ablation knobs)
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastervit_amd import _lib  # noqa: E402
from fastervit_amd.conv_runtime import frag_pack_conv128  # noqa: E402

lib = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
dt, code = torch.float16, 1
g = torch.Generator(device="cpu").manual_seed(0)
for B in [int(a) for a in sys.argv[1:]] or [86, 256]:
    H = W = 28
    xs = [torch.randn(B, 128, H, W, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last) for _ in range(3)]
    w = (torch.randn(128, 128, 3, 3, generator=g) / 34).to(dt).cuda()
    wk = w.permute(0, 2, 3, 1).contiguous()
    wf = frag_pack_conv128(wk.reshape(128, 1152))
    bias = torch.randn(128, generator=g).cuda()
    zeros = torch.zeros(256, dtype=dt, device="cuda")
    out = torch.empty_like(xs[0])
    res = torch.randn(B, 128, H, W, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last)

    def band(i, r):
        _lib.check(lib.fvit_conv3x3_c128_band(code, xs[i % 3].data_ptr(), wf.data_ptr(), bias.data_ptr(), res.data_ptr() if r else None,
                                              (res if r else out).data_ptr(), B, H, W, 0 if r else 2, zeros.data_ptr(), st), "band")

    def igemm(i, r):
        _lib.check(lib.fvit_conv3x3_nhwc(code, xs[i % 3].data_ptr(), wk.data_ptr(), bias.data_ptr(), res.data_ptr() if r else None,
                                         (res if r else out).data_ptr(), B, H, W, 128, 128, 1, 0 if r else 2, zeros.data_ptr(), st), "igemm")

    for rnd in range(2):
        for name, fn in (("band", band), ("implicit GEMM", igemm)):
            for r in (False, True):
                for i in range(3):
                    fn(i, r)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(20):
                    fn(i, r)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 20
                print(f"B={B} {name:14s} {'residual' if r else 'GELU    '}: {us:7.1f} us  {2 * B * H * W * 128 * 1152 / us / 1e6:7.1f} TFLOP/s", flush=True)

# phase timeline of the band kernel (fvit_debug_conv_band_timeline): s_memtime is per XCD, so only differences inside a wave are used
import ctypes as C  # noqa: E402

B, H, W = 86, 28, 28
x = torch.randn(B, 128, H, W, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last)
res = torch.randn(B, 128, H, W, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last)
bands = (H + 224 // (W + 2) - 1) // (224 // (W + 2))
ts = torch.zeros(B * bands * 4 * 8, dtype=torch.int64, device="cuda")
for r in (False, True):
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.fvit_debug_conv_band_timeline(x.data_ptr(), wf.data_ptr(), bias.data_ptr(), res.data_ptr() if r else None,
                                                     (res if r else out[:B]).data_ptr(), B, H, W, 0 if r else 2, zeros.data_ptr(), ts.data_ptr(), st), "tl")
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    t = ts.view(B * bands, 4, 8).double()
    d = t[..., 1:6] - t[..., 0:5]
    tot = t[..., 5] - t[..., 0]
    ghz = tot.max().item() / us / 1e3   # ticks per ns, from the longest wave against the launch's event time (upper bound of the rate)
    print(f"timeline ({'residual' if r else 'GELU'}): launch {us:.1f} us, {ghz:.3f} ticks/ns; wave total mean {tot.mean().item() / ghz / 1e3:.2f} us, max {tot.max().item() / ghz / 1e3:.2f} us")
    for i, name in enumerate(["request band + first weight steps", "wait for the band (barrier)", "K loop (36 steps)", "epilogue first half", "epilogue second half + drain"]):
        v = d[..., i] / ghz / 1e3
        print(f"   {name:36s} mean {v.mean().item():6.2f} us   min {v.min().item():6.2f}   max {v.max().item():6.2f}")
    slow = tot.view(-1, 4).max(dim=1).values
    print(f"   workgroup totals: 10 % {slow.quantile(0.1).item() / ghz / 1e3:.2f} us, median {slow.median().item() / ghz / 1e3:.2f}, 90 % {slow.quantile(0.9).item() / ghz / 1e3:.2f}")
