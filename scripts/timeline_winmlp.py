"""Phase timeline of the N-split MLP kernel (fvit_debug_win_mlp_timeline): where the time of a workgroup goes.
usage: python scripts/timeline_winmlp.py"""
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
dt = torch.float16
NAMES = ["entry->ring issued", "ring issued->rows loaded", "rows loaded->LN published", "LN->end sc0"] + \
        [f"sc{i}->sc{i + 1}" for i in range(9)] + ["last sc->pre-epilogue", "epilogue"]
for C, M, nw in ((256, 18240, 4), (512, 4214, 8), (256, 54272, 4), (512, 12544, 8)):
    hid = 4 * C
    g = torch.Generator(device="cpu").manual_seed(0)
    xs = [torch.randn(M, C, generator=g).cuda() for _ in range(3)]
    lnw, lnb = torch.ones(C).cuda(), torch.zeros(C).cuda()
    w1 = hat_runtime.frag_pack_fc1(torch.randn(hid, C, generator=g) / C ** 0.5).to(dt).cuda().contiguous()
    w2 = hat_runtime.frag_pack_fc2(torch.randn(C, hid, generator=g) / hid ** 0.5).to(dt).cuda().contiguous()
    b1, b2 = torch.zeros(hid).cuda(), torch.zeros(C).cuda()
    nwg = (M + 63) // 64
    ts = torch.zeros(nwg * nw * 16, dtype=torch.int64, device="cuda")
    junk = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")

    def call(x, stamps):
        if stamps is None:
            _lib.check(lib.fvit_win_mlp_fused(1, x.data_ptr(), M, C, hid, lnw.data_ptr(), lnb.data_ptr(), ctypes.c_float(1e-5), w1.data_ptr(),
                                              b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), None, st), "win_mlp")
        else:
            _lib.check(lib.fvit_debug_win_mlp_timeline(x.data_ptr(), M, C, hid, lnw.data_ptr(), lnb.data_ptr(), ctypes.c_float(1e-5), w1.data_ptr(),
                                                       b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), None, stamps.data_ptr(), st), "timeline")
    for i in range(3):
        call(xs[i % 3], None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for cold in (False, True):
        if cold:
            junk.fill_(1)   # evict the weights / rows from L2 and the Infinity Cache
        torch.cuda.synchronize()
        e0.record()
        call(xs[1], ts)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        raw = ts.view(nwg, nw, 16).double()
        # s_memtime counters are per XCD and not aligned with each other: only differences inside one wave mean anything
        dur = raw[..., 15] - raw[..., 0]
        ghz = dur.max().item() / us / 1e3     # the longest wave spans (nearly) the whole launch
        t = raw
        print(f"\nC={C} M={M} ({nwg} workgroups x {nw} waves) {'COLD' if cold else 'warm'}: launch {us:.1f} us; longest wave {dur.max().item():.0f} ticks "
              f"-> {ghz:.2f} ticks/ns; wave duration mean {dur.mean().item() / ghz / 1e3:.1f} us, min {dur.min().item() / ghz / 1e3:.1f} us")
        nsc = hid // 32 // nw
        idx = [0, 1, 2, 3] + [4 + i for i in range(min(nsc, 10))] + [14, 15]
        for a, b in zip(idx[:-1], idx[1:]):
            d = (t[..., b] - t[..., a]) / ghz / 1e3
            name = NAMES[a] if a < 13 else NAMES[13 if a < 14 else 14]
            if b == 14:
                name = "last sc -> pre-epilogue"
            if a == 14:
                name = "epilogue"
            dd = d.reshape(-1)
            print(f"  {a:2d}->{b:2d} {name:28s} mean {dd.mean().item():7.2f} us   max {dd.max().item():7.2f} us   (n {dd.numel()})")
