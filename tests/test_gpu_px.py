"""Two-term MAPS of the conv side and the "precise" deploy plan (r05; needs an MI355X).

A conv-side stream is a pair of 16-bit planes (value = hi + lo) or an fp32 map; conv weights are hi + lo; the Downsample convs and the first
stem conv take their input as two terms as well.  Kernel tests compare the px entry points with fp64 torch on the SAME two-term values (so
only the kernels' own arithmetic is measured: fp32 accumulation, the dropped lo.lo product, the final hi / lo split); the model tests check
the plan that bench.py times for BASELINE configs 3 and 5 against the reference's logits in ABSOLUTE terms.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fastervit_amd import _lib, hat_runtime
from tests.cases import CASES
from tests.util import build_product_model, case_input, load_golden, max_abs

pytestmark = pytest.mark.gpu


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _split(t, dt=torch.float16):
    hi = t.to(dt)
    lo = (t - hi.float()).to(dt)
    return hi, lo


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("B,Ci,Co,H,W,stride,act,in2,res,out", [
    (2, 64, 64, 14, 14, 1, 2, False, 0, "single"),     # ConvBlock conv1: one-term operand, two-term weights, erf GELU, one plane out
    (3, 64, 64, 9, 13, 1, 0, False, 2, "planes"),       # ConvBlock conv2: two-term residual, in place, two planes out
    (2, 128, 128, 7, 9, 1, 0, False, 1, "planes"),      # one-term residual
    (2, 64, 128, 12, 10, 2, 0, True, 0, "planes"),      # Downsample between conv levels: two-term input (three K segments), planes out
    (2, 128, 256, 14, 14, 2, 0, True, 0, "f32"),        # Downsample in front of a transformer level: fp32 out
    (1, 256, 448, 11, 9, 2, 0, True, 0, "f32"),         # Cout % 128 != 0: the 128 x 64 tile instance (FasterViT-4: 392 -> padded 448)
    (2, 64, 64, 16, 16, 2, 1, False, 0, "planes"),      # second stem conv: ReLU, stride 2
    (5, 64, 64, 56, 56, 1, 0, False, 2, "planes")])
def test_conv3x3_px(B, Ci, Co, H, W, stride, act, in2, res, out):
    lib = _lib.lib()
    dt = torch.float16
    g = torch.Generator(device="cpu").manual_seed(Ci + Co + H + stride)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    bias = torch.randn(Co, generator=g).cuda()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    xh, xl = _split(x)
    wh, wl = _split(w)
    xh, xl = _cl(xh.cuda()), _cl(xl.cuda())
    wk = torch.cat([wh.permute(0, 2, 3, 1).reshape(Co, -1), wl.permute(0, 2, 3, 1).reshape(Co, -1)], dim=1).contiguous().cuda()
    rh = rl = None
    if res:
        r = torch.randn(B, Co, Ho, Wo, generator=g) * 3
        rh, rl = _split(r)
        rh, rl = _cl(rh.cuda()), _cl(rl.cuda())
        if res == 1:
            rl = None
    zeros = torch.zeros(256, dtype=dt, device="cuda")
    nan = float("nan")
    oh = _cl(torch.full((B, Co, Ho, Wo), nan, dtype=dt, device="cuda")) if out != "f32" else None
    ol = _cl(torch.full((B, Co, Ho, Wo), nan, dtype=dt, device="cuda")) if out == "planes" else None
    of = _cl(torch.full((B, Co, Ho, Wo), nan, dtype=torch.float32, device="cuda")) if out == "f32" else None
    p = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
    rc = lib.fvit_conv3x3_nhwc_px(1, xh.data_ptr(), p(xl) if in2 else None, wk.data_ptr(), bias.data_ptr(), p(rh), p(rl), p(oh), p(ol), p(of),
                                  B, H, W, Ci, Co, stride, act, 2, zeros.data_ptr(), _stream())
    _lib.check(rc, "conv3x3_px")
    torch.cuda.synchronize()
    xin = xh.double() + (xl.double() if in2 else 0.0)
    ref = F.conv2d(xin.cpu(), (wh.double() + wl.double()), bias.double().cpu(), stride, 1)
    ref = [lambda t: t, torch.relu, lambda t: F.gelu(t)][act](ref)
    if res:
        ref = ref + rh.double().cpu() + (rl.double().cpu() if rl is not None else 0.0)
    scale = max(ref.abs().max().item(), 1.0)
    if out == "f32":
        got = of.double().cpu()
        tol = 2e-6 * scale          # fp32 accumulation of 9 Cin products + the dropped lo.lo term (2^-22 relative)
    elif out == "planes":
        got = oh.double().cpu() + ol.double().cpu()
        tol = 2e-6 * scale          # the two planes carry ~22 bits
    else:
        got = oh.double().cpu()
        tol = 6e-4 * scale          # one fp16 plane
    err = (got - ref).abs().max().item()
    print(f"conv3x3_px Ci {Ci} Co {Co} stride {stride} act {act} in2 {in2} res {res} out {out}: max-abs {err:.2e} (tol {tol:.1e})")
    assert torch.isfinite(got).all() and err < tol
    if res and out == "planes":   # in place over the residual planes: same bits
        rh2, rl2 = rh.clone(), (rl.clone() if rl is not None else _cl(torch.zeros_like(rh)))
        _lib.check(lib.fvit_conv3x3_nhwc_px(1, xh.data_ptr(), p(xl) if in2 else None, wk.data_ptr(), bias.data_ptr(), rh2.data_ptr(),
                                            rl2.data_ptr() if rl is not None else None, rh2.data_ptr(), rl2.data_ptr(), None, B, H, W, Ci, Co, stride,
                                            act, 2, zeros.data_ptr(), _stream()), "conv3x3_px in place")
        torch.cuda.synchronize()
        assert torch.equal(rh2, oh) and torch.equal(rl2, ol)
    # argument checks: a two-term input needs two-term weights; out_f32 excludes out_lo
    assert lib.fvit_conv3x3_nhwc_px(1, xh.data_ptr(), xl.data_ptr(), wk.data_ptr(), None, None, None, xh.data_ptr(), None, None, B, H, W, Ci, Co,
                                    stride, act, 1, zeros.data_ptr(), _stream()) != 0


@pytest.mark.parametrize("B,Ci,Cv,Co,H,W,stride,act,in2,res,out", [
    (2, 256, 200, 256, 14, 14, 1, 2, False, 0, "single"),   # FasterViT-4 level 0 conv1 (196 -> 200 of 256 channels)
    (1, 256, 200, 256, 9, 13, 1, 0, False, 2, "planes"),    # conv2: two-term residual in place
    (1, 448, 392, 832, 11, 9, 2, 0, True, 0, "f32"),        # Downsample 392 -> 784 in front of a transformer level: three K segments, fp32 out
    (2, 64, 24, 64, 12, 10, 2, 0, True, 0, "planes"),       # tiny models: a K step spans three taps, 128 x 64 tiles
    (3, 64, 16, 128, 7, 9, 1, 1, False, 1, "planes")])
def test_conv3x3_px_dense_k(B, Ci, Cv, Co, H, W, stride, act, in2, res, out):
    """r06: fvit_conv3x3_nhwc_px_dense (the two-term-map conv contracting over the real input channels only) vs F.conv2d in fp64; the pad channels
    of both input planes hold garbage that must never be read."""
    lib = _lib.lib()
    dt = torch.float16
    g = torch.Generator(device="cpu").manual_seed(Ci + Co + H + stride + Cv)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.zeros(Co, Ci, 3, 3)
    w[:, :Cv] = torch.randn(Co, Cv, 3, 3, generator=g) / (9 * Cv) ** 0.5
    bias = torch.randn(Co, generator=g).cuda()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    xh, xl = _split(x)
    xh_ref, xl_ref = xh.clone(), xl.clone()
    xh_ref[:, Cv:] = 0; xl_ref[:, Cv:] = 0
    xh[:, Cv:] = 777.0; xl[:, Cv:] = -3.0
    wh, wl = _split(w)
    xh, xl = _cl(xh.cuda()), _cl(xl.cuda())
    kd = lib.fvit_conv3x3_dense_k(Cv)

    def dense(m):
        d = torch.zeros(Co, kd, dtype=m.dtype)
        d[:, :9 * Cv] = m.permute(0, 2, 3, 1)[..., :Cv].reshape(Co, 9 * Cv)
        return d
    wk = torch.cat([dense(wh), dense(wl)], dim=1).contiguous().cuda()
    rh = rl = None
    if res:
        r = torch.randn(B, Co, Ho, Wo, generator=g) * 3
        rh, rl = _split(r)
        rh, rl = _cl(rh.cuda()), _cl(rl.cuda())
        if res == 1:
            rl = None
    zeros = torch.zeros(256, dtype=dt, device="cuda")
    nan = float("nan")
    oh = _cl(torch.full((B, Co, Ho, Wo), nan, dtype=dt, device="cuda")) if out != "f32" else None
    ol = _cl(torch.full((B, Co, Ho, Wo), nan, dtype=dt, device="cuda")) if out == "planes" else None
    of = _cl(torch.full((B, Co, Ho, Wo), nan, dtype=torch.float32, device="cuda")) if out == "f32" else None
    p = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
    _lib.check(lib.fvit_conv3x3_nhwc_px_dense(1, xh.data_ptr(), p(xl) if in2 else None, wk.data_ptr(), bias.data_ptr(), p(rh), p(rl), p(oh), p(ol), p(of),
                                              B, H, W, Ci, Cv, Co, stride, act, 2, zeros.data_ptr(), _stream()), "conv3x3_px_dense")
    torch.cuda.synchronize()
    xin = xh_ref.double() + (xl_ref.double() if in2 else 0.0)
    ref = F.conv2d(xin, (wh.double() + wl.double()), bias.double().cpu(), stride, 1)
    ref = [lambda t: t, torch.relu, lambda t: F.gelu(t)][act](ref)
    if res:
        ref = ref + rh.double().cpu() + (rl.double().cpu() if rl is not None else 0.0)
    scale = max(ref.abs().max().item(), 1.0)
    if out == "f32":
        got, tol = of.double().cpu(), 2e-6 * scale
    elif out == "planes":
        got, tol = oh.double().cpu() + ol.double().cpu(), 2e-6 * scale
    else:
        got, tol = oh.double().cpu(), 6e-4 * scale
    err = (got - ref).abs().max().item()
    print(f"conv3x3_px_dense Ci {Ci} Cv {Cv} Co {Co} stride {stride} act {act} in2 {in2} res {res} out {out}: max-abs {err:.2e} (tol {tol:.1e})")
    assert torch.isfinite(got).all() and err < tol
    assert lib.fvit_conv3x3_nhwc_px_dense(1, xh.data_ptr(), None, wk.data_ptr(), bias.data_ptr(), None, None, p(oh) or xh.data_ptr(), None, None,
                                          B, H, W, Ci, Cv + 4, Co, stride, act, 2, zeros.data_ptr(), _stream()) != 0


@pytest.mark.parametrize("B,Ci,Co,H,W,act,in2,res,out", [
    (2, 256, 256, 24, 32, 2, False, 0, "single"),     # ConvBlock conv1 on an exact patch grid
    (1, 256, 256, 20, 27, 0, False, 2, "planes"),     # conv2: two-term residual in place, ragged patches
    (1, 128, 192, 9, 13, 0, True, 1, "f32"),          # two-term input: the lo-plane blocks follow the hi-plane blocks; ragged N tile
    (2, 64, 128, 8, 16, 1, True, 0, "planes")])
def test_conv3x3_px_patch_form(B, Ci, Co, H, W, act, in2, res, out):
    """r06: the PATCH form (conv3x3_kernel<.., PX, HALO>) of fvit_conv3x3_nhwc_px vs F.conv2d in fp64: stride 1, two-term weights, optional two-term input."""
    lib = _lib.lib()
    dt = torch.float16
    g = torch.Generator(device="cpu").manual_seed(Ci + Co + H + W)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    bias = torch.randn(Co, generator=g).cuda()
    xh, xl = _split(x)
    wh, wl = _split(w)
    xh, xl = _cl(xh.cuda()), _cl(xl.cuda())
    wk = torch.cat([wh.permute(0, 2, 3, 1).reshape(Co, -1), wl.permute(0, 2, 3, 1).reshape(Co, -1)], dim=1).contiguous().cuda()
    rh = rl = None
    if res:
        r = torch.randn(B, Co, H, W, generator=g) * 3
        rh, rl = _split(r)
        rh, rl = _cl(rh.cuda()), _cl(rl.cuda())
        if res == 1:
            rl = None
    zeros = torch.zeros(256, dtype=dt, device="cuda")
    nan = float("nan")
    oh = _cl(torch.full((B, Co, H, W), nan, dtype=dt, device="cuda")) if out != "f32" else None
    ol = _cl(torch.full((B, Co, H, W), nan, dtype=dt, device="cuda")) if out == "planes" else None
    of = _cl(torch.full((B, Co, H, W), nan, dtype=torch.float32, device="cuda")) if out == "f32" else None
    p = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
    try:
        _lib.tune("conv_patch", 1)
        _lib.tune("conv_patch_max_waste_pct", 100000)
        assert lib.fvit_conv3x3_patch_form(B, H, W, Ci, Co, 1) == 1
        _lib.check(lib.fvit_conv3x3_nhwc_px(1, xh.data_ptr(), p(xl) if in2 else None, wk.data_ptr(), bias.data_ptr(), p(rh), p(rl), p(oh), p(ol), p(of),
                                            B, H, W, Ci, Co, 1, act, 2, zeros.data_ptr(), _stream()), "conv3x3_px patch form")
        torch.cuda.synchronize()
    finally:
        _lib.tune("conv_patch", 1)
        _lib.tune("conv_patch_max_waste_pct", 10)
    xin = xh.double() + (xl.double() if in2 else 0.0)
    ref = F.conv2d(xin.cpu(), (wh.double() + wl.double()), bias.double().cpu(), 1, 1)
    ref = [lambda t: t, torch.relu, lambda t: F.gelu(t)][act](ref)
    if res:
        ref = ref + rh.double().cpu() + (rl.double().cpu() if rl is not None else 0.0)
    scale = max(ref.abs().max().item(), 1.0)
    if out == "f32":
        got, tol = of.double().cpu(), 2e-6 * scale
    elif out == "planes":
        got, tol = oh.double().cpu() + ol.double().cpu(), 2e-6 * scale
    else:
        got, tol = oh.double().cpu(), 6e-4 * scale
    err = (got - ref).abs().max().item()
    print(f"conv3x3_px patch form Ci {Ci} Co {Co} {H}x{W} act {act} in2 {in2} res {res} out {out}: max-abs {err:.2e} (tol {tol:.1e})")
    assert torch.isfinite(got).all() and err < tol


@pytest.mark.parametrize("C,Cv,src", [(64, 64, "planes"), (256, 196, "planes"), (448, 392, "f32"), (832, 784, "f32"), (128, 128, "single"), (1600, 1568, "f32")])
def test_layernorm2d_px(C, Cv, src):
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(C + Cv)
    x = torch.zeros(2, 9, 7, C)
    x[..., :Cv] = torch.randn(2, 9, 7, Cv, generator=g) * 2 + 0.3
    w, b = torch.zeros(C), torch.zeros(C)
    w[:Cv] = torch.rand(Cv, generator=g) + 0.5
    b[:Cv] = torch.randn(Cv, generator=g)
    w, b = w.cuda(), b.cuda()
    xh, xl = _split(x)
    xh, xl, xf = xh.cuda(), xl.cuda(), x.cuda()
    oh = torch.full((2, 9, 7, C), float("nan"), dtype=torch.float16, device="cuda")
    ol = torch.full_like(oh, float("nan"))
    if src == "f32":
        args = (None, None, xf.data_ptr())
        val = xf.double()
    elif src == "planes":
        args = (xh.data_ptr(), xl.data_ptr(), None)
        val = xh.double() + xl.double()
    else:
        args = (xh.data_ptr(), None, None)
        val = xh.double()
    _lib.check(lib.fvit_layernorm2d_px(1, *args, oh.data_ptr(), ol.data_ptr(), w.data_ptr(), b.data_ptr(), ctypes.c_float(1e-6), 2 * 9 * 7, C, Cv,
                                       _stream()), "ln2d px")
    torch.cuda.synchronize()
    ref = F.layer_norm(val[..., :Cv], (Cv,), w[:Cv].double(), b[:Cv].double(), 1e-6)
    got = oh.double() + ol.double()
    err = (got[..., :Cv] - ref).abs().max().item()
    print(f"layernorm2d_px C {C} Cv {Cv} {src}: max-abs {err:.2e}")
    assert err < 3e-6 * max(ref.abs().max().item(), 1.0)
    assert got[..., Cv:].abs().max().item() == 0.0 if Cv < C else True
    assert lib.fvit_layernorm2d_px(1, xh.data_ptr(), None, xf.data_ptr(), oh.data_ptr(), None, w.data_ptr(), b.data_ptr(), ctypes.c_float(1e-6), 126, C, Cv,
                                   _stream()) != 0   # exactly one of in / in_f32


@pytest.mark.parametrize("in_dt,fmt,B,H,W", [(torch.float32, "nchw", 2, 32, 32), (torch.float32, "nhwc", 3, 30, 22), (torch.float16, "nhwc", 1, 64, 48)])
def test_stem_conv_px(in_dt, fmt, B, H, W):
    """First stem conv with two-term weights and the image split in registers vs fp64."""
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(H + W)
    x = torch.randn(B, 3, H, W, generator=g).to(in_dt).cuda()
    if fmt == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 3, 3, 3, generator=g) / 27 ** 0.5).cuda()
    bias = torch.randn(64, generator=g).cuda()
    wk = torch.zeros(64, 32, device="cuda")
    wk[:, :27] = w.permute(0, 2, 3, 1).reshape(64, 27)
    wkh = wk.half().contiguous()
    wkl = (wk - wkh.float()).half().contiguous()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = _cl(torch.full((B, 64, Ho, Wo), float("nan"), dtype=torch.float16, device="cuda"))
    view = hat_runtime._map_view(x)
    _lib.check(lib.fvit_stem_conv3x3s2_px(1, ctypes.byref(view), wkh.data_ptr(), wkl.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, W, _stream()),
               "stem px")
    torch.cuda.synchronize()
    weff = (wkh.double() + wkl.double())[:, :27].reshape(64, 3, 3, 3).permute(0, 3, 1, 2)
    ref = torch.relu(F.conv2d(x.double(), weff, bias.double(), 2, 1))
    err = (out.double() - ref).abs().max().item()
    # the result is ONE fp16 plane: half an ulp of the largest value, plus the ~1e-6 of the two-term contraction
    tol = 2.0 ** -11 * max(ref.abs().max().item(), 1.0) * 1.01 + 1e-5
    print(f"stem_conv_px {in_dt} {fmt}: max-abs {err:.2e} (tol {tol:.1e})")
    assert err < tol
    # against the single-term kernel: the contraction itself must be closer to fp64 than one-term weights are
    out1 = torch.empty_like(out)
    _lib.check(lib.fvit_stem_conv3x3s2(1, ctypes.byref(view), wkh.data_ptr(), bias.data_ptr(), out1.data_ptr(), B, H, W, _stream()), "stem")
    torch.cuda.synchronize()
    assert (out1.double() - ref).abs().mean().item() > (out.double() - ref).abs().mean().item()


def test_gemm_x3_dual_tiles_match_the_k_concatenated_walk():
    """The x3 GEMM's dual K loop (32 contraction indices of both terms of both operands per tile: four operand tiles for three products)
    against the K-concatenated walk [hi | hi | lo] x [hi | lo | hi] (fvit_tune gemm_x3_dual = 0) and fp64, on the tile shapes the stage path
    selects: 256 x 256 ping-pong (large M x N), 128-row and 64-row tiles; every epilogue."""
    lib = _lib.lib()
    dt = torch.float16
    for M, N, ka, epi in [(27136, 3136, 832, 1), (9116, 3072, 832, 0), (2000, 784, 1024, 2), (300, 784, 3136, 2), (130, 256, 256, 1), (6800, 2352 + 0, 832, 0)]:
        g = torch.Generator(device="cpu").manual_seed(M + N + ka)
        Mp, Np = (M + 127) // 128 * 128, (N + 127) // 128 * 128
        a = torch.randn(M, ka, generator=g)
        w = torch.randn(N, ka, generator=g) / ka ** 0.5
        ah, al = _split(a)
        wh, wl = _split(w)
        A = torch.zeros(Mp, 2 * ka, dtype=dt)
        A[:M, :ka], A[:M, ka:] = ah, al
        Wt = torch.zeros(Np, 3 * ka, dtype=dt)
        Wt[:N, :ka], Wt[:N, ka:2 * ka], Wt[:N, 2 * ka:] = wh, wl, wh
        A, Wt = A.cuda(), Wt.cuda()
        bias = torch.randn(N, generator=g).cuda()
        gamma = (torch.rand(N, generator=g) + 0.5).cuda()
        x0 = torch.randn(M, N, generator=g).cuda()
        outs = []
        try:
            for dual in (1, 0):
                _lib.tune("gemm_x3_dual", dual)
                if epi == 2:
                    out = x0.clone()
                    ldo, lo_off = N, 0
                else:
                    ldo, lo_off = 2 * Np, Np
                    out = torch.zeros(Mp, ldo, dtype=dt, device="cuda")
                _lib.check(lib.fvit_gemm_terms_lo(1, A.data_ptr(), 2 * ka, Wt.data_ptr(), 3 * ka, bias.data_ptr(), gamma.data_ptr(), out.data_ptr(), ldo, lo_off,
                                                  M, N, 3 * ka, ka, epi, _stream()), "gemm_terms_lo")
                torch.cuda.synchronize()
                outs.append(out.double().cpu() if epi == 2 else (out[:M, :N].double() + out[:M, Np:Np + N].double()).cpu())
        finally:
            _lib.tune("gemm_x3_dual", 1)
        ad, wd = ah.double() + al.double(), wh.double() + wl.double()
        ref = ad @ wd.t() + bias.double().cpu()
        if epi == 1:
            ref = F.gelu(ref)
        if epi == 2:
            ref = x0.double().cpu() + gamma.double().cpu() * ref
        scale = max(ref.abs().max().item(), 1.0)
        e_dual, e_cat, d = (outs[0] - ref).abs().max().item(), (outs[1] - ref).abs().max().item(), (outs[0] - outs[1]).abs().max().item()
        print(f"gemm x3 M {M} N {N} ka {ka} epi {epi}: dual {e_dual:.2e} concatenated {e_cat:.2e} |dual - concat| {d:.2e} (scale {scale:.1f})")
        assert e_dual < 4e-6 * scale and e_cat < 4e-6 * scale


# ---------------------------------------------------------------------------------------------------------------------------------------
# the precise plan against the reference's logits, ABSOLUTE (north_star: logits max-abs < 1e-3)
# ---------------------------------------------------------------------------------------------------------------------------------------
def _precise_logits(name, streams=1, graph=False, join_from=None, operand="f16x3", dtype=torch.float16):
    model, _ = build_product_model(name, "cuda")
    model = model.to(memory_format=torch.channels_last)
    model.set_hat_operand_dtype(operand)
    x = case_input(name).cuda().contiguous(memory_format=torch.channels_last)
    runner = model.compile_inference(x, dtype=dtype, streams=streams, graph=graph, join_from=join_from, precise=True)
    assert runner.plan.precise
    y = runner(x).float().cpu().clone()
    y2 = runner(x).float().cpu()
    assert torch.equal(y, y2)   # bitwise repeatable
    return y, runner, x


@pytest.mark.parametrize("name,bar", [("fvit4_224", 5e-4), ("fvit4_anyres_576x960", 5e-4), ("fvit0_224", 3e-4),
                                      ("fvit4_21k_384", 5e-4)])   # r06: 24 x 24 / 12 x 12 windows -- the two-term instances of the LONG attention kernel
def test_precise_deploy_plan_meets_the_absolute_bar(name, bar):
    """BASELINE configs 3 and 5 (and the headline model) through the plan bench.py TIMES for them: two-term conv streams + f16x3 HAT operands,
    logits vs the reference's CPU forward (committed goldens), ABSOLUTE.  north_star's bar is 1e-3; asserted with 2x margin
    (simulated on the fp32 oracle: conv side 1.3e-4 + HAT f16x3 2e-5 on faster_vit_4_224)."""
    g = load_golden(name)
    y, runner, x = _precise_logits(name)
    err = max_abs(y, g["logits"])
    print(f"{CASES[name]['entry']} precise deploy plan: logits max-abs err {err:.3e} ABSOLUTE (|logits| max {np.abs(g['logits']).max():.3f})")
    assert err < bar
    with open("/proc/self/maps") as f:
        assert "libfvit_hip.so" in f.read()


def test_precise_plan_with_bf16_everywhere_meets_the_bar():
    """north_star's "< 1e-3 bf16", literally: bf16 planes on the conv side, bf16 MFMA operands in the HAT stages (bf16x3), every stream two-term: measured 4.2e-4 on
    the 16 bench images of faster_vit_0_224 (r04's best bf16 point, bf16x2 on the 16-bit plan: 8.4e-4)."""
    g = load_golden("fvit0_224")
    y, _, _ = _precise_logits("fvit0_224", operand="bf16x3", dtype=torch.bfloat16)
    err = max_abs(y, g["logits"])
    print(f"faster_vit_0_224 precise plan, bf16 everywhere: logits max-abs err {err:.3e} ABSOLUTE")
    assert err < 8e-4


def test_precise_plan_shards_join_and_graph_give_the_same_bits():
    """The precise plan as stream shards with a join (tuples of planes concatenated across shards) and inside a hipGraph: same logits as the
    single-stream eager plan up to launch-shape-dependent rounding order, bitwise repeatable."""
    name = "fvit0_224"
    g = load_golden(name)
    base, _, _ = _precise_logits(name)
    for streams, jf, graph in ((2, 3, True), (2, None, True), (3, 2, False)):
        y, runner, x = _precise_logits(name, streams=streams, graph=graph, join_from=jf)
        assert max_abs(y, base) < 5e-5, (streams, jf, graph)
        assert max_abs(y, g["logits"]) < 3e-4


def test_precise_plan_tiny_configs_vs_oracle():
    """Padded / non-square / propagation geometries through the precise plan (channel counts that pad to 64: every px kernel shape)."""
    from oracle.model_reference import model_forward
    for name in ("tiny_hier", "tiny_anyres", "tiny_d40"):
        model, sd = build_product_model(name, "cuda")
        model.set_hat_operand_dtype("f16x3")
        x_cpu = case_input(name)
        x = x_cpu.cuda()
        runner = model.compile_inference(x, streams=1, graph=False, precise=True)
        y = runner(x).float().cpu()
        ref = model_forward(sd, x_cpu, CASES[name]["arch"])
        err = max_abs(y, ref)
        print(f"{name} precise plan: {err:.3e} on |{ref.abs().max().item():.2f}|")
        assert err < 3e-4 * max(ref.abs().max().item(), 1.0)   # 'stress' weights; measured 3e-5 .. 1.2e-4 relative


@pytest.mark.parametrize("dt,code,B,HW,C", [(torch.float16, 1, 5, 49, 512), (torch.float32, 0, 3, 49, 1568), (torch.bfloat16, 2, 2, 1, 64), (torch.float16, 1, 4, 196, 200)])
def test_global_avgpool_and_head(dt, code, B, HW, C):
    """Tail of the deploy plan: AdaptiveAvgPool2d(1) + flatten (fvit_global_avgpool_cl) and the head (fvit_head_logits, exact-fp32 MFMA) vs torch fp64."""
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(C + HW)
    x = torch.randn(B, HW, C, generator=g).to(dt).cuda()
    feat = torch.full((B, C), float("nan"), device="cuda")
    _lib.check(lib.fvit_global_avgpool_cl(code, x.data_ptr(), feat.data_ptr(), B, HW, C, _stream()), "avgpool")
    torch.cuda.synchronize()
    ref = x.double().mean(dim=1)
    assert (feat.double() - ref).abs().max().item() < 1e-6 * max(ref.abs().max().item(), 1.0)
    feat2 = torch.empty_like(feat)
    _lib.check(lib.fvit_global_avgpool_cl(code, x.data_ptr(), feat2.data_ptr(), B, HW, C, _stream()), "avgpool")
    torch.cuda.synchronize()
    assert torch.equal(feat, feat2)
    if C % 16 == 0:
        w = (torch.randn(1000, C, generator=g) / C ** 0.5).cuda()
        b = torch.randn(1000, generator=g).cuda()
        out = torch.empty(B, 1000, device="cuda")
        _lib.check(lib.fvit_head_logits(feat.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), B, 1000, C, _stream()), "head")
        torch.cuda.synchronize()
        refl = feat.double() @ w.double().t() + b.double()
        assert (out.double() - refl).abs().max().item() < 2e-6 * max(refl.abs().max().item(), 1.0)
