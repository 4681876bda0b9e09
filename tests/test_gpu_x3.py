"""The 'x3' operand modes (FvitStageDesc.weight_terms = 3: two-term weights AND two-term activations) on an MI355X through the C ABI.

Why they exist (tests/tools/precision_sim.py on faster_vit_4_224, |logits| max 7.1, max-abs vs the fp32 reference):
    f16 (weights and activations rounded once)             5.0e-3
    f16x2 (weights hi + lo, activations once)              9.9e-4     <- at the 1e-3 bar, not under it
    + LayerNorm outputs, GELU(fc1), attention output as hi + lo   4.3e-4
    + q, k, v as hi + lo                                    1.2e-4
    + P as hi + lo                                          2.2e-5     <- the x3 modes
i.e. once the systematic weight rounding is gone, every remaining single rounding of an ACTIVATION contributes 4-5e-4 on a model whose
logits reach |7|; the x3 modes carry every 16-bit operand of the HAT stages as two terms (each Linear layer = hi.hi + hi.lo + lo.hi with
fp32 accumulation, the attention core on two-term q / k / v / P, exact-erf GELU).  The kernel-level tests below compare against float64
at tolerances two to three orders of magnitude below the 16-bit ones of tests/test_gpu_kernels.py, so they check every index of the
[hi | lo] layouts, not just plausibility.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fastervit_amd import _lib
from tests.cases import CASES
from tests.util import build_product_model, case_input, load_golden, max_abs, rel_err

pytestmark = pytest.mark.gpu

OPS = [("f16", torch.float16, 1), ("bf16", torch.bfloat16, 2)]
# relative accuracy of a two-term value / product: fp16 2 x 11 bits, bf16 2 x 8 bits
EPS2 = {torch.float16: 2.0 ** -21, torch.bfloat16: 2.0 ** -15}


def _rup(x, m):
    return (x + m - 1) // m * m


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _split(w, dt):
    hi = w.to(dt)
    lo = (w - hi.float()).to(dt)
    return hi, lo


def _two_term_rows(t, dt, rows, width):
    """fp32 (M, K) -> 16-bit (rows, 2 * width) = [hi | lo], zero padded."""
    hi, lo = _split(t, dt)
    out = torch.zeros(rows, 2 * width, dtype=dt, device=t.device)
    out[:t.shape[0], :t.shape[1]] = hi
    out[:t.shape[0], width:width + t.shape[1]] = lo
    return out


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("M,N,K,epi", [(300, 768, 256, 0), (1000, 784, 832, 1), (77, 3136, 784, 0), (4214, 512, 2048, 2), (129, 256, 64, 2),
                                       (16384, 768, 256, 1), (16500, 1536, 512, 0)])
def test_gemm_three_terms(opname, dt, code, M, N, K, epi):
    """out = epilogue((A_hi + A_lo) . (W_hi + W_lo)^T + bias) without the lo.lo product: weight rows [hi | lo | hi] against
    activation columns [hi | hi | lo]; epilogues 0 / 1 store the result as two terms (and use the exact-erf GELU).  The last
    two shapes select the 256 x 256 ping-pong tile."""
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    gamma = (torch.rand(N, generator=g) + 0.5).cuda()
    Kp = _rup(K, 64)
    Ap = _two_term_rows(A, dt, _rup(M, 128), Kp)
    whi, wlo = _split(W, dt)
    Wp = torch.zeros(_rup(N, 128), 3 * Kp, dtype=dt, device="cuda")
    Wp[:N, :K], Wp[:N, Kp:Kp + K], Wp[:N, 2 * Kp:2 * Kp + K] = whi, wlo, whi
    y = (A.double() @ W.double().t() + bias.double())
    if epi == 1:
        y = F.gelu(y)
    scale = max(y.abs().max().item(), 1.0)
    if epi == 2:
        x0 = torch.randn(M, N, generator=g).cuda()
        out = x0.clone()
        rc = lib.fvit_gemm_terms_lo(code, Ap.data_ptr(), 2 * Kp, Wp.data_ptr(), 3 * Kp, bias.data_ptr(), gamma.data_ptr(), out.data_ptr(), N, 0,
                                    M, N, 3 * Kp, Kp, 2, _stream())
        _lib.check(rc, "gemm_terms_lo residual")
        torch.cuda.synchronize()
        ref = x0.double() + gamma.double() * y
        err = (out.double() - ref).abs().max().item()
        tol = (4 * EPS2[dt] * K ** 0.5 + 2e-7) * scale * 1.5
    else:
        ld1 = _rup(N, 64)
        out = torch.full((_rup(M, 128), 2 * ld1), float("nan"), dtype=dt, device="cuda")
        rc = lib.fvit_gemm_terms_lo(code, Ap.data_ptr(), 2 * Kp, Wp.data_ptr(), 3 * Kp, bias.data_ptr(), None, out.data_ptr(), 2 * ld1, ld1,
                                    M, N, 3 * Kp, Kp, epi, _stream())
        _lib.check(rc, "gemm_terms_lo")
        torch.cuda.synchronize()
        got = out[:M, :N].double() + out[:M, ld1:ld1 + N].double()
        assert torch.isfinite(got).all()
        err = (got - y).abs().max().item()
        tol = (4 * EPS2[dt] * K ** 0.5 + 4 * EPS2[dt] + (4e-7 if epi == 1 else 0)) * scale
    print(f"gemm x3 {opname} M={M} N={N} K={K} epi={epi}: max-abs err {err:.3e} on |{scale:.2f}| (tol {tol:.1e})")
    assert err < tol


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("rows,C", [(1000, 256), (77, 784), (5, 1568), (300, 24)])
def test_layernorm_two_term_output(opname, dt, code, rows, C):
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 2 + 0.3).cuda()
    w = (torch.rand(C, generator=g) + 0.5).cuda()
    b = torch.randn(C, generator=g).cuda()
    lw = _rup(C, 64)
    out = torch.full((_rup(rows, 128), 2 * lw), float("nan"), dtype=dt, device="cuda")
    rc = lib.fvit_gather_layernorm_terms(code, x.data_ptr(), 0, None, 0, None, None, None, None, out.data_ptr(), 2 * lw, lw, w.data_ptr(),
                                         b.data_ptr(), 1e-5, rows, 1, C, _stream())
    _lib.check(rc, "gather_layernorm_terms")
    torch.cuda.synchronize()
    ref = F.layer_norm(x.double(), (C,), w.double(), b.double(), 1e-5)
    got = out[:rows, :C].double() + out[:rows, lw:lw + C].double()
    err = (got - ref).abs().max().item()
    print(f"layernorm two-term {opname} rows={rows} C={C}: max-abs err {err:.3e}")
    assert err < 4 * EPS2[dt] * ref.abs().max().item() + 2e-6
    # pad columns of both images are zero (they are GEMM operand columns)
    assert (out[:rows, C:lw] == 0).all() and (out[:rows, lw + C:] == 0).all()


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("S,nwin,heads,d,dpad", [(49, 5, 3, 32, 32), (53, 3, 2, 49, 64), (148, 2, 2, 49, 64), (36, 4, 2, 49, 64), (64, 2, 4, 24, 32),
                                                 (16, 3, 1, 80, 96), (200, 1, 1, 32, 32)])
def test_window_attention_two_term(opname, dt, code, S, nwin, heads, d, dpad):
    """softmax(q k^T * scale + bias) v on two-term q / k / v rows, two-term output: against float64 attention of the SAME fp32 q, k, v."""
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(S * 7 + heads)
    rows = nwin * S
    q, k, v = (torch.randn(rows, heads, d, generator=g).cuda() for _ in range(3))
    spad = lib.fvit_attention_spad(S)
    bias = torch.zeros(heads, spad, spad, device="cuda")
    bias[:, :S, :S] = torch.randn(heads, S, S, generator=g).cuda()
    bias[:, :, S:] = _lib.FVIT_MASK_BIAS
    bias[:, S:, :] = 0.0
    bias[:, S:, S:] = _lib.FVIT_MASK_BIAS if S < spad else 0.0
    ld1 = 3 * heads * dpad
    qkv = torch.zeros(rows, 3, heads, dpad, device="cuda")
    qkv[:, 0, :, :d], qkv[:, 1, :, :d], qkv[:, 2, :, :d] = q, k, v
    buf = _two_term_rows(qkv.reshape(rows, ld1), dt, _rup(rows, 128), ld1)
    ldo1 = _rup(heads * dpad, 64)
    out = torch.zeros(_rup(rows, 128), 2 * ldo1, dtype=dt, device="cuda")
    scale = d ** -0.5
    rc = lib.fvit_window_attention_terms(code, buf.data_ptr(), 2 * ld1, ld1, out.data_ptr(), 2 * ldo1, ldo1, bias.data_ptr(), nwin, S, heads, dpad,
                                         scale, _stream())
    _lib.check(rc, "window_attention_terms")
    torch.cuda.synchronize()
    qd, kd, vd = (t.double().view(nwin, S, heads, d).permute(0, 2, 1, 3) for t in (q, k, v))
    att = (qd @ kd.transpose(-1, -2)) * scale + bias[:, :S, :S].double()
    ref = (att.softmax(-1) @ vd).permute(0, 2, 1, 3).reshape(rows, heads, d)
    got = (out[:rows, :heads * dpad].double() + out[:rows, ldo1:ldo1 + heads * dpad].double()).view(rows, heads, dpad)
    err = (got[:, :, :d] - ref).abs().max().item()
    print(f"attention two-term {opname} S={S} heads={heads} d={d}: max-abs err {err:.3e} on |{ref.abs().max().item():.2f}|")
    # scores carry |q.k| ~ sqrt(d) * few: their two-term error enters exp(); __expf itself is ~2e-7 relative
    assert err < (40 * EPS2[dt] + 2e-6) * max(ref.abs().max().item(), 1.0)
    assert (got[:, :, d:] == 0).all()   # padded head channels stay zero


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("S,w,nwin,heads,d,dpad", [(256, 16, 3, 4, 32, 32), (260, 16, 2, 3, 49, 64), (580, 24, 1, 2, 49, 64), (209, 14, 5, 2, 32, 32),
                                                   (1024, 32, 1, 2, 49, 64), (233, 15, 2, 3, 24, 32), (260, 16, 2, 2, 80, 96), (148, 12, 3, 2, 80, 96)])
def test_window_attention_long_two_term(opname, dt, code, S, w, nwin, heads, d, dpad):
    """r06: the online-softmax attention of LONG windows (> 208 tokens; > 128 at the 96-wide head padding) on two-term q / k / v rows with a two-term
    output (fvit_window_attention_long_terms): against float64 attention of the SAME fp32 q, k, v with the densely gathered compact bias table."""
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(S * 13 + heads + d)
    ng = S - w * w
    rows = nwin * S
    q, k, v = (torch.randn(rows, heads, d, generator=g).cuda() for _ in range(3))
    rel = torch.randn(heads, (2 * w - 1) ** 2, generator=g) * 2
    tw = 2 * w - 1
    pos = torch.arange(w * w)
    yy, xx = pos // w, pos % w
    idx = (yy[:, None] - yy[None, :] + w - 1) * tw + (xx[:, None] - xx[None, :] + w - 1)
    dense = torch.zeros(heads, S, S, dtype=torch.float64)
    dense[:, ng:, ng:] = rel[:, idx.view(-1)].view(-1, w * w, w * w).double()
    ld1 = 3 * heads * dpad
    qkv = torch.zeros(rows, 3, heads, dpad, device="cuda")
    qkv[:, 0, :, :d], qkv[:, 1, :, :d], qkv[:, 2, :, :d] = q, k, v
    buf = _two_term_rows(qkv.reshape(rows, ld1), dt, _rup(rows, 128), ld1)
    ldo1 = _rup(heads * dpad, 64)
    out = torch.full((_rup(rows, 128), 2 * ldo1), float("nan"), dtype=dt, device="cuda")
    scale = d ** -0.5
    relc = rel.cuda()
    rc = lib.fvit_window_attention_long_terms(code, buf.data_ptr(), 2 * ld1, ld1, out.data_ptr(), 2 * ldo1, ldo1, relc.data_ptr(), w, ng, nwin, S, heads, dpad,
                                              scale, _stream())
    _lib.check(rc, "window_attention_long_terms")
    torch.cuda.synchronize()
    qd, kd, vd = (t.double().cpu().view(nwin, S, heads, d).permute(0, 2, 1, 3) for t in (q, k, v))
    att = (qd @ kd.transpose(-1, -2)) * scale + dense
    ref = (att.softmax(-1) @ vd).permute(0, 2, 1, 3).reshape(rows, heads, d)
    got = (out[:rows, :heads * dpad].double() + out[:rows, ldo1:ldo1 + heads * dpad].double()).cpu().view(rows, heads, dpad)
    err = (got[:, :, :d] - ref).abs().max().item()
    # the single-term long kernel on the hi image of the same rows: the two-term result must be far closer
    out1 = torch.zeros(_rup(rows, 128), ldo1, dtype=dt, device="cuda")
    _lib.check(lib.fvit_window_attention_long(code, buf.data_ptr(), 2 * ld1, out1.data_ptr(), ldo1, relc.data_ptr(), w, ng, nwin, S, heads, dpad, scale, _stream()),
               "window_attention_long")
    torch.cuda.synchronize()
    err1 = (out1[:rows, :heads * dpad].double().cpu().view(rows, heads, dpad)[:, :, :d] - ref).abs().max().item()
    print(f"long attention two-term {opname} S={S} heads={heads} d={d}: max-abs err {err:.3e} (single-term {err1:.3e}) on |{ref.abs().max().item():.2f}|")
    assert err < (40 * EPS2[dt] + 2e-6) * max(ref.abs().max().item(), 1.0)
    assert err1 > 20 * err
    assert (got[:, :, d:] == 0).all()   # padded head channels stay zero
    # argument checks: the lo images must fit the rows
    assert lib.fvit_window_attention_long_terms(code, buf.data_ptr(), ld1, ld1, out.data_ptr(), 2 * ldo1, ldo1, relc.data_ptr(), w, ng, nwin, S, heads, dpad,
                                                scale, _stream()) != 0


TINY = [n for n, c in CASES.items() if c["per_block"]]   # (r06: long windows included -- tiny_21k_384, tiny_anyres_w16 run the two-term long attention kernel)


@pytest.mark.parametrize("name", TINY)
def test_hat_blocks_x3_vs_reference_goldens(name):
    """Every HAT block of the tiny configurations on the reference's OWN block inputs ('stress' weights), f16x3: the 16-bit modes are
    asserted at 5e-3 relative (tests/test_gpu_parity.py), this one two orders of magnitude tighter."""
    from oracle import hat_reference as hr
    g = load_golden(name)
    model, _ = build_product_model(name, "cuda")
    model.set_hat_operand_dtype("f16x3")
    worst = 0.0
    for li in (2, 3):
        lvl = model.levels[li]
        ws = lvl.window_size
        xin = torch.from_numpy(g[f"level{li}_in"])
        H, W = xin.shape[2:]
        pad_b, pad_r = (ws - H % ws) % ws, (ws - W % ws) % ws
        xw = hr.window_partition(torch.nn.functional.pad(xin, (0, pad_r, 0, pad_b)), ws)
        ct = torch.from_numpy(g[f"l{li}_ct0"]) if f"l{li}_ct0" in g and lvl.blocks[0].do_sr_hat else None
        for bi, blk in enumerate(lvl.blocks):
            with torch.no_grad():
                xo, cto = blk(xw.cuda(), None if ct is None else ct.cuda())
            ref_x = g[f"l{li}b{bi}_x"]
            e = rel_err(xo.cpu(), ref_x)
            worst = max(worst, e)
            assert e < 5e-5, f"{name} level {li} block {bi} x: {e:.2e}"
            if ct is not None:
                ref_ct = g[f"l{li}b{bi}_ct"]
                e = rel_err(cto.cpu(), ref_ct)
                worst = max(worst, e)
                assert e < 5e-5, f"{name} level {li} block {bi} ct: {e:.2e}"
                ct = torch.from_numpy(ref_ct)
            xw = torch.from_numpy(ref_x)
    print(f"{name} f16x3: worst block rel err {worst:.2e}")


@pytest.mark.parametrize("name", ["tiny_21k_384", "tiny_anyres_w16", "fvit4_21k_384"])
def test_long_windows_run_x3(name):
    """r06: windows beyond the dense attention kernel (> 208 tokens) run the x3 modes on fvit_attnlong.hip's two-term instances (r05 refused them by name):
    the tiny long-window configurations and faster_vit_4_21k_384 (one 24 x 24 window + carriers per image at level 2, 12 x 12 at level 3) against the
    reference goldens, module mode (fp32 conv side), ABSOLUTE: an order of magnitude and more below the 16-bit result of the same model."""
    g = load_golden(name)
    model, _ = build_product_model(name, "cuda")
    x = case_input(name).cuda()
    scale = max(float(np.abs(g["logits"]).max()), 1.0)
    with torch.no_grad():
        e16 = max_abs(model(x).float().cpu(), g["logits"])
        model.set_hat_operand_dtype("f16x3")
        assert model.hat_operand_dtype == "f16x3"
        e3 = max_abs(model(x).float().cpu(), g["logits"])
        again = max_abs(model(x).float().cpu(), g["logits"])
    print(f"{name} module mode: f16 {e16:.3e} -> f16x3 {e3:.3e} ABSOLUTE (|logits| max {scale:.3f})")
    assert e3 < 2e-4 * scale and e3 < 0.1 * e16
    assert again == e3 or abs(again - e3) < 1e-4 * scale   # (module mode: the MIOpen conv side is not bitwise repeatable)


@pytest.mark.parametrize("case,mode,tol", [("fvit0_224", "f16x3", 1e-4), ("fvit0_224", "bf16x3", 4e-4), ("fvit4_224", "f16x3", 2e-4),
                                           ("fvit4_anyres_576x960", "f16x3", 2e-4)])
def test_x3_module_mode_logits_absolute(case, mode, tol):
    """north_star tolerance, ABSOLUTE (logits max-abs < 1e-3) on all three single-GPU BASELINE configurations, with margin: module mode
    (fp32 conv side) + x3 HAT stages.  faster_vit_4_224 / any-res reach |logits| 7 with the gamma ~ U(0.5, 1.5) test weights; simulated
    2.2e-5 for the HAT roundings alone (precision_sim); measured 2.6e-5 / 2.4e-5 on the 8 bench images of faster_vit_4_224 / any-res."""
    g = load_golden(case)
    model, _ = build_product_model(case, "cuda")
    x = case_input(case).cuda()
    with torch.no_grad():
        base = max_abs(model(x).float().cpu(), g["logits"])
        model.set_hat_operand_dtype(mode)
        model(x)
        err = max_abs(model(x).float().cpu(), g["logits"])
    print(f"{case} module mode: f16 {base:.3e} -> {mode} {err:.3e} (|logits| max {np.abs(g['logits']).max():.3f})")
    assert err < tol and err < 1e-3


@pytest.mark.parametrize("entry,batch", [("faster_vit_0_224", 5), ("faster_vit_4_224", 2)])
def test_x3_stages_are_bitwise_repeatable(entry, batch):
    import fastervit_amd
    from fastervit_amd import hat_runtime
    torch.manual_seed(0)
    model = fastervit_amd.create_model(entry).eval().cuda()
    model.set_hat_operand_dtype("f16x3")
    g = torch.Generator(device="cpu").manual_seed(11)
    for li in (2, 3):
        lvl = model.levels[li]
        C = lvl.blocks[0].attn.qkv.in_features
        R = 14 if li == 2 else 7
        x = torch.randn(batch, C, R, R, generator=g).cuda()
        outs = [hat_runtime.stage_forward(lvl, x.clone()).clone() for _ in range(3)]
        torch.cuda.synchronize()
        assert torch.isfinite(outs[0]).all()
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), f"{entry} level {li}: repeat call differs"


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("B,Ci,Co,H,W,stride", [(5, 64, 128, 56, 56, 2), (7, 128, 256, 28, 28, 2), (3, 256, 512, 14, 14, 2), (2, 64, 64, 9, 11, 1), (2, 256, 448, 15, 9, 2)])
def test_conv3x3_two_weight_terms(opname, dt, code, B, Ci, Co, H, W, stride):
    """fvit_conv3x3_nhwc_terms: the Downsample.reduction convs of the deploy plan with weights as hi + lo ([Cout][hi (3,3,Cin) | lo (3,3,Cin)]).  Against
    the convolution of the SAME 16-bit map with the fp32 weights: the two-term result differs from it by the output rounding only, and is closer to
    it than the single-term result (whose weight rounding is systematic); one term through the new entry point = fvit_conv3x3_nhwc bitwise."""
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(Ci + Co + H)
    x = torch.randn(B, Ci, H, W, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5).cuda()            # fp32 weights
    hi = w.to(dt)
    lo = (w - hi.float()).to(dt)
    k1 = hi.permute(0, 2, 3, 1).contiguous()
    k2 = torch.cat([k1.reshape(Co, -1), lo.permute(0, 2, 3, 1).reshape(Co, -1)], dim=1).contiguous()
    zeros = torch.zeros(256, dtype=dt, device="cuda")
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    outs = []
    for wk, terms in ((k2, 2), (k1, 1)):
        out = torch.full((B, Co, Ho, Wo), float("nan"), dtype=dt, device="cuda").contiguous(memory_format=torch.channels_last)
        _lib.check(lib.fvit_conv3x3_nhwc_terms(code, x.data_ptr(), wk.data_ptr(), None, None, out.data_ptr(), B, H, W, Ci, Co, stride, 0, terms,
                                               zeros.data_ptr(), _stream()), "conv3x3_nhwc_terms")
        outs.append(out)
    ref1 = torch.full_like(outs[1], float("nan"))
    _lib.check(lib.fvit_conv3x3_nhwc(code, x.data_ptr(), k1.data_ptr(), None, None, ref1.data_ptr(), B, H, W, Ci, Co, stride, 0, zeros.data_ptr(), _stream()), "conv3x3")
    torch.cuda.synchronize()
    assert torch.equal(outs[1], ref1)
    exact = F.conv2d(x.double(), w.double(), None, stride, 1)
    e2, e1 = (outs[0].double() - exact).abs(), (outs[1].double() - exact).abs()
    scale = exact.abs().max().item()
    eps1 = 2.0 ** -11 if dt == torch.float16 else 2.0 ** -8
    assert torch.isfinite(outs[0].float()).all() and e2.max().item() < 1.2 * eps1 * scale          # output rounding (half an ulp of the largest value) + slack
    print(f"conv two-term {opname} {Ci}->{Co} {H}x{W}: mean |err| two terms {e2.mean().item():.3e}, one term {e1.mean().item():.3e}")
    assert e2.mean().item() < e1.mean().item()


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("M,N,K", [(688, 784, 3136), (240, 784, 1024), (2107, 1568, 6272), (2048, 256, 1024), (77, 512, 512), (5000, 784, 3136)])
def test_gemm_residual_split_k(opname, dt, code, M, N, K):
    """Deterministic split-K of the small-grid residual GEMMs (the carrier-token branch of FasterViT-4: 688 x 784 x 3136 = 42 workgroups x 49 K tiles):
    against the fp32 product, against the unsplit kernel (same products, other fp32 summation order), bitwise repeatable; a grid that fills the chip
    (the last shape) is not split and returns the unsplit kernel's bits."""
    lib = _lib.lib()
    lib.fvit_tune(b"gemm_splitk", 1)   # opt-in knob (negative end to end under the stream shards, see fvit_gemm.hip)
    try:
        _split_k_body(lib, dt, code, M, N, K)
    finally:
        lib.fvit_tune(b"gemm_splitk", 0)


def _split_k_body(lib, dt, code, M, N, K):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dt).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dt).cuda()
    bias, gamma = torch.randn(N, generator=g).cuda(), (torch.rand(N, generator=g) + 0.5).cuda()
    x0 = torch.randn(M, N, generator=g).cuda()
    Ap = torch.zeros(_rup(M, 128), K, dtype=dt, device="cuda")
    Ap[:M] = A
    Wp = torch.zeros(_rup(N, 128), K, dtype=dt, device="cuda")
    Wp[:N] = W
    slab = torch.full((8 * M * N,), float("nan"), dtype=torch.float32, device="cuda")
    outs = []
    for _ in range(2):
        out = x0.clone()
        _lib.check(lib.fvit_gemm_residual_splitk(code, Ap.data_ptr(), K, Wp.data_ptr(), K, bias.data_ptr(), gamma.data_ptr(), out.data_ptr(), N, M, N, K,
                                                 slab.data_ptr(), slab.numel() * 4, _stream()), "gemm_residual_splitk")
        outs.append(out)
    plain = x0.clone()
    _lib.check(lib.fvit_gemm_residual(code, Ap.data_ptr(), K, Wp.data_ptr(), K, bias.data_ptr(), gamma.data_ptr(), plain.data_ptr(), N, M, N, K, _stream()), "gemm_residual")
    torch.cuda.synchronize()
    ref = x0.double() + gamma.double() * (A.double() @ W.double().t() + bias.double())
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])
    scale = ref.abs().max().item()
    assert (outs[0].double() - ref).abs().max().item() < 2e-5 * scale
    assert (outs[0] - plain).abs().max().item() < 2e-5 * scale
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles > 230:
        assert torch.equal(outs[0], plain)
    else:
        assert torch.isfinite(slab[:2 * M * N]).all()   # the partials went through the scratch
