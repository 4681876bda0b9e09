"""Weight-term operand modes ('f16x2' / 'bf16x2': every Linear weight as hi + lo, FvitStageDesc.weight_terms = 2) and the fp16
saturating narrowing, on an MI355X through the C ABI.

Why the WEIGHTS are split and not the activations (tests/tools/precision_sim.py): a rounded weight is wrong by
the same amount for every token of every image, so its error survives the attention / pooling averages and reaches the logits
coherently; a rounded activation is wrong independently per token and averages out.  Simulated on the fp32 oracle for FasterViT-0:
bf16 2.9e-3, activations hi + lo 2.9e-3 (no gain), weights hi + lo 7.2e-4.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fastervit_amd import _lib, hat_runtime
from tests.util import build_product_model, case_input, load_golden, max_abs

pytestmark = pytest.mark.gpu

OPS = [("f16", torch.float16, 1), ("bf16", torch.bfloat16, 2)]
GEMM_PP_DEFAULT = 1      # default of the "gemm_pp" tuning knob (fvit_gemm.hip)
CT_VARIANT_DEFAULT = 3   # default of the "ct_variant" knob (fvit_ctblk.hip)


def _rup(x, m):
    return (x + m - 1) // m * m


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _padded(t, rows, cols):
    out = torch.zeros(rows, cols, dtype=t.dtype, device=t.device)
    out[:t.shape[0], :t.shape[1]] = t
    return out


def _split(w, dt):
    hi = w.to(dt)
    lo = (w - hi.float()).to(dt)
    return hi, lo


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("M,N,K,epi", [(300, 768, 256, 0), (1000, 784, 832, 1), (77, 3136, 784, 0), (4214, 512, 2048, 2), (129, 256, 64, 2)])
def test_gemm_two_weight_terms(opname, dt, code, M, N, K, epi):
    """out = epilogue(A . (W_hi + W_lo)^T + bias): K-concatenated weight rows [hi | lo], the activation column wraps at ka."""
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dt).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()          # fp32 weights
    bias = torch.randn(N, generator=g).cuda()
    gamma = (torch.rand(N, generator=g) + 0.5).cuda()
    Kp = _rup(K, 64)
    hi, lo = _split(W, dt)
    Ap = _padded(A, _rup(M, 128), Kp)
    Wp = torch.cat([_padded(hi, _rup(N, 128), Kp), _padded(lo, _rup(N, 128), Kp)], dim=1).contiguous()
    if epi == 2:
        x0 = torch.randn(M, N, generator=g).cuda()
        out = x0.clone()
        ldo = N
    else:
        ldo = _rup(N, 64)
        out = torch.full((_rup(M, 128), ldo), float("nan"), dtype=dt, device="cuda")
    rc = lib.fvit_gemm_terms(code, Ap.data_ptr(), Kp, Wp.data_ptr(), 2 * Kp, bias.data_ptr(), gamma.data_ptr(), out.data_ptr(), ldo,
                             M, N, 2 * Kp, Kp, epi, _stream())
    _lib.check(rc, "gemm_terms")
    torch.cuda.synchronize()
    y = A.float() @ (hi.float() + lo.float()).t() + bias          # what the kernel computes, in fp32
    y_exact = A.float() @ W.t() + bias                              # ... which is the fp32-weight product to ~2^-16
    assert (y - y_exact).abs().max().item() < (2e-4 if dt == torch.bfloat16 else 2e-5) * y_exact.abs().max().item()
    if epi == 1:
        y = F.gelu(y)
    if epi == 2:
        ref, got = x0 + gamma * y, out
        tol = 2e-5 * ref.abs().max().item() + 1e-4
    else:
        ref, got = y, out[:M, :N].float()
        tol = (1.5e-3 if dt == torch.float16 else 1e-2) * max(ref.abs().max().item(), 1.0)   # output rounding only
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < tol
    # one term through the same entry point = the plain GEMM
    if epi != 2:
        out1 = torch.zeros_like(out)
        W1 = _padded(hi, _rup(N, 128), Kp)
        _lib.check(lib.fvit_gemm_terms(code, Ap.data_ptr(), Kp, W1.data_ptr(), Kp, bias.data_ptr(), None, out1.data_ptr(), ldo, M, N, Kp, Kp, epi,
                                       _stream()), "gemm_terms 1")
        out2 = torch.zeros_like(out)
        _lib.check(lib.fvit_gemm_bias_act(code, Ap.data_ptr(), Kp, W1.data_ptr(), Kp, bias.data_ptr(), out2.data_ptr(), ldo, M, N, Kp, epi, _stream()),
                   "gemm")
        torch.cuda.synchronize()
        assert torch.equal(out1[:M, :N], out2[:M, :N])


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("M,use_gamma,C", [(300, True, 256), (18020, True, 256), (4214, True, 512), (70, False, 512)])
def test_win_mlp_two_weight_terms(opname, dt, code, M, use_gamma, C):
    lib = _lib.lib()
    hid = 4 * C
    g = torch.Generator(device="cpu").manual_seed(M + C)
    x0 = (torch.randn(M, C, generator=g) * 1.5 + 0.3).cuda()
    lnw = (torch.rand(C, generator=g) + 0.5).cuda()
    lnb = (torch.randn(C, generator=g) * 0.2).cuda()
    w1 = (torch.randn(hid, C, generator=g) / C ** 0.5).cuda()
    b1 = (torch.randn(hid, generator=g) * 0.3).cuda()
    w2 = (torch.randn(C, hid, generator=g) / hid ** 0.5).cuda()
    b2 = (torch.randn(C, generator=g) * 0.3).cuda()
    gamma = (torch.rand(C, generator=g) + 0.5).cuda() if use_gamma else None
    keep = hat_runtime._Keep(dt, 2)
    w1p, w2p = keep.frag16(hat_runtime.frag_pack_fc1(w1)), keep.frag16(hat_runtime.frag_pack_fc2(w2))
    assert w1p.numel() == 2 * w1.numel() and w1p.dtype == dt
    xw = torch.cat([x0, torch.full((5, C), float("nan"), device="cuda")])
    _lib.check(lib.fvit_win_mlp_fused_terms(code, xw.data_ptr(), M, C, hid, lnw.data_ptr(), lnb.data_ptr(), ctypes.c_float(1e-5), w1p.data_ptr(),
                                            b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(), gamma.data_ptr() if use_gamma else None, 2, _stream()),
               "win_mlp_fused_terms")
    torch.cuda.synchronize()
    h1, l1 = _split(w1, dt)
    h2, l2 = _split(w2, dt)
    xn = F.layer_norm(x0, (C,), lnw, lnb, 1e-5).to(dt).float()
    h = F.gelu(xn @ (h1.float() + l1.float()).t() + b1).to(dt).float()
    y = h @ (h2.float() + l2.float()).t() + b2
    ref = x0 + (gamma * y if use_gamma else y)
    tol = (3e-3 if dt == torch.float16 else 2e-2) * ref.abs().max().item()
    assert torch.isfinite(xw[:M]).all() and torch.isnan(xw[M:]).all()
    err = (xw[:M] - ref).abs().max().item()
    assert err < tol, f"{err} vs {tol}"
    # the second term matters: against single-rounded weights the result differs measurably in bf16
    if dt == torch.bfloat16:
        y1 = F.gelu(xn @ h1.float().t() + b1).to(dt).float() @ h2.float().t() + b2
        ref1 = x0 + (gamma * y1 if use_gamma else y1)
        assert (ref1 - ref).abs().max().item() > 1e-3


@pytest.mark.parametrize("mode,deploy,tol", [("bf16x2", False, 1e-3), ("bf16x2", True, 1e-3), ("f16x2", False, 2.5e-4), ("f16x2", True, 8e-4)])
def test_fvit0_224_weight_term_modes_meet_the_bar(mode, deploy, tol):
    """north_star: logits max-abs < 1e-3 -- with bf16 operands through the two-term weights (module mode: conv side fp32; deploy
    plan: fp16 conv kernels).  Simulated (precision_sim.py): bf16x2 7.2e-4 / 8.3e-4, f16x2 9.6e-5 / 4.7e-4."""
    g = load_golden("fvit0_224")
    model, _ = build_product_model("fvit0_224", "cuda")
    model.set_hat_operand_dtype(mode)
    x = case_input("fvit0_224").cuda()
    if deploy:
        model.switch_to_deploy(torch.float16)
    with torch.no_grad():
        model(x)   # module mode: MIOpen's first call of a shape (find mode) may run another fp32 solver than the later ones
        logits = model(x).float().cpu()
        again = model(x).float().cpu()
    err = max_abs(logits, g["logits"])
    print(f"faster_vit_0_224 {mode} {'deploy plan (fp16 conv side)' if deploy else 'module mode (fp32 conv side)'}: logits max-abs err {err:.3e}"
          f" (repeat call differs by {max_abs(logits, again):.1e})")
    if deploy:   # our kernels only: bitwise repeatable (module mode contains MIOpen's fp32 convolutions, tests/test_gpu_determinism.py)
        assert torch.equal(logits, again)
    assert err < tol


@pytest.mark.parametrize("mode", ["f16x2", "bf16x2"])
@pytest.mark.parametrize("entry,batch", [("faster_vit_0_224", 5), ("faster_vit_4_224", 2)])
def test_weight_term_stages_are_bitwise_repeatable(mode, entry, batch):
    import fastervit_amd
    torch.manual_seed(0)
    model = fastervit_amd.create_model(entry).eval().cuda()
    model.set_hat_operand_dtype(mode)
    g = torch.Generator(device="cpu").manual_seed(11)
    for li in (2, 3):
        lvl = model.levels[li]
        C = lvl.blocks[0].attn.qkv.in_features
        R = 14 if li == 2 else 7
        x = torch.randn(batch, C, R, R, generator=g).cuda()
        outs = [hat_runtime.stage_forward(lvl, x.clone()).clone() for _ in range(3)]
        torch.cuda.synchronize()
        assert torch.isfinite(outs[0]).all()
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), f"{entry} level {li} {mode}: repeat call differs"


def test_fvit4_224_f16x2_module_mode_absolute_error():
    """faster_vit_4_224 with gamma ~ U(0.5, 1.5) reaches |logits| 7: fp16 operands give 5-6e-3 absolute (8e-4 relative).  Two-term
    fp16 weights + the fp32 conv side bring the ABSOLUTE error to the 1e-3 level (simulated 9.9e-4; asserted < 1.5e-3)."""
    g = load_golden("fvit4_224")
    model, _ = build_product_model("fvit4_224", "cuda")
    with torch.no_grad():
        base = max_abs(model(case_input("fvit4_224").cuda()).float().cpu(), g["logits"])
        model.set_hat_operand_dtype("f16x2")
        err = max_abs(model(case_input("fvit4_224").cuda()).float().cpu(), g["logits"])
    print(f"faster_vit_4_224 module mode: f16 {base:.3e}, f16x2 {err:.3e} (|logits| max {np.abs(g['logits']).max():.3f})")
    assert err < 1.5e-3 and err < 0.5 * base


@pytest.mark.parametrize("opname,dt,code", OPS)
def test_f16_operands_saturate(opname, dt, code):
    """Activations beyond the fp16 range: the narrowing saturates at +-65504 (no inf, hence no NaN downstream); bf16 keeps the value."""
    lib = _lib.lib()
    M, N, K = 256, 256, 256
    g = torch.Generator(device="cpu").manual_seed(5)
    A = torch.randn(M, K, generator=g).to(dt).cuda()
    W = (torch.randn(N, K, generator=g) * 40.0).to(dt).cuda()       # |A W^T| ~ 40 * 16 = 640 per unit; bias pushes it over
    bias = (torch.randn(N, generator=g) * 1e5).cuda()                 # +-1e5 > 65504
    out = torch.zeros(M, N, dtype=dt, device="cuda")
    _lib.check(lib.fvit_gemm_bias_act(code, A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), out.data_ptr(), N, M, N, K, 0, _stream()), "gemm")
    torch.cuda.synchronize()
    ref = A.float() @ W.float().t() + bias
    assert torch.isfinite(out.float()).all()
    if dt == torch.float16:
        assert (ref.abs() > 65504).any()
        assert torch.equal(out.float(), ref.clamp(-65504, 65504).to(dt).float()) or (out.float() - ref.clamp(-65504, 65504)).abs().max() < 64
    else:
        assert (out.float() - ref).abs().max().item() < 1e-2 * ref.abs().max().item()


@pytest.mark.parametrize("mode", ["f16", "bf16"])
def test_large_activation_scales_stay_finite(mode):
    """LayerNorm gamma x 50 and a large fc1 / qkv bias push the fp16 operands of a whole HAT stage toward 6e4.
    The fp16 path must return finite logits (saturating narrowing); bf16 operands must still track the fp32 oracle."""
    from oracle.model_reference import model_forward
    from tests.cases import CASES
    model, sd = build_product_model("fvit0_224", "cuda")
    sd = {k: v.clone() for k, v in sd.items()}
    for k in sd:
        if ".blocks." in k and k.startswith(("levels.2.", "levels.3.")):
            if k.endswith(("norm1.weight", "norm2.weight")):
                sd[k] = sd[k] * 50.0
            if k.endswith(("fc1.bias", "qkv.bias")):
                sd[k] = sd[k] + 3000.0 * torch.sign(sd[k])
            if k.endswith(("fc1.weight",)):
                sd[k] = sd[k] * 40.0
    model.load_state_dict(sd)
    model.set_hat_operand_dtype(mode)
    x = case_input("fvit0_224")[:2]
    with torch.no_grad():
        y = model(x.cuda()).float().cpu()
    assert torch.isfinite(y).all(), f"{mode}: non-finite logits"
    ref = model_forward({k: v.cpu() for k, v in sd.items()}, x, CASES["fvit0_224"]["arch"])
    rel = (y - ref).abs().max().item() / ref.abs().max().item()
    print(f"large-scale weights, {mode} operands: logits rel err {rel:.3e} (|logits| max {ref.abs().max().item():.3f})")
    if mode == "bf16":
        assert rel < 5e-2


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("M,N,K,epi", [(300, 768, 256, 0), (1000, 784, 832, 1), (257, 3136, 832, 0), (4214, 512, 2048, 2), (9116, 784, 3136, 2),
                                       (2107, 6272, 1600, 1), (6272, 1568, 6272, 2), (511, 272, 256, 2)])
@pytest.mark.parametrize("pp", [0, 1])
def test_gemm_256x256_tile(opname, dt, code, M, N, K, epi, pp):
    """The 256 x 256 x 64 tile of gemm_kernel (8 waves, r03) forced on through its knob, on ragged shapes (M, N not multiples of 256,
    N not a multiple of 64) and the FasterViT-4 layer shapes: same contract as the 128 x 128 tile, compared against it and fp32 torch.
    pp = 1: the ping-pong form of that tile (gemm_pp_kernel: two wave groups alternate between MFMA and LDS / request sections)."""
    lib = _lib.lib()
    _lib.tune("gemm_pp", pp)
    g = torch.Generator(device="cpu").manual_seed(M + N + K + epi)
    A = torch.randn(M, K, generator=g).to(dt).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dt).cuda()
    bias = torch.randn(N, generator=g).cuda()
    gamma = (torch.rand(N, generator=g) + 0.5).cuda()
    Kp = _rup(K, 64)
    Ap, Wp = _padded(A, _rup(M, 256), Kp), _padded(W, _rup(N, 256), Kp)
    x0 = torch.randn(M, N, generator=g).cuda()
    outs = []
    try:
        for knob in (1, 0):
            _lib.tune("gemm256_min_tiles", knob)
            if epi == 2:
                out = torch.cat([x0.clone(), torch.full((3, N), float("nan"), device="cuda")])
                ldo = N
            else:
                ldo = _rup(N, 64)
                out = torch.full((_rup(M, 256), ldo), float("nan"), dtype=dt, device="cuda")
            _lib.check(lib.fvit_gemm_terms(code, Ap.data_ptr(), Kp, Wp.data_ptr(), Kp, bias.data_ptr(), gamma.data_ptr(), out.data_ptr(), ldo,
                                           M, N, Kp, Kp, epi, _stream()), "gemm_terms")
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        _lib.tune("gemm256_min_tiles", 192)
        _lib.tune("gemm_pp", GEMM_PP_DEFAULT)
    y = A.float() @ W.float().t() + bias
    if epi == 1:
        y = F.gelu(y)
    if epi == 2:
        ref = x0 + gamma * y
        got, got128 = outs[0][:M], outs[1][:M]
        tol = 2e-5 * ref.abs().max().item() + 1e-4
        assert torch.isnan(outs[0][M:]).all()
    else:
        ref = y
        got, got128 = outs[0][:M, :N].float(), outs[1][:M, :N].float()
        tol = (1.5e-3 if dt == torch.float16 else 1e-2) * max(ref.abs().max().item(), 1.0)
        assert torch.isnan(outs[0][M:].float()).all() and torch.isnan(outs[0][:M, N:].float()).all()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < tol
    assert torch.equal(got, got128)   # same K order, same fp32 accumulation chain per output: bitwise the same as the 128 x 128 tile


def test_gemm_ping_pong_repeatable_with_other_streams_on_the_chip():
    """The ping-pong tile's LDS hand-offs are ordered by counted vmcnt + barriers only: the same GEMM, repeated while two other streams keep the CUs
    busy with different kernels (other LDS-DMA traffic, other timing), must return the same bits every time and match the 128 x 128 tile."""
    lib = _lib.lib()
    dt, code = torch.float16, 1
    g = torch.Generator(device="cpu").manual_seed(99)
    M, N, K = 9116, 3072, 832
    A = _padded(torch.randn(M, K, generator=g).to(dt).cuda(), _rup(M, 256), K)
    W = _padded((torch.randn(N, K, generator=g) / K ** 0.5).to(dt).cuda(), _rup(N, 256), K)
    bias = torch.randn(N, generator=g).cuda()
    A2 = torch.randn(4096, 2048, generator=g).to(dt).cuda()
    W2 = torch.randn(4096, 2048, generator=g).to(dt).cuda()
    side = [torch.cuda.Stream() for _ in range(2)]

    def run():
        out = torch.empty(_rup(M, 256), N, dtype=dt, device="cuda")
        _lib.check(lib.fvit_gemm_bias_act(code, A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), out.data_ptr(), N, M, N, K, 1, _stream()), "gemm")
        return out

    try:
        _lib.tune("gemm_pp", 0)
        _lib.tune("gemm256_min_tiles", 0)
        ref = run()[:M].clone()
        _lib.tune("gemm_pp", 1)
        _lib.tune("gemm256_min_tiles", 192)
        torch.cuda.synchronize()
        for it in range(24):
            for i, s2 in enumerate(side):
                with torch.cuda.stream(s2):
                    o2 = torch.empty(4096, 4096, dtype=dt, device="cuda")
                    _lib.check(lib.fvit_gemm_bias_act(code, A2.data_ptr(), 2048, W2.data_ptr(), 2048, None, o2.data_ptr(), 4096, 4096 - 128 * i, 4096, 2048, 0,
                                                      s2.cuda_stream), "side gemm")
            got = run()[:M]
            torch.cuda.synchronize()
            assert torch.equal(got, ref), f"iteration {it}: {(got.float() - ref.float()).abs().max().item()}"
    finally:
        _lib.tune("gemm_pp", GEMM_PP_DEFAULT)
        _lib.tune("gemm256_min_tiles", 192)


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("M,nsplit,terms", [(4214, 2, 1), (70, 2, 1), (12544, 2, 1), (513, 2, 2), (4165, 2, 2)])
def test_win_mlp_split_hidden(opname, dt, code, M, nsplit, terms):
    """C = 512 MLP kernel with the hidden units of each 64-row group split over 2 sibling workgroups that meet in L2 (last-arriver
    reduction in split order): same contract as the unsplit kernel; repeated launches (counters must return to zero) are bitwise equal."""
    lib = _lib.lib()
    C, hid = 512, 2048
    g = torch.Generator(device="cpu").manual_seed(M + nsplit)
    x0 = (torch.randn(M, C, generator=g) * 1.5 + 0.3).cuda()
    lnw = (torch.rand(C, generator=g) + 0.5).cuda()
    lnb = (torch.randn(C, generator=g) * 0.2).cuda()
    w1 = (torch.randn(hid, C, generator=g) / C ** 0.5).cuda()
    b1 = (torch.randn(hid, generator=g) * 0.3).cuda()
    w2 = (torch.randn(C, hid, generator=g) / hid ** 0.5).cuda()
    b2 = (torch.randn(C, generator=g) * 0.3).cuda()
    gamma = (torch.rand(C, generator=g) + 0.5).cuda()
    keep = hat_runtime._Keep(dt, terms)
    w1p, w2p = keep.frag16(hat_runtime.frag_pack_fc1(w1)), keep.frag16(hat_runtime.frag_pack_fc2(w2))
    slab = torch.full((lib.fvit_win_mlp_split_bytes(M, C, nsplit) // 4,), float("nan"), device="cuda")   # scratch content must not matter
    cnt = torch.zeros((M + 63) // 64, dtype=torch.int32, device="cuda")
    outs = []
    for rep in range(3):
        xw = torch.cat([x0, torch.full((5, C), float("nan"), device="cuda")])
        _lib.check(lib.fvit_win_mlp_fused_split(code, xw.data_ptr(), M, C, hid, lnw.data_ptr(), lnb.data_ptr(), ctypes.c_float(1e-5),
                                                w1p.data_ptr(), b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(), gamma.data_ptr(), terms,
                                                slab.data_ptr(), cnt.data_ptr(), nsplit, _stream()), "win_mlp_fused_split")
        torch.cuda.synchronize()
        assert int(cnt.abs().sum().item()) == 0, "arrival counters must be zero again after the launch"
        outs.append(xw)
    if terms == 2:
        (h1, l1), (h2, l2) = _split(w1, dt), _split(w2, dt)
        w1e, w2e = h1.float() + l1.float(), h2.float() + l2.float()
    else:
        w1e, w2e = w1.to(dt).float(), w2.to(dt).float()
    xn = F.layer_norm(x0, (C,), lnw, lnb, 1e-5).to(dt).float()
    h = F.gelu(xn @ w1e.t() + b1).to(dt).float()
    ref = x0 + gamma * (h @ w2e.t() + b2)
    tol = (3e-3 if dt == torch.float16 else 2e-2) * ref.abs().max().item()
    assert torch.isfinite(outs[0][:M]).all() and torch.isnan(outs[0][M:]).all()
    err = (outs[0][:M] - ref).abs().max().item()
    assert err < tol, f"{err} vs {tol}"
    assert torch.equal(outs[0][:M], outs[1][:M]) and torch.equal(outs[0][:M], outs[2][:M])
    # against the unsplit kernel: same numbers up to the fp32 summation order of the partial sums
    x1 = x0.clone()
    _lib.check(lib.fvit_win_mlp_fused_terms(code, x1.data_ptr(), M, C, hid, lnw.data_ptr(), lnb.data_ptr(), ctypes.c_float(1e-5), w1p.data_ptr(),
                                            b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(), gamma.data_ptr(), terms, _stream()), "win_mlp_fused_terms")
    torch.cuda.synchronize()
    assert (x1 - outs[0][:M]).abs().max().item() < 1e-4 * ref.abs().max().item()



def _attention_ref(xin, lnw, lnb, wqkv, bqkv, wproj, bproj, gamma, bias, heads, dt):
    n, S, C = xin.shape
    d = C // heads
    xn = F.layer_norm(xin, (C,), lnw, lnb, 1e-5).to(dt).float()
    qkv = (xn @ wqkv.t() + bqkv).view(n, S, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0].to(dt).float(), qkv[1].to(dt).float(), qkv[2].to(dt).float()
    att = ((q @ k.transpose(-1, -2)) * d ** -0.5 + bias).softmax(-1)
    o = (att @ v).transpose(1, 2).reshape(n, S, C).to(dt).float()
    y = o @ wproj.t() + bproj
    return xin + (gamma * y if gamma is not None else y)


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("S,nwin,use_gamma", [(49, 66, True), (53, 9, False), (64, 3, True)])
def test_win_block_two_weight_terms(opname, dt, code, S, nwin, use_gamma):
    """fvit_win_block_fused_terms (C = 512, stage 3 of FasterViT-0): the fused attention sub-block reading [hi image | lo image] weights
    vs fp32 torch on hi + lo weights; the lo image must matter (bf16: closer to the two-term reference than to the single-term one)."""
    lib = _lib.lib()
    C, heads = 512, 16
    g = torch.Generator(device="cpu").manual_seed(S * 7 + nwin)
    X = (torch.randn(nwin, S, C, generator=g) * 1.3 + 0.2).cuda()
    lnw = (torch.rand(C, generator=g) + 0.5).cuda()
    lnb = (torch.randn(C, generator=g) * 0.2).cuda()
    wqkv = (torch.randn(3 * C, C, generator=g) / C ** 0.5).cuda()
    bqkv = (torch.randn(3 * C, generator=g) * 0.3).cuda()
    wproj = (torch.randn(C, C, generator=g) / C ** 0.5).cuda()
    bproj = (torch.randn(C, generator=g) * 0.3).cuda()
    gamma = (torch.rand(C, generator=g) + 0.5).cuda() if use_gamma else None
    bias = (torch.randn(heads, S, S, generator=g) * 2).cuda()
    bp = torch.zeros(heads, 64, 64, device="cuda")
    bp[:, :S, :S] = bias
    bp[:, :, S:] = _lib.FVIT_MASK_BIAS
    keep = hat_runtime._Keep(dt, 2)
    wqf = keep.frag16(hat_runtime.frag_pack_qkv(wqkv, heads))
    wpf = keep.frag16(hat_runtime.frag_pack_fc2(wproj))
    assert wqf.numel() == 2 * wqkv.numel() and wpf.numel() == 2 * wproj.numel()
    bqh = bqkv.view(3, heads, 32).permute(1, 0, 2).reshape(heads, 96).contiguous()
    out = torch.full((nwin * S + 2, C), float("nan"), device="cuda")
    args = (code, X.data_ptr(), S, None, 0, None, None, None, lnw.data_ptr(), lnb.data_ptr(), ctypes.c_float(1e-5), S, wqf.data_ptr(),
            bqh.data_ptr(), wpf.data_ptr(), bproj.data_ptr(), gamma.data_ptr() if use_gamma else None, bp.data_ptr(), out.data_ptr(), nwin, S,
            heads, C, ctypes.c_float(32 ** -0.5))
    _lib.check(lib.fvit_win_block_fused_terms(*args, 2, _stream()), "win_block_fused_terms")
    torch.cuda.synchronize()
    hq, lq = _split(wqkv, dt)
    hp, lp = _split(wproj, dt)
    ref2 = _attention_ref(X, lnw, lnb, hq.float() + lq.float(), bqkv, hp.float() + lp.float(), bproj, gamma, bias, heads, dt).reshape(-1, C)
    ref1 = _attention_ref(X, lnw, lnb, hq.float(), bqkv, hp.float(), bproj, gamma, bias, heads, dt).reshape(-1, C)
    got = out[:nwin * S]
    assert torch.isfinite(got).all() and torch.isnan(out[nwin * S:]).all()
    e2, e1 = (got - ref2).abs().max().item(), (got - ref1).abs().max().item()
    tol = (4e-3 if dt == torch.float16 else 3e-2) * ref2.abs().max().item()
    assert e2 < tol, f"{e2} vs {tol}"
    if dt == torch.bfloat16:
        assert (got - ref2).abs().mean().item() < (got - ref1).abs().mean().item(), (e2, e1)
    # terms = 1 on the same arrays reads the hi image only: the single-term result
    out1 = torch.full((nwin * S, C), float("nan"), device="cuda")
    args1 = args[:18] + (out1.data_ptr(),) + args[19:]
    _lib.check(lib.fvit_win_block_fused_terms(*args1, 1, _stream()), "win_block_fused_terms(1)")
    torch.cuda.synchronize()
    assert (out1 - ref1).abs().max().item() < tol
    assert lib.fvit_win_block_fused_terms(*args1, 3, _stream()) != 0
    args256 = args1[:21] + (8, 256) + args1[23:]
    assert lib.fvit_win_block_fused_terms(*args256, 2, _stream()) != 0   # two-term weights: C = 512 only


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("batch,G,use_add,use_gamma", [(86, 16, True, True), (7, 9, False, False), (2, 1, True, True)])
@pytest.mark.parametrize("variant", [0, 3])
def test_ct_block_two_weight_terms(opname, dt, code, batch, G, use_add, use_gamma, variant):
    """fvit_ct_block_fused_terms: the carrier-token branch in one kernel with [hi image | lo image] weights in all four fragment arrays
    (variant 0: the 4-wave form, 3: the 8-wave form)."""
    lib = _lib.lib()
    _lib.tune("ct_variant", variant)
    C, heads, hid = 256, 8, 1024
    g = torch.Generator(device="cpu").manual_seed(batch * 19 + G)
    rowsA = 4 * 53
    X = (torch.randn(batch * rowsA, C, generator=g) * 1.3 + 0.2).cuda()
    src_idx = torch.randperm(rowsA, generator=g)[:G].int().cuda()
    add = torch.randn(G, C, generator=g).cuda() if use_add else None
    ln1w, ln2w = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    ln1b, ln2b = (torch.randn(C, generator=g) * 0.2).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
    wqkv = (torch.randn(3 * C, C, generator=g) / C ** 0.5).cuda()
    bqkv = (torch.randn(3 * C, generator=g) * 0.3).cuda()
    wproj = (torch.randn(C, C, generator=g) / C ** 0.5).cuda()
    bproj = (torch.randn(C, generator=g) * 0.3).cuda()
    w1 = (torch.randn(hid, C, generator=g) / C ** 0.5).cuda()
    b1 = (torch.randn(hid, generator=g) * 0.3).cuda()
    w2 = (torch.randn(C, hid, generator=g) / hid ** 0.5).cuda()
    b2 = (torch.randn(C, generator=g) * 0.3).cuda()
    g1 = (torch.rand(C, generator=g) + 0.5).cuda() if use_gamma else None
    g2 = (torch.rand(C, generator=g) + 0.5).cuda() if use_gamma else None
    bias = (torch.randn(heads, G, G, generator=g) * 2).cuda()
    bp = torch.zeros(heads, 16, 16, device="cuda")
    bp[:, :G, :G] = bias
    bp[:, :, G:] = _lib.FVIT_MASK_BIAS
    keep = hat_runtime._Keep(dt, 2)
    wqf = keep.frag16(hat_runtime.frag_pack_qkv(wqkv, heads))
    wpf = keep.frag16(hat_runtime.frag_pack_fc2(wproj))
    w1f = keep.frag16(hat_runtime.frag_pack_fc1(w1))
    w2f = keep.frag16(hat_runtime.frag_pack_fc2(w2))
    bqh = bqkv.view(3, heads, 32).permute(1, 0, 2).reshape(heads, 96).contiguous()
    out = torch.full((batch * G + 3, C), float("nan"), device="cuda")
    p = lambda t: t.data_ptr() if t is not None else None   # noqa: E731

    def call(terms):
        out.fill_(float("nan"))
        return lib.fvit_ct_block_fused_terms(code, X.data_ptr(), rowsA, src_idx.data_ptr(), p(add), out.data_ptr(), batch, G, heads, C, hid,
                                             ln1w.data_ptr(), ln1b.data_ptr(), wqf.data_ptr(), bqh.data_ptr(), wpf.data_ptr(), bproj.data_ptr(),
                                             p(g1), bp.data_ptr(), ctypes.c_float(32 ** -0.5), ln2w.data_ptr(), ln2b.data_ptr(), w1f.data_ptr(),
                                             b1.data_ptr(), w2f.data_ptr(), b2.data_ptr(), p(g2), ctypes.c_float(1e-5), terms, _stream())

    def ref(two):
        ws = []
        for w in (wqkv, wproj, w1, w2):
            hi, lo = _split(w, dt)
            ws.append(hi.float() + lo.float() if two else hi.float())
        ct = X.view(batch, rowsA, C)[:, src_idx.long()]
        if use_add:
            ct = ct + add[None]
        ct = _attention_ref(ct, ln1w, ln1b, ws[0], bqkv, ws[1], bproj, g1, bias, heads, dt)
        xn2 = F.layer_norm(ct, (C,), ln2w, ln2b, 1e-5).to(dt).float()
        y2 = F.gelu(xn2 @ ws[2].t() + b1).to(dt).float() @ ws[3].t() + b2
        return (ct + (g2 * y2 if use_gamma else y2)).reshape(batch * G, C)

    _lib.check(call(2), "ct_block_fused_terms")
    torch.cuda.synchronize()
    got = out[:batch * G].clone()
    assert torch.isfinite(got).all() and torch.isnan(out[batch * G:]).all()
    ref2, ref1 = ref(True), ref(False)
    tol = (4e-3 if dt == torch.float16 else 3e-2) * ref2.abs().max().item()
    assert (got - ref2).abs().max().item() < tol
    if dt == torch.bfloat16:
        assert (got - ref2).abs().mean().item() < (got - ref1).abs().mean().item()
    _lib.check(call(1), "ct_block_fused_terms(1)")
    torch.cuda.synchronize()
    assert (out[:batch * G] - ref1).abs().max().item() < tol
    assert call(0) != 0
    _lib.tune("ct_variant", CT_VARIANT_DEFAULT)


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("S,nwin,use_gamma,gather", [(53, 344, True, True), (53, 9, False, False), (64, 3, True, False), (49, 2, True, False)])
def test_attn_block_two_weight_terms(opname, dt, code, S, nwin, use_gamma, gather):
    """fvit_attn_block_fused_terms (r04; C = 256, the stage-2 window attention of FasterViT-0 in the x2 operand modes): [hi image | lo image] weights on the
    double-buffered 8-wave form, vs fp32 torch on hi + lo weights; odd window counts (two windows per workgroup), the gather / position-embedding prologue,
    and bitwise repeatability."""
    lib = _lib.lib()
    C, heads = 256, 8
    g = torch.Generator(device="cpu").manual_seed(S * 11 + nwin)
    X = (torch.randn(nwin, S, C, generator=g) * 1.3 + 0.2).cuda()
    lnw = (torch.rand(C, generator=g) + 0.5).cuda()
    lnb = (torch.randn(C, generator=g) * 0.2).cuda()
    wqkv = (torch.randn(3 * C, C, generator=g) / C ** 0.5).cuda()
    bqkv = (torch.randn(3 * C, generator=g) * 0.3).cuda()
    wproj = (torch.randn(C, C, generator=g) / C ** 0.5).cuda()
    bproj = (torch.randn(C, generator=g) * 0.3).cuda()
    gamma = (torch.rand(C, generator=g) + 0.5).cuda() if use_gamma else None
    bias = (torch.randn(heads, S, S, generator=g) * 2).cuda()
    bp = torch.zeros(heads, 64, 64, device="cuda")
    bp[:, :S, :S] = bias
    bp[:, :, S:] = _lib.FVIT_MASK_BIAS
    keep = hat_runtime._Keep(dt, 2)
    wqf = keep.frag16(hat_runtime.frag_pack_qkv(wqkv, heads))
    wpf = keep.frag16(hat_runtime.frag_pack_fc2(wproj))
    bqh = bqkv.view(3, heads, 32).permute(1, 0, 2).reshape(heads, 96).contiguous()
    xin = X
    src_idx = add_idx = add = None
    if gather:   # rows of a window come from a permuted source, every row gets a position-embedding row added (fvit_gather_layernorm semantics)
        perm = torch.randperm(S, generator=g)
        src_idx = perm.to(torch.int32).cuda()
        add = (torch.randn(S, C, generator=g) * 0.5).cuda()
        add_idx = torch.arange(S, dtype=torch.int32).cuda()
        xin = X[:, perm.cuda()] + add
    outs = []
    for _ in range(2):
        out = torch.full((nwin * S + 2, C), float("nan"), device="cuda")
        args = (code, X.data_ptr(), S, None, 0, src_idx.data_ptr() if gather else None, add_idx.data_ptr() if gather else None, add.data_ptr() if gather else None,
                lnw.data_ptr(), lnb.data_ptr(), ctypes.c_float(1e-5), S, wqf.data_ptr(), bqh.data_ptr(), wpf.data_ptr(), bproj.data_ptr(),
                gamma.data_ptr() if use_gamma else None, bp.data_ptr(), out.data_ptr(), nwin, S, heads, C, ctypes.c_float(32 ** -0.5))
        _lib.check(lib.fvit_attn_block_fused_terms(*args, 2, _stream()), "attn_block_fused_terms")
        outs.append(out)
    torch.cuda.synchronize()
    hq, lq = _split(wqkv, dt)
    hp, lp = _split(wproj, dt)
    ref2 = _attention_ref(xin, lnw, lnb, hq.float() + lq.float(), bqkv, hp.float() + lp.float(), bproj, gamma, bias, heads, dt).reshape(-1, C)
    ref1 = _attention_ref(xin, lnw, lnb, hq.float(), bqkv, hp.float(), bproj, gamma, bias, heads, dt).reshape(-1, C)
    got = outs[0][:nwin * S]
    assert torch.isfinite(got).all() and torch.isnan(outs[0][nwin * S:]).all() and torch.equal(outs[0][:nwin * S], outs[1][:nwin * S])
    e2 = (got - ref2).abs().max().item()
    tol = (4e-3 if dt == torch.float16 else 3e-2) * ref2.abs().max().item()
    assert e2 < tol, f"{e2} vs {tol}"
    if dt == torch.bfloat16:
        assert (got - ref2).abs().mean().item() < (got - ref1).abs().mean().item()
    # one term through the same entry point = fvit_attn_block_fused on the hi image
    out1 = torch.full((nwin * S, C), float("nan"), device="cuda")
    args1 = args[:18] + (out1.data_ptr(),) + args[19:]
    _lib.check(lib.fvit_attn_block_fused_terms(*args1, 1, _stream()), "attn_block_fused_terms(1)")
    torch.cuda.synchronize()
    assert (out1 - ref1).abs().max().item() < tol
    assert lib.fvit_attn_block_fused_terms(*args1, 3, _stream()) != 0
