"""Kernel-level parity through the C ABI (needs an MI355X): each HIP kernel against a plain
PyTorch fp32 statement of the same op on the same (16-bit rounded) operands."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from fastervit_amd import _lib, hat_runtime
from oracle import hat_reference as hr

pytestmark = pytest.mark.gpu

OPS = [("f16", torch.float16, 1), ("bf16", torch.bfloat16, 2)]


def _rup(x, m):
    return (x + m - 1) // m * m


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _padded(t, rows, cols):
    out = torch.zeros(rows, cols, dtype=t.dtype, device=t.device)
    out[:t.shape[0], :t.shape[1]] = t
    return out


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("M,N,K", [(300, 768, 256), (128, 256, 1024), (1000, 784, 784), (77, 3136, 784), (54272, 1024, 256),
                                   (4096, 16, 64)])
@pytest.mark.parametrize("act", [0, 1])
def test_gemm_bias_act(opname, dt, code, M, N, K, act):
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g)).to(dt).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dt).cuda()
    bias = torch.randn(N, generator=g).cuda()
    Kp = _rup(K, 64)
    Ap, Wp = _padded(A, _rup(M, 128), Kp), _padded(W, _rup(N, 128), Kp)
    ldo = _rup(N, 64)
    out = torch.full((_rup(M, 128), ldo), float("nan"), dtype=dt, device="cuda")
    rc = lib.fvit_gemm_bias_act(code, Ap.data_ptr(), Kp, Wp.data_ptr(), Kp, bias.data_ptr(), out.data_ptr(), ldo, M, N, Kp, act,
                                _stream())
    _lib.check(rc, "gemm")
    torch.cuda.synchronize()
    ref = A.float() @ W.float().t() + bias
    if act:
        ref = F.gelu(ref)
    got = out[:M, :N].float()
    tol = (4e-3 if dt == torch.float16 else 2e-2) * max(ref.abs().max().item(), 1.0)
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < tol
    # rows beyond M and columns beyond N are never written
    assert torch.isnan(out[M:].float()).all() and torch.isnan(out[:M, N:].float()).all()


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("M,N,K,use_gamma", [(300, 256, 256, True), (212, 784, 3136, True), (1000, 512, 2048, False),
                                              (54272, 256, 1024, True)])
def test_gemm_residual(opname, dt, code, M, N, K, use_gamma):
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(M * 3 + N + K)
    A = torch.randn(M, K, generator=g).to(dt).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dt).cuda()
    bias = torch.randn(N, generator=g).cuda()
    gamma = (torch.rand(N, generator=g) + 0.5).cuda() if use_gamma else None
    x0 = torch.randn(M, N, generator=g).cuda()
    x = x0.clone()
    Kp = _rup(K, 64)
    Ap, Wp = _padded(A, _rup(M, 128), Kp), _padded(W, _rup(N, 128), Kp)
    rc = lib.fvit_gemm_residual(code, Ap.data_ptr(), Kp, Wp.data_ptr(), Kp, bias.data_ptr(),
                                gamma.data_ptr() if use_gamma else None, x.data_ptr(), N, M, N, Kp, _stream())
    _lib.check(rc, "gemm_residual")
    torch.cuda.synchronize()
    y = A.float() @ W.float().t() + bias
    ref = x0 + (gamma * y if use_gamma else y)
    assert (x - ref).abs().max().item() < 2e-4 * max(ref.abs().max().item(), 1.0) * (K / 256) ** 0.5


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("S,dpad,d,heads,nwin", [(53, 32, 32, 8, 7), (49, 32, 32, 16, 5), (16, 32, 32, 8, 6), (53, 64, 49, 16, 3),
                                                  (148, 64, 49, 4, 3), (36, 64, 49, 3, 5), (60, 64, 49, 4, 2),
                                                  (196, 32, 16, 2, 2), (13, 32, 24, 4, 9), (32, 32, 32, 2, 4),
                                                  # head_dim 80 (FasterViT-5 / -6) -> padded to 96
                                                  (53, 96, 80, 4, 3), (49, 96, 80, 2, 5), (16, 96, 80, 2, 6), (128, 96, 72, 2, 2)])
def test_window_attention(opname, dt, code, S, dpad, d, heads, nwin):
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(S * 7 + d)
    rows = nwin * S
    q, k, v = (torch.randn(nwin, heads, S, d, generator=g).to(dt) for _ in range(3))
    bias = torch.randn(heads, S, S, generator=g) * 2
    spad = lib.fvit_attention_spad(S)
    ldq = 3 * heads * dpad
    ldo = _rup(heads * dpad, 64)
    qkv = torch.zeros(_rup(rows, 128), ldq, dtype=dt)
    for si, t in enumerate((q, k, v)):
        qkv[:rows].view(nwin, S, 3, heads, dpad)[:, :, si, :, :d] = t.permute(0, 2, 1, 3)
    bp = torch.zeros(heads, spad, spad)
    bp[:, :S, :S] = bias
    bp[:, :, S:] = _lib.FVIT_MASK_BIAS
    qkv, bp = qkv.cuda(), bp.cuda()
    out = torch.zeros(_rup(rows, 128), ldo, dtype=dt, device="cuda")
    scale = d ** -0.5
    rc = lib.fvit_window_attention(code, qkv.data_ptr(), ldq, out.data_ptr(), ldo, bp.data_ptr(), nwin, S, heads, dpad,
                                   ctypes.c_float(scale), _stream())
    _lib.check(rc, "attention")
    torch.cuda.synchronize()
    att = (q.float() @ k.float().transpose(-1, -2)) * scale + bias
    ref = att.softmax(-1) @ v.float()                       # (nwin, heads, S, d)
    got = out[:rows, :heads * dpad].float().cpu().view(nwin, S, heads, dpad).permute(0, 2, 1, 3)
    assert torch.isfinite(got).all()
    assert (got[..., :d] - ref).abs().max().item() < (4e-3 if dt == torch.float16 else 2.5e-2) * max(ref.abs().max().item(), 1.0)
    assert got[..., d:].abs().max().item() == 0.0 if d < dpad else True


def _rel_bias_dense(rel, w, ng, S):
    """(heads, S, S) bias from the compact table the way the reference gathers it (FV:243-258, 276-299): token ng + y*w + x."""
    tw = 2 * w - 1
    pos = torch.arange(w * w)
    y, x = pos // w, pos % w
    idx = (y[:, None] - y[None, :] + w - 1) * tw + (x[:, None] - x[None, :] + w - 1)     # [query][key]
    dense = torch.zeros(rel.shape[0], S, S)
    dense[:, ng:, ng:] = rel[:, idx.view(-1)].view(-1, w * w, w * w)
    return dense


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("S,w,dpad,d,heads,nwin", [(256, 16, 32, 32, 4, 3), (260, 16, 64, 49, 3, 2), (580, 24, 64, 49, 2, 1), (209, 14, 32, 32, 2, 5),
                                                    (1024, 32, 64, 49, 2, 1), (233, 15, 32, 24, 3, 2), (53, 7, 32, 32, 8, 4),
                                                    (260, 16, 96, 80, 2, 2), (148, 12, 96, 80, 2, 3)])
def test_window_attention_long(opname, dt, code, S, w, dpad, d, heads, nwin):
    """Online-softmax attention for long windows: bias looked up arithmetically from the compact (heads, (2w-1)^2) table, n_g = S - w^2
    leading tokens without bias; vs PyTorch fp32 with the densely gathered bias.  Any S is accepted (the 53-token case cross-checks the
    short-window kernel's domain)."""
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(S * 11 + d)
    ng = S - w * w
    rows = nwin * S
    q, k, v = (torch.randn(nwin, heads, S, d, generator=g).to(dt) for _ in range(3))
    rel = torch.randn(heads, (2 * w - 1) ** 2, generator=g) * 2
    ldq = 3 * heads * dpad
    ldo = _rup(heads * dpad, 64)
    qkv = torch.zeros(_rup(rows, 128), ldq, dtype=dt)
    for si, t in enumerate((q, k, v)):
        qkv[:rows].view(nwin, S, 3, heads, dpad)[:, :, si, :, :d] = t.permute(0, 2, 1, 3)
    qkv, relc = qkv.cuda(), rel.cuda()
    out = torch.zeros(_rup(rows, 128), ldo, dtype=dt, device="cuda")
    scale = d ** -0.5
    rc = lib.fvit_window_attention_long(code, qkv.data_ptr(), ldq, out.data_ptr(), ldo, relc.data_ptr(), w, ng, nwin, S, heads, dpad,
                                        ctypes.c_float(scale), _stream())
    _lib.check(rc, "attention_long")
    torch.cuda.synchronize()
    att = (q.float() @ k.float().transpose(-1, -2)) * scale + _rel_bias_dense(rel, w, ng, S)
    ref = att.softmax(-1) @ v.float()
    got = out[:rows, :heads * dpad].float().cpu().view(nwin, S, heads, dpad).permute(0, 2, 1, 3)
    assert torch.isfinite(got).all()
    assert (got[..., :d] - ref).abs().max().item() < (4e-3 if dt == torch.float16 else 2.5e-2) * max(ref.abs().max().item(), 1.0)
    if d < dpad:
        assert got[..., d:].abs().max().item() == 0.0
    # no bias table at all
    rc = lib.fvit_window_attention_long(code, qkv.data_ptr(), ldq, out.data_ptr(), ldo, None, 0, 0, nwin, S, heads, dpad,
                                        ctypes.c_float(scale), _stream())
    _lib.check(rc, "attention_long (no bias)")
    torch.cuda.synchronize()
    ref0 = ((q.float() @ k.float().transpose(-1, -2)) * scale).softmax(-1) @ v.float()
    got0 = out[:rows, :heads * dpad].float().cpu().view(nwin, S, heads, dpad).permute(0, 2, 1, 3)
    assert (got0[..., :d] - ref0).abs().max().item() < (4e-3 if dt == torch.float16 else 2.5e-2) * max(ref0.abs().max().item(), 1.0)


def test_window_attention_long_rejects_bad_geometry():
    lib = _lib.lib()
    t = torch.zeros(1024, 192, dtype=torch.float16, device="cuda")
    rel = torch.zeros(2, 31 * 31, device="cuda")
    rc = lib.fvit_window_attention_long(1, t.data_ptr(), 192, t.data_ptr(), 64, rel.data_ptr(), 16, 3, 1, 256, 2, 32, ctypes.c_float(1.0), _stream())
    assert rc != 0 and b"n_g + w^2" in lib.fvit_last_error()
    rc = lib.fvit_window_attention(1, t.data_ptr(), 192, t.data_ptr(), 64, rel.data_ptr(), 1, 256, 2, 32, ctypes.c_float(1.0), _stream())
    assert rc != 0 and b"fvit_window_attention_long" in lib.fvit_last_error()


@pytest.mark.parametrize("C", [256, 784, 1568, 64, 2560])
def test_gather_layernorm_plain(C):
    lib = _lib.lib()
    rows = 203
    g = torch.Generator(device="cpu").manual_seed(C)
    x = (torch.randn(rows, C, generator=g) * 2 + 0.5).cuda()
    w = (torch.rand(C, generator=g) + 0.5).cuda()
    b = torch.randn(C, generator=g).cuda()
    ldn = _rup(C, 64)
    n = torch.full((rows, ldn), float("nan"), dtype=torch.float16, device="cuda")
    rc = lib.fvit_gather_layernorm(1, x.data_ptr(), 0, None, 0, None, None, None, None, n.data_ptr(), ldn, w.data_ptr(),
                                   b.data_ptr(), ctypes.c_float(1e-5), rows, 1, C, _stream())
    _lib.check(rc, "layernorm")
    torch.cuda.synchronize()
    ref = F.layer_norm(x, (C,), w, b, 1e-5)
    assert (n[:, :C].float() - ref).abs().max().item() < 4e-3 * ref.abs().max().item()
    if ldn > C:
        assert (n[:, C:] == 0).all()


def test_gather_layernorm_tables():
    """The norm1 pass of a hierarchical block: gather carrier rows from R, add pe to local rows, write X back."""
    lib = _lib.lib()
    sr0, sr1, ws, cw, C, B = 2, 4, 3, 2, 64, 3
    tb = hat_runtime.build_tables(sr0, sr1, ws, cw, True)
    nW, S, G = tb["nW"], tb["S"], tb["G"]
    g = torch.Generator(device="cpu").manual_seed(5)
    X = torch.randn(B, nW * S, C, generator=g).cuda()
    R = torch.randn(B, G, C, generator=g).cuda()
    pe = torch.randn(ws * ws, C, generator=g).cuda()
    w = (torch.rand(C, generator=g) + 0.5).cuda()
    b = torch.randn(C, generator=g).cuda()
    src, add = tb["ln1_src"].cuda(), tb["ln1_add"].cuda()
    xo = X.clone()
    n = torch.zeros(B * nW * S, 64, dtype=torch.float16, device="cuda")
    rc = lib.fvit_gather_layernorm(1, xo.data_ptr(), nW * S, R.data_ptr(), G, src.data_ptr(), add.data_ptr(), pe.data_ptr(),
                                   xo.data_ptr(), n.data_ptr(), 64, w.data_ptr(), b.data_ptr(), ctypes.c_float(1e-5),
                                   B * nW * S, nW * S, C, _stream())
    _lib.check(rc, "layernorm")
    torch.cuda.synchronize()
    srcl, addl = src.long(), add.long()
    gathered = torch.where((srcl >= 0)[None, :, None], X[:, srcl.clamp(min=0)], R[:, (-srcl - 1).clamp(min=0)])
    gathered = gathered + torch.where((addl >= 0)[:, None], pe[addl.clamp(min=0)], torch.zeros_like(pe[:1]))[None]
    assert torch.equal(xo, gathered)
    ref = F.layer_norm(gathered, (C,), w, b, 1e-5).view(-1, C)
    assert (n.float() - ref).abs().max().item() < 4e-3 * ref.abs().max().item()


@pytest.mark.parametrize("fmt", ["nchw", "nhwc"])
@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,C,Hp,Wp,H,W,ws", [(3, 64, 14, 14, 14, 14, 7), (2, 48, 6, 12, 6, 10, 3), (1, 784, 36, 60, 36, 60, 12)])
def test_window_partition_reverse(fmt, dt, B, C, Hp, Wp, H, W, ws):
    lib = _lib.lib()
    x = torch.randn(B, C, Hp, Wp).to(dt).cuda()
    if fmt == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    win = torch.full((B * (Hp // ws) * (Wp // ws), ws * ws, C), float("nan"), device="cuda")
    v = hat_runtime._map_view(x)
    _lib.check(lib.fvit_window_partition(ctypes.byref(v), B, C, Hp, Wp, ws, win.data_ptr(), _stream()), "partition")
    torch.cuda.synchronize()
    assert torch.equal(win, hr.window_partition(x.float(), ws))  # bit exact: pure data movement
    out = torch.full((B, C, H, W), float("nan"), dtype=dt, device="cuda")
    if fmt == "nhwc":
        out = out.contiguous(memory_format=torch.channels_last)
    vo = hat_runtime._map_view(out)
    _lib.check(lib.fvit_window_reverse(win.data_ptr(), B, C, Hp, Wp, H, W, ws, ctypes.byref(vo), _stream()), "reverse")
    torch.cuda.synchronize()
    assert torch.equal(out, x[:, :, :H, :W])


def test_errors_are_reported_not_swallowed():
    lib = _lib.lib()
    rc = lib.fvit_gemm_bias_act(1, None, 100, None, 100, None, None, 64, 10, 10, 100, 0, _stream())
    assert rc == -1 and b"gemm" in lib.fvit_last_error()
    with pytest.raises(RuntimeError, match="gemm"):
        _lib.check(rc, "gemm")
    rc = lib.fvit_window_attention(1, None, 96, None, 64, None, 1, 300, 1, 32, ctypes.c_float(1.0), _stream())
    assert rc == -1 and b"fvit_window_attention_long" in lib.fvit_last_error()
    rc = lib.fvit_window_attention(1, None, 96, None, 64, None, 1, 64, 1, 48, ctypes.c_float(1.0), _stream())   # dpad must be 32 / 64 / 96
    assert rc == -1 and b"unsupported geometry" in lib.fvit_last_error()


@pytest.mark.parametrize("dt,code", [(torch.float16, 1), (torch.bfloat16, 2)])
@pytest.mark.parametrize("C", [64, 196, 256])
def test_glue_kernels(dt, code, C):
    """bias+activation, bias+residual, LayerNorm2d on channels_last 16-bit maps vs PyTorch."""
    lib = _lib.lib()
    B, H, W = 3, 14, 9
    g = torch.Generator(device="cpu").manual_seed(C)
    x = torch.randn(B, C, H, W, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last)
    y = torch.randn(B, C, H, W, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last)
    bias = torch.randn(C, generator=g).cuda()
    tol = 1e-2 if dt == torch.float16 else 6e-2
    for act, fn in [(0, lambda t: t), (1, torch.relu), (2, F.gelu)]:
        z = x.clone()
        _lib.check(lib.fvit_bias_act_cl(code, z.data_ptr(), bias.data_ptr(), B * H * W, C, act, _stream()), "bias_act")
        ref = fn(x.float() + bias.view(1, -1, 1, 1))
        assert (z.float() - ref).abs().max().item() < tol
    z = x.clone()
    _lib.check(lib.fvit_bias_residual_cl(code, z.data_ptr(), y.data_ptr(), bias.data_ptr(), B * H * W, C, _stream()), "bias_res")
    assert (z.float() - (x.float() + y.float() + bias.view(1, -1, 1, 1))).abs().max().item() < 2 * tol
    if C % 8 == 0:
        w = (torch.rand(C, generator=g) + 0.5).cuda()
        out = torch.empty_like(x)
        _lib.check(lib.fvit_layernorm2d_cl(code, x.data_ptr(), out.data_ptr(), w.data_ptr(), bias.data_ptr(),
                                           ctypes.c_float(1e-6), B * H * W, C, 0, _stream()), "ln2d")
        ref = F.layer_norm(x.float().permute(0, 2, 3, 1), (C,), w, bias, 1e-6).permute(0, 3, 1, 2)
        assert (out.float() - ref).abs().max().item() < 4 * tol
    torch.cuda.synchronize()


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("M,use_gamma,C", [(128, True, 256), (300, False, 256), (54272, True, 256), (4096, True, 256), (17, True, 256),
                                           (4165, True, 512), (12544, False, 512), (70, True, 512)])
def test_mlp_fused(opname, dt, code, M, use_gamma, C):
    """x += gamma * fc2(GELU(fc1(LN(x)))) in one kernel vs PyTorch fp32 on 16-bit-rounded weights (C = 256: stage 2, 512: stage 3)."""
    lib = _lib.lib()
    hid = 4 * C
    assert lib.fvit_mlp_fused_supported(C, hid) == 1 and lib.fvit_mlp_fused_supported(784, 3136) == 0
    g = torch.Generator(device="cpu").manual_seed(M)
    x0 = (torch.randn(M, C, generator=g) * 1.5 + 0.3).cuda()
    lnw = (torch.rand(C, generator=g) + 0.5).cuda()
    lnb = (torch.randn(C, generator=g) * 0.2).cuda()
    w1 = (torch.randn(hid, C, generator=g) / C ** 0.5).to(dt).cuda()
    b1 = (torch.randn(hid, generator=g) * 0.3).cuda()
    w2 = (torch.randn(C, hid, generator=g) / hid ** 0.5).to(dt).cuda()
    b2 = (torch.randn(C, generator=g) * 0.3).cuda()
    gamma = (torch.rand(C, generator=g) + 0.5).cuda() if use_gamma else None
    w1p = hat_runtime.frag_pack_fc1(w1.float()).to(dt).contiguous()
    w2c = hat_runtime.frag_pack_fc2(w2.float()).to(dt).contiguous()
    x = x0.clone()
    rc = lib.fvit_mlp_fused(code, x.data_ptr(), M, C, hid, lnw.data_ptr(), lnb.data_ptr(), ctypes.c_float(1e-5), w1p.data_ptr(),
                            b1.data_ptr(), w2c.data_ptr(), b2.data_ptr(), gamma.data_ptr() if use_gamma else None, _stream())
    _lib.check(rc, "mlp_fused")
    torch.cuda.synchronize()
    xn = F.layer_norm(x0, (C,), lnw, lnb, 1e-5).to(dt).float()
    h = F.gelu(xn @ w1.float().t() + b1).to(dt).float()
    y = h @ w2.float().t() + b2
    ref = x0 + (gamma * y if use_gamma else y)
    tol = (3e-3 if dt == torch.float16 else 2e-2) * ref.abs().max().item()
    assert torch.isfinite(x).all()
    assert (x - ref).abs().max().item() < tol
    assert lib.fvit_win_mlp_supported(C, hid) == 1 and lib.fvit_win_mlp_supported(784, 3136) == 0
    # the same contract with the N-split work split (fvit_winmlp.hip: 64-row workgroups), in its plain and its software-pipelined main loop (r06: the default at
    # C = 512; C = 256 always runs the plain loop, the knob is a no-op there)
    outs = {}
    for pipe in (0, 1):
        _lib.tune("win_mlp_pipe", pipe)
        xw = torch.cat([x0, torch.full((5, C), float("nan"), device="cuda")])   # rows beyond M stay untouched
        rc = lib.fvit_win_mlp_fused(code, xw.data_ptr(), M, C, hid, lnw.data_ptr(), lnb.data_ptr(), ctypes.c_float(1e-5), w1p.data_ptr(),
                                    b1.data_ptr(), w2c.data_ptr(), b2.data_ptr(), gamma.data_ptr() if use_gamma else None, _stream())
        _lib.tune("win_mlp_pipe", 1)
        _lib.check(rc, "win_mlp_fused")
        torch.cuda.synchronize()
        assert torch.isfinite(xw[:M]).all() and torch.isnan(xw[M:]).all()
        assert (xw[:M] - ref).abs().max().item() < tol, f"win_mlp (pipe {pipe}): {(xw[:M] - ref).abs().max().item()} vs {tol}"
        outs[pipe] = xw[:M].clone()
    # the pipelined loop runs the same operations per value in the same order: bitwise the plain form
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("C,Cv", [(256, 196), (448, 392), (64, 16), (128, 80)])
def test_layernorm2d_channel_padded(C, Cv):
    """LayerNorm2d over the first Cv of C channels (zero pad channels in, zero weight/bias there, zeros out) vs F.layer_norm on Cv."""
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(C + Cv)
    x = torch.zeros(2, 9, 7, C)
    x[..., :Cv] = torch.randn(2, 9, 7, Cv, generator=g) * 2 + 0.3
    x = x.half().cuda()
    w = torch.zeros(C)
    b = torch.zeros(C)
    w[:Cv] = torch.rand(Cv, generator=g) + 0.5
    b[:Cv] = torch.randn(Cv, generator=g)
    w, b = w.cuda(), b.cuda()
    out = torch.full_like(x, float("nan"))
    _lib.check(lib.fvit_layernorm2d_cl(1, x.data_ptr(), out.data_ptr(), w.data_ptr(), b.data_ptr(), ctypes.c_float(1e-6), 2 * 9 * 7, C, Cv,
                                       _stream()), "ln2d padded")
    torch.cuda.synchronize()
    ref = F.layer_norm(x[..., :Cv].float(), (Cv,), w[:Cv], b[:Cv], 1e-6)
    assert (out[..., :Cv].float() - ref).abs().max().item() < 1e-2 * max(ref.abs().max().item(), 1.0)
    assert out[..., Cv:].abs().max().item() == 0.0


@pytest.mark.parametrize("dt,code", [(torch.float16, 1), (torch.bfloat16, 2)])
@pytest.mark.parametrize("B,Ci,Co,H,W,stride,act,res", [
    (2, 64, 64, 14, 14, 1, 2, False), (3, 64, 64, 9, 13, 1, 0, True), (2, 64, 64, 16, 16, 2, 1, False),
    (2, 64, 128, 12, 10, 2, 0, False), (2, 128, 128, 7, 9, 1, 2, False), (1, 128, 128, 28, 28, 1, 0, True),
    (2, 128, 256, 14, 14, 2, 0, False), (1, 256, 512, 14, 14, 2, 0, False), (5, 64, 64, 56, 56, 1, 0, True),
    # Cout % 128 == 64: 128 x 128 tiles with a ragged last N tile (r05; FasterViT-4's 448 / 832 padded channels)
    (2, 128, 192, 9, 11, 1, 2, True), (1, 256, 448, 14, 14, 1, 0, True), (2, 448, 832, 8, 6, 2, 0, False),
    # halo-tiled kernel (Cin = Cout = 64, stride 1): ragged tiles in both directions, single pixel rows/columns, many tiles per workgroup
    (1, 64, 64, 8, 16, 1, 0, False), (2, 64, 64, 1, 1, 1, 1, True), (1, 64, 64, 3, 40, 1, 2, True), (3, 64, 64, 37, 5, 1, 0, False),
    (2, 64, 64, 17, 33, 1, 2, True), (40, 64, 64, 56, 56, 1, 1, True)])
def test_conv3x3_fused(dt, code, B, Ci, Co, H, W, stride, act, res):
    """Implicit-GEMM 3x3 conv + bias + activation (+ residual) on channels_last 16-bit maps vs F.conv2d in fp32."""
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(Ci + Co + H)
    x = torch.randn(B, Ci, H, W, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5).to(dt).cuda()
    bias = torch.randn(Co, generator=g).cuda()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    r = torch.randn(B, Co, Ho, Wo, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last) if res else None
    wk = w.permute(0, 2, 3, 1).contiguous()
    zeros = torch.zeros(256, dtype=dt, device="cuda")
    out = torch.full((B, Co, Ho, Wo), float("nan"), dtype=dt, device="cuda").contiguous(memory_format=torch.channels_last)
    rc = lib.fvit_conv3x3_nhwc(code, x.data_ptr(), wk.data_ptr(), bias.data_ptr(), r.data_ptr() if res else None, out.data_ptr(), B, H, W,
                               Ci, Co, stride, act, zeros.data_ptr(), _stream())
    _lib.check(rc, "conv3x3")
    torch.cuda.synchronize()
    ref = F.conv2d(x.float(), w.float(), bias, stride, 1)
    ref = [lambda t: t, torch.relu, F.gelu][act](ref)
    if res:
        ref = ref + r.float()
    assert torch.isfinite(out.float()).all()
    tol = (5e-3 if dt == torch.float16 else 3e-2) * max(ref.abs().max().item(), 1.0)
    assert (out.float() - ref).abs().max().item() < tol
    if res:  # in place into the residual buffer
        r2 = r.clone()
        _lib.check(lib.fvit_conv3x3_nhwc(code, x.data_ptr(), wk.data_ptr(), bias.data_ptr(), r2.data_ptr(), r2.data_ptr(), B, H, W, Ci, Co,
                                         stride, act, zeros.data_ptr(), _stream()), "conv3x3 in place")
        torch.cuda.synchronize()
        assert torch.equal(r2, out)
    if Co % 128 == 64 and Co > 128:   # the 128 x 64-tile walk (fvit_tune conv_n128_ragged = 0) sums every output over the same K steps: same bits
        out2 = torch.full_like(out, float("nan"))
        try:
            _lib.tune("conv_n128_ragged", 0)
            _lib.check(lib.fvit_conv3x3_nhwc(code, x.data_ptr(), wk.data_ptr(), bias.data_ptr(), r.data_ptr() if res else None, out2.data_ptr(), B, H, W,
                                             Ci, Co, stride, act, zeros.data_ptr(), _stream()), "conv3x3 128x64 tiles")
        finally:
            _lib.tune("conv_n128_ragged", 1)
        torch.cuda.synchronize()
        assert torch.equal(out2, out)


def _dense_rows(w_cl, cv):
    """[Co][3][3][Ci] -> the dense-K rows of fvit_conv3x3_nhwc_dense: column t * cv + c, zero tail up to fvit_conv3x3_dense_k(cv)."""
    co = w_cl.shape[0]
    kd = _lib.lib().fvit_conv3x3_dense_k(cv)
    d = torch.zeros(co, kd, dtype=w_cl.dtype, device=w_cl.device)
    d[:, :9 * cv] = w_cl[..., :cv].reshape(co, 9 * cv)
    return d


@pytest.mark.parametrize("dt,code", [(torch.float16, 1), (torch.bfloat16, 2)])
@pytest.mark.parametrize("B,Ci,Cv,Co,H,W,stride,act,res,terms", [
    (2, 256, 200, 256, 14, 14, 1, 2, False, 1),    # FasterViT-4 level 0 (196 -> 200 of 256): K steps straddle taps, 29 instead of 36
    (1, 448, 392, 448, 9, 11, 1, 0, True, 1),      # level 1 (392 of 448), ragged last N tile, residual in place
    (2, 448, 392, 832, 8, 6, 2, 0, False, 2),      # Downsample 392 -> 784 with two-term weights
    (3, 64, 24, 64, 9, 13, 1, 0, True, 1),         # the tiny test models: cv < 64, a K step spans three taps; 128 x 64 tiles
    (2, 64, 16, 64, 16, 16, 2, 1, False, 2), (2, 128, 72, 192, 12, 10, 2, 2, False, 1), (1, 64, 8, 128, 5, 7, 1, 0, False, 1),
    (2, 128, 128, 256, 7, 9, 1, 2, False, 1)])     # cin_valid == Cin: the classic kernel behind the same entry point
def test_conv3x3_dense_k(dt, code, B, Ci, Cv, Co, H, W, stride, act, res, terms):
    """r06: the implicit-GEMM conv contracting over the cin_valid real channels of a channel-padded map (fvit_conv3x3_nhwc_dense) vs F.conv2d in
    fp32 on those channels, and vs the classic kernel on the zero-padded weight matrix.  The pad channels of the INPUT hold garbage here: the dense
    kernel must never read them."""
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(Ci + Co + H + Cv)
    x = torch.randn(B, Ci, H, W, generator=g)
    x[:, Cv:] = 1000.0                                    # poison: a classic contraction would see it (its weights there are zero, but 1000 * 0 must not even be formed from a NaN)
    xz = x.clone(); xz[:, Cv:] = 0
    x, xz = [t.to(dt).cuda().contiguous(memory_format=torch.channels_last) for t in (x, xz)]
    w = torch.zeros(Co, Ci, 3, 3)
    w[:, :Cv] = torch.randn(Co, Cv, 3, 3, generator=g) / (9 * Cv) ** 0.5
    bias = torch.randn(Co, generator=g).cuda()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    r = torch.randn(B, Co, Ho, Wo, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last) if res else None
    wh = w.to(dt)
    wl = (w - wh.float()).to(dt)
    cl = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()   # noqa: E731
    dense = _dense_rows(cl(wh), Cv) if Cv < Ci else cl(wh).reshape(Co, -1)
    classic = cl(wh).reshape(Co, -1)
    if terms == 2:
        dense = torch.cat([dense, _dense_rows(cl(wl), Cv) if Cv < Ci else cl(wl).reshape(Co, -1)], dim=1).contiguous()
        classic = torch.cat([classic, cl(wl).reshape(Co, -1)], dim=1).contiguous()
    assert dense.shape[1] == terms * (lib.fvit_conv3x3_dense_k(Cv) if Cv < Ci else 9 * Ci)
    zeros = torch.zeros(256, dtype=dt, device="cuda")
    out = torch.full((B, Co, Ho, Wo), float("nan"), dtype=dt, device="cuda").contiguous(memory_format=torch.channels_last)
    _lib.check(lib.fvit_conv3x3_nhwc_dense(code, x.data_ptr(), dense.data_ptr(), bias.data_ptr(), r.data_ptr() if res else None, out.data_ptr(),
                                           B, H, W, Ci, Cv, Co, stride, act, terms, zeros.data_ptr(), _stream()), "conv3x3 dense")
    old = torch.full_like(out, float("nan"))
    _lib.check(lib.fvit_conv3x3_nhwc_terms(code, xz.data_ptr(), classic.data_ptr(), bias.data_ptr(), r.data_ptr() if res else None, old.data_ptr(),
                                           B, H, W, Ci, Co, stride, act, terms, zeros.data_ptr(), _stream()), "conv3x3 classic")
    torch.cuda.synchronize()
    wref = (wh.float() + wl.float()) if terms == 2 else wh.float()
    ref = F.conv2d(xz.float(), wref.cuda(), bias, stride, 1)
    ref = [lambda t: t, torch.relu, F.gelu][act](ref)
    if res:
        ref = ref + r.float()
    scale = max(ref.abs().max().item(), 1.0)
    assert torch.isfinite(out.float()).all()
    assert (out.float() - ref).abs().max().item() < (5e-3 if dt == torch.float16 else 3e-2) * scale
    # same products, another grouping into MFMA steps: the two kernels differ by fp32 summation order + one output rounding at most
    assert (out.float() - old.float()).abs().max().item() <= (2.0 ** -10 if dt == torch.float16 else 2.0 ** -7) * scale
    if res:
        r2 = r.clone()
        _lib.check(lib.fvit_conv3x3_nhwc_dense(code, x.data_ptr(), dense.data_ptr(), bias.data_ptr(), r2.data_ptr(), r2.data_ptr(), B, H, W, Ci, Cv, Co,
                                               stride, act, terms, zeros.data_ptr(), _stream()), "conv3x3 dense in place")
        torch.cuda.synchronize()
        assert torch.equal(r2, out)
    # argument checks: cin_valid must be a multiple of 8 in (0, Cin]
    for bad in (0, Cv + 4, Ci + 8):
        assert lib.fvit_conv3x3_nhwc_dense(code, x.data_ptr(), dense.data_ptr(), bias.data_ptr(), None, out.data_ptr(), B, H, W, Ci, bad, Co, stride, act,
                                           terms, zeros.data_ptr(), _stream()) != 0
    assert lib.fvit_conv3x3_dense_k(Cv + 4) == -1


@pytest.mark.parametrize("dt,code", [(torch.float16, 1), (torch.bfloat16, 2)])
@pytest.mark.parametrize("B,Ci,Co,H,W,act,res,terms", [
    (2, 256, 256, 56, 56, 2, False, 1),     # FasterViT-4 level 0: 7 x 3.5 patches per image (ragged right column)
    (1, 128, 128, 24, 32, 0, True, 1),      # exact patch grid, residual in place
    (3, 128, 192, 9, 13, 1, True, 1),       # ragged in both directions + ragged last N tile
    (2, 448, 448, 14, 20, 0, False, 2),     # two-term weights: both terms read the same halo tile
    (2, 64, 128, 8, 16, 2, False, 1),       # one chunk, one patch per image
    (1, 192, 256, 1, 1, 0, True, 2)])       # a single pixel
def test_conv3x3_patch_form(dt, code, B, Ci, Co, H, W, act, res, terms):
    """r06: the PATCH form of the implicit-GEMM conv (conv3x3_kernel<.., HALO>: 8 x 16 output patches, the 10 x 18 halo of a 64-channel chunk staged once for nine
    taps and both weight terms; fvit_tune conv_patch) behind the same entry points: vs F.conv2d in fp32 and vs the classic form (another summation order)."""
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(Ci + Co + H + W)
    x = torch.randn(B, Ci, H, W, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (9 * Ci) ** 0.5
    wh = w.to(dt)
    wl = (w - wh.float()).to(dt)
    bias = torch.randn(Co, generator=g).cuda()
    r = torch.randn(B, Co, H, W, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last) if res else None
    cl = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda().reshape(Co, -1)   # noqa: E731
    wk = cl(wh) if terms == 1 else torch.cat([cl(wh), cl(wl)], dim=1).contiguous()
    zeros = torch.zeros(256, dtype=dt, device="cuda")
    outs = {}
    try:
        for form in (0, 1):
            _lib.tune("conv_patch", form)
            _lib.tune("conv_patch_max_waste_pct", 100000)
            assert lib.fvit_conv3x3_patch_form(B, H, W, Ci, Co, 1) == form
            out = torch.full((B, Co, H, W), float("nan"), dtype=dt, device="cuda").contiguous(memory_format=torch.channels_last)
            _lib.check(lib.fvit_conv3x3_nhwc_terms(code, x.data_ptr(), wk.data_ptr(), bias.data_ptr(), r.data_ptr() if res else None, out.data_ptr(),
                                                   B, H, W, Ci, Co, 1, act, terms, zeros.data_ptr(), _stream()), "conv3x3")
            torch.cuda.synchronize()
            outs[form] = out
            if res and form == 1:
                r2 = r.clone()
                _lib.check(lib.fvit_conv3x3_nhwc_terms(code, x.data_ptr(), wk.data_ptr(), bias.data_ptr(), r2.data_ptr(), r2.data_ptr(), B, H, W, Ci, Co, 1,
                                                       act, terms, zeros.data_ptr(), _stream()), "conv3x3 patch in place")
                torch.cuda.synchronize()
                assert torch.equal(r2, out)
        assert lib.fvit_conv3x3_patch_form(B, H, W, Ci, Co, 2) == 0        # stride 2: never the patch form
    finally:
        _lib.tune("conv_patch", 1)
        _lib.tune("conv_patch_max_waste_pct", 10)
    wref = (wh.float() + wl.float()) if terms == 2 else wh.float()
    ref = F.conv2d(x.float(), wref.cuda(), bias, 1, 1)
    ref = [lambda t: t, torch.relu, F.gelu][act](ref)
    if res:
        ref = ref + r.float()
    scale = max(ref.abs().max().item(), 1.0)
    for form in (0, 1):
        assert torch.isfinite(outs[form].float()).all()
        assert (outs[form].float() - ref).abs().max().item() < (5e-3 if dt == torch.float16 else 3e-2) * scale
    assert (outs[1].float() - outs[0].float()).abs().max().item() <= (2.0 ** -10 if dt == torch.float16 else 2.0 ** -7) * scale


@pytest.mark.parametrize("dt,code", [(torch.float16, 1), (torch.bfloat16, 2)])
@pytest.mark.parametrize("B,H,W,act,res", [(86, 28, 28, 2, False), (86, 28, 28, 0, True), (3, 28, 28, 1, True), (2, 30, 30, 0, False), (5, 1, 1, 2, True),
                                           (2, 9, 14, 0, True), (3, 33, 7, 2, False), (1, 14, 14, 0, True), (4, 5, 30, 1, False)])
def test_conv3x3_c128_band(dt, code, B, H, W, act, res):
    """Row-band 128 -> 128 conv (level 1 of FasterViT-0) vs F.conv2d in fp32 and vs the implicit-GEMM kernel; ragged last bands, single
    band images, the widest supported map, in-place residual."""
    from fastervit_amd.conv_runtime import frag_pack_conv128
    lib = _lib.lib()
    assert lib.fvit_conv3x3_c128_band_supported(H, W) == 1 and lib.fvit_conv3x3_c128_band_supported(28, 31) == 0
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + H * 31 + W)
    x = torch.randn(B, 128, H, W, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(128, 128, 3, 3, generator=g) / (9 * 128) ** 0.5).to(dt).cuda()
    bias = torch.randn(128, generator=g).cuda()
    r = torch.randn(B, 128, H, W, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last) if res else None
    wk = w.permute(0, 2, 3, 1).contiguous()
    wf = frag_pack_conv128(wk.reshape(128, 1152))
    zeros = torch.zeros(256, dtype=dt, device="cuda")
    out = torch.full((B, 128, H, W), float("nan"), dtype=dt, device="cuda").contiguous(memory_format=torch.channels_last)
    _lib.check(lib.fvit_conv3x3_c128_band(code, x.data_ptr(), wf.data_ptr(), bias.data_ptr(), r.data_ptr() if res else None, out.data_ptr(), B, H, W,
                                          act, zeros.data_ptr(), _stream()), "conv3x3_c128_band")
    torch.cuda.synchronize()
    ref = F.conv2d(x.float(), w.float(), bias, 1, 1)
    ref = [lambda t: t, torch.relu, F.gelu][act](ref)
    if res:
        ref = ref + r.float()
    assert torch.isfinite(out.float()).all()
    tol = (5e-3 if dt == torch.float16 else 3e-2) * max(ref.abs().max().item(), 1.0)
    assert (out.float() - ref).abs().max().item() < tol
    # the implicit-GEMM kernel accumulates the same products in fp32 (other order): equal to 16-bit rounding
    out2 = torch.empty_like(out)
    _lib.check(lib.fvit_conv3x3_nhwc(code, x.data_ptr(), wk.data_ptr(), bias.data_ptr(), r.data_ptr() if res else None, out2.data_ptr(), B, H, W,
                                     128, 128, 1, act, zeros.data_ptr(), _stream()), "conv3x3")
    torch.cuda.synchronize()
    assert (out.float() - out2.float()).abs().max().item() <= (2e-3 if dt == torch.float16 else 1.6e-2) * max(ref.abs().max().item(), 1.0)
    if res:  # in place into the residual buffer, and bitwise repeatable
        r2 = r.clone()
        _lib.check(lib.fvit_conv3x3_c128_band(code, x.data_ptr(), wf.data_ptr(), bias.data_ptr(), r2.data_ptr(), r2.data_ptr(), B, H, W, act,
                                              zeros.data_ptr(), _stream()), "conv3x3_c128_band in place")
        torch.cuda.synchronize()
        assert torch.equal(r2, out)
    assert lib.fvit_conv3x3_c128_band(code, x.data_ptr(), wf.data_ptr(), bias.data_ptr(), None, out.data_ptr(), B, H, 31, act, zeros.data_ptr(),
                                      _stream()) != 0


@pytest.mark.parametrize("B,H,W,grid", [(3, 56, 56, 512), (2, 24, 50, 8), (9, 9, 17, 16), (1, 40, 40, 1)])
def test_conv3x3_halo_matches_implicit_gemm_kernel(B, H, W, grid):
    """The halo-tiled 64->64 kernel and the implicit-GEMM kernel accumulate the same products in fp32: outputs agree to 16-bit rounding,
    for any persistent grid size (tile ranges per XCD, several tiles per workgroup)."""
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + H)
    x = torch.randn(B, H, W, 64, generator=g).half().cuda()
    r = torch.randn(B, H, W, 64, generator=g).half().cuda()
    wk = (torch.randn(64, 3, 3, 64, generator=g) / 24).half().cuda()
    bias = torch.randn(64, generator=g).cuda()
    zeros = torch.zeros(256, dtype=torch.float16, device="cuda")
    outs = []
    try:
        for halo in (0, 1):
            _lib.tune("conv_halo", halo)
            _lib.tune("conv_halo_grid", grid)
            o = torch.full_like(x, float("nan"))
            _lib.check(lib.fvit_conv3x3_nhwc(1, x.data_ptr(), wk.data_ptr(), bias.data_ptr(), r.data_ptr(), o.data_ptr(), B, H, W, 64, 64, 1, 2,
                                             zeros.data_ptr(), _stream()), "conv3x3")
            torch.cuda.synchronize()
            outs.append(o.float())
    finally:
        _lib.tune("conv_halo", 1)
        _lib.tune("conv_halo_grid", 512)
    assert torch.isfinite(outs[1]).all()
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-3 * max(outs[0].abs().max().item(), 1.0)


@pytest.mark.parametrize("dt,code", [(torch.float16, 1), (torch.bfloat16, 2)])
@pytest.mark.parametrize("in_dt,fmt", [(torch.float32, "nchw"), (torch.float16, "nhwc"), (torch.float32, "nhwc")])
@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (3, 30, 22), (1, 224, 224)])
def test_stem_conv(dt, code, in_dt, fmt, B, H, W):
    """Fused stem: conv3x3 s2 (3 -> 64) + bias + ReLU straight from the caller's image, vs F.conv2d."""
    import ctypes as C
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(H + W)
    x = torch.randn(B, 3, H, W, generator=g).to(in_dt).cuda()
    if fmt == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 3, 3, 3, generator=g) / 27 ** 0.5).cuda()
    bias = torch.randn(64, generator=g).cuda()
    wk = torch.zeros(64, 32, device="cuda")
    wk[:, :27] = w.permute(0, 2, 3, 1).reshape(64, 27)
    wk = wk.to(dt).contiguous()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.full((B, 64, Ho, Wo), float("nan"), dtype=dt, device="cuda").contiguous(memory_format=torch.channels_last)
    view = hat_runtime._map_view(x)
    _lib.check(lib.fvit_stem_conv3x3s2(code, C.byref(view), wk.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, W, _stream()), "stem")
    torch.cuda.synchronize()
    ref = torch.relu(F.conv2d(x.float().to(dt).float(), wk[:, :27].float().view(64, 3, 3, 3).permute(0, 3, 1, 2), bias, 2, 1))
    assert torch.isfinite(out.float()).all()
    assert (out.float() - ref).abs().max().item() < (4e-3 if dt == torch.float16 else 3e-2) * max(ref.abs().max().item(), 1.0)


@pytest.mark.parametrize("dt,code", [(torch.float16, 1), (torch.bfloat16, 2)])
@pytest.mark.parametrize("in_dt,fmt,B,H,W", [(torch.float32, "nchw", 2, 224, 224), (torch.float16, "nhwc", 3, 64, 96), (torch.float32, "nchw", 1, 50, 38),
                                            (torch.float32, "nhwc", 2, 36, 132), (torch.float32, "nchw", 5, 8, 8),
                                            # r06: fp32 channels-last takes the contiguous-run gather (buffer loads, re-ordered contraction): odd sizes = every border case
                                            (torch.float32, "nhwc", 3, 51, 37), (torch.float32, "nhwc", 2, 224, 224), (torch.float32, "nhwc", 1, 7, 5)])
def test_stem_fused(dt, code, in_dt, fmt, B, H, W):
    """Both PatchEmbed convs in one kernel (conv1 3->64 s2 + ReLU kept in LDS, conv2 64->64 s2 + ReLU) vs the two separate kernels
    (same 16-bit rounding of the intermediate) and vs PyTorch fp32."""
    import ctypes as C
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(H * 7 + W)
    x = torch.randn(B, 3, H, W, generator=g).to(in_dt).cuda()
    if fmt == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    w1 = (torch.randn(64, 3, 3, 3, generator=g) / 27 ** 0.5).cuda()
    b1 = torch.randn(64, generator=g).cuda()
    w2 = (torch.randn(64, 64, 3, 3, generator=g) / 576 ** 0.5).to(dt).cuda()
    b2 = torch.randn(64, generator=g).cuda()
    wk1 = torch.zeros(64, 32, device="cuda")
    wk1[:, :27] = w1.permute(0, 2, 3, 1).reshape(64, 27)
    wk1 = wk1.to(dt).contiguous()
    wk2 = w2.permute(0, 2, 3, 1).contiguous()
    H1, W1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    H2, W2 = (H1 - 1) // 2 + 1, (W1 - 1) // 2 + 1
    view = hat_runtime._map_view(x)
    out = torch.full((B, H2, W2, 64), float("nan"), dtype=dt, device="cuda")
    _lib.check(lib.fvit_stem_fused(code, C.byref(view), wk1.data_ptr(), b1.data_ptr(), wk2.data_ptr(), b2.data_ptr(), out.data_ptr(), B, H, W,
                                   _stream()), "stem_fused")
    # the two-kernel path
    mid = torch.empty(B, H1, W1, 64, dtype=dt, device="cuda")
    _lib.check(lib.fvit_stem_conv3x3s2(code, C.byref(view), wk1.data_ptr(), b1.data_ptr(), mid.data_ptr(), B, H, W, _stream()), "stem")
    two = torch.empty(B, H2, W2, 64, dtype=dt, device="cuda")
    zeros = torch.zeros(256, dtype=dt, device="cuda")
    _lib.check(lib.fvit_conv3x3_nhwc(code, mid.data_ptr(), wk2.data_ptr(), b2.data_ptr(), None, two.data_ptr(), B, H1, W1, 64, 64, 2, 1,
                                     zeros.data_ptr(), _stream()), "conv2")
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    scale = max(two.float().abs().max().item(), 1.0)
    assert (out.float() - two.float()).abs().max().item() <= (2e-3 if dt == torch.float16 else 1.6e-2) * scale
    ref1 = torch.relu(F.conv2d(x.float().to(dt).float(), wk1[:, :27].float().view(64, 3, 3, 3).permute(0, 3, 1, 2), b1, 2, 1)).to(dt).float()
    ref = torch.relu(F.conv2d(ref1, w2.float(), b2, 2, 1)).permute(0, 2, 3, 1)
    assert (out.float() - ref).abs().max().item() < (6e-3 if dt == torch.float16 else 4e-2) * max(ref.abs().max().item(), 1.0)


@pytest.mark.parametrize("fmt,dt", [("nchw", torch.float32), ("nhwc", torch.float16), ("nhwc", torch.bfloat16)])
@pytest.mark.parametrize("C,res,ws,cw", [(64, (14, 14), 7, 2), (48, (6, 12), 3, 2), (32, (36, 60), 12, 2), (16, (8, 8), 4, 1)])
def test_token_init(fmt, dt, C, res, ws, cw):
    """fvit_token_init vs the oracle's TokenInitializer (depthwise conv + avg-pool + per-window reorder)."""
    import ctypes as Ct
    from fastervit_amd.models.faster_vit import TokenInitializer
    lib = _lib.lib()
    tok = TokenInitializer(C, list(res), ws, ct_size=cw).cuda()
    g = torch.Generator(device="cpu").manual_seed(C)
    with torch.no_grad():
        tok.pos_embed.weight.copy_(torch.randn(C, 1, 3, 3, generator=g))
        tok.pos_embed.bias.copy_(torch.randn(C, generator=g))
    x = torch.randn(3, C, res[0], res[1], generator=g).to(dt).cuda()
    if fmt == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    got = hat_runtime.token_init(tok, x)
    torch.cuda.synchronize()
    sd = {"t.pos_embed.weight": tok.pos_embed.weight.detach().cpu(), "t.pos_embed.bias": tok.pos_embed.bias.detach().cpu()}
    ref = hr.token_initializer(x.float().cpu(), sd, "t.", res, ws, cw)
    assert got.shape == ref.shape
    assert (got.cpu() - ref).abs().max().item() < 2e-4 * max(ref.abs().max().item(), 1.0)


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("S,nwin,use_tables,use_gamma,C", [(53, 9, True, True, 256), (49, 4, False, False, 256), (16, 13, True, True, 256),
                                                           (64, 3, False, True, 256), (53, 1024, True, True, 256), (16, 256, True, False, 256),
                                                           (49, 85, False, True, 512), (49, 3, False, False, 512), (53, 8, True, True, 512)])
def test_attn_block_fused(opname, dt, code, S, nwin, use_tables, use_gamma, C):
    """gather + LayerNorm + qkv + attention + proj + gamma-residual in one kernel vs PyTorch fp32 on 16-bit-rounded weights
    (C = 256 / 8 heads: stage 2 of FasterViT-0 and its carrier branch; C = 512 / 16 heads: stage 3)."""
    lib = _lib.lib()
    heads, d = C // 32, 32
    assert lib.fvit_attn_block_supported(C, heads, S) == 1 and lib.fvit_attn_block_supported(512, 16, 16) == 0 and \
        lib.fvit_attn_block_supported(784, 16, 53) == 0
    g = torch.Generator(device="cpu").manual_seed(S * 131 + nwin)
    rows = nwin * S
    wins_per_img = 1 if S == 16 else (4 if nwin % 4 == 0 else 1)
    rpi = wins_per_img * S
    nimg = nwin // wins_per_img
    X = (torch.randn(rows, C, generator=g) * 1.3 + 0.2).cuda()
    R = torch.randn(nimg, 7, C, generator=g).cuda()                       # srcB: 7 extra rows per image
    src_idx = add_idx = add = None
    if use_tables:
        si = torch.arange(rpi)
        si[1] = -3                                                        # row 1 of every image comes from R[b, 2]
        si[5 % rpi] = (rpi - 1)                                           # row 5 reads the image's last X row
        ai = torch.full((rpi,), -1, dtype=torch.int64)
        ai[2:] = torch.arange(rpi - 2) % 11
        src_idx, add_idx = si.int().cuda(), ai.int().cuda()
        add = torch.randn(11, C, generator=g).cuda()
    lnw = (torch.rand(C, generator=g) + 0.5).cuda()
    lnb = (torch.randn(C, generator=g) * 0.2).cuda()
    wqkv = (torch.randn(3 * C, C, generator=g) / C ** 0.5).to(dt).cuda()
    bqkv = (torch.randn(3 * C, generator=g) * 0.3).cuda()
    wproj = (torch.randn(C, C, generator=g) / C ** 0.5).to(dt).cuda()
    bproj = (torch.randn(C, generator=g) * 0.3).cuda()
    gamma = (torch.rand(C, generator=g) + 0.5).cuda() if use_gamma else None
    bias = (torch.randn(heads, S, S, generator=g) * 2).cuda()
    spad = lib.fvit_attention_spad(S)
    bp = torch.zeros(heads, spad, spad, device="cuda")
    bp[:, :S, :S] = bias
    bp[:, :, S:] = _lib.FVIT_MASK_BIAS
    wqf = hat_runtime.frag_pack_qkv(wqkv.float(), heads).to(dt).contiguous()
    bqh = bqkv.view(3, heads, 32).permute(1, 0, 2).reshape(heads, 96).contiguous()
    wpf = hat_runtime.frag_pack_fc2(wproj.float()).to(dt).contiguous()
    out = torch.full((rows, C), float("nan"), device="cuda")
    scale = d ** -0.5
    args = (code, X.data_ptr(), rpi, R.data_ptr(), 7, src_idx.data_ptr() if use_tables else None,
            add_idx.data_ptr() if use_tables else None, add.data_ptr() if use_tables else None, lnw.data_ptr(),
            lnb.data_ptr(), ctypes.c_float(1e-5), rpi, wqf.data_ptr(), bqh.data_ptr(), wpf.data_ptr(), bproj.data_ptr(),
            gamma.data_ptr() if use_gamma else None, bp.data_ptr())
    rc = lib.fvit_attn_block_fused(*args, out.data_ptr(), nwin, S, heads, C, ctypes.c_float(scale), _stream())
    _lib.check(rc, "attn_block_fused")
    torch.cuda.synchronize()
    out_win = None
    assert lib.fvit_win_block_supported(C, heads, S) == (1 if S > 48 else 0) and lib.fvit_win_block_supported(784, 16, 49) == 0
    if S > 48:   # the same contract with the N-split work split (fvit_winblk.hip: 8 waves for C = 512, 4 waves for C = 256)
        out_win = torch.full((rows, C), float("nan"), device="cuda")
        _lib.check(lib.fvit_win_block_fused(*args, out_win.data_ptr(), nwin, S, heads, C, ctypes.c_float(scale), _stream()), "win_block_fused")
        torch.cuda.synchronize()
    # reference
    xin = X.view(nimg, rpi, C).clone()
    if use_tables:
        sil, ail = src_idx.long(), add_idx.long()
        xin = torch.where((sil >= 0)[None, :, None], X.view(nimg, rpi, C)[:, sil.clamp(min=0)], R[:, (-sil - 1).clamp(min=0)])
        xin = xin + torch.where((ail >= 0)[:, None], add[ail.clamp(min=0)], torch.zeros_like(add[:1]))[None]
    xin = xin.reshape(nwin, S, C)
    xn = F.layer_norm(xin, (C,), lnw, lnb, 1e-5).to(dt).float()
    qkv = (xn @ wqkv.float().t() + bqkv).view(nwin, S, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0].to(dt).float(), qkv[1].to(dt).float(), qkv[2].to(dt).float()
    att = ((q @ k.transpose(-1, -2)) * scale + bias).softmax(-1)
    o = (att @ v).transpose(1, 2).reshape(nwin, S, C).to(dt).float()
    y = o @ wproj.float().t() + bproj
    ref = (xin + (gamma * y if use_gamma else y)).reshape(rows, C)
    assert torch.isfinite(out).all()
    tol = (4e-3 if dt == torch.float16 else 3e-2) * ref.abs().max().item()
    assert (out - ref).abs().max().item() < tol
    if out_win is not None:
        assert torch.isfinite(out_win).all()
        assert (out_win - ref).abs().max().item() < tol, f"win_block: {(out_win - ref).abs().max().item()} vs {tol}"
    if S > 48 and C == 256:   # r06: the wave-per-(window, head) cut (fvit_attnblk2.hip) behind fvit_tune "ab_variant" = 3, one and two windows per workgroup
        try:
            for nw in (1, 2):
                _lib.tune("ab_variant", 3)
                _lib.tune("ab2_nwin", nw)
                o3 = torch.full((rows, C), float("nan"), device="cuda")
                _lib.check(lib.fvit_attn_block_fused(*args, o3.data_ptr(), nwin, S, heads, C, ctypes.c_float(scale), _stream()), "attn_block_fused (attnblk2)")
                torch.cuda.synchronize()
                assert torch.isfinite(o3).all()
                assert (o3 - ref).abs().max().item() < tol, f"attnblk2 nwin={nw}: {(o3 - ref).abs().max().item()} vs {tol}"
                o4 = torch.full((rows, C), float("nan"), device="cuda")
                _lib.check(lib.fvit_attn_block_fused(*args, o4.data_ptr(), nwin, S, heads, C, ctypes.c_float(scale), _stream()), "attn_block_fused (attnblk2)")
                torch.cuda.synchronize()
                assert torch.equal(o3, o4)   # fixed reduction order: bitwise repeatable
        finally:
            _lib.tune("ab_variant", 0)
            _lib.tune("ab2_nwin", 1)
    if S > 48 and C == 512:   # r03: the 16 heads split over two sibling workgroups that meet in L2 (last-arriver reduction in split order)
        slab = torch.full((nwin * 2 * 64 * C,), float("nan"), device="cuda")
        cnt = torch.zeros(nwin, dtype=torch.int32, device="cuda")
        outs = []
        for _ in range(3):
            o2 = torch.full((rows, C), float("nan"), device="cuda")
            _lib.check(lib.fvit_win_block_fused_split(*args, o2.data_ptr(), nwin, S, heads, C, ctypes.c_float(scale), slab.data_ptr(), cnt.data_ptr(), 2,
                                                      _stream()), "win_block_fused_split")
            torch.cuda.synchronize()
            assert int(cnt.abs().sum().item()) == 0
            outs.append(o2)
        assert torch.isfinite(outs[0]).all()
        assert (outs[0] - ref).abs().max().item() < tol
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        assert (outs[0] - out_win).abs().max().item() < 1e-4 * ref.abs().max().item()   # same numbers up to the order of the two partial sums


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("batch,G,use_add,use_gamma", [(86, 16, True, False), (5, 16, False, True), (33, 9, True, True), (2, 1, True, True)])
def test_ct_block_fused(opname, dt, code, batch, G, use_add, use_gamma):
    """The carrier-token branch of a HAT block in one kernel (gather + pos-embed, LN, qkv, attention over the image's G carrier tokens,
    proj, residual, LN, fc1, GELU, fc2, residual; AR:679-686) vs PyTorch fp32 on 16-bit-rounded weights; G < 16 exercises the masked keys
    and the unwritten padded rows."""
    lib = _lib.lib()
    C, heads, d, hid = 256, 8, 32, 1024
    assert lib.fvit_ct_block_supported(C, heads, G, hid) == 1 and lib.fvit_ct_block_supported(C, heads, 17, hid) == 0 and \
        lib.fvit_ct_block_supported(512, 16, 16, 2048) == 0
    g = torch.Generator(device="cpu").manual_seed(batch * 17 + G)
    rowsA = 4 * 53
    X = (torch.randn(batch * rowsA, C, generator=g) * 1.3 + 0.2).cuda()
    src_idx = torch.randperm(rowsA, generator=g)[:G].int().cuda()
    add = torch.randn(G, C, generator=g).cuda() if use_add else None
    ln1w, ln2w = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    ln1b, ln2b = (torch.randn(C, generator=g) * 0.2).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
    wqkv = (torch.randn(3 * C, C, generator=g) / C ** 0.5).to(dt).cuda()
    bqkv = (torch.randn(3 * C, generator=g) * 0.3).cuda()
    wproj = (torch.randn(C, C, generator=g) / C ** 0.5).to(dt).cuda()
    bproj = (torch.randn(C, generator=g) * 0.3).cuda()
    w1 = (torch.randn(hid, C, generator=g) / C ** 0.5).to(dt).cuda()
    b1 = (torch.randn(hid, generator=g) * 0.3).cuda()
    w2 = (torch.randn(C, hid, generator=g) / hid ** 0.5).to(dt).cuda()
    b2 = (torch.randn(C, generator=g) * 0.3).cuda()
    g1 = (torch.rand(C, generator=g) + 0.5).cuda() if use_gamma else None
    g2 = (torch.rand(C, generator=g) + 0.5).cuda() if use_gamma else None
    bias = (torch.randn(heads, G, G, generator=g) * 2).cuda()
    bp = torch.zeros(heads, 16, 16, device="cuda")
    bp[:, :G, :G] = bias
    bp[:, :, G:] = _lib.FVIT_MASK_BIAS
    wqf = hat_runtime.frag_pack_qkv(wqkv.float(), heads).to(dt).contiguous()
    bqh = bqkv.view(3, heads, 32).permute(1, 0, 2).reshape(heads, 96).contiguous()
    wpf = hat_runtime.frag_pack_fc2(wproj.float()).to(dt).contiguous()
    w1f = hat_runtime.frag_pack_fc1(w1.float()).to(dt).contiguous()
    w2f = hat_runtime.frag_pack_fc2(w2.float()).to(dt).contiguous()
    out = torch.full((batch * G + 3, C), float("nan"), device="cuda")
    scale = d ** -0.5
    p = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
    for variant in (0, 1, 2, 3):   # 3: the 8-wave form (waves split output channels)
        _lib.tune("ct_variant", variant)
        out.fill_(float("nan"))
        rc = lib.fvit_ct_block_fused(code, X.data_ptr(), rowsA, src_idx.data_ptr(), p(add), out.data_ptr(), batch, G, heads, C, hid,
                                     ln1w.data_ptr(), ln1b.data_ptr(), wqf.data_ptr(), bqh.data_ptr(), wpf.data_ptr(), bproj.data_ptr(), p(g1),
                                     bp.data_ptr(), ctypes.c_float(scale), ln2w.data_ptr(), ln2b.data_ptr(), w1f.data_ptr(), b1.data_ptr(),
                                     w2f.data_ptr(), b2.data_ptr(), p(g2), ctypes.c_float(1e-5), _stream())
        _lib.tune("ct_variant", 3)   # the default
        _lib.check(rc, "ct_block_fused")
        torch.cuda.synchronize()
        ct = X.view(batch, rowsA, C)[:, src_idx.long()]
        if use_add:
            ct = ct + add[None]
        xn = F.layer_norm(ct, (C,), ln1w, ln1b, 1e-5).to(dt).float()
        qkv = (xn @ wqkv.float().t() + bqkv).view(batch, G, 3, heads, d).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0].to(dt).float(), qkv[1].to(dt).float(), qkv[2].to(dt).float()
        att = ((q @ k.transpose(-1, -2)) * scale + bias).softmax(-1)
        o = (att @ v).transpose(1, 2).reshape(batch, G, C).to(dt).float()
        y = o @ wproj.float().t() + bproj
        ct = ct + (g1 * y if use_gamma else y)
        xn2 = F.layer_norm(ct, (C,), ln2w, ln2b, 1e-5).to(dt).float()
        h = F.gelu(xn2 @ w1.float().t() + b1).to(dt).float()
        y2 = h @ w2.float().t() + b2
        ref = (ct + (g2 * y2 if use_gamma else y2)).reshape(batch * G, C)
        got = out[:batch * G]
        assert torch.isfinite(got).all() and torch.isnan(out[batch * G:]).all()
        tol = (4e-3 if dt == torch.float16 else 3e-2) * ref.abs().max().item()
        assert (got - ref).abs().max().item() < tol, f"variant {variant}: {(got - ref).abs().max().item()} vs {tol}"


@pytest.mark.parametrize("opname,dt,code", OPS)
@pytest.mark.parametrize("M,C,N,act,gather", [(1360, 256, 768, 0, True), (1360, 256, 1024, 1, False), (4165, 512, 2048, 1, False),
                                               (4165, 512, 1536, 0, False), (77, 256, 48, 0, True), (300, 512, 272, 1, True)])
def test_ln_gemm(opname, dt, code, M, C, N, act, gather):
    """fvit_ln_gemm (LayerNorm in the GEMM's A staging) vs fp32 torch: F.layer_norm(gathered rows + add rows) @ W^T + bias (+ GELU),
    plus the fp32 copy of the gathered rows; rows beyond M / columns beyond N untouched."""
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(M + C + N)
    rows_per_image = next(d for d in (17, 10, 7, 5, 1) if M % d == 0) if gather else 1
    B = M // rows_per_image
    if gather:
        rowsA, rowsB = 23, 9
        srcA = torch.randn(B * rowsA, C, generator=g) * 2 + 0.5
        srcB = torch.randn(B * rowsB, C, generator=g)
        src_idx = torch.randint(-rowsB, rowsA, (rows_per_image,), generator=g).to(torch.int32)
        add = torch.randn(11, C, generator=g)
        add_idx = torch.randint(-1, 11, (rows_per_image,), generator=g).to(torch.int32)
        rows = []
        for b in range(B):
            for pr in range(rows_per_image):
                si = int(src_idx[pr])
                v = srcA[b * rowsA + si] if si >= 0 else srcB[b * rowsB + (-si - 1)]
                if int(add_idx[pr]) >= 0:
                    v = v + add[int(add_idx[pr])]
                rows.append(v)
        v = torch.stack(rows)
    else:
        rowsA = rowsB = 0
        srcA = torch.randn(M, C, generator=g) * 2 + 0.5
        srcB = src_idx = add = add_idx = None
        v = srcA
    ln_w, ln_b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    W = (torch.randn(N, C, generator=g) / C ** 0.5).to(dt)
    bias = torch.randn(N, generator=g)
    Wp = _padded(W, _rup(N, 128), C).cuda()
    ldo = _rup(N, 64)
    out = torch.full((_rup(M, 128), ldo), float("nan"), dtype=dt, device="cuda")
    x_out = torch.full((M, C), float("nan"), device="cuda") if gather else None
    dev = lambda t: None if t is None else t.cuda()
    sA, sB, si_d, ai_d, ad = dev(srcA), dev(srcB), dev(src_idx), dev(add_idx), dev(add)
    lw, lb, bd = ln_w.cuda(), ln_b.cuda(), bias.cuda()
    ptr = lambda t: None if t is None else t.data_ptr()
    assert lib.fvit_ln_gemm_supported(C, N, C, ldo)
    rc = lib.fvit_ln_gemm(code, ptr(sA), rowsA, ptr(sB), rowsB, ptr(si_d), ptr(ai_d), ptr(ad), ptr(x_out), lw.data_ptr(), lb.data_ptr(),
                          ctypes.c_float(1e-5), M, rows_per_image, C, Wp.data_ptr(), C, bd.data_ptr(), out.data_ptr(), ldo, N, act, _stream())
    _lib.check(rc, "ln_gemm")
    torch.cuda.synchronize()
    xn = F.layer_norm(v, (C,), ln_w, ln_b, 1e-5).to(dt).float()       # the kernel rounds the normalised rows to the operand type
    ref = xn @ W.float().t() + bias
    if act:
        ref = F.gelu(ref)
    got = out[:M, :N].float().cpu()
    tol = (4e-3 if dt == torch.float16 else 2e-2) * max(ref.abs().max().item(), 1.0)
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < tol
    assert torch.isnan(out[M:].float()).all() and torch.isnan(out[:M, N:].float()).all()
    if gather:
        assert torch.equal(x_out.cpu(), v)
    # bit-repeatable
    out2 = torch.full_like(out, float("nan"))
    rc = lib.fvit_ln_gemm(code, ptr(sA), rowsA, ptr(sB), rowsB, ptr(si_d), ptr(ai_d), ptr(ad), ptr(x_out), lw.data_ptr(), lb.data_ptr(),
                          ctypes.c_float(1e-5), M, rows_per_image, C, Wp.data_ptr(), C, bd.data_ptr(), out2.data_ptr(), ldo, N, act, _stream())
    _lib.check(rc, "ln_gemm")
    torch.cuda.synchronize()
    assert torch.equal(out[:M, :N], out2[:M, :N])
