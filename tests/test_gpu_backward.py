"""Backward of the MLP sub-block (fastervit_amd.hat_backward, csrc/fvit_bwd.hip + the GEMM kernels) against torch.autograd on the fp32 form of
the reference's sub-block  y = x + gamma * fc2(GELU(fc1(LayerNorm(x))))  (FV:398-407, 691); SURVEY.md section 8 row f-4.

Tolerance: the GEMM operands are 16-bit (activations, weights and upstream gradients are rounded once each, accumulation fp32), so every gradient
is compared at a few 1e-3 (fp16) / 1e-2 (bf16) of its own largest entry, and dx additionally carries the exact fp32 pass-through of dy."""
import pytest
import torch
import torch.nn.functional as F

from fastervit_amd import _lib, hat_backward

pytestmark = pytest.mark.gpu

# per-tensor bound of the whole-model gradient tests: max-abs error of a parameter gradient relative to that gradient's largest entry (fp16 operands,
# eleven HAT blocks deep).  Measured r04 (profiles/r04_backward_tests.log): worst tensor 3.0e-3, relative L2 over all 367 tensors 3.6e-4 -- before the
# power-of-two scaling of dy inside the autograd bridge 16 carrier-branch tensors sat at 4-6e-2 (their gradients were fp16 subnormals)
PER_TENSOR = 1e-2


def _reference(x, dy, lnw, lnb, w1, b1, w2, b2, gamma):
    ps = [t.clone().requires_grad_(True) for t in (x, lnw, lnb, w1, b1, w2, b2)] + ([gamma.clone().requires_grad_(True)] if gamma is not None else [])
    xr, lw, lb, W1, B1, W2, B2 = ps[:7]
    y = F.linear(F.gelu(F.linear(F.layer_norm(xr, (xr.shape[1],), lw, lb, 1e-5), W1, B1)), W2, B2)
    out = xr + (ps[7] * y if gamma is not None else y)
    out.backward(dy)
    return [t.grad for t in ps]


@pytest.mark.parametrize("dt,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("M,C,hid,use_gamma", [(300, 256, 1024, True), (4214, 512, 2048, True), (77, 256, 1024, False), (1376, 256, 1024, True)])
def test_mlp_block_backward_vs_autograd(dt, tol, M, C, hid, use_gamma):
    g = torch.Generator(device="cpu").manual_seed(M + C)
    x = (torch.randn(M, C, generator=g) * 1.3 + 0.2).cuda()
    dy = torch.randn(M, C, generator=g).cuda()
    lnw, lnb = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
    w1, b1 = (torch.randn(hid, C, generator=g) / C ** 0.5).cuda(), (torch.randn(hid, generator=g) * 0.3).cuda()
    w2, b2 = (torch.randn(C, hid, generator=g) / hid ** 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    gamma = (torch.rand(C, generator=g) + 0.5).cuda() if use_gamma else None
    ref = _reference(x, dy, lnw, lnb, w1, b1, w2, b2, gamma)
    grads = hat_backward.MlpGrads.zeros(C, hid, x.device, with_gamma=use_gamma)
    dx = hat_backward.mlp_block_backward(x, dy, lnw, lnb, w1, b1, w2, b2, gamma, grads, operand_dtype=dt)
    torch.cuda.synchronize()
    got = [dx, grads.ln_w, grads.ln_b, grads.fc1_w, grads.fc1_b, grads.fc2_w, grads.fc2_b] + ([grads.gamma] if use_gamma else [])
    names = ["dx", "d ln_w", "d ln_b", "dW1", "db1", "dW2", "db2", "dgamma"]
    for name, a, b in zip(names, got, ref):
        assert torch.isfinite(a).all(), name
        err, scale = (a - b).abs().max().item(), b.abs().max().item()
        assert err < tol * scale, f"{name}: max-abs err {err:.3e} vs scale {scale:.3e}"
    # accumulation: a second call adds the same gradients again; bit-reproducible
    dx2 = hat_backward.mlp_block_backward(x, dy, lnw, lnb, w1, b1, w2, b2, gamma, grads, operand_dtype=dt)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx2)
    assert (grads.fc1_b - 2 * ref[4]).abs().max().item() < 2 * tol * ref[4].abs().max().item()
    grads_b = hat_backward.MlpGrads.zeros(C, hid, x.device, with_gamma=use_gamma)
    hat_backward.mlp_block_backward(x, dy, lnw, lnb, w1, b1, w2, b2, gamma, grads_b, operand_dtype=dt)
    hat_backward.mlp_block_backward(x, dy, lnw, lnb, w1, b1, w2, b2, gamma, grads_b, operand_dtype=dt)
    torch.cuda.synchronize()
    assert torch.equal(grads.fc1_w, grads_b.fc1_w) and torch.equal(grads.fc2_w, grads_b.fc2_w) and torch.equal(grads.ln_w, grads_b.ln_w)


def test_backward_primitives_reject_bad_arguments():
    lib = _lib.lib()
    assert lib.fvit_bwd_blocks(65) == 2
    assert lib.fvit_bwd_transpose16(1, None, 0, None, 0, 1, 1, None) != 0
    assert lib.fvit_bwd_window_attention(1, None, 0, None, 0, None, 0, 1.0, None, None, 1, 65, 8, 32, None) != 0


def _attn_reference(x, dy, lnw, lnb, wq, bq, wp, bp, gamma, bias, heads, S):
    leaves = [t.clone().requires_grad_(True) for t in (x, lnw, lnb, wq, bq, wp, bp)]
    gl = gamma.clone().requires_grad_(True) if gamma is not None else None
    bl = bias.clone().requires_grad_(True) if bias is not None else None
    xr, lw, lb, Wq, Bq, Wp, Bp = leaves
    M, C = xr.shape
    n, d = M // S, C // heads
    xn = F.layer_norm(xr, (C,), lw, lb, 1e-5)
    qkv = F.linear(xn, Wq, Bq).view(n, S, 3, heads, d).permute(2, 0, 3, 1, 4)
    att = (qkv[0] @ qkv[1].transpose(-1, -2)) * d ** -0.5
    if bl is not None:
        att = att + bl
    o = (att.softmax(-1) @ qkv[2]).transpose(1, 2).reshape(M, C)
    y = F.linear(o, Wp, Bp)
    (xr + (gl * y if gl is not None else y)).backward(dy)
    return [t.grad for t in leaves], (gl.grad if gl is not None else None), (bl.grad if bl is not None else None)


@pytest.mark.parametrize("dt,tol", [(torch.float16, 5e-3), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize("nwin,S,C,use_gamma,use_bias", [(86, 49, 512, True, True), (7, 50, 256, True, True), (3, 64, 256, False, True), (5, 16, 256, True, False)])
def test_attn_block_backward_vs_autograd(dt, tol, nwin, S, C, use_gamma, use_bias):
    heads = C // 32
    g = torch.Generator(device="cpu").manual_seed(nwin * 100 + S)
    M = nwin * S
    x = (torch.randn(M, C, generator=g) * 1.3 + 0.2).cuda()
    dy = torch.randn(M, C, generator=g).cuda()
    lnw, lnb = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
    wq, bq = (torch.randn(3 * C, C, generator=g) / C ** 0.5).cuda(), (torch.randn(3 * C, generator=g) * 0.3).cuda()
    wp, bp = (torch.randn(C, C, generator=g) / C ** 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    gamma = (torch.rand(C, generator=g) + 0.5).cuda() if use_gamma else None
    bias = torch.randn(heads, S, S, generator=g).cuda() if use_bias else None
    ref, dgam, dbias = _attn_reference(x, dy, lnw, lnb, wq, bq, wp, bp, gamma, bias, heads, S)
    grads = hat_backward.AttnGrads.zeros(C, heads, S, x.device, with_gamma=use_gamma, with_bias=use_bias)
    dx = hat_backward.attn_block_backward(x, dy, lnw, lnb, wq, bq, wp, bp, gamma, bias, heads, S, grads, operand_dtype=dt)
    torch.cuda.synchronize()
    pairs = [("dx", dx, ref[0]), ("d ln_w", grads.ln_w, ref[1]), ("d ln_b", grads.ln_b, ref[2]), ("dWqkv", grads.qkv_w, ref[3]),
             ("dbqkv", grads.qkv_b, ref[4]), ("dWproj", grads.proj_w, ref[5]), ("dbproj", grads.proj_b, ref[6])]
    if use_gamma:
        pairs.append(("dgamma", grads.gamma, dgam))
    if use_bias:
        pairs.append(("dbias", grads.bias, dbias))
    for name, a, b in pairs:
        assert torch.isfinite(a).all(), name
        err, scale = (a - b).abs().max().item(), b.abs().max().item()
        assert err < tol * scale, f"{name}: max-abs err {err:.3e} vs scale {scale:.3e}"
    grads2 = hat_backward.AttnGrads.zeros(C, heads, S, x.device, with_gamma=use_gamma, with_bias=use_bias)
    dx2 = hat_backward.attn_block_backward(x, dy, lnw, lnb, wq, bq, wp, bp, gamma, bias, heads, S, grads2, operand_dtype=dt)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx2) and torch.equal(grads.qkv_w, grads2.qkv_w) and torch.equal(grads.proj_w, grads2.proj_w)


@pytest.mark.parametrize("dt,tol", [(torch.float16, 6e-3), (torch.bfloat16, 5e-2)])
def test_local_hat_block_backward_vs_autograd(dt, tol):
    """One HAT block without carrier tokens (stage 3 of FasterViT-0: 7 x 7 windows, C = 512, 16 heads): x1 = x + g3 attn(LN(x)), y = x1 + g4 mlp(LN(x1));
    dx and all 15 parameter gradients against torch.autograd."""
    nwin, S, C, hid = 12, 49, 512, 2048
    heads, M = C // 32, nwin * S
    g = torch.Generator(device="cpu").manual_seed(11)
    rnd = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).cuda()   # noqa: E731
    x, dy = rnd(M, C, k=1.3), rnd(M, C)
    attn = dict(ln_w=(torch.rand(C, generator=g) + 0.5).cuda(), ln_b=rnd(C, k=0.2), qkv_w=rnd(3 * C, C, k=C ** -0.5), qkv_b=rnd(3 * C, k=0.3),
                proj_w=rnd(C, C, k=C ** -0.5), proj_b=rnd(C, k=0.3), gamma=(torch.rand(C, generator=g) + 0.5).cuda(), bias=rnd(heads, S, S))
    mlp = dict(ln_w=(torch.rand(C, generator=g) + 0.5).cuda(), ln_b=rnd(C, k=0.2), fc1_w=rnd(hid, C, k=C ** -0.5), fc1_b=rnd(hid, k=0.3),
               fc2_w=rnd(C, hid, k=hid ** -0.5), fc2_b=rnd(C, k=0.3), gamma=(torch.rand(C, generator=g) + 0.5).cuda())
    xr = x.clone().requires_grad_(True)
    A = {k: v.clone().requires_grad_(True) for k, v in attn.items()}
    Mm = {k: v.clone().requires_grad_(True) for k, v in mlp.items()}
    xn = F.layer_norm(xr, (C,), A["ln_w"], A["ln_b"], 1e-5)
    qkv = F.linear(xn, A["qkv_w"], A["qkv_b"]).view(nwin, S, 3, heads, 32).permute(2, 0, 3, 1, 4)
    o = (((qkv[0] @ qkv[1].transpose(-1, -2)) * 32 ** -0.5 + A["bias"]).softmax(-1) @ qkv[2]).transpose(1, 2).reshape(M, C)
    x1 = xr + A["gamma"] * F.linear(o, A["proj_w"], A["proj_b"])
    y = x1 + Mm["gamma"] * F.linear(F.gelu(F.linear(F.layer_norm(x1, (C,), Mm["ln_w"], Mm["ln_b"], 1e-5), Mm["fc1_w"], Mm["fc1_b"])), Mm["fc2_w"], Mm["fc2_b"])
    y.backward(dy)
    ag, mg = hat_backward.AttnGrads.zeros(C, heads, S, x.device), hat_backward.MlpGrads.zeros(C, hid, x.device)
    dx = hat_backward.local_block_backward(x, dy, attn, mlp, heads, S, ag, mg, operand_dtype=dt)
    torch.cuda.synchronize()
    pairs = [("dx", dx, xr.grad)] + [(f"attn.{k}", getattr(ag, k), A[k].grad) for k in attn] + [(f"mlp.{k}", getattr(mg, k), Mm[k].grad) for k in mlp]
    for name, a, b in pairs:
        err, scale = (a - b).abs().max().item(), b.abs().max().item()
        assert torch.isfinite(a).all() and err < tol * scale, f"{name}: max-abs err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("sr,dt,tol", [((2, 2), torch.float16, 8e-3), ((2, 2), torch.bfloat16, 6e-2), ((1, 2), torch.float16, 8e-3)])
def test_hier_hat_block_backward_vs_autograd(sr, dt, tol):
    """One HAT block WITH carrier tokens (stage 2 of FasterViT-0: 7 x 7 windows, 2 x 2 carrier tokens per window, C = 256; also a non-square 1 x 2 window
    grid, where ct_window is not the inverse of ct_dewindow): dx, dct and the parameter gradients of all four sub-blocks against torch.autograd on a
    functional restatement that uses the oracle's carrier reshuffles (oracle.hat_reference.ct_dewindow / ct_window)."""
    from oracle import hat_reference as hr
    B, ws, cw, C, hid = 6, 7, 2, 256, 1024
    heads = C // 32
    nW, ncw, nloc = sr[0] * sr[1], cw * cw, ws * ws
    G, S, Bw = ncw * nW, ncw + nloc, B * nW
    g = torch.Generator(device="cpu").manual_seed(sr[0] * 10 + sr[1])
    rnd = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).cuda()   # noqa: E731
    pos = lambda n: (torch.rand(n, generator=g) + 0.5).cuda()   # noqa: E731

    def attn_params(Sa):
        return dict(ln_w=pos(C), ln_b=rnd(C, k=0.2), qkv_w=rnd(3 * C, C, k=C ** -0.5), qkv_b=rnd(3 * C, k=0.3), proj_w=rnd(C, C, k=C ** -0.5), proj_b=rnd(C, k=0.3),
                    gamma=pos(C), bias=rnd(heads, Sa, Sa))

    def mlp_params():
        return dict(ln_w=pos(C), ln_b=rnd(C, k=0.2), fc1_w=rnd(hid, C, k=C ** -0.5), fc1_b=rnd(hid, k=0.3), fc2_w=rnd(C, hid, k=hid ** -0.5), fc2_b=rnd(C, k=0.3),
                    gamma=pos(C))

    P = dict(hat_attn=attn_params(G), hat_mlp=mlp_params(), attn=attn_params(S), mlp=mlp_params())
    x, ct = rnd(Bw, nloc, C, k=1.3), rnd(B, G, C, k=1.3)
    dx_out, dct_out = rnd(Bw, nloc, C), rnd(B, G, C)
    pe_x, pe_ct = rnd(nloc, C, k=0.5), (rnd(G, C, k=0.5) if sr[0] == sr[1] else None)

    # ---- reference: autograd over the functional form (AR:668-707) ----
    L = {k: {n: t.clone().requires_grad_(True) for n, t in d.items()} for k, d in P.items()}
    xr, ctr = x.clone().requires_grad_(True), ct.clone().requires_grad_(True)

    def attn_f(t, p, Sa):
        n = t.shape[0] // Sa
        q = F.linear(F.layer_norm(t, (C,), p["ln_w"], p["ln_b"], 1e-5), p["qkv_w"], p["qkv_b"]).view(n, Sa, 3, heads, 32).permute(2, 0, 3, 1, 4)
        o = (((q[0] @ q[1].transpose(-1, -2)) * 32 ** -0.5 + p["bias"]).softmax(-1) @ q[2]).transpose(1, 2).reshape(n * Sa, C)
        return t + p["gamma"] * F.linear(o, p["proj_w"], p["proj_b"])

    def mlp_f(t, p):
        return t + p["gamma"] * F.linear(F.gelu(F.linear(F.layer_norm(t, (C,), p["ln_w"], p["ln_b"], 1e-5), p["fc1_w"], p["fc1_b"])), p["fc2_w"], p["fc2_b"])

    c0 = hr.ct_dewindow(ctr, cw * sr[0], cw * sr[1], cw)
    if pe_ct is not None:
        c0 = c0 + pe_ct
    c2 = mlp_f(attn_f(c0.reshape(B * G, C), L["hat_attn"], G), L["hat_mlp"]).view(B, G, C)
    cwn = hr.ct_window(c2, cw * sr[0], cw * sr[1], cw).reshape(Bw, -1, C)
    xin = torch.cat((cwn, xr + pe_x), dim=1).reshape(Bw * S, C)
    y2 = mlp_f(attn_f(xin, L["attn"], S), L["mlp"]).view(Bw, S, C)
    ((y2[:, ncw:] * dx_out).sum() + (y2[:, :ncw].reshape(B, G, C) * dct_out).sum()).backward()

    grads = dict(hat_attn=hat_backward.AttnGrads.zeros(C, heads, G, x.device), hat_mlp=hat_backward.MlpGrads.zeros(C, hid, x.device),
                 attn=hat_backward.AttnGrads.zeros(C, heads, S, x.device), mlp=hat_backward.MlpGrads.zeros(C, hid, x.device))
    dx, dct = hat_backward.hier_block_backward(x, ct, dx_out, dct_out, P["hat_attn"], P["hat_mlp"], P["attn"], P["mlp"], heads, ws, cw, sr, pe_x, pe_ct, grads,
                                               operand_dtype=dt)
    torch.cuda.synchronize()
    pairs = [("dx", dx, xr.grad), ("dct", dct, ctr.grad)]
    for k in P:
        pairs += [(f"{k}.{n}", getattr(grads[k], n), L[k][n].grad) for n in P[k]]
    for name, a, b in pairs:
        err, scale = (a - b).abs().max().item(), b.abs().max().item()
        assert torch.isfinite(a).all() and err < tol * scale, f"{name}: max-abs err {err:.3e} vs scale {scale:.3e}"


def test_stage3_backward_of_the_model_vs_oracle_autograd():
    """fastervit_amd.hat_backward.local_stage_backward on the REAL stage 3 of faster_vit_0_224 (five HAT blocks without carrier tokens, synthetic 'init'
    weights): dx and the .grad of every parameter of layer.blocks -- including the cpb_mlp of the relative-position bias and of the 1-D position embedding,
    reached through the folded tables -- against torch.autograd through the CPU oracle's hat_stage (oracle/hat_reference.py) on the same weights."""
    import fastervit_amd
    from oracle import hat_reference as hr
    from tests.synth import synth_state_dict
    torch.manual_seed(0)
    model = fastervit_amd.create_model("faster_vit_0_224").eval()
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234, family="init"))
    layer = model.levels[3]
    g = torch.Generator(device="cpu").manual_seed(3)
    B = 6
    x = torch.randn(B, 512, 7, 7, generator=g)
    dy = torch.randn(B, 512, 7, 7, generator=g)
    # ---- reference: autograd through the oracle on CPU, fp32 ----
    sd = {k: v.detach().clone().float().requires_grad_(v.dtype.is_floating_point) for k, v in layer.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    blk0 = layer.blocks[0]
    out = hr.hat_stage(xr, sd, "", depth=len(layer.blocks), heads=blk0.attn.num_heads, ws=layer.window_size, cw=blk0.cr_window, input_resolution=[7, 7],
                       only_local=True, do_propagation=False, any_res=layer.any_res)
    out.backward(dy)
    ref = {k: v.grad for k, v in sd.items() if k.startswith("blocks.") and v.requires_grad and v.grad is not None}
    # ---- product ----
    layer = layer.cuda()
    for p in layer.parameters():
        p.grad = None
    dx = hat_backward.local_stage_backward(layer, x.cuda(), dy.cuda())
    torch.cuda.synchronize()
    err, scale = (dx.cpu() - xr.grad).abs().max().item(), xr.grad.abs().max().item()
    assert err < 1e-2 * scale, f"dx: {err:.3e} vs {scale:.3e}"
    got = {k: p.grad for k, p in layer.named_parameters() if k.startswith("blocks.")}
    checked = 0
    for k, r in ref.items():
        assert k in got and got[k] is not None, f"no gradient for {k}"
        a = got[k].float().cpu()
        e, sc = (a - r).abs().max().item(), r.abs().max().item()
        assert torch.isfinite(a).all() and e < 1.5e-2 * sc + 1e-7, f"{k}: max-abs err {e:.3e} vs scale {sc:.3e}"
        checked += 1
    assert checked >= 5 * 14   # 14 weight / bias / gamma tensors per block at least, plus the two cpb MLPs


def test_stage2_backward_of_the_model_vs_oracle_autograd():
    """hat_backward.hier_stage_backward on the REAL stage 2 of faster_vit_0_224 (TokenInitializer + six HAT blocks with carrier tokens, 14 x 14 map, 2 x 2
    windows of 7 x 7 with 2 x 2 carrier tokens each): dx and the .grad of every parameter of layer.blocks and layer.global_tokenizer against torch.autograd
    through the CPU oracle's hat_stage on the same synthetic 'init' weights."""
    import fastervit_amd
    from oracle import hat_reference as hr
    from tests.synth import synth_state_dict
    torch.manual_seed(0)
    model = fastervit_amd.create_model("faster_vit_0_224").eval()
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234, family="init"))
    layer = model.levels[2]
    g = torch.Generator(device="cpu").manual_seed(4)
    B = 4
    x = torch.randn(B, 256, 14, 14, generator=g)
    dy = torch.randn(B, 256, 14, 14, generator=g)
    sd = {k: v.detach().clone().float().requires_grad_(v.dtype.is_floating_point) for k, v in layer.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    blk0 = layer.blocks[0]
    out = hr.hat_stage(xr, sd, "", depth=len(layer.blocks), heads=blk0.attn.num_heads, ws=layer.window_size, cw=blk0.cr_window, input_resolution=[14, 14],
                       only_local=False, do_propagation=False, any_res=layer.any_res)
    out.backward(dy)
    ref = {k: v.grad for k, v in sd.items() if (k.startswith("blocks.") or k.startswith("global_tokenizer.")) and v.requires_grad and v.grad is not None}
    layer = layer.cuda()
    for p in layer.parameters():
        p.grad = None
    dx = hat_backward.hier_stage_backward(layer, x.cuda(), dy.cuda())
    torch.cuda.synchronize()
    err, scale = (dx.cpu() - xr.grad).abs().max().item(), xr.grad.abs().max().item()
    assert err < 1.5e-2 * scale, f"dx: {err:.3e} vs {scale:.3e}"
    got = dict(layer.named_parameters())
    checked = 0
    for k, r in ref.items():
        if k not in got:      # the tokenizer's conv is registered under two names (to_global_feature.pos == pos_embed): named_parameters lists it once
            continue
        assert got[k].grad is not None, f"no gradient for {k}"
        a = got[k].grad.float().cpu()
        e, sc = (a - r).abs().max().item(), r.abs().max().item()
        assert torch.isfinite(a).all() and e < 2e-2 * sc + 1e-7, f"{k}: max-abs err {e:.3e} vs scale {sc:.3e}"
        checked += 1
    assert checked >= 6 * 28


def test_whole_model_gradients_through_the_hip_hat_stages_vs_oracle_autograd():
    """model.enable_hat_backward(): faster_vit_0_224 in eval mode, loss.backward() through the conv side (PyTorch modules) AND both HAT stages (HIP forward +
    the kernel-sequence backward as one autograd node each): d loss / d input and the gradient of every parameter of the model against torch.autograd
    through the CPU oracle's model_forward on the same synthetic 'init' weights."""
    import fastervit_amd
    from oracle import model_reference as mr
    from tests.cases import CASES
    from tests.synth import synth_state_dict
    torch.manual_seed(0)
    model = fastervit_amd.create_model("faster_vit_0_224").eval()
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234, family="init"))
    g = torch.Generator(device="cpu").manual_seed(8)
    x = torch.randn(2, 3, 224, 224, generator=g)
    r = torch.randn(2, 1000, generator=g)
    sd = {k: (v.detach().clone().float().requires_grad_(True) if v.dtype.is_floating_point and "running_" not in k and "num_batches" not in k
              else v.detach().clone()) for k, v in model.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    (mr.model_forward(sd, xr, CASES["fvit0_224"]["arch"]) * r).sum().backward()
    model = model.cuda().enable_hat_backward(True)
    for p in model.parameters():
        p.grad = None
    xg = x.cuda().requires_grad_(True)
    logits = model(xg)
    (logits * r.cuda()).sum().backward()
    torch.cuda.synchronize()
    err, scale = (xg.grad.cpu() - xr.grad).abs().max().item(), xr.grad.abs().max().item()
    assert err < 3e-2 * scale, f"d/dx: {err:.3e} vs {scale:.3e}"   # (through the conv side: MIOpen backward kernels; measured 2.2e-2)
    # per tensor: max-abs error within PER_TENSOR of the tensor's largest entry; over ALL parameters together: relative L2 error below 1.5 %
    bad, n, num, den, worst = [], 0, 0.0, 0.0, 0.0
    for k, p in model.named_parameters():
        ref = sd[k].grad if k in sd and isinstance(sd[k], torch.Tensor) and sd[k].requires_grad else None
        if ref is None:
            continue
        assert p.grad is not None, f"no gradient for {k}"
        diff = p.grad.float().cpu() - ref
        e, sc = diff.abs().max().item(), ref.abs().max().item()
        num += diff.double().pow(2).sum().item()
        den += ref.double().pow(2).sum().item()
        n += 1
        worst = max(worst, e / (sc + 1e-30))
        if not (e < PER_TENSOR * sc + 1e-6):
            bad.append((k, e, sc))
    print(f"whole-model gradients: {n} tensors, worst per-tensor max-abs / max {worst:.3e}, relative L2 over all {(num / den) ** 0.5:.3e}")
    assert n > 300 and not bad, f"{len(bad)} of {n} parameter gradients off: {bad[:5]}"
    assert (num / den) ** 0.5 < 2e-3, f"relative L2 error of all parameter gradients {(num / den) ** 0.5:.3e}"


def _fvit0_with_reference_grads(r_scale=1.0, batch=2, seed=8):
    """faster_vit_0_224 ('init' weights) + the gradient of every parameter of sum(logits * r) through the CPU oracle."""
    import fastervit_amd
    from oracle import model_reference as mr
    from tests.cases import CASES
    from tests.synth import synth_state_dict
    torch.manual_seed(0)
    model = fastervit_amd.create_model("faster_vit_0_224").eval()
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=1234, family="init"))
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(batch, 3, 224, 224, generator=g)
    r = torch.randn(batch, 1000, generator=g) * r_scale
    sd = {k: (v.detach().clone().float().requires_grad_(True) if v.dtype.is_floating_point and "running_" not in k and "num_batches" not in k
              else v.detach().clone()) for k, v in model.state_dict().items()}
    (mr.model_forward(sd, x, CASES["fvit0_224"]["arch"]) * r).sum().backward()
    ref = {k: v.grad for k, v in sd.items() if isinstance(v, torch.Tensor) and v.requires_grad and v.grad is not None}
    return model, x, r, ref


def _grad_errors(model, ref):
    num = den = 0.0
    worst = 0.0
    for k, p in model.named_parameters():
        if k not in ref:
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        diff = p.grad.float().cpu() - ref[k]
        worst = max(worst, diff.abs().max().item() / (ref[k].abs().max().item() + 1e-30))
        num += diff.double().pow(2).sum().item()
        den += ref[k].double().pow(2).sum().item()
    return worst, (num / den) ** 0.5


def test_tiny_upstream_gradients_survive_the_fp16_backward():
    """with a mean-reduced loss the gradient entering a HAT stage is ~1e-6 .. 1e-7 -- inside the fp16 subnormals (spacing 6e-8).  The autograd
    bridge scales dy by a power of two per stage call (the backward is linear in dy) and unscales in fp32: the gradients of sum(logits * r), r ~ 1e-6, must
    be as accurate (relative to their own size) as those of r ~ 1."""
    model, x, r, ref = _fvit0_with_reference_grads(r_scale=1e-6)
    model = model.cuda().enable_hat_backward(True)
    (model(x.cuda()) * r.cuda()).sum().backward()
    torch.cuda.synchronize()
    worst, l2 = _grad_errors(model, ref)
    print(f"upstream gradient 1e-6: worst per-tensor {worst:.3e}, relative L2 {l2:.3e}")
    assert l2 < 2e-3 and worst < PER_TENSOR


def test_bf16_operand_mode_differentiates_in_bf16():
    """The backward's operand type follows the layer's forward operand mode (it used to be fp16 whatever the forward ran in)."""
    model, x, r, ref = _fvit0_with_reference_grads()
    model = model.cuda().set_hat_operand_dtype("bf16x2").enable_hat_backward(True)
    assert hat_backward.operand_torch_dtype(model.levels[2]) == torch.bfloat16
    (model(x.cuda()) * r.cuda()).sum().backward()
    torch.cuda.synchronize()
    worst, l2 = _grad_errors(model, ref)
    print(f"bf16 backward: worst per-tensor {worst:.3e}, relative L2 {l2:.3e}")
    assert l2 < 1e-2 and worst < 5e-2   # 8-bit mantissas: ~8 x the fp16 figure (measured 2.7e-3 / 1.5e-2)


def test_gradient_hooks_fire_once_and_ddp_step_runs():
    """the stage backward used to call torch.autograd.backward re-entrantly on leaf parameters (cpb MLPs, tokenizer): their AccumulateGrad
    nodes -- and with them DistributedDataParallel's reducer hooks -- ran from inside the outer backward and again from it.  Now every parameter gradient
    is returned to the outer engine: each post-accumulate hook fires exactly once, and ONE DDP-wrapped step (RCCL backend, world size 1: the single-GPU
    box) runs and yields the gradients of the unwrapped model."""
    import os
    import torch.distributed as dist
    model, x, r, ref = _fvit0_with_reference_grads()
    model = model.cuda().enable_hat_backward(True)
    calls = {}
    hooks = [p.register_post_accumulate_grad_hook(lambda p_, k=k: calls.__setitem__(k, calls.get(k, 0) + 1)) for k, p in model.named_parameters()]
    (model(x.cuda()) * r.cuda()).sum().backward()
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    multi = {k: c for k, c in calls.items() if c != 1}
    assert not multi, f"hooks fired more than once: {list(multi.items())[:5]}"
    assert len(calls) > 300
    plain = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    for p in model.parameters():
        p.grad = None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0])
        (ddp(x.cuda()) * r.cuda()).sum().backward()
        torch.cuda.synchronize()
        for k, p in model.named_parameters():
            if k in plain:
                # same kernels, same order on the HAT side; MIOpen's backward-weight convolutions are not run-to-run deterministic: per-tensor bound
                assert p.grad is not None and (p.grad - plain[k]).abs().max().item() <= 2e-3 * plain[k].abs().max().item() + 1e-9, k
    finally:
        dist.destroy_process_group()


def test_unsupported_geometries_fail_at_forward_time():
    """a geometry the kernel-sequence backward does not cover used to raise from inside loss.backward(), on the autograd engine's thread, after a
    forward that succeeded.  Now: enable_hat_backward() raises for a model with such a stage (windows of more than 64 tokens), a map that does not fit a
    hierarchical stage's window grid raises at FORWARD time, and everything the backward covers -- head_dim 40 / 48 / 49 / 64 / 80, propagation, padded
    maps -- is accepted."""
    import fastervit_amd
    for name in ("faster_vit_1_224", "faster_vit_3_224", "faster_vit_5_224"):
        m = fastervit_amd.create_model(name, depths=[1, 1, 1, 1]).eval()
        assert m.enable_hat_backward(True) is m
    big = fastervit_amd.create_model("faster_vit_4_21k_384", depths=[1, 1, 1, 1], num_heads=[1, 1, 2, 4], dim=16, in_dim=16).eval().cuda()   # 576-token windows
    with pytest.raises(RuntimeError, match="no kernel-sequence backward"):
        big.enable_hat_backward(True)
    m0 = fastervit_amd.create_model("faster_vit_0_224").eval().cuda().enable_hat_backward(True)
    with pytest.raises(RuntimeError, match="does not pad into"):
        m0.levels[2](torch.randn(1, 256, 21, 21, device="cuda", requires_grad=True))     # 3 x 3 windows on a stage built for 2 x 2
    y = m0.levels[3](torch.randn(1, 512, 9, 9, device="cuda", requires_grad=True))        # padded to 14 x 14 inside, cropped back
    assert y.shape[-2:] == (9, 9) and y.grad_fn is not None
    # a supported model differentiates w.r.t. its input even without enable_hat_backward (the stage input carries a graph): no silent zero
    m0.enable_hat_backward(False)
    xg = torch.randn(1, 3, 224, 224, device="cuda", requires_grad=True)
    m0(xg).sum().backward()
    assert xg.grad is not None and xg.grad.abs().max().item() > 0
    # the deploy plan is inference-only: asking it for a gradient raises instead of returning detached logits
    m0.switch_to_deploy(torch.float16)
    with pytest.raises(RuntimeError, match="inference-only"):
        m0(xg)


# ------------------------------------------------------------------------------------------------------------------------------------------------
# r04: every head_dim / channel count of the reference entrypoints, the last block's carrier propagation, padded stages, stochastic depth
# ------------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nwin,S,C,heads", [(5, 53, 320, 8), (3, 49, 784, 16), (4, 16, 160, 2), (6, 53, 96, 4)])
def test_attn_block_backward_padded_head_dims(nwin, S, C, heads):
    """head_dim 40 (FasterViT-1), 49 (FasterViT-4; C = 784 is not a multiple of 64 either), 80 (FasterViT-5, padded to 96) and 24: the kernels run on the
    head_dim padded to 32 / 64 / 96 with zero channels; gradients come back in the module's own (3C, C) / (C, C) layouts."""
    dt, tol = torch.float16, 6e-3
    g = torch.Generator(device="cpu").manual_seed(nwin * 100 + S + C)
    M = nwin * S
    x = (torch.randn(M, C, generator=g) * 1.3 + 0.2).cuda()
    dy = torch.randn(M, C, generator=g).cuda()
    lnw, lnb = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
    wq, bq = (torch.randn(3 * C, C, generator=g) / C ** 0.5).cuda(), (torch.randn(3 * C, generator=g) * 0.3).cuda()
    wp, bp = (torch.randn(C, C, generator=g) / C ** 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    gamma = (torch.rand(C, generator=g) + 0.5).cuda()
    bias = torch.randn(heads, S, S, generator=g).cuda()
    ref, dgam, dbias = _attn_reference(x, dy, lnw, lnb, wq, bq, wp, bp, gamma, bias, heads, S)
    grads = hat_backward.AttnGrads.zeros(C, heads, S, x.device)
    dx = hat_backward.attn_block_backward(x, dy, lnw, lnb, wq, bq, wp, bp, gamma, bias, heads, S, grads, operand_dtype=dt)
    torch.cuda.synchronize()
    pairs = [("dx", dx, ref[0]), ("d ln_w", grads.ln_w, ref[1]), ("d ln_b", grads.ln_b, ref[2]), ("dWqkv", grads.qkv_w, ref[3]), ("dbqkv", grads.qkv_b, ref[4]),
             ("dWproj", grads.proj_w, ref[5]), ("dbproj", grads.proj_b, ref[6]), ("dgamma", grads.gamma, dgam), ("dbias", grads.bias, dbias)]
    for name, a, b in pairs:
        assert torch.isfinite(a).all(), name
        err, scale = (a - b).abs().max().item(), b.abs().max().item()
        assert err < tol * scale, f"{name}: max-abs err {err:.3e} vs scale {scale:.3e}"


def test_mlp_block_backward_channel_count_not_a_multiple_of_64():
    """C = 784 (FasterViT-4 stage 2): as a GEMM K dimension the channel axis is zero-padded to 832; outputs and gradients keep C columns."""
    M, C, hid = 300, 784, 3136
    g = torch.Generator(device="cpu").manual_seed(5)
    x = (torch.randn(M, C, generator=g) * 1.3 + 0.2).cuda()
    dy = torch.randn(M, C, generator=g).cuda()
    lnw, lnb = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
    w1, b1 = (torch.randn(hid, C, generator=g) / C ** 0.5).cuda(), (torch.randn(hid, generator=g) * 0.3).cuda()
    w2, b2 = (torch.randn(C, hid, generator=g) / hid ** 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    gamma = (torch.rand(C, generator=g) + 0.5).cuda()
    ref = _reference(x, dy, lnw, lnb, w1, b1, w2, b2, gamma)
    grads = hat_backward.MlpGrads.zeros(C, hid, x.device)
    dx = hat_backward.mlp_block_backward(x, dy, lnw, lnb, w1, b1, w2, b2, gamma, grads)
    torch.cuda.synchronize()
    got = [dx, grads.ln_w, grads.ln_b, grads.fc1_w, grads.fc1_b, grads.fc2_w, grads.fc2_b, grads.gamma]
    for name, a, b in zip(["dx", "d ln_w", "d ln_b", "dW1", "db1", "dW2", "db2", "dgamma"], got, ref):
        err, scale = (a - b).abs().max().item(), b.abs().max().item()
        assert torch.isfinite(a).all() and err < 5e-3 * scale, f"{name}: max-abs err {err:.3e} vs scale {scale:.3e}"
    y = hat_backward.mlp_block_forward(x, lnw, lnb, w1, b1, w2, b2, gamma)
    yr = x + gamma * F.linear(F.gelu(F.linear(F.layer_norm(x, (C,), lnw, lnb, 1e-5), w1, b1)), w2, b2)
    assert (y - yr).abs().max().item() < 5e-3 * yr.abs().max().item()


def test_drop_path_row_scale_in_the_sub_block_backwards():
    """Stochastic depth (FV:690-691): y = x + m[row group] * gamma * f(x) with m in {0, 1 / keep}: forward and every gradient against autograd on the
    masked form (the sub-block sees dy * m, the skip connection sees dy; dropped groups contribute nothing to the parameter gradients)."""
    nwin, S, C, hid = 9, 49, 256, 1024
    heads, M = C // 32, nwin * S
    g = torch.Generator(device="cpu").manual_seed(21)
    rnd = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).cuda()   # noqa: E731
    x, dy = rnd(M, C, k=1.3), rnd(M, C)
    m = (torch.tensor([1, 0, 1, 1, 0, 1, 0, 1, 1], dtype=torch.float32) / 0.75).cuda()   # per window
    rows = m.repeat_interleave(S)
    lnw, lnb = (torch.rand(C, generator=g) + 0.5).cuda(), rnd(C, k=0.2)
    w1, b1, w2, b2 = rnd(hid, C, k=C ** -0.5), rnd(hid, k=0.3), rnd(C, hid, k=hid ** -0.5), rnd(C, k=0.3)
    gamma = (torch.rand(C, generator=g) + 0.5).cuda()
    leaves = [t.clone().requires_grad_(True) for t in (x, lnw, lnb, w1, b1, w2, b2, gamma)]
    xr, lw, lb, W1, B1, W2, B2, G = leaves
    yr = xr + rows[:, None] * (G * F.linear(F.gelu(F.linear(F.layer_norm(xr, (C,), lw, lb, 1e-5), W1, B1)), W2, B2))
    yr.backward(dy)
    y = hat_backward.mlp_block_forward(x, lnw, lnb, w1, b1, w2, b2, gamma, row_scale=rows)
    assert (y - yr.detach()).abs().max().item() < 5e-3 * yr.abs().max().item()
    assert torch.equal(y.view(nwin, S, C)[1], x.view(nwin, S, C)[1])   # a dropped window passes through untouched
    grads = hat_backward.MlpGrads.zeros(C, hid, x.device)
    dx = hat_backward.mlp_block_backward(x, dy, lnw, lnb, w1, b1, w2, b2, gamma, grads, row_scale=rows)
    torch.cuda.synchronize()
    got = [dx, grads.ln_w, grads.ln_b, grads.fc1_w, grads.fc1_b, grads.fc2_w, grads.fc2_b, grads.gamma]
    for name, a, b in zip(["dx", "d ln_w", "d ln_b", "dW1", "db1", "dW2", "db2", "dgamma"], got, [t.grad for t in leaves]):
        err, scale = (a - b).abs().max().item(), b.abs().max().item()
        assert torch.isfinite(a).all() and err < 5e-3 * scale, f"mlp {name}: max-abs err {err:.3e} vs scale {scale:.3e}"
    assert torch.equal(dx.view(nwin, S, C)[1], dy.view(nwin, S, C)[1])
    # attention sub-block
    wq, bq, wp, bp = rnd(3 * C, C, k=C ** -0.5), rnd(3 * C, k=0.3), rnd(C, C, k=C ** -0.5), rnd(C, k=0.3)
    bias = rnd(heads, S, S)
    L = [t.clone().requires_grad_(True) for t in (x, lnw, lnb, wq, bq, wp, bp, gamma, bias)]
    xr, lw, lb, Wq, Bq, Wp, Bp, G, Bt = L
    qkv = F.linear(F.layer_norm(xr, (C,), lw, lb, 1e-5), Wq, Bq).view(nwin, S, 3, heads, 32).permute(2, 0, 3, 1, 4)
    o = (((qkv[0] @ qkv[1].transpose(-1, -2)) * 32 ** -0.5 + Bt).softmax(-1) @ qkv[2]).transpose(1, 2).reshape(M, C)
    ya = xr + rows[:, None] * (G * F.linear(o, Wp, Bp))
    ya.backward(dy)
    y = hat_backward.attn_block_forward(x, lnw, lnb, wq, bq, wp, bp, gamma, bias, heads, S, row_scale=rows)
    assert (y - ya.detach()).abs().max().item() < 5e-3 * ya.abs().max().item()
    ag = hat_backward.AttnGrads.zeros(C, heads, S, x.device)
    dx = hat_backward.attn_block_backward(x, dy, lnw, lnb, wq, bq, wp, bp, gamma, bias, heads, S, ag, row_scale=rows)
    torch.cuda.synchronize()
    got = [dx, ag.ln_w, ag.ln_b, ag.qkv_w, ag.qkv_b, ag.proj_w, ag.proj_b, ag.gamma, ag.bias]
    for name, a, b in zip(["dx", "d ln_w", "d ln_b", "dWqkv", "dbqkv", "dWproj", "dbproj", "dgamma", "dbias"], got, [t.grad for t in L]):
        err, scale = (a - b).abs().max().item(), b.abs().max().item()
        assert torch.isfinite(a).all() and err < 6e-3 * scale, f"attn {name}: max-abs err {err:.3e} vs scale {scale:.3e}"


GENERAL = [  # (entry, kwargs, input size): reduced-depth builds of the reference entrypoints' geometries
    ("faster_vit_1_224", dict(depths=[1, 1, 2, 2]), 224),                                        # head_dim 40
    ("faster_vit_3_224", dict(depths=[1, 1, 2, 2], dim=64, in_dim=32, num_heads=[1, 2, 4, 8]), 224),   # head_dim 64, last-block propagation
    ("faster_vit_4_224", dict(depths=[1, 1, 2, 1]), 224),                                        # head_dim 49, C = 784 / 1568, propagation
    ("faster_vit_4_any_res", dict(depths=[1, 1, 2, 2], num_heads=[1, 1, 2, 4], dim=16, in_dim=16, resolution=[96, 160], window_size=[7, 7, 3, 3], ct_size=2), (96, 160)),
]


@pytest.mark.parametrize("entry,kwargs,hw", GENERAL)
def test_stage_backward_of_every_geometry_vs_oracle_autograd(entry, kwargs, hw):
    """hier_stage_backward / local_stage_backward on stages 2 and 3 of FasterViT-1 / -3 / -4 geometries and of a padded non-square any-res model (6 x 10 map
    padded to 6 x 12, carrier grid 2 x 4 windows: the ct_window scramble): dx and the gradient of every parameter against torch.autograd through the CPU
    oracle's hat_stage on the same synthetic 'stress' weights."""
    import fastervit_amd
    from oracle import hat_reference as hr
    from tests.synth import synth_state_dict
    torch.manual_seed(0)
    model = fastervit_amd.create_model(entry, **kwargs).eval()
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=77, family="stress"))
    H0, W0 = (hw, hw) if isinstance(hw, int) else hw
    g = torch.Generator(device="cpu").manual_seed(9)
    for li in (2, 3):
        layer = model.levels[li]
        b0 = layer.blocks[0]
        C = b0.attn.qkv.in_features
        H, W = H0 // (4 * 2 ** li), W0 // (4 * 2 ** li)
        x = torch.randn(2, C, H, W, generator=g)
        dy = torch.randn(2, C, H, W, generator=g)
        sd = {k: v.detach().clone().float().requires_grad_(v.dtype.is_floating_point) for k, v in layer.state_dict().items()}
        xr = x.clone().requires_grad_(True)
        out = hr.hat_stage(xr, sd, "", depth=len(layer.blocks), heads=b0.attn.num_heads, ws=layer.window_size, cw=b0.cr_window, input_resolution=[H, W],
                           only_local=not b0.do_sr_hat, do_propagation=bool(b0.do_propagation), any_res=layer.any_res)
        out.backward(dy)
        ref = {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None and (k.startswith("blocks.") or k.startswith("global_tokenizer."))}
        layer = layer.cuda()
        for p in layer.parameters():
            p.grad = None
        fn = hat_backward.hier_stage_backward if b0.do_sr_hat else hat_backward.local_stage_backward
        dx = fn(layer, x.cuda(), dy.cuda())
        torch.cuda.synchronize()
        # the train-mode forward chain (drop probabilities 0) reproduces the oracle's forward
        y = hat_backward.stage_forward_train(layer, x.cuda()).cpu()
        assert (y - out.detach()).abs().max().item() < 1e-2 * out.abs().max().item(), f"{entry} level {li}: forward chain"
        err, scale = (dx.cpu() - xr.grad).abs().max().item(), xr.grad.abs().max().item()
        assert err < 2e-2 * scale, f"{entry} level {li} dx: {err:.3e} vs {scale:.3e}"
        got = dict(layer.named_parameters())
        checked, worst = 0, 0.0
        for k, r in ref.items():
            if k not in got:
                continue
            assert got[k].grad is not None, f"{entry} level {li}: no gradient for {k}"
            a = got[k].grad.float().cpu()
            e, sc = (a - r).abs().max().item(), r.abs().max().item()
            worst = max(worst, e / (sc + 1e-30))
            assert torch.isfinite(a).all() and e < 3e-2 * sc + 1e-7, f"{entry} level {li} {k}: max-abs err {e:.3e} vs scale {sc:.3e}"
            checked += 1
        print(f"{entry} level {li}: {checked} parameter gradients, worst {worst:.3e}; dx {err / scale:.3e}")
        assert checked >= 14 * len(layer.blocks)
        model.levels[li] = layer.cpu()


def test_train_mode_runs_with_stochastic_depth_and_matches_eval_when_nothing_is_dropped():
    """model.train(): conv side = PyTorch modules (BatchNorm in train mode, DropPath), HAT stages = the unit-kernel chain with per-window / per-image
    DropPath draws as one autograd node each.  With every drop probability 0 the HAT stages reproduce their eval output; with drop_path_rate > 0 a step
    runs, every parameter receives a finite gradient, and the draws differ between calls."""
    import fastervit_amd
    from fastervit_amd import hat_backward as hb
    torch.manual_seed(0)
    model = fastervit_amd.create_model("faster_vit_0_224", drop_path_rate=0.3).cuda()
    probs = sorted({round(m.drop_prob, 4) for m in model.modules() if hasattr(m, "drop_prob")})
    assert probs and max(probs) > 0.25 and len(probs) > 5      # the linear schedule of FV:918 reached the blocks
    x = torch.randn(4, 3, 224, 224, device="cuda")
    lvl = model.levels[2]
    xs = torch.randn(4, 256, 14, 14, device="cuda")
    model.train()
    masks = hb.drop_path_masks(lvl, 4, 4, xs.device)
    assert masks[-1]["attn"] is not None and tuple(masks[-1]["attn"].shape) == (16,) and tuple(masks[-1]["hat_attn"].shape) == (4,)
    vals = torch.cat([m for d in masks for m in d.values() if m is not None]).unique()
    assert (vals == 0).any() and all(v == 0 or v > 1 for v in vals.tolist())
    y1 = lvl(xs)
    y2 = lvl(xs)
    assert y1.grad_fn is not None and not torch.equal(y1, y2)        # different draws
    logits = model(x)
    logits.square().mean().backward()
    missing = [k for k, p in model.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    assert not missing, missing[:5]
    # nothing dropped: the train-mode chain equals the eval path of the same stage (up to the 16-bit rounding of different kernels)
    for m in model.modules():
        if hasattr(m, "drop_prob"):
            m.drop_prob = 0.0
    with torch.no_grad():
        yt = hb.stage_forward_with_grad(lvl, xs)
        model.eval()
        ye = hb.stage_forward_with_grad(lvl, xs)
    assert (yt - ye).abs().max().item() < 1e-2 * ye.abs().max().item()


def test_a_few_training_steps_reduce_the_loss():
    """End to end: faster_vit_0_224 in train mode (BatchNorm batch statistics, stochastic depth on the conv side and inside the HIP stages), AdamW on ALL
    parameters, a fixed batch of 16 images with random labels: the loss falls step over step (what train.py:820-951 does per iteration, minus its data /
    scheduler / EMA plumbing, which is out of scope)."""
    import fastervit_amd
    torch.manual_seed(0)
    model = fastervit_amd.create_model("faster_vit_0_224", drop_path_rate=0.1, num_classes=10).cuda().train()
    opt = torch.optim.AdamW(model.parameters(), lr=2e-4, weight_decay=0.0)
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(16, 3, 224, 224, generator=g).cuda()
    y = torch.randint(0, 10, (16,), generator=g).cuda()
    losses = []
    for _ in range(8):
        opt.zero_grad(set_to_none=True)
        loss = F.cross_entropy(model(x), y)
        loss.backward()
        assert all(p.grad is None or torch.isfinite(p.grad).all() for p in model.parameters())
        opt.step()
        losses.append(loss.item())
    print("training losses:", [round(v, 4) for v in losses])
    assert losses[-1] < 0.7 * losses[0], losses
    model.eval()
    with torch.no_grad():
        assert torch.isfinite(model(x)).all()   # the updated weights re-pack for the inference kernels


def test_dropout_masks_in_the_sub_blocks_vs_autograd():
    """Dropout inside the blocks in train mode (r05; ``drop_rate`` of the entrypoints): ``Mlp.drop`` after GELU and after fc2 (FV:404-406) and
    ``WindowAttention.proj_drop`` (FV:567) enter the sub-block forward / backward as explicit masks (``hat_backward._RS``): forward and every gradient vs
    torch.autograd on the fp32 sub-block with the SAME masks, combined with a DropPath row factor."""
    dt, tol = torch.float16, 5e-3
    g = torch.Generator(device="cpu").manual_seed(11)
    M, C, hid, keep = 300, 256, 1024, 0.8
    x = (torch.randn(M, C, generator=g) * 1.3 + 0.2).cuda()
    dy = torch.randn(M, C, generator=g).cuda()
    lnw, lnb = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
    w1, b1 = (torch.randn(hid, C, generator=g) / C ** 0.5).cuda(), (torch.randn(hid, generator=g) * 0.3).cuda()
    w2, b2 = (torch.randn(C, hid, generator=g) / hid ** 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    gamma = (torch.rand(C, generator=g) + 0.5).cuda()
    rows = (torch.bernoulli(torch.full((M,), 0.7), generator=g) / 0.7).cuda()
    m_hid = (torch.bernoulli(torch.full((M, hid), keep), generator=g) / keep).to(dt).cuda()
    m_out = (torch.bernoulli(torch.full((M, C), keep), generator=g) / keep).cuda()
    rs = hat_backward._RS(rows, m_out, m_hid)
    # reference
    ps = [t.clone().requires_grad_(True) for t in (x, lnw, lnb, w1, b1, w2, b2, gamma)]
    xr, lw, lb, W1, B1, W2, B2, G = ps
    h = F.gelu(F.linear(F.layer_norm(xr, (C,), lw, lb, 1e-5), W1, B1)) * m_hid.float()
    out = xr + rows.view(M, 1) * (G * (F.linear(h, W2, B2) * m_out))
    out.backward(dy)
    ref = [t.grad for t in ps]
    y = hat_backward.mlp_block_forward(x, lnw, lnb, w1, b1, w2, b2, gamma, operand_dtype=dt, row_scale=rs)
    assert (y - out.detach()).abs().max().item() < tol * out.abs().max().item()
    grads = hat_backward.MlpGrads.zeros(C, hid, x.device, with_gamma=True)
    dx = hat_backward.mlp_block_backward(x, dy, lnw, lnb, w1, b1, w2, b2, gamma, grads, operand_dtype=dt, row_scale=rs)
    torch.cuda.synchronize()
    got = [dx, grads.ln_w, grads.ln_b, grads.fc1_w, grads.fc1_b, grads.fc2_w, grads.fc2_b, grads.gamma]
    for name, a, b in zip(["dx", "d ln_w", "d ln_b", "dW1", "db1", "dW2", "db2", "dgamma"], got, ref):
        err, scale = (a - b).abs().max().item(), b.abs().max().item()
        assert torch.isfinite(a).all() and err < tol * scale, f"mlp {name}: {err:.3e} vs {scale:.3e}"
    # attention sub-block with proj_drop (an output mask) and DropPath
    nwin, S, heads = 6, 49, 8
    M2 = nwin * S
    x2 = (torch.randn(M2, C, generator=g) * 1.2).cuda()
    dy2 = torch.randn(M2, C, generator=g).cuda()
    wq, bq = (torch.randn(3 * C, C, generator=g) / C ** 0.5).cuda(), (torch.randn(3 * C, generator=g) * 0.2).cuda()
    wp, bp = (torch.randn(C, C, generator=g) / C ** 0.5).cuda(), (torch.randn(C, generator=g) * 0.2).cuda()
    bias = (torch.rand(heads, S, S, generator=g) * 4).cuda()
    rows2 = (torch.bernoulli(torch.full((nwin,), 0.7), generator=g) / 0.7).cuda().repeat_interleave(S)
    m2 = (torch.bernoulli(torch.full((M2, C), keep), generator=g) / keep).cuda()
    spad = _lib.lib().fvit_attention_spad(S)
    mp = torch.zeros(nwin * heads, S, spad)
    mp[:, :, :S] = torch.bernoulli(torch.full((nwin * heads, S, S), keep), generator=g) / keep      # attn_drop: Dropout on the softmax probabilities
    mp = mp.to(dt).cuda()
    rs2 = hat_backward._RS(rows2, m2, None, mp)
    leaves = [t.clone().requires_grad_(True) for t in (x2, lnw, lnb, wq, bq, wp, bp, gamma)]
    xr, lw, lb, Wq, Bq, Wp, Bp, G = leaves
    d = C // heads
    qkv = F.linear(F.layer_norm(xr, (C,), lw, lb, 1e-5), Wq, Bq).view(nwin, S, 3, heads, d).permute(2, 0, 3, 1, 4)
    att = ((qkv[0] @ qkv[1].transpose(-1, -2)) * d ** -0.5 + bias).softmax(-1) * mp[:, :, :S].float().view(nwin, heads, S, S)
    o = (att @ qkv[2]).transpose(1, 2).reshape(M2, C)
    out2 = xr + rows2.view(M2, 1) * (G * (F.linear(o, Wp, Bp) * m2))
    out2.backward(dy2)
    y2 = hat_backward.attn_block_forward(x2, lnw, lnb, wq, bq, wp, bp, gamma, bias, heads, S, operand_dtype=dt, row_scale=rs2)
    assert (y2 - out2.detach()).abs().max().item() < tol * out2.abs().max().item()
    ag = hat_backward.AttnGrads.zeros(C, heads, S, x2.device, with_gamma=True)
    dx2 = hat_backward.attn_block_backward(x2, dy2, lnw, lnb, wq, bq, wp, bp, gamma, bias, heads, S, ag, operand_dtype=dt, row_scale=rs2)
    torch.cuda.synchronize()
    for name, a, b in zip(["dx", "d ln_w", "d ln_b", "dWqkv", "dbqkv", "dWproj", "dbproj", "dgamma"],
                          [dx2, ag.ln_w, ag.ln_b, ag.qkv_w, ag.qkv_b, ag.proj_w, ag.proj_b, ag.gamma], [t.grad for t in leaves]):
        err, scale = (a - b).abs().max().item(), b.abs().max().item()
        assert torch.isfinite(a).all() and err < 1.5 * tol * scale, f"attn {name}: {err:.3e} vs {scale:.3e}"


def test_training_with_drop_rate_and_attn_drop_rate_runs():
    """``drop_rate`` > 0 (Mlp.drop + proj_drop) and ``attn_drop_rate`` > 0 (Dropout on the softmax probabilities, inside the attention kernels) train end to
    end: finite gradients for every parameter, different draws call to call, the loss falls on a fixed batch."""
    import fastervit_amd
    torch.manual_seed(0)
    model = fastervit_amd.create_model("faster_vit_0_224", drop_path_rate=0.1, drop_rate=0.1, attn_drop_rate=0.1, num_classes=10).cuda().train()
    opt = torch.optim.AdamW(model.parameters(), lr=2e-4, weight_decay=0.0)
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(8, 3, 224, 224, generator=g).cuda()
    y = torch.randint(0, 10, (8,), generator=g).cuda()
    with torch.no_grad():
        a, b = model(x), model(x)
    assert not torch.equal(a, b)   # Dropout draws differ call to call
    losses = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        loss = F.cross_entropy(model(x), y)
        loss.backward()
        assert all(p.grad is None or torch.isfinite(p.grad).all() for p in model.parameters())
        opt.step()
        losses.append(loss.item())
    print("training losses with drop_rate 0.1, attn_drop_rate 0.1:", [round(v, 4) for v in losses])
    assert losses[-1] < 0.85 * losses[0], losses
    model.eval()
    with torch.no_grad():
        a, b = model(x), model(x)
    assert (a - b).abs().max().item() < 1e-3 * a.abs().max().item()   # eval mode: no Dropout (MIOpen's fp32 convs are not bitwise repeatable call to call)
