"""Backward of the MLP sub-block (fastervit_amd.hat_backward, csrc/fvit_bwd.hip + the GEMM kernels) against torch.autograd on the fp32 form of
the reference's sub-block  y = x + gamma * fc2(GELU(fc1(LayerNorm(x))))  (FV:398-407, 691); SURVEY.md section 8 row f-4.

Tolerance: the GEMM operands are 16-bit (activations, weights and upstream gradients are rounded once each, accumulation fp32), so every gradient
is compared at a few 1e-3 (fp16) / 1e-2 (bf16) of its own largest entry, and dx additionally carries the exact fp32 pass-through of dy."""
import pytest
import torch
import torch.nn.functional as F

from fastervit_amd import _lib, hat_backward

pytestmark = pytest.mark.gpu


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


def test_mlp_block_backward_rejects_cpu_and_bad_shapes():
    x = torch.zeros(4, 256)
    grads = None
    with pytest.raises(RuntimeError):
        hat_backward.mlp_block_backward(x, x, x[0], x[0], torch.zeros(1024, 256), torch.zeros(1024), torch.zeros(256, 1024), torch.zeros(256), None, grads)
    lib = _lib.lib()
    assert lib.fvit_bwd_blocks(65) == 2
    assert lib.fvit_bwd_transpose16(1, None, 0, None, 0, 1, 1, None) != 0
