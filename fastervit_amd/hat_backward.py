"""Backward (and train-mode forward) of the HAT stages on the MI355X kernels.

    y = x + gamma * fc2(GELU(fc1(LayerNorm(x))))            Mlp.forward FV:398-407 inside HAT.forward FV:691 / AR:697        mlp_block_backward
    y = x + gamma * proj(attention(qkv(LayerNorm(x))))      WindowAttention.forward FV:557-568 inside HAT.forward FV:690      attn_block_backward
    a whole HAT block, without (stage 3) and WITH carrier tokens (stage 2: FV:662-701 / AR:668-707, incl. the last block's
    carrier propagation FV:697-700)                                                                       local_block_backward / hier_block_backward
    a whole stage (zero padding, TokenInitializer, partition, depth x block, reverse, crop: FV:832-841 / AR:848-869)   local_stage_backward / hier_stage_backward
    the stage as ONE autograd node, eval or TRAIN mode (stochastic depth, FV:630, 652, 690-691)                      HatStageFunction / stage_forward_with_grad

The reference differentiates this with autograd (train.py:820-951, the model wrapped in DDP at train.py:542-551).  Here the backward of ONE
sub-block is a fixed kernel sequence behind the C ABI, checked against torch.autograd (tests/test_gpu_backward.py):

    recompute   xn = LN(x)               fvit_gather_layernorm        a = xn W1^T + b1         fvit_gemm_bias_act
                h  = GELU(a)             fvit_bwd_gelu                z = h W2^T + b2          fvit_gemm_bias_act
    gamma / b2  dz = gamma * dy, column sums of dy * z and dz         fvit_bwd_scale_cols + fvit_bwd_colsum_finish
    fc2         dh  = dz W2              fvit_gemm_bias_act (weight operand = W2^T)
                dW2 += dz^T h            fvit_bwd_transpose16 x 2 + fvit_gemm_residual (fp32 accumulation INTO the gradient buffer)
    GELU / b1   da = dh * GELU'(a), column sums of da                 fvit_bwd_gelu + fvit_bwd_colsum_finish
    fc1         dW1 += da^T xn           fvit_bwd_transpose16 x 2 + fvit_gemm_residual
                dxn = da W1              fvit_gemm_residual into a zeroed fp32 buffer (weight operand = W1^T)
    LayerNorm   dx = dy + LN'(dxn), d ln_w, d ln_b                    fvit_bwd_layernorm + fvit_bwd_colsum_finish

Operands of the GEMMs are 16-bit (fp16 / bf16) with fp32 accumulation, like the forward path; activations are recomputed, nothing is saved by
the forward.  Parameter gradients ACCUMULATE into the buffers of ``MlpGrads`` / ``AttnGrads`` (zero them for a fresh step); no atomics anywhere
(column sums are per-64-row partials added in block order), so gradients are bit-reproducible.  Geometry: head_dim <= 96 (the attention kernels
run on the head_dim padded to 32 / 64 / 96 with zero channels, like the forward path), any C that is a multiple of 16 (a K dimension is zero-padded
to a multiple of 64), windows / carrier grids of at most 64 tokens -- every reference entrypoint at 224 x 224.  What this is NOT: backward KERNELS
for the conv side (PyTorch autograd differentiates those modules) or the optimizer / data loop of train.py; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from .hat_runtime import FVIT_BF16, FVIT_F16

_CODE = {torch.float16: FVIT_F16, torch.bfloat16: FVIT_BF16}


def _rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class MlpGrads:
    """fp32 gradient buffers of one MLP sub-block; ``mlp_block_backward`` adds into them."""
    fc1_w: torch.Tensor
    fc1_b: torch.Tensor
    fc2_w: torch.Tensor
    fc2_b: torch.Tensor
    ln_w: torch.Tensor
    ln_b: torch.Tensor
    gamma: Optional[torch.Tensor]

    @staticmethod
    def zeros(C_: int, hidden: int, device, with_gamma: bool = True) -> "MlpGrads":
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)   # noqa: E731
        return MlpGrads(z(hidden, C_), z(hidden), z(C_, hidden), z(C_), z(C_), z(C_), z(C_) if with_gamma else None)


def _pad_rows(w: torch.Tensor, dt, cols: Optional[int] = None) -> torch.Tensor:
    """16-bit GEMM weight operand: rows zero-padded to a multiple of 128 (fvit_gemm_bias_act's Wt contract), columns (the K dimension) to ``cols``."""
    cols = w.shape[1] if cols is None else cols
    out = torch.zeros(_rup(w.shape[0], 128), cols, dtype=dt, device=w.device)
    out[:w.shape[0], :w.shape[1]] = w.to(dt)
    return out


def _dpad(d: int) -> int:
    """Padded head_dim of the attention kernels (hat_runtime._prepare): 32 / 64 / 96."""
    if d > 96:
        raise RuntimeError(f"head_dim {d} > 96 has no attention kernel instance")
    return 32 if d <= 32 else (64 if d <= 64 else 96)


def _pack_qkv(wq: torch.Tensor, bq: torch.Tensor, heads: int, d: int, dp: int):
    """qkv.weight (3C, C) / bias (3C) -> rows [q|k|v][head][dpad] (zero rows / entries on the padded head channels)."""
    if d == dp:
        return wq, bq
    C_ = wq.shape[1]
    w = wq.new_zeros(3, heads, dp, C_)
    w[:, :, :d] = wq.view(3, heads, d, C_)
    b = bq.new_zeros(3, heads, dp)
    b[:, :, :d] = bq.view(3, heads, d)
    return w.view(3 * heads * dp, C_), b.view(-1)


def _unpack_qkv(g: torch.Tensor, heads: int, d: int, dp: int) -> torch.Tensor:
    """Inverse of ``_pack_qkv`` on dim 0 (gradient rows / bias entries of the real head channels)."""
    if d == dp:
        return g
    return g.view(3, heads, dp, *g.shape[1:])[:, :, :d].reshape(3 * heads * d, *g.shape[1:])


def _pack_proj(wp: torch.Tensor, heads: int, d: int, dp: int, K: int) -> torch.Tensor:
    """proj.weight (C, C) -> (C, K) with columns [head][dpad] (+ zero columns up to K, a multiple of 64)."""
    C_ = wp.shape[0]
    if d == dp and K == C_:
        return wp
    w = wp.new_zeros(C_, K)
    w[:, :heads * dp].view(C_, heads, dp)[:, :, :d] = wp.view(C_, heads, d)
    return w


def _unpack_proj(g: torch.Tensor, heads: int, d: int, dp: int) -> torch.Tensor:
    C_ = g.shape[0]
    if d == dp and g.shape[1] == C_:
        return g
    return g[:, :heads * dp].reshape(C_, heads, dp)[:, :, :d].reshape(C_, heads * d)


class _RS:
    """What scales a sub-block's residual update in TRAIN mode: ``rows`` fp32 [M] = DropPath (one factor per window / image, repeated per row), ``out`` fp32
    [M][C] = the Dropout mask of the sub-block's OUTPUT (0 or 1 / keep: ``Mlp.drop`` after fc2, FV:406; ``WindowAttention.proj_drop``, FV:567), ``hid``
    operand-dtype [M][hidden] = the Dropout mask of GELU(fc1) (``Mlp.drop`` after the activation, FV:404).  y = x + rows * out * gamma * f(x): the output mask is
    an elementwise form of the DropPath factor, so forward and backward treat the two alike; ``hid`` multiplies the hidden activation (and dh in the backward);
    ``attn`` operand-dtype [windows * heads][S][Spad] = the Dropout mask of the softmax probabilities (``WindowAttention.attn_drop``, FV:564), applied inside the
    attention kernels (fvit_window_attention_drop / fvit_bwd_window_attention_drop)."""
    __slots__ = ("rows", "out", "hid", "attn")

    def __init__(self, rows=None, out=None, hid=None, attn=None):
        self.rows, self.out, self.hid, self.attn = rows, out, hid, attn


def _rows(scale, group: int, out=None, hid=None, attn=None):
    """Per-group DropPath factors (one per window / per image) -> one per row; with Dropout masks an ``_RS``."""
    rows = None if scale is None else scale.repeat_interleave(group)
    if out is None and hid is None and attn is None:
        return rows
    return _RS(rows, out, hid, attn)


def _rs(mk: Optional[dict], key: str, group: int):
    mk = mk or {}
    return _rows(mk.get(key), group, mk.get(key + "_out"), mk.get(key + "_hid"), mk.get(key + "_p"))


def _attn_mask(rs):
    return rs.attn if isinstance(rs, _RS) else None


def _out_scale(rs, M: int) -> Optional[torch.Tensor]:
    """fp32 factor of the sub-block's output, broadcastable to [M][C] (None: 1)."""
    if rs is None:
        return None
    if isinstance(rs, torch.Tensor):
        return rs.float().view(M, 1)
    sc = None if rs.rows is None else rs.rows.float().view(M, 1)
    if rs.out is not None:
        sc = rs.out.float() if sc is None else sc * rs.out.float()
    return sc


def _hid_mask(rs):
    return rs.hid if isinstance(rs, _RS) else None


def mlp_block_backward(x: torch.Tensor, dy: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor, fc1_w: torch.Tensor, fc1_b: torch.Tensor,
                       fc2_w: torch.Tensor, fc2_b: torch.Tensor, gamma: Optional[torch.Tensor], grads: MlpGrads, eps: float = 1e-5,
                       operand_dtype=torch.float16, row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Returns dx (fp32 [M][C]) and adds the parameter gradients of the sub-block into ``grads``.  x, dy: fp32 [M][C] on a HIP device.
    ``row_scale`` (fp32 [M], optional): stochastic depth -- y = x + row_scale[m] * gamma * (...) (timm DropPath: 0 or 1 / keep_prob per sample,
    FV:691 ``self.drop_path(self.gamma4 * self.mlp(...))``): the sub-block sees dy * row_scale, the skip connection sees dy."""
    if not x.is_cuda:
        raise RuntimeError("mlp_block_backward runs only on a HIP device (libfvit_hip.so kernels); there is no CPU fallback")
    if operand_dtype not in _CODE:
        raise ValueError("operand_dtype must be torch.float16 or torch.bfloat16")
    M, C_ = x.shape
    hid = fc1_w.shape[0]
    if C_ % 16 or hid % 64:
        raise RuntimeError(f"mlp_block_backward: C = {C_} must be a multiple of 16 and hidden = {hid} a multiple of 64")
    for t, name in ((x, "x"), (dy, "dy")):
        if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != (M, C_) or t.device != x.device:
            raise RuntimeError(f"mlp_block_backward: {name} must be a contiguous fp32 [M][C] tensor on {x.device}")
    dev, dt, code = x.device, operand_dtype, _CODE[operand_dtype]
    lib = _lib.lib()
    Mp, Mk, Ck = _rup(M, 128), _rup(M, 64), _rup(C_, 64)   # Ck: C as a GEMM K dimension (zero columns beyond C: FasterViT-4's 784 / 1568)
    f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()   # noqa: E731
    ln_w, ln_b, b1, b2 = f32(ln_w), f32(ln_b), f32(fc1_b), f32(fc2_b)
    g = f32(gamma) if gamma is not None else None
    w1, w2 = f32(fc1_w), f32(fc2_w)
    W1, W2 = _pad_rows(w1, dt, Ck), _pad_rows(w2, dt)                  # [pad(hid)][Ck], [pad(C)][hid]
    W1T, W2T = _pad_rows(w1.t().contiguous(), dt), _pad_rows(w2.t().contiguous(), dt, Ck)   # [pad(C)][hid], [pad(hid)][Ck]
    e16 = lambda r, c: torch.zeros(r, c, dtype=dt, device=dev)   # noqa: E731
    xn, a, h, z = e16(Mp, Ck), e16(Mp, hid), e16(Mp, hid), e16(Mp, C_)
    dz, dh, da = e16(Mp, Ck), e16(Mp, hid), e16(Mp, hid)
    dzT, xnT = e16(_rup(C_, 128), Mk), e16(_rup(C_, 128), Mk)
    hT, daT = e16(_rup(hid, 128), Mk), e16(_rup(hid, 128), Mk)
    blocks = lib.fvit_bwd_blocks(M)
    part = torch.empty(blocks * 2 * max(C_, hid), dtype=torch.float32, device=dev)
    dxn = torch.zeros(M, C_, dtype=torch.float32, device=dev)
    dx = torch.empty(M, C_, dtype=torch.float32, device=dev)
    stats = torch.empty(M, 2, dtype=torch.float32, device=dev)
    osc, hmask = _out_scale(row_scale, M), _hid_mask(row_scale)
    dyi = dy if osc is None else (dy * osc).contiguous()   # what the sub-block sees
    p = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream
        ck = _lib.check
        # ---- recompute the forward intermediates ----
        ck(lib.fvit_gather_layernorm(code, x.data_ptr(), M, None, 0, None, None, None, None, xn.data_ptr(), Ck, ln_w.data_ptr(), ln_b.data_ptr(),
                                     C.c_float(eps), M, M, C_, st), "layernorm")
        ck(lib.fvit_gemm_bias_act(code, xn.data_ptr(), Ck, W1.data_ptr(), Ck, b1.data_ptr(), a.data_ptr(), hid, M, hid, Ck, 0, st), "fc1")
        ck(lib.fvit_bwd_gelu(code, a.data_ptr(), hid, None, 0, h.data_ptr(), hid, None, M, hid, st), "gelu")
        if hmask is not None:
            h[:M] *= hmask          # Dropout on GELU(fc1) (FV:404): the same mask as in the forward
        ck(lib.fvit_gemm_bias_act(code, h.data_ptr(), hid, W2.data_ptr(), hid, b2.data_ptr(), z.data_ptr(), C_, M, C_, hid, 0, st), "fc2")
        # ---- gamma, fc2 bias, dz = gamma * dy ----
        ck(lib.fvit_bwd_scale_cols(code, dyi.data_ptr(), z.data_ptr(), C_, p(g), dz.data_ptr(), Ck, part.data_ptr(), M, C_, st), "scale_cols")
        if grads.gamma is not None:
            ck(lib.fvit_bwd_colsum_finish(part.data_ptr(), blocks, 2 * C_, grads.gamma.data_ptr(), C_, 1, st), "dgamma")
        ck(lib.fvit_bwd_colsum_finish(part.data_ptr() + 4 * C_, blocks, 2 * C_, grads.fc2_b.data_ptr(), C_, 1, st), "db2")
        # ---- fc2: dh = dz W2, dW2 += dz^T h ----
        ck(lib.fvit_gemm_bias_act(code, dz.data_ptr(), Ck, W2T.data_ptr(), Ck, None, dh.data_ptr(), hid, M, hid, Ck, 0, st), "dh")
        ck(lib.fvit_bwd_transpose16(code, dz.data_ptr(), Ck, dzT.data_ptr(), Mk, M, C_, st), "dz^T")
        ck(lib.fvit_bwd_transpose16(code, h.data_ptr(), hid, hT.data_ptr(), Mk, M, hid, st), "h^T")
        ck(lib.fvit_gemm_residual(code, dzT.data_ptr(), Mk, hT.data_ptr(), Mk, None, None, grads.fc2_w.data_ptr(), hid, C_, hid, Mk, st), "dW2")
        # ---- GELU, fc1 bias ----
        if hmask is not None:
            dh[:M] *= hmask         # adjoint of the hidden Dropout
        ck(lib.fvit_bwd_gelu(code, a.data_ptr(), hid, dh.data_ptr(), hid, da.data_ptr(), hid, part.data_ptr(), M, hid, st), "gelu_bwd")
        ck(lib.fvit_bwd_colsum_finish(part.data_ptr(), blocks, hid, grads.fc1_b.data_ptr(), hid, 1, st), "db1")
        # ---- fc1: dW1 += da^T xn, dxn = da W1 ----
        ck(lib.fvit_bwd_transpose16(code, da.data_ptr(), hid, daT.data_ptr(), Mk, M, hid, st), "da^T")
        ck(lib.fvit_bwd_transpose16(code, xn.data_ptr(), Ck, xnT.data_ptr(), Mk, M, C_, st), "xn^T")
        ck(lib.fvit_gemm_residual(code, daT.data_ptr(), Mk, xnT.data_ptr(), Mk, None, None, grads.fc1_w.data_ptr(), C_, hid, C_, Mk, st), "dW1")
        ck(lib.fvit_gemm_residual(code, da.data_ptr(), hid, W1T.data_ptr(), hid, None, None, dxn.data_ptr(), C_, M, C_, hid, st), "dxn")
        # ---- LayerNorm ----
        ck(lib.fvit_bwd_layernorm(x.data_ptr(), dxn.data_ptr(), dy.data_ptr(), ln_w.data_ptr(), C.c_float(eps), dx.data_ptr(), stats.data_ptr(),
                                  part.data_ptr(), M, C_, st), "layernorm_bwd")
        ck(lib.fvit_bwd_colsum_finish(part.data_ptr(), blocks, 2 * C_, grads.ln_w.data_ptr(), C_, 1, st), "dln_w")
        ck(lib.fvit_bwd_colsum_finish(part.data_ptr() + 4 * C_, blocks, 2 * C_, grads.ln_b.data_ptr(), C_, 1, st), "dln_b")
    return dx


@dataclass
class AttnGrads:
    """fp32 gradient buffers of one attention sub-block; ``attn_block_backward`` adds into them.  ``bias`` is the gradient of the folded
    relative-position bias table (heads, S, S) -- the table is a constant of the inference path (PosEmbMLPSwinv2D folded at load, FV:213-310);
    its own small MLP is differentiated from this on the host."""
    qkv_w: torch.Tensor
    qkv_b: torch.Tensor
    proj_w: torch.Tensor
    proj_b: torch.Tensor
    ln_w: torch.Tensor
    ln_b: torch.Tensor
    gamma: Optional[torch.Tensor]
    bias: Optional[torch.Tensor]

    @staticmethod
    def zeros(C_: int, heads: int, S: int, device, with_gamma: bool = True, with_bias: bool = True) -> "AttnGrads":
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)   # noqa: E731
        return AttnGrads(z(3 * C_, C_), z(3 * C_), z(C_, C_), z(C_), z(C_), z(C_), z(C_) if with_gamma else None, z(heads, S, S) if with_bias else None)


def _attn_geometry(C_: int, heads: int):
    d = C_ // heads
    dp = _dpad(d)
    HD = heads * dp
    return d, dp, HD, 3 * HD, _rup(3 * HD, 64), _rup(HD, 64), _rup(C_, 64)


def attn_block_backward(x: torch.Tensor, dy: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor, qkv_w: torch.Tensor, qkv_b: Optional[torch.Tensor],
                        proj_w: torch.Tensor, proj_b: torch.Tensor, gamma: Optional[torch.Tensor], bias: Optional[torch.Tensor], heads: int, S: int,
                        grads: AttnGrads, eps: float = 1e-5, qk_scale: Optional[float] = None, operand_dtype=torch.float16,
                        row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Backward of y = x + gamma * proj(softmax(q k^T * scale + bias) v), [q|k|v] = qkv(LayerNorm(x)), per window of S consecutive rows.
    x, dy: fp32 [nwin * S][C]; bias: fp32 (heads, S, S) or None; head_dim = C / heads <= 96 (the kernels run on the head_dim padded to 32 / 64 / 96, as
    the forward path does; C a multiple of 16) and S <= 64.  Returns dx; parameter gradients are added into ``grads``.  ``row_scale``: see
    ``mlp_block_backward`` (FV:690 ``self.drop_path(self.gamma3 * self.attn(...))``).  Kernel sequence as in ``mlp_block_backward`` with
    fvit_window_attention (recompute) / fvit_bwd_window_attention in the middle."""
    if not x.is_cuda:
        raise RuntimeError("attn_block_backward runs only on a HIP device (libfvit_hip.so kernels); there is no CPU fallback")
    if operand_dtype not in _CODE:
        raise ValueError("operand_dtype must be torch.float16 or torch.bfloat16")
    M, C_ = x.shape
    if C_ % 16 or C_ % heads or C_ // heads > 96 or S < 1 or S > 64 or M % S:
        raise RuntimeError(f"attn_block_backward: C = {C_}, heads = {heads}, S = {S}, rows = {M}: need head_dim <= 96, C % 16 == 0, S <= 64, rows % S == 0")
    for t, name in ((x, "x"), (dy, "dy")):
        if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != (M, C_) or t.device != x.device:
            raise RuntimeError(f"attn_block_backward: {name} must be a contiguous fp32 [rows][C] tensor on {x.device}")
    dev, dt, code = x.device, operand_dtype, _CODE[operand_dtype]
    lib = _lib.lib()
    d, dp, HD, C3p, Kq, Kao, Ck = _attn_geometry(C_, heads)
    padded = d != dp or Kao != C_
    nwin = M // S
    scale = float(qk_scale) if qk_scale else d ** -0.5
    Mp, Mk = _rup(M, 128), _rup(M, 64)
    f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()   # noqa: E731
    ln_w, ln_b, bp = f32(ln_w), f32(ln_b), f32(proj_b)
    g = f32(gamma) if gamma is not None else None
    wq, bq = _pack_qkv(f32(qkv_w), f32(qkv_b) if qkv_b is not None else torch.zeros(3 * C_, device=dev), heads, d, dp)   # [C3p][C], [C3p]
    wp = _pack_proj(f32(proj_w), heads, d, dp, Kao)                                                                      # [C][Kao]
    bq = bq.contiguous()
    Wq, Wp = _pad_rows(wq, dt, Ck), _pad_rows(wp, dt)                           # [pad(C3p)][Ck], [pad(C)][Kao]
    WqT, WpT = _pad_rows(wq.t().contiguous(), dt, Kq), _pad_rows(wp.t().contiguous(), dt, Ck)   # [pad(C)][Kq], [pad(Kao)][Ck]
    spad = lib.fvit_attention_spad(S)
    btab = torch.zeros(heads, spad, spad, dtype=torch.float32, device=dev)   # fvit_window_attention always takes a table (mask on padded keys)
    if bias is not None:
        btab[:, :S, :S] = f32(bias)
    btab[:, :, S:] = _lib.FVIT_MASK_BIAS
    e16 = lambda r, c: torch.zeros(r, c, dtype=dt, device=dev)   # noqa: E731
    xn, qkv, o, z = e16(Mp, Ck), e16(Mp, Kq), e16(Mp, Kao), e16(Mp, C_)
    dz, do, dqkv = e16(Mp, Ck), e16(Mp, Kao), e16(Mp, Kq)
    dzT, oT, xnT = e16(_rup(C_, 128), Mk), e16(_rup(Kao, 128), Mk), e16(_rup(C_, 128), Mk)
    dqkvT = e16(_rup(C3p, 128), Mk)
    blocks = lib.fvit_bwd_blocks(M)
    part = torch.empty(blocks * 2 * max(C3p, C_), dtype=torch.float32, device=dev)
    dbias_part = torch.empty(nwin * heads * S * S, dtype=torch.float32, device=dev) if grads.bias is not None else None
    dxn = torch.zeros(M, C_, dtype=torch.float32, device=dev)
    dx = torch.empty(M, C_, dtype=torch.float32, device=dev)
    stats = torch.empty(M, 2, dtype=torch.float32, device=dev)
    # gradient buffers in the padded layout (the accumulating GEMMs write straight into ``grads`` when no padding is involved)
    gq_w = torch.zeros(C3p, C_, dtype=torch.float32, device=dev) if padded else grads.qkv_w
    gq_b = torch.zeros(C3p, dtype=torch.float32, device=dev) if padded else grads.qkv_b
    gp_w = torch.zeros(C_, Kao, dtype=torch.float32, device=dev) if padded else grads.proj_w
    osc = _out_scale(row_scale, M)
    dyi = dy if osc is None else (dy * osc).contiguous()
    p = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream
        ck = _lib.check
        # ---- recompute the forward intermediates ----
        ck(lib.fvit_gather_layernorm(code, x.data_ptr(), M, None, 0, None, None, None, None, xn.data_ptr(), Ck, ln_w.data_ptr(), ln_b.data_ptr(),
                                     C.c_float(eps), M, M, C_, st), "layernorm")
        ck(lib.fvit_gemm_bias_act(code, xn.data_ptr(), Ck, Wq.data_ptr(), Ck, bq.data_ptr(), qkv.data_ptr(), Kq, M, C3p, Ck, 0, st), "qkv")
        pmask = _attn_mask(row_scale)
        ck(lib.fvit_window_attention_drop(code, qkv.data_ptr(), Kq, o.data_ptr(), Kao, p(btab), nwin, S, heads, dp, C.c_float(scale), p(pmask), st), "attention")
        ck(lib.fvit_gemm_bias_act(code, o.data_ptr(), Kao, Wp.data_ptr(), Kao, bp.data_ptr(), z.data_ptr(), C_, M, C_, Kao, 0, st), "proj")
        # ---- gamma, proj bias, dz = gamma * dy ----
        ck(lib.fvit_bwd_scale_cols(code, dyi.data_ptr(), z.data_ptr(), C_, p(g), dz.data_ptr(), Ck, part.data_ptr(), M, C_, st), "scale_cols")
        if grads.gamma is not None:
            ck(lib.fvit_bwd_colsum_finish(part.data_ptr(), blocks, 2 * C_, grads.gamma.data_ptr(), C_, 1, st), "dgamma")
        ck(lib.fvit_bwd_colsum_finish(part.data_ptr() + 4 * C_, blocks, 2 * C_, grads.proj_b.data_ptr(), C_, 1, st), "dbproj")
        # ---- proj: dO = dz Wp, dWp += dz^T O ----
        ck(lib.fvit_gemm_bias_act(code, dz.data_ptr(), Ck, WpT.data_ptr(), Ck, None, do.data_ptr(), Kao, M, Kao, Ck, 0, st), "dO")
        ck(lib.fvit_bwd_transpose16(code, dz.data_ptr(), Ck, dzT.data_ptr(), Mk, M, C_, st), "dz^T")
        ck(lib.fvit_bwd_transpose16(code, o.data_ptr(), Kao, oT.data_ptr(), Mk, M, Kao, st), "O^T")
        ck(lib.fvit_gemm_residual(code, dzT.data_ptr(), Mk, oT.data_ptr(), Mk, None, None, gp_w.data_ptr(), Kao, C_, Kao, Mk, st), "dWproj")
        # ---- attention core ----
        ck(lib.fvit_bwd_window_attention_drop(code, qkv.data_ptr(), Kq, do.data_ptr(), Kao, p(btab), spad, C.c_float(scale), dqkv.data_ptr(), p(dbias_part),
                                              nwin, S, heads, dp, p(pmask), st), "attention_bwd")
        if grads.bias is not None:
            ck(lib.fvit_bwd_colsum_finish(dbias_part.data_ptr(), nwin, heads * S * S, grads.bias.data_ptr(), heads * S * S, 1, st), "dbias")
        # ---- qkv: bias, dWqkv += dqkv^T xn, dxn = dqkv Wqkv ----
        ck(lib.fvit_bwd_colsum16(code, dqkv.data_ptr(), Kq, part.data_ptr(), M, C3p, st), "colsum dqkv")
        ck(lib.fvit_bwd_colsum_finish(part.data_ptr(), blocks, C3p, gq_b.data_ptr(), C3p, 1, st), "dbqkv")
        ck(lib.fvit_bwd_transpose16(code, dqkv.data_ptr(), Kq, dqkvT.data_ptr(), Mk, M, C3p, st), "dqkv^T")
        ck(lib.fvit_bwd_transpose16(code, xn.data_ptr(), Ck, xnT.data_ptr(), Mk, M, C_, st), "xn^T")
        ck(lib.fvit_gemm_residual(code, dqkvT.data_ptr(), Mk, xnT.data_ptr(), Mk, None, None, gq_w.data_ptr(), C_, C3p, C_, Mk, st), "dWqkv")
        ck(lib.fvit_gemm_residual(code, dqkv.data_ptr(), Kq, WqT.data_ptr(), Kq, None, None, dxn.data_ptr(), C_, M, C_, Kq, st), "dxn")
        # ---- LayerNorm ----
        ck(lib.fvit_bwd_layernorm(x.data_ptr(), dxn.data_ptr(), dy.data_ptr(), ln_w.data_ptr(), C.c_float(eps), dx.data_ptr(), stats.data_ptr(),
                                  part.data_ptr(), M, C_, st), "layernorm_bwd")
        ck(lib.fvit_bwd_colsum_finish(part.data_ptr(), blocks, 2 * C_, grads.ln_w.data_ptr(), C_, 1, st), "dln_w")
        ck(lib.fvit_bwd_colsum_finish(part.data_ptr() + 4 * C_, blocks, 2 * C_, grads.ln_b.data_ptr(), C_, 1, st), "dln_b")
    if padded:   # the real rows / columns of the padded-layout gradients
        grads.qkv_w += _unpack_qkv(gq_w, heads, d, dp)
        grads.qkv_b += _unpack_qkv(gq_b, heads, d, dp)
        grads.proj_w += _unpack_proj(gp_w, heads, d, dp)
    return dx


def _lerp_rows(x: torch.Tensor, y0: torch.Tensor, row_scale: Optional[torch.Tensor]) -> torch.Tensor:
    """Stochastic depth on top of the residual epilogue: y0 = x + f  ->  x + row_scale[m] * f."""
    sc = _out_scale(row_scale, x.shape[0])
    if sc is None:
        return y0
    return torch.addcmul(x, y0 - x, sc.to(x.dtype).expand_as(x))


def attn_block_forward(x: torch.Tensor, ln_w, ln_b, qkv_w, qkv_b, proj_w, proj_b, gamma, bias, heads: int, S: int, eps: float = 1e-5,
                       qk_scale: Optional[float] = None, operand_dtype=torch.float16, row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x + row_scale * gamma * proj(attention(qkv(LayerNorm(x)))) through the unit kernels of the forward path (LayerNorm, GEMM, attention core, GEMM
    with the residual epilogue): the activation recompute a block-level backward starts from, and the train-mode forward (DropPath = ``row_scale``)."""
    if not x.is_cuda:
        raise RuntimeError("attn_block_forward runs only on a HIP device (libfvit_hip.so kernels); there is no CPU fallback")
    M, C_ = x.shape
    dev, dt, code = x.device, operand_dtype, _CODE[operand_dtype]
    lib = _lib.lib()
    d, dp, HD, C3p, Kq, Kao, Ck = _attn_geometry(C_, heads)
    nwin, Mp = M // S, _rup(M, 128)
    scale = float(qk_scale) if qk_scale else d ** -0.5
    f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()   # noqa: E731
    wq, bq = _pack_qkv(f32(qkv_w), f32(qkv_b) if qkv_b is not None else torch.zeros(3 * C_, device=dev), heads, d, dp)
    bq = bq.contiguous()
    Wq, Wp = _pad_rows(wq, dt, Ck), _pad_rows(_pack_proj(f32(proj_w), heads, d, dp, Kao), dt)
    spad = lib.fvit_attention_spad(S)
    btab = torch.zeros(heads, spad, spad, dtype=torch.float32, device=dev)
    if bias is not None:
        btab[:, :S, :S] = f32(bias)
    btab[:, :, S:] = _lib.FVIT_MASK_BIAS
    xn, qkv, o = (torch.zeros(Mp, n, dtype=dt, device=dev) for n in (Ck, Kq, Kao))
    y = x.clone()
    lw, lb, bp = f32(ln_w), f32(ln_b), f32(proj_b)
    g = f32(gamma) if gamma is not None else None
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream
        ck = _lib.check
        ck(lib.fvit_gather_layernorm(code, x.data_ptr(), M, None, 0, None, None, None, None, xn.data_ptr(), Ck, lw.data_ptr(), lb.data_ptr(),
                                     C.c_float(eps), M, M, C_, st), "layernorm")
        ck(lib.fvit_gemm_bias_act(code, xn.data_ptr(), Ck, Wq.data_ptr(), Ck, bq.data_ptr(), qkv.data_ptr(), Kq, M, C3p, Ck, 0, st), "qkv")
        pm = _attn_mask(row_scale)
        ck(lib.fvit_window_attention_drop(code, qkv.data_ptr(), Kq, o.data_ptr(), Kao, btab.data_ptr(), nwin, S, heads, dp, C.c_float(scale),
                                          None if pm is None else pm.data_ptr(), st), "attention")
        ck(lib.fvit_gemm_residual(code, o.data_ptr(), Kao, Wp.data_ptr(), Kao, bp.data_ptr(), None if g is None else g.data_ptr(), y.data_ptr(), C_,
                                  M, C_, Kao, st), "proj")
    return _lerp_rows(x, y, row_scale)


def local_block_backward(x: torch.Tensor, dy: torch.Tensor, attn: dict, mlp: dict, heads: int, S: int, attn_grads: AttnGrads, mlp_grads: MlpGrads,
                         eps: float = 1e-5, operand_dtype=torch.float16, masks: Optional[dict] = None) -> torch.Tensor:
    """Backward of one HAT block WITHOUT carrier tokens (the stage-3 form, FV:690-691 with ct = None):
         x1 = x + gamma3 * attn(norm1(x));  y = x1 + gamma4 * mlp(norm2(x1)).
    ``attn`` = dict(ln_w, ln_b, qkv_w, qkv_b, proj_w, proj_b, gamma, bias[, scale]), ``mlp`` = dict(ln_w, ln_b, fc1_w, fc1_b, fc2_w, fc2_b, gamma);
    ``masks`` = dict(attn=, mlp=) per-WINDOW DropPath factors or None (train mode).  x1 is recomputed by the forward kernels, then the two sub-block
    backwards run in reverse order.  Returns dx."""
    ra, rm = _rs(masks, "attn", S), _rs(masks, "mlp", S)
    x1 = attn_block_forward(x, attn["ln_w"], attn["ln_b"], attn["qkv_w"], attn.get("qkv_b"), attn["proj_w"], attn["proj_b"], attn.get("gamma"),
                            attn.get("bias"), heads, S, eps, attn.get("scale"), operand_dtype, ra)
    dx1 = mlp_block_backward(x1, dy, mlp["ln_w"], mlp["ln_b"], mlp["fc1_w"], mlp["fc1_b"], mlp["fc2_w"], mlp["fc2_b"], mlp.get("gamma"), mlp_grads, eps,
                             operand_dtype, rm)
    return attn_block_backward(x, dx1, attn["ln_w"], attn["ln_b"], attn["qkv_w"], attn.get("qkv_b"), attn["proj_w"], attn["proj_b"], attn.get("gamma"),
                               attn.get("bias"), heads, S, attn_grads, eps, attn.get("scale"), operand_dtype, ra)


def mlp_block_forward(x: torch.Tensor, ln_w, ln_b, fc1_w, fc1_b, fc2_w, fc2_b, gamma, eps: float = 1e-5, operand_dtype=torch.float16,
                      row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x + row_scale * gamma * fc2(GELU(fc1(LayerNorm(x)))) through the unit kernels of the forward path (activation recompute for the block-level
    backwards; train-mode forward with DropPath = ``row_scale``)."""
    if not x.is_cuda:
        raise RuntimeError("mlp_block_forward runs only on a HIP device (libfvit_hip.so kernels); there is no CPU fallback")
    M, C_ = x.shape
    hid = fc1_w.shape[0]
    dev, dt, code = x.device, operand_dtype, _CODE[operand_dtype]
    lib = _lib.lib()
    Mp, Ck = _rup(M, 128), _rup(C_, 64)
    f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()   # noqa: E731
    W1, W2 = _pad_rows(f32(fc1_w), dt, Ck), _pad_rows(f32(fc2_w), dt)
    xn, h = torch.zeros(Mp, Ck, dtype=dt, device=dev), torch.zeros(Mp, hid, dtype=dt, device=dev)
    y = x.clone()
    lw, lb, b1, b2 = f32(ln_w), f32(ln_b), f32(fc1_b), f32(fc2_b)
    g = f32(gamma) if gamma is not None else None
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream
        ck = _lib.check
        ck(lib.fvit_gather_layernorm(code, x.data_ptr(), M, None, 0, None, None, None, None, xn.data_ptr(), Ck, lw.data_ptr(), lb.data_ptr(),
                                     C.c_float(eps), M, M, C_, st), "layernorm")
        ck(lib.fvit_gemm_bias_act(code, xn.data_ptr(), Ck, W1.data_ptr(), Ck, b1.data_ptr(), h.data_ptr(), hid, M, hid, Ck, 1, st), "fc1 + GELU")
        if _hid_mask(row_scale) is not None:
            h[:M] *= _hid_mask(row_scale)   # Dropout on GELU(fc1) (FV:404)
        ck(lib.fvit_gemm_residual(code, h.data_ptr(), hid, W2.data_ptr(), hid, b2.data_ptr(), None if g is None else g.data_ptr(), y.data_ptr(), C_,
                                  M, C_, hid, st), "fc2")
    return _lerp_rows(x, y, row_scale)


def _carrier_permutations(sr0: int, sr1: int, cw: int, device):
    """Index form of the reference's carrier-token reshuffles for G = cw^2 * sr0 * sr1 tokens per image (the view / permute chains of ct_dewindow
    AR:97-102 and ct_window AR:105-110 applied to an arange; the two are inverses only on square grids, which is reproduced):
    dewindowed[:, r] = ct[:, dew[r]];  windowed[:, p] = ct2[:, win[p]]."""
    G = cw * cw * sr0 * sr1
    ar = torch.arange(G, device=device)
    dew = ar.view(sr0, sr1, cw, cw).permute(0, 2, 1, 3).reshape(G)
    win = ar.view(sr1, cw, sr0, cw).permute(0, 2, 1, 3).reshape(G)
    return dew, win


def _upsample_index(ws: int, cw: int, device) -> torch.Tensor:
    """nn.Upsample(size=ws, mode='nearest') of the cw x cw carrier grid of a window (FV:656, 699): carrier slot of window token t."""
    near = (torch.arange(ws, device=device) * cw) // ws
    return (near[:, None] * cw + near[None, :]).reshape(ws * ws)


def _hier_forward_parts(x, ct, hat_attn, hat_mlp, attn, mlp, heads, ws, cw, sr, pe_x, pe_ct, eps, od, masks):
    """Forward of one carrier-token HAT block through the unit kernels, keeping the intermediates the backward starts from."""
    Bw, nloc, C_ = x.shape
    B, G, _ = ct.shape
    sr0, sr1 = int(sr[0]), int(sr[1])
    ncw = cw * cw
    S = ncw + nloc
    dev = x.device
    dew, win = _carrier_permutations(sr0, sr1, cw, dev)
    mk = masks or {}
    r_ha, r_hm, r_a, r_m = _rs(mk, "hat_attn", G), _rs(mk, "hat_mlp", G), _rs(mk, "attn", S), _rs(mk, "mlp", S)
    x0 = (x + pe_x.to(dev)) if pe_x is not None else x
    ct0 = ct[:, dew]
    if pe_ct is not None:
        ct0 = ct0 + pe_ct.to(dev)
    ct0 = ct0.reshape(B * G, C_).contiguous()
    ha, hm = hat_attn, hat_mlp
    ct1 = attn_block_forward(ct0, ha["ln_w"], ha["ln_b"], ha["qkv_w"], ha.get("qkv_b"), ha["proj_w"], ha["proj_b"], ha.get("gamma"), ha.get("bias"), heads, G, eps,
                             ha.get("scale"), od, r_ha)
    ct2 = mlp_block_forward(ct1, hm["ln_w"], hm["ln_b"], hm["fc1_w"], hm["fc1_b"], hm["fc2_w"], hm["fc2_b"], hm.get("gamma"), eps, od, r_hm)
    ctw = ct2.view(B, G, C_)[:, win].reshape(Bw, ncw, C_)
    xin = torch.cat((ctw, x0), dim=1).reshape(Bw * S, C_).contiguous()
    y1 = attn_block_forward(xin, attn["ln_w"], attn["ln_b"], attn["qkv_w"], attn.get("qkv_b"), attn["proj_w"], attn["proj_b"], attn.get("gamma"), attn.get("bias"),
                            heads, S, eps, attn.get("scale"), od, r_a)
    return dict(ct0=ct0, ct1=ct1, xin=xin, y1=y1, dew=dew, win=win, rows=(r_ha, r_hm, r_a, r_m), S=S, ncw=ncw)


def hier_block_backward(x: torch.Tensor, ct: torch.Tensor, dx_out: torch.Tensor, dct_out: torch.Tensor, hat_attn: dict, hat_mlp: dict, attn: dict, mlp: dict,
                        heads: int, ws: int, cw: int, sr, pe_x: Optional[torch.Tensor], pe_ct: Optional[torch.Tensor], grads: dict, eps: float = 1e-5,
                        operand_dtype=torch.float16, masks: Optional[dict] = None, prop_gamma=None, prop_grad: Optional[torch.Tensor] = None):
    """Backward of one HAT block WITH carrier tokens (HAT.forward FV:662-701 / AR:668-707):
         x (B nW, ws^2, C), ct (B, G, C)  ->  x_out, ct_out  with the carrier branch (ct_dewindow, hat_pos_embed, hat_attn, hat_mlp, ct_window), the
         concatenation [carrier | window] per window, the window attention and MLP sub-blocks, the final split and -- for the LAST block of a stage built
         with ``do_propagation`` (``prop_gamma``: the block's gamma1, a tensor or the number 1; None = no propagation) -- FV:697-700
         x_out += gamma1 * upsample_nearest(carrier tokens of the window).
    ``hat_attn`` / ``attn`` / ``hat_mlp`` / ``mlp``: parameter dicts as in ``local_block_backward`` (bias = the folded relative-position table of that attention);
    ``pe_x`` (ws^2, C) / ``pe_ct`` (G, C or None): the constant-folded 1-D position embeddings (PosEmbMLPSwinv1D, input independent).
    ``grads`` = dict(hat_attn=AttnGrads, hat_mlp=MlpGrads, attn=AttnGrads, mlp=MlpGrads); ``masks`` = per-image (hat_attn, hat_mlp) / per-window (attn, mlp)
    DropPath factors; ``prop_grad`` (fp32 [C], optional) receives the propagation's share of d gamma1.  Returns (dx, dct).
    The sub-block backwards run on the kernels; the carrier reshuffles, the concatenation / split and the propagation are row gathers / scatters applied
    with torch indexing."""
    if not x.is_cuda:
        raise RuntimeError("hier_block_backward runs only on a HIP device (libfvit_hip.so kernels); there is no CPU fallback")
    Bw, nloc, C_ = x.shape
    B, G, _ = ct.shape
    sr0, sr1 = int(sr[0]), int(sr[1])
    nW, ncw = sr0 * sr1, cw * cw
    if nloc != ws * ws or Bw != B * nW or G != ncw * nW:
        raise RuntimeError(f"hier_block_backward: x {tuple(x.shape)} / ct {tuple(ct.shape)} do not match ws={ws} cw={cw} sr={sr}")
    od = operand_dtype
    f = _hier_forward_parts(x, ct, hat_attn, hat_mlp, attn, mlp, heads, ws, cw, sr, pe_x, pe_ct, eps, od, masks)
    S, dew, win = f["S"], f["dew"], f["win"]
    r_ha, r_hm, r_a, r_m = f["rows"]
    ha, hm = hat_attn, hat_mlp
    # ---- backward ----
    dctw_out = dct_out.reshape(Bw, ncw, C_)
    if prop_gamma is not None:
        # x_out[w, t] = x2[w, t] + gamma1 * ctr[w, up[t]], ctr = the window's carrier rows AFTER the MLP: recompute them
        y2 = mlp_block_forward(f["y1"], mlp["ln_w"], mlp["ln_b"], mlp["fc1_w"], mlp["fc1_b"], mlp["fc2_w"], mlp["fc2_b"], mlp.get("gamma"), eps, od, r_m).view(Bw, S, C_)
        up = _upsample_index(ws, cw, x.device)
        gam = prop_gamma.detach().float().to(x.device) if isinstance(prop_gamma, torch.Tensor) else None
        if prop_grad is not None and gam is not None:
            prop_grad += (dx_out * y2[:, :ncw][:, up]).sum(dim=(0, 1))
        contrib = dx_out if gam is None else dx_out * gam
        dctw_out = dctw_out.clone().index_add_(1, up, contrib)
    dy2 = torch.cat((dctw_out, dx_out), dim=1).reshape(Bw * S, C_).contiguous()
    dy1 = mlp_block_backward(f["y1"], dy2, mlp["ln_w"], mlp["ln_b"], mlp["fc1_w"], mlp["fc1_b"], mlp["fc2_w"], mlp["fc2_b"], mlp.get("gamma"), grads["mlp"], eps, od, r_m)
    dxin = attn_block_backward(f["xin"], dy1, attn["ln_w"], attn["ln_b"], attn["qkv_w"], attn.get("qkv_b"), attn["proj_w"], attn["proj_b"], attn.get("gamma"),
                               attn.get("bias"), heads, S, grads["attn"], eps, attn.get("scale"), od, r_a).view(Bw, S, C_)
    dx = dxin[:, ncw:].contiguous()
    dctw = dxin[:, :ncw].reshape(B, G, C_)
    dct2 = torch.empty_like(dctw)
    dct2[:, win] = dctw                                   # adjoint of windowed[:, p] = ct2[:, win[p]]
    dct2 = dct2.reshape(B * G, C_).contiguous()
    dct1 = mlp_block_backward(f["ct1"], dct2, hm["ln_w"], hm["ln_b"], hm["fc1_w"], hm["fc1_b"], hm["fc2_w"], hm["fc2_b"], hm.get("gamma"), grads["hat_mlp"], eps, od, r_hm)
    dct0 = attn_block_backward(f["ct0"], dct1, ha["ln_w"], ha["ln_b"], ha["qkv_w"], ha.get("qkv_b"), ha["proj_w"], ha["proj_b"], ha.get("gamma"), ha.get("bias"), heads, G,
                               grads["hat_attn"], eps, ha.get("scale"), od, r_ha).view(B, G, C_)
    dct = torch.empty_like(dct0)
    dct[:, dew] = dct0                                    # adjoint of dewindowed[:, r] = ct[:, dew[r]]
    return dx, dct


def hier_block_forward(x: torch.Tensor, ct: torch.Tensor, hat_attn: dict, hat_mlp: dict, attn: dict, mlp: dict, heads: int, ws: int, cw: int, sr,
                       pe_x: Optional[torch.Tensor], pe_ct: Optional[torch.Tensor], eps: float = 1e-5, operand_dtype=torch.float16,
                       masks: Optional[dict] = None, prop_gamma=None):
    """Forward of one HAT block with carrier tokens through the unit kernels (the recompute ``hier_block_backward`` starts from; the train-mode forward):
    returns (x_out, ct_out) in the reference's layouts, x (B nW, ws^2, C), ct (B, G, C).  ``prop_gamma``: see ``hier_block_backward``."""
    Bw, nloc, C_ = x.shape
    B, G, _ = ct.shape
    f = _hier_forward_parts(x, ct, hat_attn, hat_mlp, attn, mlp, heads, ws, cw, sr, pe_x, pe_ct, eps, operand_dtype, masks)
    S, ncw = f["S"], f["ncw"]
    y2 = mlp_block_forward(f["y1"], mlp["ln_w"], mlp["ln_b"], mlp["fc1_w"], mlp["fc1_b"], mlp["fc2_w"], mlp["fc2_b"], mlp.get("gamma"), eps, operand_dtype,
                           f["rows"][3]).view(Bw, S, C_)
    xo, cto = y2[:, ncw:].contiguous(), y2[:, :ncw].reshape(B, G, C_).contiguous()
    if prop_gamma is not None:
        up = _upsample_index(ws, cw, x.device)
        ctr = y2[:, :ncw][:, up]
        xo = xo + (ctr * prop_gamma.detach().float().to(x.device) if isinstance(prop_gamma, torch.Tensor) else ctr)
    return xo, cto


def _table_with_grad(mod, *args):
    """The module's folded-table function (PosEmbMLPSwinv1D.table / PosEmbMLPSwinv2D.table: inference code under @torch.no_grad) run WITH autograd, so
    that a table gradient can continue into the small MLP behind it."""
    fn = type(mod).table
    fn = getattr(fn, "__wrapped__", fn)
    with torch.enable_grad():
        return fn(mod, *args)


def _acc_grad(param, g: torch.Tensor) -> None:
    if isinstance(param, torch.nn.Parameter) and param.requires_grad:
        g = g.to(param.dtype).reshape(param.shape)
        param.grad = g.clone() if param.grad is None else param.grad + g


def _emit(sink: Optional[dict], param, g: torch.Tensor) -> None:
    """Hand one parameter gradient over: into ``sink`` (param -> fp32 gradient; the autograd bridge returns these to the engine, so that
    DistributedDataParallel's reducer and post-accumulate hooks fire exactly once, from the OUTER backward) or, for the direct callers of the
    ``*_stage_backward`` functions (sink None), accumulated into ``.grad``."""
    if not (isinstance(param, torch.nn.Parameter) and param.requires_grad):
        return
    if sink is None:
        _acc_grad(param, g)
        return
    g = g.detach().float().reshape(param.shape)
    sink[param] = g.clone() if param not in sink else sink[param] + g


def _module_params(mods):
    seen, out = set(), []
    for m in mods:
        if m is None:
            continue
        for prm in m.parameters():
            if prm.requires_grad and id(prm) not in seen:
                seen.add(id(prm))
                out.append(prm)
    return out


def _table_grads(sink: Optional[dict], outs, gouts, mods) -> None:
    """Gradients of the folded tables continue into the small MLPs behind them (PosEmbMLPSwinv1D / PosEmbMLPSwinv2D, FV:213-367) with
    ``torch.autograd.grad`` -- NOT ``torch.autograd.backward``: a nested backward would run the parameters' AccumulateGrad nodes (and with them
    DDP's reducer hooks) from inside the outer backward, once on a partial gradient and again from the outer pass."""
    live = [(o, g) for o, g in zip(outs, gouts) if o is not None and g is not None and o.requires_grad]
    params = _module_params(mods)
    if not live or not params:
        return
    gs = torch.autograd.grad([o for o, _ in live], params, [g.to(o.dtype) for o, g in live], allow_unused=True)
    for prm, g in zip(params, gs):
        if g is not None:
            _emit(sink, prm, g)


def drop_path_masks(layer, batch: int, windows_per_image: int, device, generator=None, operand_dtype=torch.float16):
    """Per-block stochastic factors of a stage in TRAIN mode: a list with one dict per block.
      * DropPath (timm semantics, FV:630, 652, 690-691: one Bernoulli(keep) / keep draw per sample of the tensor the DropPath is applied to -- per WINDOW for
        the window branch (x is (B nW, S, C)), per IMAGE for the carrier branch): entries ``attn`` / ``mlp`` / ``hat_attn`` / ``hat_mlp``, None where p = 0;
      * Dropout inside the blocks (r05; ``drop_rate`` of the entrypoints -> ``Mlp.drop`` after GELU and after fc2, FV:404-406, and ``WindowAttention.proj_drop``,
        FV:567): elementwise masks ``<key>_out`` fp32 [rows][C] and ``mlp_hid`` / ``hat_mlp_hid`` operand-dtype [rows][hidden], values 0 or 1 / keep, only where p > 0
        and ``attn_p`` / ``hat_attn_p`` operand-dtype [windows * heads][S][Spad] for ``WindowAttention.attn_drop`` (FV:564: Dropout on the softmax probabilities,
        applied inside the attention kernels)."""
    out = []
    for blk in layer.blocks:
        def draw(n, mod):
            p_ = float(getattr(mod, "drop_prob", 0.0) or 0.0)
            if p_ <= 0.0:
                return None
            keep = 1.0 - p_
            return torch.empty(n, dtype=torch.float32, device=device).bernoulli_(keep, generator=generator) / keep

        def unbiased(keep, dt):
            # the kept value 1 / keep is STORED in dt: in bf16 1 / 0.9 rounds to 1.109375 and E[mask] would be 0.9984 instead of 1.  Draw with the
            # keep probability whose reciprocal is exactly representable -- 1 / round_dt(1 / keep), within 2^-9 of the requested one -- so that E[mask] = 1
            inv = torch.tensor(1.0 / keep, dtype=torch.float32).to(dt).float().item()
            return 1.0 / inv, inv

        def elem(rows, cols, mod, dt=torch.float32):
            p_ = float(getattr(mod, "p", 0.0) or 0.0)
            if p_ <= 0.0:
                return None
            keep, inv = unbiased(1.0 - p_, dt)
            return (torch.empty(rows, cols, dtype=torch.float32, device=device).bernoulli_(keep, generator=generator) * inv).to(dt)

        C_, hid = blk.attn.qkv.in_features, blk.mlp.fc1.out_features
        ncw = blk.cr_window ** 2 if blk.do_sr_hat else 0
        rows = batch * windows_per_image * (blk.window_size ** 2 + ncw)
        lib = _lib.lib()
        S_ = blk.window_size ** 2 + ncw
        heads = blk.attn.num_heads

        def pmask(items, S, mod):   # [items][S][Spad]: the attention kernels index the key axis at the padded stride
            p_ = float(getattr(mod, "p", 0.0) or 0.0)
            if p_ <= 0.0:
                return None
            keep, inv = unbiased(1.0 - p_, operand_dtype)
            return (torch.empty(items, S, lib.fvit_attention_spad(S), dtype=torch.float32, device=device).bernoulli_(keep, generator=generator) * inv).to(operand_dtype)

        m = dict(attn=draw(batch * windows_per_image, blk.drop_path), mlp=draw(batch * windows_per_image, blk.drop_path),
                 attn_out=elem(rows, C_, blk.attn.proj_drop), mlp_out=elem(rows, C_, blk.mlp.drop), mlp_hid=elem(rows, hid, blk.mlp.drop, operand_dtype),
                 attn_p=pmask(batch * windows_per_image * heads, S_, blk.attn.attn_drop))
        if blk.do_sr_hat:
            crow = batch * windows_per_image * ncw
            m.update(hat_attn=draw(batch, blk.hat_drop_path), hat_mlp=draw(batch, blk.hat_drop_path),
                     hat_attn_out=elem(crow, C_, blk.hat_attn.proj_drop), hat_mlp_out=elem(crow, C_, blk.hat_mlp.drop),
                     hat_mlp_hid=elem(crow, hid, blk.hat_mlp.drop, operand_dtype),
                     hat_attn_p=pmask(batch * heads, windows_per_image * ncw, blk.hat_attn.attn_drop))
        out.append(m)
    return out


def _pad_map(t: torch.Tensor, ws: int):
    H, W = t.shape[2:]
    pb, pr = (ws - H % ws) % ws, (ws - W % ws) % ws
    return (torch.nn.functional.pad(t, (0, pr, 0, pb)) if (pb or pr) else t), H + pb, W + pr


def _gm(v):
    return v if isinstance(v, torch.Tensor) else None


def _local_params(blk, S):
    bias_t = _table_with_grad(blk.attn.pos_emb_funct, S)   # (heads, S, S), differentiable w.r.t. cpb_mlp
    pe_t = _table_with_grad(blk.pos_embed)                 # (S, C)
    a = dict(ln_w=blk.norm1.weight, ln_b=blk.norm1.bias, qkv_w=blk.attn.qkv.weight, qkv_b=blk.attn.qkv.bias, proj_w=blk.attn.proj.weight,
             proj_b=blk.attn.proj.bias, gamma=_gm(blk.gamma3), bias=bias_t.detach(), scale=float(blk.attn.scale))
    m = dict(ln_w=blk.norm2.weight, ln_b=blk.norm2.bias, fc1_w=blk.mlp.fc1.weight, fc1_b=blk.mlp.fc1.bias, fc2_w=blk.mlp.fc2.weight, fc2_b=blk.mlp.fc2.bias,
             gamma=_gm(blk.gamma4))
    return a, m, bias_t, pe_t


def _local_stage_run(layer, x: torch.Tensor, operand_dtype, masks=None):
    """Forward of a carrier-free stage through the unit kernels: (output map, per-block inputs, per-block parameter dicts, geometry)."""
    B, C_, H, W = x.shape
    blocks = list(layer.blocks)
    if not blocks or any(b.do_sr_hat for b in blocks):
        raise RuntimeError("local stage: the stage must consist of HAT blocks without carrier tokens")
    ws = blocks[0].window_size
    xp, Hp, Wp = _pad_map(x.float(), ws)                  # F.pad of AR:851-853 (zeros); the output is cropped back (AR:866-867)
    heads, S = blocks[0].attn.num_heads, ws * ws
    if S > 64:
        raise RuntimeError(f"local stage: windows of {S} tokens (the attention-core backward holds at most 64)")
    nh, nw = Hp // ws, Wp // ws

    def partition(t):   # (B, C, Hp, Wp) -> (B nW S, C) rows in window order (window_partition FV:83-87)
        return t.reshape(B, C_, nh, ws, nw, ws).permute(0, 2, 4, 3, 5, 1).reshape(B * nh * nw * S, C_).contiguous()

    def reverse(r):     # rows -> (B, C, Hp, Wp) (window_reverse FV:90-93); the two are each other's adjoint
        return r.view(B, nh, nw, ws, ws, C_).permute(0, 5, 1, 3, 2, 4).reshape(B, C_, Hp, Wp).contiguous()

    rows = partition(xp)
    ins, ps = [], []
    for bi, blk in enumerate(blocks):
        a, m, bias_t, pe_t = _local_params(blk, S)
        mk = masks[bi] if masks is not None else None
        xin = (rows.view(-1, S, C_) + pe_t.detach().to(rows.dtype)).reshape(-1, C_).contiguous()
        ins.append(xin)
        ps.append((a, m, bias_t, pe_t))
        x1 = attn_block_forward(xin, a["ln_w"], a["ln_b"], a["qkv_w"], a["qkv_b"], a["proj_w"], a["proj_b"], a["gamma"], a["bias"], heads, S, 1e-5,
                                a["scale"], operand_dtype, _rs(mk, "attn", S))
        rows = mlp_block_forward(x1, m["ln_w"], m["ln_b"], m["fc1_w"], m["fc1_b"], m["fc2_w"], m["fc2_b"], m["gamma"], 1e-5, operand_dtype,
                                 _rs(mk, "mlp", S))
    return reverse(rows)[:, :, :H, :W], ins, ps, (partition, reverse, heads, S, Hp, Wp)


def local_stage_backward(layer, x: torch.Tensor, dy: torch.Tensor, operand_dtype=torch.float16, sink: Optional[dict] = None, masks=None) -> torch.Tensor:
    """Backward of the transformer branch of a FasterViTLayer WITHOUT carrier tokens (FasterViTLayer.forward FV:832-841 / AR:848-869 with only-local HAT blocks:
    stage 3) -- zero padding to a multiple of the window, window_partition, depth x HAT block, window_reverse, crop -- for the module's own parameters:

      x, dy: (B, C, H, W) fp32 maps;  returns dx and ADDS every parameter gradient of ``layer.blocks`` into ``.grad`` (``sink`` None) or into the dict
      ``sink`` (param -> gradient; nothing is written to ``.grad`` then: the autograd bridge below) -- norm1 / norm2, attn.qkv / attn.proj, mlp.fc1 / fc2,
      gamma3 / gamma4 when they are parameters, and -- through the modules' differentiable ``table()`` functions on the host -- the cpb_mlp of the
      relative-position bias and of the 1-D position embedding, from the table gradients the kernels return.  ``masks``: ``drop_path_masks`` of a
      train-mode forward (the same draws must be passed to the forward and the backward).

    Forward activations are recomputed block by block with the unit kernels (one fp32 row checkpoint per block boundary); the block backwards are
    ``local_block_backward``.  The downsample conv of the layer is not part of this path."""
    if not x.is_cuda:
        raise RuntimeError("local_stage_backward runs only on a HIP device (libfvit_hip.so kernels); there is no CPU fallback")
    B, C_, H, W = x.shape
    blocks = list(layer.blocks)
    _, ins, ps, (partition, reverse, heads, S, Hp, Wp) = _local_stage_run(layer, x, operand_dtype, masks)
    dyp = torch.nn.functional.pad(dy.float(), (0, Wp - W, 0, Hp - H)) if (Hp != H or Wp != W) else dy.float()   # adjoint of the crop
    d = partition(dyp)
    for bi in range(len(blocks) - 1, -1, -1):
        blk, xin, (a, m, bias_t, pe_t) = blocks[bi], ins[bi], ps[bi]
        ag = AttnGrads.zeros(C_, heads, S, x.device, with_gamma=a["gamma"] is not None)
        mg = MlpGrads.zeros(C_, m["fc1_w"].shape[0], x.device, with_gamma=m["gamma"] is not None)
        d = local_block_backward(xin, d, a, m, heads, S, ag, mg, 1e-5, operand_dtype, masks[bi] if masks is not None else None)
        for prm, g in ((blk.norm1.weight, ag.ln_w), (blk.norm1.bias, ag.ln_b), (blk.attn.qkv.weight, ag.qkv_w), (blk.attn.qkv.bias, ag.qkv_b),
                       (blk.attn.proj.weight, ag.proj_w), (blk.attn.proj.bias, ag.proj_b), (blk.gamma3, ag.gamma), (blk.norm2.weight, mg.ln_w),
                       (blk.norm2.bias, mg.ln_b), (blk.mlp.fc1.weight, mg.fc1_w), (blk.mlp.fc1.bias, mg.fc1_b), (blk.mlp.fc2.weight, mg.fc2_w),
                       (blk.mlp.fc2.bias, mg.fc2_b), (blk.gamma4, mg.gamma)):
            if g is not None:
                _emit(sink, prm, g)
        # the folded tables are functions of small MLPs: their gradients continue on the host through the modules' own table() code
        _table_grads(sink, [bias_t, pe_t], [ag.bias, d.view(-1, S, C_).sum(0)], [blk.attn.pos_emb_funct, blk.pos_embed])
    return reverse(d)[:, :, :H, :W].contiguous()   # adjoint of the zero padding


def _hier_params(blk, S, G):
    t = dict(bias=_table_with_grad(blk.attn.pos_emb_funct, S), hat_bias=_table_with_grad(blk.hat_attn.pos_emb_funct, G), pe_x=_table_with_grad(blk.pos_embed),
             pe_ct=_table_with_grad(blk.hat_pos_embed) if hasattr(blk, "hat_pos_embed") and blk.square else None)
    mk_a = lambda n, at, g, bias: dict(ln_w=n.weight, ln_b=n.bias, qkv_w=at.qkv.weight, qkv_b=at.qkv.bias, proj_w=at.proj.weight, proj_b=at.proj.bias,   # noqa: E731
                                       gamma=_gm(g), bias=bias.detach(), scale=float(at.scale))
    mk_m = lambda n, ml, g: dict(ln_w=n.weight, ln_b=n.bias, fc1_w=ml.fc1.weight, fc1_b=ml.fc1.bias, fc2_w=ml.fc2.weight, fc2_b=ml.fc2.bias, gamma=_gm(g))   # noqa: E731
    return dict(hat_attn=mk_a(blk.hat_norm1, blk.hat_attn, blk.gamma1, t["hat_bias"]), hat_mlp=mk_m(blk.hat_norm2, blk.hat_mlp, blk.gamma2),
                attn=mk_a(blk.norm1, blk.attn, blk.gamma3, t["bias"]), mlp=mk_m(blk.norm2, blk.mlp, blk.gamma4)), t


def _prop_gamma(blk):
    """gamma1 of the last block when the stage propagates carrier tokens into the map (FV:697-700; the reference re-uses gamma1 here), else None."""
    if blk.last and blk.do_propagation and blk.do_sr_hat:
        return blk.gamma1 if isinstance(blk.gamma1, torch.Tensor) else 1.0
    return None


def _hier_stage_run(layer, x: torch.Tensor, operand_dtype, masks=None):
    B, C_, H, W = x.shape
    blocks = list(layer.blocks)
    if not blocks or not all(b.do_sr_hat for b in blocks):
        raise RuntimeError("hier stage: the stage must consist of carrier-token HAT blocks")
    b0 = blocks[0]
    ws, cw, sr = b0.window_size, b0.cr_window, tuple(b0.sr_ratio)
    x_leaf = x.detach().float().requires_grad_(True)
    with torch.enable_grad():
        xp_leaf, Hp, Wp = _pad_map(x_leaf, ws)           # the TokenInitializer sees the padded map (AR:851-858)
        if (Hp // ws, Wp // ws) != sr:
            raise RuntimeError(f"hier stage: map {H}x{W} (padded {Hp}x{Wp}) does not tile into the stage's {sr[0]}x{sr[1]} windows of {ws}")
        ct_init = layer.global_tokenizer(xp_leaf)
    heads, nloc, ncw = b0.attn.num_heads, ws * ws, cw * cw
    nh, nw = sr
    G, S = ncw * nh * nw, ncw + nloc
    if S > 64 or G > 64:
        raise RuntimeError(f"hier stage: {S} tokens per window / {G} carrier tokens per image (the attention-core backward holds at most 64)")

    def partition(t):
        return t.reshape(B, C_, nh, ws, nw, ws).permute(0, 2, 4, 3, 5, 1).reshape(B * nh * nw, nloc, C_).contiguous()

    def reverse(r):
        return r.view(B, nh, nw, ws, ws, C_).permute(0, 5, 1, 3, 2, 4).reshape(B, C_, Hp, Wp).contiguous()

    rows, ct = partition(xp_leaf.detach()), ct_init.detach().float().contiguous()
    ckpt, ps = [], []
    for bi, blk in enumerate(blocks):
        P, t = _hier_params(blk, S, G)
        ckpt.append((rows, ct))
        ps.append((P, t))
        rows, ct = hier_block_forward(rows, ct, P["hat_attn"], P["hat_mlp"], P["attn"], P["mlp"], heads, ws, cw, sr, t["pe_x"].detach(),
                                      None if t["pe_ct"] is None else t["pe_ct"].detach(), 1e-5, operand_dtype,
                                      masks[bi] if masks is not None else None, _prop_gamma(blk))
    return reverse(rows)[:, :, :H, :W], ckpt, ps, dict(partition=partition, reverse=reverse, heads=heads, S=S, G=G, ws=ws, cw=cw, sr=sr, Hp=Hp, Wp=Wp,
                                                          x_leaf=x_leaf, ct_init=ct_init)


def hier_stage_backward(layer, x: torch.Tensor, dy: torch.Tensor, operand_dtype=torch.float16, sink: Optional[dict] = None, masks=None) -> torch.Tensor:
    """Backward of the transformer branch of a FasterViTLayer WITH carrier tokens (FasterViTLayer.forward FV:832-841 / AR:848-869: zero padding, TokenInitializer,
    window_partition, depth x hierarchical HAT block incl. the last block's carrier propagation (``do_propagation``: FasterViT-3 and up), window_reverse, crop):
    returns dx (B, C, H, W) and hands over the gradient of every parameter of ``layer.blocks`` and ``layer.global_tokenizer`` (``.grad`` / ``sink`` as in
    ``local_stage_backward``).  The HAT blocks run on the kernels (``hier_block_forward`` checkpoints per block, ``hier_block_backward``); the TokenInitializer
    (one depthwise conv + average pool, FV:704-738) and the four folded tables per block are small torch modules differentiated by autograd on the host side."""
    if not x.is_cuda:
        raise RuntimeError("hier_stage_backward runs only on a HIP device (libfvit_hip.so kernels); there is no CPU fallback")
    B, C_, H, W = x.shape
    blocks = list(layer.blocks)
    _, ckpt, ps, gm = _hier_stage_run(layer, x, operand_dtype, masks)
    heads, S, G, ws, cw, sr, Hp, Wp = gm["heads"], gm["S"], gm["G"], gm["ws"], gm["cw"], gm["sr"], gm["Hp"], gm["Wp"]
    dyp = torch.nn.functional.pad(dy.float(), (0, Wp - W, 0, Hp - H)) if (Hp != H or Wp != W) else dy.float()
    d, dct = gm["partition"](dyp), torch.zeros(B, G, C_, dtype=torch.float32, device=x.device)   # the stage's final carrier tokens are dropped (FV:841)
    dew, _ = _carrier_permutations(sr[0], sr[1], cw, x.device)
    for bi in range(len(blocks) - 1, -1, -1):
        blk, (xb, ctb), (P, t) = blocks[bi], ckpt[bi], ps[bi]
        hid = P["mlp"]["fc1_w"].shape[0]
        grads = dict(hat_attn=AttnGrads.zeros(C_, heads, G, x.device, with_gamma=P["hat_attn"]["gamma"] is not None),
                     hat_mlp=MlpGrads.zeros(C_, hid, x.device, with_gamma=P["hat_mlp"]["gamma"] is not None),
                     attn=AttnGrads.zeros(C_, heads, S, x.device, with_gamma=P["attn"]["gamma"] is not None),
                     mlp=MlpGrads.zeros(C_, hid, x.device, with_gamma=P["mlp"]["gamma"] is not None))
        pg = _prop_gamma(blk)
        prop_grad = torch.zeros(C_, dtype=torch.float32, device=x.device) if isinstance(pg, torch.Tensor) else None
        d, dct = hier_block_backward(xb, ctb, d, dct, P["hat_attn"], P["hat_mlp"], P["attn"], P["mlp"], heads, ws, cw, sr, t["pe_x"].detach(),
                                     None if t["pe_ct"] is None else t["pe_ct"].detach(), grads, 1e-5, operand_dtype,
                                     masks[bi] if masks is not None else None, pg, prop_grad)
        if prop_grad is not None:
            _emit(sink, blk.gamma1, prop_grad)
        for key, norm, at in (("hat_attn", blk.hat_norm1, blk.hat_attn), ("attn", blk.norm1, blk.attn)):
            gr = grads[key]
            for prm, g in ((norm.weight, gr.ln_w), (norm.bias, gr.ln_b), (at.qkv.weight, gr.qkv_w), (at.qkv.bias, gr.qkv_b), (at.proj.weight, gr.proj_w),
                           (at.proj.bias, gr.proj_b), (blk.gamma1 if key == "hat_attn" else blk.gamma3, gr.gamma)):
                if g is not None:
                    _emit(sink, prm, g)
        for key, norm, ml in (("hat_mlp", blk.hat_norm2, blk.hat_mlp), ("mlp", blk.norm2, blk.mlp)):
            gr = grads[key]
            for prm, g in ((norm.weight, gr.ln_w), (norm.bias, gr.ln_b), (ml.fc1.weight, gr.fc1_w), (ml.fc1.bias, gr.fc1_b), (ml.fc2.weight, gr.fc2_w),
                           (ml.fc2.bias, gr.fc2_b), (blk.gamma2 if key == "hat_mlp" else blk.gamma4, gr.gamma)):
                if g is not None:
                    _emit(sink, prm, g)
        # folded tables -> their small MLPs (host autograd): d pe_x = sum over windows of dx, d pe_ct = sum over images of the dewindowed carrier gradient
        outs, gouts = [t["bias"], t["hat_bias"], t["pe_x"]], [grads["attn"].bias, grads["hat_attn"].bias, d.sum(0)]
        if t["pe_ct"] is not None:
            outs.append(t["pe_ct"])
            gouts.append(dct[:, dew].sum(0))
        _table_grads(sink, outs, gouts, [blk.attn.pos_emb_funct, blk.hat_attn.pos_emb_funct, blk.pos_embed, getattr(blk, "hat_pos_embed", None)])
    # ---- the carrier tokens came from the tokenizer: its conv parameters and its share of dx (autograd.grad: no .grad is touched) ----
    x_leaf, ct_init = gm["x_leaf"], gm["ct_init"]
    tok_params = _module_params([layer.global_tokenizer])
    gs = torch.autograd.grad([ct_init], tok_params + [x_leaf], [dct.to(ct_init.dtype)], allow_unused=True)
    for prm, g in zip(tok_params, gs[:-1]):
        if g is not None:
            _emit(sink, prm, g)
    dx_tok = gs[-1]
    dx = gm["reverse"](d)[:, :, :H, :W]
    return (dx + dx_tok if dx_tok is not None else dx).contiguous()


def stage_forward_train(layer, x: torch.Tensor, operand_dtype=torch.float16, masks=None) -> torch.Tensor:
    """The transformer branch of a FasterViTLayer through the unit kernels with stochastic depth (``masks`` = ``drop_path_masks``): the TRAIN-mode forward
    (the fused inference kernels implement eval semantics).  Same checkpoints as the backward's recompute."""
    fn = _hier_stage_run if layer.blocks[0].do_sr_hat else _local_stage_run
    return fn(layer, x, operand_dtype, masks)[0].to(x.dtype).contiguous()


# --------------------------------------------------------------------------------------------------------------------------------------
# autograd bridge: the transformer branch of a FasterViTLayer as ONE autograd node (forward: fvit_hat_stage_forward, backward: the functions above)
# --------------------------------------------------------------------------------------------------------------------------------------
def _stage_params(layer):
    seen, out = set(), []
    mods = [layer.blocks] + ([layer.global_tokenizer] if getattr(layer, "do_gt", False) and hasattr(layer, "global_tokenizer") else [])
    for m in mods:
        for prm in m.parameters():
            if id(prm) not in seen:
                seen.add(id(prm))
                out.append(prm)
    return out


def operand_torch_dtype(layer) -> torch.dtype:
    """16-bit operand type of the backward kernels for this layer: the type of its forward operand mode (``hat_operand_dtype``; the two- and
    three-term modes recompute and differentiate with single 16-bit terms of the same type)."""
    return torch.bfloat16 if str(getattr(layer, "hat_operand_dtype", "f16")).startswith("bf16") else torch.float16


def backward_unsupported_reason(layer, H: Optional[int] = None, W: Optional[int] = None) -> Optional[str]:
    """None if the kernel-sequence backward (and train-mode forward) covers this stage (and, when given, this map size); otherwise the reason, as text.
    Checked at FORWARD time (``stage_forward_with_grad``) and by ``FasterViT.enable_hat_backward``: a stage that cannot be differentiated must not fail
    from inside ``loss.backward()`` on the autograd engine's thread after a forward that succeeded.
    Covered: head_dim <= 96 (run padded to 32 / 64 / 96), C a multiple of 16, hidden a multiple of 64, windows and carrier grids of at most 64 tokens,
    maps padded up to a multiple of the window, the last block's carrier propagation -- i.e. every reference entrypoint at its native 224 x 224
    resolution; not the 384+ / any-res geometries with longer windows."""
    blocks = list(layer.blocks)
    if not blocks:
        return None
    b0 = blocks[0]
    C_, heads = b0.attn.qkv.in_features, b0.attn.num_heads
    hid = b0.mlp.fc1.out_features
    ws = b0.window_size
    hier = bool(b0.do_sr_hat)
    if C_ % heads or C_ // heads > 96:
        return f"head_dim {C_ // max(heads, 1)} (the attention kernels cover head_dim <= 96)"
    if C_ % 16 or hid % 64:
        return f"C = {C_} must be a multiple of 16 and hidden = {hid} a multiple of 64"
    if any(bool(b.do_sr_hat) != hier for b in blocks):
        return "mixed hierarchical / local blocks in one stage"
    ncw = b0.cr_window ** 2 if hier else 0
    if ws * ws + ncw > 64:
        return f"windows of {ws * ws + ncw} tokens (the attention-core backward holds at most 64 in LDS)"
    if hier:
        sr = tuple(b0.sr_ratio)
        if ncw * sr[0] * sr[1] > 64:
            return f"{ncw * sr[0] * sr[1]} carrier tokens per image (at most 64)"
        if H is not None and (-(-H // ws), -(-W // ws)) != sr:
            return f"map {H}x{W} does not pad into the stage's {sr[0]}x{sr[1]} windows of {ws}"
    return None


class HatStageFunction(torch.autograd.Function):
    """y = FasterViTLayer transformer branch (x) with the module's parameters as differentiable inputs.
    Forward: eval mode = the HIP inference path (``fvit_hat_stage_forward``, fused kernels); TRAIN mode = the unit-kernel chain with stochastic depth
    (``stage_forward_train``; the DropPath draws are kept for the backward).  Backward = local_stage_backward / hier_stage_backward, whose parameter
    gradients are collected in a dict and RETURNED to autograd (so hooks such as DistributedDataParallel's fire once, from the outer engine); neither
    ``.grad`` nor any AccumulateGrad node is touched from inside this backward (tables and tokenizer: ``torch.autograd.grad``).

    fp16 operands: every gradient inside the kernel sequence (dz, dO, dqkv, dh, da and their transposes) is narrowed to 16 bits.  The backward is linear
    in dy, so dy is scaled by a power of two that brings its largest entry to ~2^10 and dx / the parameter gradients are scaled back in fp32: a mean-reduced
    loss (dy ~ 1e-6) or a 1e-5 layer scale would otherwise sit in or below the fp16 subnormals (spacing 6e-8).  bf16 operands need no scaling."""

    @staticmethod
    def forward(ctx, x, layer, operand_dtype, *params):
        from . import hat_runtime
        ctx.layer, ctx.operand_dtype, ctx.params = layer, operand_dtype, params
        ctx.save_for_backward(x)
        ctx.masks = None
        if layer.training:
            b0 = layer.blocks[0]
            ws = b0.window_size
            nW = (-(-x.shape[2] // ws)) * (-(-x.shape[3] // ws))
            ctx.masks = drop_path_masks(layer, x.shape[0], nW, x.device, operand_dtype=operand_dtype)
            return stage_forward_train(layer, x.detach(), operand_dtype, ctx.masks)
        return hat_runtime.stage_forward(layer, x.detach())

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        layer, params = ctx.layer, ctx.params
        dy32 = dy.float().contiguous()
        inv = None
        if ctx.operand_dtype == torch.float16:
            amax = dy32.abs().amax().clamp_min(1e-30)
            scale = torch.exp2(torch.floor(torch.log2(1024.0 / amax)))   # 0-dim tensor on the device: no host synchronisation
            dy32 = dy32 * scale
            inv = 1.0 / scale
        sink: dict = {}
        fn = hier_stage_backward if layer.blocks[0].do_sr_hat else local_stage_backward
        dx = fn(layer, x.float().contiguous(), dy32, ctx.operand_dtype, sink=sink, masks=ctx.masks)
        if inv is not None:
            dx = dx * inv
        grads = []
        for prm in params:
            g = sink.get(prm)
            if g is not None and inv is not None:
                g = g * inv
            grads.append(None if g is None else g.to(prm.dtype))
        return (dx.to(x.dtype), None, None, *grads)


def stage_forward_with_grad(layer, x: torch.Tensor, operand_dtype=None) -> torch.Tensor:
    """``hat_runtime.stage_forward`` as a differentiable op (see HatStageFunction); in TRAIN mode the forward has train semantics (stochastic depth).
    ``operand_dtype`` None = the 16-bit type of the layer's forward operand mode.  Unsupported geometries raise HERE, at forward time."""
    if len(layer.blocks) == 0:
        return x
    why = backward_unsupported_reason(layer, x.shape[2], x.shape[3])
    if why is not None:
        raise RuntimeError(f"stage_forward_with_grad: this HAT stage has no kernel-sequence backward / train-mode forward: {why}. Run it forward-only in eval "
                           "mode under torch.no_grad() (model.enable_hat_backward(False)).")
    if operand_dtype is None:
        operand_dtype = operand_torch_dtype(layer)
    return HatStageFunction.apply(x, layer, operand_dtype, *_stage_params(layer))
