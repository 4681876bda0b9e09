"""CPU oracle for the FasterViT Hierarchical-Attention (HAT) hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``fastervit_amd/`` may import this module; it is used by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` as the checker
the HIP path is compared against, never as the thing measured or shipped.

It is a *functional* restatement (plain functions over a flat ``state_dict``; no nn.Module tree) of
the reference algorithm, written against the any-resolution variant of the reference, which is
bit-identical to the base file at square resolutions.  Each function cites the
reference lines it follows; ``AR:`` = /root/reference/fastervit/models/faster_vit_any_res.py,
``FV:`` = /root/reference/fastervit/models/faster_vit.py.

Parity pinning: this oracle is checked in ``tests/test_oracle_golden.py`` against golden vectors
produced by importing the *real* reference in the build container (``tests/golden/make_golden.py``,
vectors committed under ``tests/golden/*.npz``).  The reference ships no golden vectors of its own
(SURVEY.md §4), so those generated fixtures are the pin.

All arithmetic is floating point; ``dtype`` selects float32 (what the reference CPU path runs) or
float64 (a tighter truth for tolerance studies).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------------------------
# a-1 .. a-4: layout functions
# --------------------------------------------------------------------------------------------
def window_partition(x: Tensor, ws: int) -> Tensor:
    """AR:84-88 / FV:83-87.  NCHW -> (B*nW, ws*ws, C); window (wh, ww), token (ih, iw), C fastest."""
    B, C, H, W = x.shape
    t = x.reshape(B, C, H // ws, ws, W // ws, ws)
    return t.permute(0, 2, 4, 3, 5, 1).reshape(-1, ws * ws, C)


def window_reverse(win: Tensor, ws: int, H: int, W: int, B: int) -> Tensor:
    """AR:91-94 / FV:90-93.  Inverse of window_partition."""
    C = win.shape[2]
    t = win.reshape(B, H // ws, W // ws, ws, ws, C)
    return t.permute(0, 5, 1, 3, 2, 4).reshape(B, C, H, W)


def ct_dewindow(ct: Tensor, W: int, H: int, cw: int) -> Tensor:
    """AR:97-102 / FV:96-101.  Per-window carrier order -> raster over a (W x H) grid.

    Note the argument names: the first extent is the *row* extent (callers pass cw*sr[0] first).
    """
    bs, _, N = ct.shape
    t = ct.reshape(-1, W // cw, H // cw, cw, cw, N).permute(0, 5, 1, 3, 2, 4)
    return t.reshape(bs, N, W * H).transpose(1, 2)


def ct_window(ct: Tensor, W: int, H: int, cw: int) -> Tensor:
    """AR:105-110 / FV:104-109.  Treats the raster as (H x W) -- NOT the inverse of ct_dewindow on
    non-square grids; reproduced as is."""
    bs, _, N = ct.shape
    t = ct.reshape(bs, H // cw, cw, W // cw, cw, N)
    return t.permute(0, 1, 3, 2, 4, 5)


# --------------------------------------------------------------------------------------------
# a-6, a-7: position terms (input independent)
# --------------------------------------------------------------------------------------------
def pos_embed_1d(sd: SD, prefix: str, seq_length: int, dtype) -> Tensor:
    """PosEmbMLPSwinv1D, rank 2, conv False (AR:340-368 / FV:339-367).  Returns (1, L*L, C)."""
    L = int(seq_length ** 0.5)
    ch = torch.arange(0, L, dtype=dtype)
    cwd = torch.arange(0, L, dtype=dtype)
    table = torch.stack(torch.meshgrid([ch, cwd], indexing="ij")).contiguous().unsqueeze(0)
    table = table - (L // 2)
    table = table / (L // 2)
    t = table.flatten(2).transpose(1, 2)                      # (1, L*L, 2)
    w0 = sd[prefix + "cpb_mlp.0.weight"].to(dtype)
    b0 = sd[prefix + "cpb_mlp.0.bias"].to(dtype)
    w2 = sd[prefix + "cpb_mlp.2.weight"].to(dtype)
    return F.linear(torch.relu(F.linear(t, w0, b0)), w2)


def rel_coords_table(w0: int, w1: int, dtype) -> Tensor:
    """Log-spaced relative coordinate table (AR:227-244 / FV:226-243), pretrained window = window."""
    rh = torch.arange(-(w0 - 1), w0, dtype=torch.float32)
    rw = torch.arange(-(w1 - 1), w1, dtype=torch.float32)
    t = torch.stack(torch.meshgrid([rh, rw], indexing="ij")).permute(1, 2, 0).contiguous().unsqueeze(0)
    t[:, :, :, 0] /= (w0 - 1)
    t[:, :, :, 1] /= (w1 - 1)
    t *= 8
    t = torch.sign(t) * torch.log2(torch.abs(t) + 1.0) / math.log2(8)
    return t.to(dtype)


def rel_position_index(w0: int, w1: int) -> Tensor:
    """AR:245-255 / FV:244-254."""
    coords = torch.stack(torch.meshgrid([torch.arange(w0), torch.arange(w1)], indexing="ij"))
    cf = torch.flatten(coords, 1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += w0 - 1
    rel[:, :, 1] += w1 - 1
    rel[:, :, 0] *= 2 * w1 - 1
    return rel.sum(-1)


def attn_bias(sd: SD, prefix: str, res: int, heads: int, S: int, dtype) -> Tensor:
    """PosEmbMLPSwinv2D non-deploy path, ct_correct False (AR:267-311 / FV:266-310).

    Returns (heads, S, S): 16*sigmoid(cpb_mlp(table))[index] on the trailing res^2 x res^2 block,
    zero on the first S - res^2 rows/cols (top/left zero pad)."""
    table = rel_coords_table(res, res, dtype)
    idx = rel_position_index(res, res)
    w0 = sd[prefix + "cpb_mlp.0.weight"].to(dtype)
    b0 = sd[prefix + "cpb_mlp.0.bias"].to(dtype)
    w2 = sd[prefix + "cpb_mlp.2.weight"].to(dtype)
    tab = F.linear(torch.relu(F.linear(table, w0, b0)), w2).view(-1, heads)
    bias = tab[idx.view(-1)].view(res * res, res * res, -1).permute(2, 0, 1).contiguous()
    bias = 16 * torch.sigmoid(bias)
    n_g = S - res * res
    return F.pad(bias, (n_g, 0, n_g, 0)).contiguous()


# --------------------------------------------------------------------------------------------
# a-8, a-9, a-10
# --------------------------------------------------------------------------------------------
def window_attention(x: Tensor, sd: SD, prefix: str, heads: int, res: int, qk_scale: Optional[float] = None) -> Tensor:
    """WindowAttention.forward (AR:558-569 / FV:557-568).  x: (Bw, S, C).  scale = qk_scale or head_dim ** -0.5 (FV:538)."""
    dtype = x.dtype
    Bw, S, C = x.shape
    d = C // heads
    qkv = F.linear(x, sd[prefix + "qkv.weight"].to(dtype), sd[prefix + "qkv.bias"].to(dtype))
    qkv = qkv.reshape(Bw, -1, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (qk_scale or d ** -0.5)
    attn = attn + attn_bias(sd, prefix + "pos_emb_funct.", res, heads, S, dtype).unsqueeze(0)
    attn = attn.softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(Bw, -1, C)
    return F.linear(out, sd[prefix + "proj.weight"].to(dtype), sd[prefix + "proj.bias"].to(dtype))


def mlp(x: Tensor, sd: SD, prefix: str) -> Tensor:
    """Mlp.forward (AR:399-408 / FV:398-407): fc2(GELU_erf(fc1(x)))."""
    dtype = x.dtype
    h = F.linear(x, sd[prefix + "fc1.weight"].to(dtype), sd[prefix + "fc1.bias"].to(dtype))
    h = F.gelu(h)
    return F.linear(h, sd[prefix + "fc2.weight"].to(dtype), sd[prefix + "fc2.bias"].to(dtype))


def layer_norm(x: Tensor, sd: SD, prefix: str, eps: float = 1e-5) -> Tensor:
    dtype = x.dtype
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + "weight"].to(dtype), sd[prefix + "bias"].to(dtype), eps)


def _gamma(sd: SD, key: str, dtype):
    """gamma is the Python int 1 when layer_scale is None, else a (C,) parameter (AR:641-643,660-661)."""
    return sd[key].to(dtype) if key in sd else 1


# --------------------------------------------------------------------------------------------
# a-11: HAT.forward
# --------------------------------------------------------------------------------------------
def hat_block(x: Tensor, ct: Optional[Tensor], sd: SD, prefix: str, *, heads: int, ws: int, cw: int,
              sr: Tuple[int, int], last: bool, do_propagation: bool, qk_scale: Optional[float] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """HAT.forward (AR:668-707 / FV:662-701).  x: (B*nW, ws^2, C); ct: (B, G, C) or None."""
    dtype = x.dtype
    Bw, T, C = x.shape
    do_sr_hat = sr[0] > 1 or sr[1] > 1
    square = sr[0] == sr[1]
    x = x + pos_embed_1d(sd, prefix + "pos_embed.", ws * ws, dtype)
    if do_sr_hat:
        Bg, Ng, Hg = ct.shape
        ct = ct_dewindow(ct, cw * sr[0], cw * sr[1], cw)
        if square:
            ct = ct + pos_embed_1d(sd, prefix + "hat_pos_embed.", Ng, dtype)
        g1 = _gamma(sd, prefix + "gamma1", dtype)
        g2 = _gamma(sd, prefix + "gamma2", dtype)
        res_ct = int((cw * cw * sr[0] * sr[1]) ** 0.5)
        ct = ct + g1 * window_attention(layer_norm(ct, sd, prefix + "hat_norm1."), sd, prefix + "hat_attn.", heads, res_ct, qk_scale)
        ct = ct + g2 * mlp(layer_norm(ct, sd, prefix + "hat_norm2."), sd, prefix + "hat_mlp.")
        ct = ct_window(ct, cw * sr[0], cw * sr[1], cw)
        ct = ct.reshape(Bw, -1, C)
        x = torch.cat((ct, x), dim=1)
    g3 = _gamma(sd, prefix + "gamma3", dtype)
    g4 = _gamma(sd, prefix + "gamma4", dtype)
    x = x + g3 * window_attention(layer_norm(x, sd, prefix + "norm1."), sd, prefix + "attn.", heads, ws, qk_scale)
    x = x + g4 * mlp(layer_norm(x, sd, prefix + "norm2."), sd, prefix + "mlp.")
    if do_sr_hat:
        ctr, x = x.split([x.shape[1] - ws * ws, ws * ws], dim=1)
        ct = ctr.reshape(Bg, Ng, Hg)
        if last and do_propagation:
            img = ctr.transpose(1, 2).reshape(Bw, C, cw, cw)
            up = F.interpolate(img, size=(ws, ws), mode="nearest")
            x = x + g1 * up.flatten(2).transpose(1, 2)
    return x, ct


# --------------------------------------------------------------------------------------------
# a-5: TokenInitializer, a-12: FasterViTLayer transformer branch
# --------------------------------------------------------------------------------------------
def token_initializer(x: Tensor, sd: SD, prefix: str, res: Tuple[int, int], ws: int, cw: int) -> Tensor:
    """TokenInitializer (AR:715-750 / FV:709-738).  x: padded NCHW map.  Returns (B, G, C)."""
    dtype = x.dtype
    C = x.shape[1]
    ks, ss = [], []
    for r in res:
        out = int(cw * r / ws)
        s = int(r / out)
        ks.append(r - (out - 1) * s)
        ss.append(s)
    y = F.conv2d(x, sd[prefix + "pos_embed.weight"].to(dtype), sd[prefix + "pos_embed.bias"].to(dtype), padding=1, groups=C)
    y = F.avg_pool2d(y, kernel_size=tuple(ks), stride=tuple(ss))
    B, C, H, W = y.shape
    t = y.reshape(B, C, H // cw, cw, W // cw, cw)
    return t.permute(0, 2, 4, 3, 5, 1).reshape(-1, H * W, C)


def padded_resolution(res: Sequence[int], ws: int) -> Tuple[int, int]:
    """AR:806-808."""
    return tuple(r + (ws - r % ws) % ws for r in res)


def hat_stage(x: Tensor, sd: SD, prefix: str, *, depth: int, heads: int, ws: int, cw: int,
              input_resolution: Sequence[int], only_local: bool, do_propagation: bool,
              any_res: bool = True, capture: Optional[list] = None, qk_scale: Optional[float] = None) -> Tensor:
    """Transformer branch of FasterViTLayer.forward WITHOUT the Downsample (AR:848-869 / FV:832-841).

    ``any_res`` False applies the base-file rule for the tokenizer (FV:821: needs
    input_resolution // window_size > 1); True applies AR:837 (truthy list => always when hierarchical).
    ``capture`` (optional list) receives (x, ct) after every block, for per-block parity tests.
    """
    B, C, H, W = x.shape
    Hr, Wr = padded_resolution(input_resolution, ws)
    sr = (1, 1) if only_local else (Hr // ws, Wr // ws)
    if any_res:
        do_gt = depth > 0 and not only_local
    else:
        do_gt = depth > 0 and not only_local and (input_resolution[0] // ws > 1)
    pad_r = (ws - W % ws) % ws
    pad_b = (ws - H % ws) % ws
    if pad_r > 0 or pad_b > 0:
        x = F.pad(x, (0, pad_r, 0, pad_b))
    Hp, Wp = x.shape[2], x.shape[3]
    ct = token_initializer(x, sd, prefix + "global_tokenizer.", (Hr, Wr), ws, cw) if do_gt else None
    x = window_partition(x, ws)
    for i in range(depth):
        x, ct = hat_block(x, ct, sd, f"{prefix}blocks.{i}.", heads=heads, ws=ws, cw=cw, sr=sr,
                          last=(i == depth - 1), do_propagation=do_propagation, qk_scale=qk_scale)
        if capture is not None:
            capture.append((x.clone(), None if ct is None else ct.clone()))
    x = window_reverse(x, ws, Hp, Wp, B)
    if pad_r > 0 or pad_b > 0:
        x = x[:, :, :H, :W].contiguous()
    return x
