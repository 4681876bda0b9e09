"""Host side of the HAT hot path: packs a FasterViTLayer's parameters into the layout the gfx950
kernels consume, folds the input-independent position terms, builds the index tables and calls the
C ABI (include/fvit_hip.h) on the caller's HIP stream.

PyTorch is used here for device memory, streams and the few input-independent tensor ops of the
packing step; the per-forward arithmetic of the stage is entirely inside libfvit_hip.so.  There is
no CPU / eager fallback: a non-GPU tensor or a missing library raises RuntimeError.

Constant folding (done once per weight version):
  * PosEmbMLPSwinv1D (AR:340-368)  -> pe_x (ws^2, C), pe_ct (G, C) fp32 tables
  * PosEmbMLPSwinv2D (AR:267-311)  -> bias (h, Spad, Spad) fp32, zero on carrier rows/cols,
                                       FVIT_MASK_BIAS on padded key columns
  * ct_dewindow / ct_window / torch.cat / nn.Upsample(nearest) (AR:97-110, 693, 662)
                                    -> int32 row-gather tables
  * qkv / proj weights are re-laid out per head with head_dim padded to 32 or 64, so every MFMA
    fragment load is 16-byte aligned also for head_dim 49 (FasterViT-4).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import (FVIT_BF16, FVIT_F16, FVIT_F32, FVIT_MASK_BIAS, FVIT_TILE_K, FVIT_TILE_N, FvitAttnWeights,
                   FvitBlockWeights, FvitMapView, FvitMlpWeights, FvitStageDesc, FvitStageTables)

_DT = {torch.float32: FVIT_F32, torch.float16: FVIT_F16, torch.bfloat16: FVIT_BF16}
_OP = {"f16": (FVIT_F16, torch.float16), "bf16": (FVIT_BF16, torch.bfloat16)}


def _rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: the FasterViT HAT path runs only on a HIP device (got a {t.device.type} tensor); "
            "there is no CPU or eager fallback. Move the model and input to the GPU.")


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _map_view(t: torch.Tensor) -> FvitMapView:
    if t.dtype not in _DT:
        raise RuntimeError(f"unsupported feature-map dtype {t.dtype}")
    sb, sc, sh, sw = t.stride()
    return FvitMapView(t.data_ptr(), sb, sc, sh, sw, _DT[t.dtype], 0)


# --------------------------------------------------------------------------------------------
# index tables
# --------------------------------------------------------------------------------------------
def build_tables(sr0: int, sr1: int, ws: int, cw: int, hier: bool):
    """Row-gather tables for one geometry (CPU int32 tensors).

    Derived by running the reference's view/permute chains on aranges: ct_dewindow
    (AR:97-102: view(-1, W/cw, H/cw, cw, cw, N).permute(0,5,1,3,2,4)) and ct_window
    (AR:105-110: view(bs, H/cw, cw, W/cw, cw, N).permute(0,1,3,2,4,5)) with W = cw*sr0,
    H = cw*sr1 -- the two are inverses only on square grids, which is reproduced here."""
    nW, nloc = sr0 * sr1, ws * ws
    ncw = cw * cw if hier else 0
    S, G = nloc + ncw, ncw * nW
    ln1_src = torch.arange(nW * S, dtype=torch.int64).view(nW, S).clone()
    ln1_add = torch.full((nW, S), -1, dtype=torch.int64)
    ln1_add[:, ncw:] = torch.arange(nloc)
    ct_src = torch.zeros(max(G, 1), dtype=torch.int64)
    up_idx = torch.zeros(nloc, dtype=torch.int64)
    if hier:
        ar = torch.arange(G)
        dew = ar.view(sr0, sr1, cw, cw).permute(0, 2, 1, 3).reshape(G)   # raster r  -> windowed p
        win = ar.view(sr1, cw, sr0, cw).permute(0, 2, 1, 3).reshape(G)   # windowed p -> raster r
        ct_src = (dew // ncw) * S + dew % ncw                             # X row of windowed p
        ln1_src[:, :ncw] = -(win.view(nW, ncw) + 1)
        near = (torch.arange(ws) * cw) // ws                              # nn.Upsample(size=ws, 'nearest')
        up_idx = (near[:, None] * cw + near[None, :]).reshape(nloc)
    i32 = torch.int32
    return dict(ln1_src=ln1_src.reshape(-1).to(i32), ln1_add=ln1_add.reshape(-1).to(i32), ct_src=ct_src.to(i32),
                up_idx=up_idx.to(i32), nW=nW, S=S, G=G, ncw=ncw)


# --------------------------------------------------------------------------------------------
# weight packing
# --------------------------------------------------------------------------------------------
def _f32(t) -> torch.Tensor:
    return t.detach().float().contiguous()


def _gamma(g) -> Optional[torch.Tensor]:
    return _f32(g) if isinstance(g, torch.Tensor) else None


class _Keep:
    """Holds packed device tensors alive and hands out their pointers (dtype / layout checked:
    the kernels reinterpret raw pointers, a silently down-cast table would be read out of bounds)."""

    def __init__(self, op_dtype=None):
        self.tensors = []
        self.op_dtype = op_dtype

    def ptr(self, t: Optional[torch.Tensor], op16: bool = False) -> Optional[int]:
        if t is None:
            return None
        want = self.op_dtype if op16 else torch.float32
        if t.dtype != want or not t.is_contiguous():
            raise RuntimeError(f"packed tensor has dtype {t.dtype} / contiguous={t.is_contiguous()}, expected {want}")
        self.tensors.append(t)
        return t.data_ptr()


def pack_attention(attn, norm, gamma, S: int, dpad: int, op_dtype, keep: _Keep) -> FvitAttnWeights:
    lib = _lib.lib()
    C_ = attn.qkv.in_features
    h = attn.num_heads
    d = C_ // h
    dev = attn.qkv.weight.device
    ldn = _rup(C_, FVIT_TILE_K)
    nq = 3 * h * dpad
    wq = torch.zeros(_rup(nq, FVIT_TILE_N), ldn, device=dev, dtype=torch.float32)
    wq[:nq].view(3, h, dpad, ldn)[:, :, :d, :C_] = _f32(attn.qkv.weight).view(3, h, d, C_)
    bq = torch.zeros(nq, device=dev, dtype=torch.float32)
    if attn.qkv.bias is not None:
        bq.view(3, h, dpad)[:, :, :d] = _f32(attn.qkv.bias).view(3, h, d)
    ldao = _rup(h * dpad, FVIT_TILE_K)
    wp = torch.zeros(_rup(C_, FVIT_TILE_N), ldao, device=dev, dtype=torch.float32)
    wp[:C_, :h * dpad].view(C_, h, dpad)[:, :, :d] = _f32(attn.proj.weight).view(C_, h, d)
    bias = rel = None
    rel_w = rel_ng = 0
    if lib.fvit_attention_dense(S, dpad):
        spad = lib.fvit_attention_spad(S)
        bias = torch.zeros(h, spad, spad, device=dev, dtype=torch.float32)
        bias[:, :S, :S] = attn.pos_emb_funct.table(S)
        bias[:, :, S:] = FVIT_MASK_BIAS
        bias[:, S:, :] = 0.0
        bias[:, S:, S:] = FVIT_MASK_BIAS if S < spad else 0.0
    else:
        # long windows (21k 384/512/768 fine-tunes, large carrier grids): the dense table would be heads*S*S floats per block;
        # the kernel evaluates relative_position_index arithmetically on the un-gathered (heads, (2w-1)^2) table instead
        rel, rel_w = attn.pos_emb_funct.rel_table()
        rel_ng = S - rel_w * rel_w
        if rel_ng < 0:
            raise RuntimeError(f"bias window {rel_w}x{rel_w} larger than the sequence ({S} tokens)")
    wqf = bqh = wpf = None
    if d == 32 and lib.fvit_attn_block_supported(C_, h, S):
        wqkv32 = _f32(attn.qkv.weight)
        bqkv32 = _f32(attn.qkv.bias) if attn.qkv.bias is not None else torch.zeros(3 * C_, device=dev)
        wqf = frag_pack_qkv(wqkv32, h).to(op_dtype)
        bqh = bqkv32.view(3, h, 32).permute(1, 0, 2).reshape(h, 96).contiguous()
        wpf = frag_pack_fc2(_f32(attn.proj.weight)).to(op_dtype)   # chunks of 32 input columns = heads
    return FvitAttnWeights(keep.ptr(wq.to(op_dtype), True), keep.ptr(bq), keep.ptr(wp.to(op_dtype), True), keep.ptr(_f32(attn.proj.bias)),
                           keep.ptr(bias), keep.ptr(_f32(norm.weight)), keep.ptr(_f32(norm.bias)), keep.ptr(_gamma(gamma)),
                           keep.ptr(wqf, True), keep.ptr(bqh), keep.ptr(wpf, True), keep.ptr(rel), rel_w, rel_ng)


def frag_pack_qkv(wqkv: torch.Tensor, heads: int) -> torch.Tensor:
    """qkv.weight (3C, C), head_dim 32 -> [heads][6][C/32][64][8] (include/fvit_hip.h: w_qkv_frag): element e of lane 16g + s of
    fragment (head, ub, kk) = wqkv[(ub>>1)*C + head*32 + (ub&1)*16 + s][kch(kk, g, e)] (kslot_channels)."""
    C3, C_ = wqkv.shape
    wqkv = wqkv[:, kslot_channels(C_, wqkv.device)]
    t = wqkv.view(3, heads, 2, 16, C_ // 32, 4, 8)          # sec, head, half, s, kk, g, e
    t = t.permute(1, 0, 2, 4, 5, 3, 6).contiguous()          # head, sec, half, kk, g, s, e
    return t.view(heads, 6, C_ // 32, 64, 8)


def pack_mlp(mlp, norm, gamma, op_dtype, keep: _Keep) -> FvitMlpWeights:
    C_ = mlp.fc1.in_features
    hid = mlp.fc1.out_features
    dev = mlp.fc1.weight.device
    ldn, ldh = _rup(C_, FVIT_TILE_K), _rup(hid, FVIT_TILE_K)
    w1 = torch.zeros(_rup(hid, FVIT_TILE_N), ldn, device=dev, dtype=torch.float32)
    w1[:hid, :C_] = _f32(mlp.fc1.weight)
    w2 = torch.zeros(_rup(C_, FVIT_TILE_N), ldh, device=dev, dtype=torch.float32)
    w2[:C_, :hid] = _f32(mlp.fc2.weight)
    w1f = w2f = None
    if _lib.lib().fvit_mlp_fused_supported(C_, hid):
        w1f = frag_pack_fc1(_f32(mlp.fc1.weight)).to(op_dtype)
        w2f = frag_pack_fc2(_f32(mlp.fc2.weight)).to(op_dtype)
    return FvitMlpWeights(keep.ptr(w1.to(op_dtype), True), keep.ptr(_f32(mlp.fc1.bias)), keep.ptr(w2.to(op_dtype), True),
                          keep.ptr(_f32(mlp.fc2.bias)), keep.ptr(_f32(norm.weight)), keep.ptr(_f32(norm.bias)),
                          keep.ptr(_gamma(gamma)), keep.ptr(w1f, True), keep.ptr(w2f, True))


def kslot_channels(C_: int, device=None) -> torch.Tensor:
    """Input channel of GEMM k slot kk*32 + 8g + e in the fused kernels: kch = (kk>>1)*64 + g*16 + (kk&1)*8 + e.  With this order the
    64 input values a lane loads per row are the 64 output channels it owns in the epilogue, so the residual needs no re-read of X."""
    k = torch.arange(C_, device=device)
    kk, g, e = k >> 5, (k >> 3) & 3, k & 7
    return (kk >> 1) * 64 + g * 16 + (kk & 1) * 8 + e


def frag_pack_fc1(w1: torch.Tensor) -> torch.Tensor:
    """fc1.weight (hidden, C) -> [hidden/32][2][C/32][64][8] in MFMA A-fragment order (include/fvit_hip.h: w_fc1_frag):
    element e of lane 16g + s of fragment (j, hb, kk) = w1[j*32 + hb*16 + s][kch(kk, g, e)] (kslot_channels)."""
    hid, C_ = w1.shape
    w1 = w1[:, kslot_channels(C_, w1.device)]
    t = w1.view(hid // 32, 2, 16, C_ // 32, 4, 8)       # j, hb, s, kk, g, e
    return t.permute(0, 1, 3, 4, 2, 5).contiguous()      # j, hb, kk, g, s, e


def frag_pack_fc2(w2: torch.Tensor) -> torch.Tensor:
    """fc2.weight (C, hidden) -> [hidden/32][C/16][64][8] (include/fvit_hip.h: w_fc2_frag): element e of lane 16g + s of
    fragment (j, cb) = w2[ch(cb, s)][j*32 + (e>>2)*16 + 4g + (e&3)], ch(cb, s) = (cb>>2)*64 + (s>>2)*16 + (cb&3)*4 + (s&3)."""
    C_, hid = w2.shape
    dev = w2.device
    cb = torch.arange(C_ // 16, device=dev).view(-1, 1)
    s = torch.arange(16, device=dev).view(1, -1)
    ch = ((cb >> 2) * 64 + (s >> 2) * 16 + (cb & 3) * 4 + (s & 3)).reshape(-1)          # (CB*16,)
    g = torch.arange(4, device=dev).view(-1, 1)
    e = torch.arange(8, device=dev).view(1, -1)
    col = ((e >> 2) * 16 + 4 * g + (e & 3)).reshape(-1)                                   # (32,) indexed by 8g + e
    t = w2.view(C_, hid // 32, 32)[ch][:, :, col]                                          # (CB*16, nch, 32)
    t = t.view(C_ // 16, 16, hid // 32, 4, 8)                                              # cb, s, j, g, e
    return t.permute(2, 0, 3, 1, 4).contiguous()                                           # j, cb, g, s, e


def pack_block(blk, S: int, G: int, dpad: int, op_dtype, keep: _Keep) -> FvitBlockWeights:
    w = FvitBlockWeights()
    w.attn = pack_attention(blk.attn, blk.norm1, blk.gamma3, S, dpad, op_dtype, keep)
    w.mlp = pack_mlp(blk.mlp, blk.norm2, blk.gamma4, op_dtype, keep)
    w.pe_x = keep.ptr(blk.pos_embed.table(blk.window_size ** 2))
    w.pe_ct = None
    if blk.do_sr_hat:
        w.hat_attn = pack_attention(blk.hat_attn, blk.hat_norm1, blk.gamma1, G, dpad, op_dtype, keep)
        w.hat_mlp = pack_mlp(blk.hat_mlp, blk.hat_norm2, blk.gamma2, op_dtype, keep)
        if hasattr(blk, "hat_pos_embed"):
            w.pe_ct = keep.ptr(blk.hat_pos_embed.table(G))
    w.last = 1 if blk.last else 0
    return w


def _signature(blocks, device, op_name):
    sig = [str(device), op_name]
    for blk in blocks:
        for p in blk.parameters():
            sig.append((p.data_ptr(), p._version))
    return tuple(sig)


# --------------------------------------------------------------------------------------------
# per-layer runtime state
# --------------------------------------------------------------------------------------------
class StageState:
    """Packed weights + tables + workspaces of one FasterViTLayer (cached on the module)."""

    def __init__(self):
        self.sig = None
        self.keep = None
        self.blocks_c = None
        self.tables = {}      # (Hp, Wp) -> (dict of device tensors, FvitStageTables)
        self.workspaces = {}  # (B, Hp, Wp, H, W) -> (desc, uint8 tensor)


def _state(layer) -> StageState:
    st = layer.__dict__.get("_fvit_state")
    if st is None:
        st = StageState()
        layer.__dict__["_fvit_state"] = st  # plain attribute: not a submodule, not in state_dict
    return st


def _geometry(layer, Hp: int, Wp: int):
    blk = layer.blocks[0]
    ws = layer.window_size
    hier = bool(blk.do_sr_hat)
    sr0, sr1 = Hp // ws, Wp // ws
    if hier and [sr0, sr1] != list(blk.sr_ratio):
        raise ValueError(f"hierarchical stage was built for {blk.sr_ratio[0]}x{blk.sr_ratio[1]} windows of {ws}; "
                         f"the (padded) input has {sr0}x{sr1}")
    return ws, hier, sr0, sr1


def _prepare(layer, x_dev, Hp: int, Wp: int):
    st = _state(layer)
    op_name = getattr(layer, "hat_operand_dtype", "f16")
    op_code, op_dtype = _OP[op_name]
    ws, hier, sr0, sr1 = _geometry(layer, Hp, Wp)
    blk0 = layer.blocks[0]
    cw = blk0.cr_window
    Cdim = blk0.attn.qkv.in_features
    heads = blk0.attn.num_heads
    d = Cdim // heads
    if d > 96:
        raise NotImplementedError(f"head_dim {d} > 96 has no attention kernel instance")
    dpad = 32 if d <= 32 else (64 if d <= 64 else 96)   # 96: head_dim 80 of FasterViT-5 / -6
    tkey = (Hp, Wp)
    if tkey not in st.tables:
        tb = build_tables(sr0, sr1, ws, cw, hier)
        dev_t = {k: v.to(x_dev) for k, v in tb.items() if isinstance(v, torch.Tensor)}
        ct = FvitStageTables(dev_t["ln1_src"].data_ptr(), dev_t["ln1_add"].data_ptr(), dev_t["ct_src"].data_ptr(),
                             dev_t["up_idx"].data_ptr())
        st.tables[tkey] = (tb, dev_t, ct)
    tb, _, ctables = st.tables[tkey]
    sig = _signature(layer.blocks, x_dev, op_name) + (tb["S"], tb["G"])
    if st.sig != sig:
        keep = _Keep(op_dtype)
        arr = (FvitBlockWeights * len(layer.blocks))()
        # constant folding must not run under the caller's autocast: the tables are fp32 by contract
        with torch.autocast(device_type="cuda", enabled=False):
            for i, blk in enumerate(layer.blocks):
                arr[i] = pack_block(blk, tb["S"], tb["G"], dpad, op_dtype, keep)
        st.keep, st.blocks_c, st.sig = keep, arr, sig
    lib = _lib.lib()
    desc_common = dict(C=Cdim, heads=heads, dpad=dpad, ws=ws, Hp=Hp, Wp=Wp, cw=cw if hier else 0, hier=int(hier),
                       square=int(hier and hasattr(blk0, "hat_pos_embed")), hidden=blk0.mlp.fc1.out_features,
                       depth=len(layer.blocks), do_propagation=int(bool(blk0.do_propagation)), operand_dtype=op_code,
                       spad=lib.fvit_attention_spad(tb["S"]), gpad=lib.fvit_attention_spad(tb["G"]) if hier else 0,
                       qk_scale=float(blk0.attn.scale))
    return st, tb, ctables, desc_common


_WS_SLOT = [0]   # workspace slot of the current caller: concurrent forwards on different streams must not share scratch


def set_workspace_slot(slot: int) -> None:
    _WS_SLOT[0] = int(slot)


def _workspace(st: StageState, desc_common: dict, B: int, H: int, W: int, device):
    key = (B, desc_common["Hp"], desc_common["Wp"], H, W, desc_common["operand_dtype"], _WS_SLOT[0])
    hit = st.workspaces.get(key)
    if hit is not None:
        return hit
    lib = _lib.lib()
    desc = FvitStageDesc(batch=B, H=H, W=W, **desc_common)
    nbytes = lib.fvit_stage_workspace_bytes(C.byref(desc))
    if nbytes == 0:
        _lib.check(-1, "fvit_stage_workspace_bytes")
    ws_t = torch.zeros(nbytes, dtype=torch.uint8, device=device)  # zero-filled once, dedicated to this geometry
    st.workspaces[key] = (desc, ws_t)
    return desc, ws_t


@torch.no_grad()
def token_init(tok, xp: torch.Tensor) -> torch.Tensor:
    """TokenInitializer.forward (AR:745-750) through fvit_token_init: depthwise conv + bias + avg-pool + per-window reorder
    in one HIP kernel, f32 (B, G, C) out.  (MIOpen runs this depthwise conv on its naive path: ~150 us at B = 256.)"""
    _require_gpu(xp, "TokenInitializer")
    st = tok.__dict__.get("_fvit_tok")
    sig = (tok.pos_embed.weight.data_ptr(), tok.pos_embed.weight._version, tok.pos_embed.bias._version, str(xp.device))
    if st is None or st[0] != sig:
        w = tok.pos_embed.weight.detach().float().reshape(-1, 9).contiguous().to(xp.device)
        b = tok.pos_embed.bias.detach().float().contiguous().to(xp.device)
        st = (sig, w, b)
        tok.__dict__["_fvit_tok"] = st
    _, w, b = st
    pool = tok.to_global_feature.pool
    kh, kw = pool.kernel_size if isinstance(pool.kernel_size, (tuple, list)) else (pool.kernel_size,) * 2
    sh, sw = pool.stride if isinstance(pool.stride, (tuple, list)) else (pool.stride,) * 2
    B, Cc, Hp, Wp = xp.shape
    Ho, Wo = (Hp - kh) // sh + 1, (Wp - kw) // sw + 1
    ct = torch.empty((B, Ho * Wo, Cc), dtype=torch.float32, device=xp.device)
    view = _map_view(xp)
    rc = _lib.lib().fvit_token_init(C.byref(view), w.data_ptr(), b.data_ptr(), ct.data_ptr(), B, Cc, Hp, Wp, kh, kw, sh, sw,
                                    tok.window_size, _stream_ptr())
    _lib.check(rc, "fvit_token_init")
    return ct


def _check_mode(layer):
    if layer.training and torch.is_grad_enabled():
        raise RuntimeError("the MI355X HAT path is inference-only (forward kernels); call model.eval() and run under "
                           "torch.no_grad(). Backward kernels are listed under SURVEY.md §8(f).")


@torch.no_grad()
def stage_forward(layer, x: torch.Tensor, tokenizer=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Transformer branch of FasterViTLayer.forward (AR:848-869) minus the Downsample.  ``x`` and the optional preallocated
    ``out`` (same shape and dtype) may be arbitrary strided views, e.g. the leading channels of a channel-padded map.

    ``tokenizer`` (optional) replaces the layer's TokenInitializer module with an equivalent callable
    returning f32 (B, G, C) carrier tokens (deploy mode passes a 16-bit channels_last version)."""
    _check_mode(layer)
    _require_gpu(x, "FasterViTLayer")
    lib = _lib.lib()
    B, Cc, H, W = x.shape
    ws = layer.window_size
    pad_r = (ws - W % ws) % ws
    pad_b = (ws - H % ws) % ws
    xp = F.pad(x, (0, pad_r, 0, pad_b)) if (pad_r or pad_b) else x
    Hp, Wp = H + pad_b, W + pad_r
    ct = None
    blk0 = layer.blocks[0] if len(layer.blocks) else None
    if blk0 is None:
        return x
    if layer.do_gt and blk0.do_sr_hat:
        # TokenInitializer stays a PyTorch-ROCm dwconv + pool (north_star); runs in the model's dtype
        if tokenizer is not None:
            ct = tokenizer(xp)
        else:
            ct = token_init(layer.global_tokenizer, xp)
    st, tb, ctables, dc = _prepare(layer, x.device, Hp, Wp)
    desc, ws_t = _workspace(st, dc, B, H, W, x.device)
    if out is None:
        out = torch.empty_like(x)  # keeps dtype and memory format (NCHW or channels_last)
    elif out.shape != x.shape or out.dtype != x.dtype or out.device != x.device:
        raise RuntimeError(f"stage_forward: out {tuple(out.shape)} {out.dtype} does not match x {tuple(x.shape)} {x.dtype}")
    vin, vout = _map_view(xp), _map_view(out)
    rc = lib.fvit_hat_stage_forward(C.byref(desc), st.blocks_c, C.byref(ctables), C.byref(vin),
                                    ct.data_ptr() if ct is not None else None, C.byref(vout), ws_t.data_ptr(),
                                    ws_t.numel(), _stream_ptr())
    _lib.check(rc, "fvit_hat_stage_forward")
    return out


@torch.no_grad()
def block_forward(blk, x: torch.Tensor, carrier_tokens: Optional[torch.Tensor]):
    """HAT.forward(x, carrier_tokens) with the reference signature (AR:668-707)."""
    _check_mode(blk)
    _require_gpu(x, "HAT")
    lib = _lib.lib()
    Bw, T, Cc = x.shape
    ws = blk.window_size
    hier = bool(blk.do_sr_hat)
    sr0, sr1 = (blk.sr_ratio if hier else (1, 1))
    nW = sr0 * sr1
    if Bw % nW:
        raise ValueError(f"HAT.forward: {Bw} windows is not a multiple of {nW} windows per image")
    B = Bw // nW
    # a one-block pseudo layer so packing / tables are shared with the stage path
    holder = blk.__dict__.get("_fvit_holder")
    if holder is None:
        holder = _BlockHolder(blk)
        blk.__dict__["_fvit_holder"] = holder
    st, tb, ctables, dc = _prepare(holder, x.device, sr0 * ws, sr1 * ws)
    dc = dict(dc, depth=1)
    desc, ws_t = _workspace(st, dc, B, sr0 * ws, sr1 * ws, x.device)
    xf = x.float().contiguous().clone()
    ctf = None
    if hier:
        if carrier_tokens is None:
            raise ValueError("hierarchical HAT block needs carrier tokens")
        ctf = carrier_tokens.float().contiguous().clone()
    rc = lib.fvit_hat_block_forward(C.byref(desc), st.blocks_c, C.byref(ctables), xf.data_ptr(),
                                    ctf.data_ptr() if ctf is not None else None, ws_t.data_ptr(), ws_t.numel(), _stream_ptr())
    _lib.check(rc, "fvit_hat_block_forward")
    if hier:
        return xf.to(x.dtype), ctf.to(carrier_tokens.dtype)
    return xf.to(x.dtype), carrier_tokens


class _BlockHolder:
    """Minimal stand-in for a FasterViTLayer around a single HAT block (block-level API)."""

    def __init__(self, blk):
        self.blocks = [blk]
        self.window_size = blk.window_size
        self.training = False

    @property
    def hat_operand_dtype(self):
        return getattr(self.blocks[0], "hat_operand_dtype", "f16")


def workspace_bytes(layer) -> int:
    """Bytes of HIP workspace currently held for this layer (all cached geometries)."""
    st = layer.__dict__.get("_fvit_state")
    return 0 if st is None else sum(t.numel() for _, t in st.workspaces.values())
