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

import contextlib
import ctypes as C
import threading
import warnings
from typing import Optional

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import (FVIT_BF16, FVIT_F16, FVIT_F32, FVIT_MASK_BIAS, FVIT_TILE_K, FVIT_TILE_N, FvitAttnWeights,
                   FvitBlockWeights, FvitMapView, FvitMlpWeights, FvitStageDesc, FvitStageTables)

_DT = {torch.float32: FVIT_F32, torch.float16: FVIT_F16, torch.bfloat16: FVIT_BF16}
# operand modes: (MFMA operand code, torch dtype, weight terms).  "x2" = every Linear weight packed as TWO 16-bit terms
# (hi = round(w), lo = round(w - hi); FvitStageDesc.weight_terms): the route to logits max-abs < 1e-3 with bf16 operands.
# "x3" (r04) = two-term weights AND two-term activations (every Linear layer hi.hi + hi.lo + lo.hi, the attention core on two-term
# q / k / v / P, exact-erf GELU): ~22 significant bits through the HAT stages -- the route to logits max-abs < 1e-3 ABSOLUTE on
# FasterViT-4 / any-res, whose logits reach |7| (the single rounding of the fp16 activations alone is ~1e-3 there).  Runs the
# unfused kernel chain (LayerNorm, GEMM, attention) with three times the MFMA work.
from ._lib import FVIT_MAX_DENSE_SEQ  # noqa: E402

_OP = {"f16": (FVIT_F16, torch.float16, 1), "bf16": (FVIT_BF16, torch.bfloat16, 1),
       "f16x2": (FVIT_F16, torch.float16, 2), "bf16x2": (FVIT_BF16, torch.bfloat16, 2),
       "f16x3": (FVIT_F16, torch.float16, 3), "bf16x3": (FVIT_BF16, torch.bfloat16, 3)}
OPERAND_MODES = tuple(_OP)


def _rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: the FasterViT HAT path runs only on a HIP device (got a {t.device.type} tensor); "
            "there is no CPU or eager fallback. Move the model and input to the GPU.")


def _stream_ptr(device=None) -> int:
    """Raw hipStream_t of torch's current stream ON ``device`` (not on the thread's current device: a model moved with
    ``.to('cuda:1')`` and called while cuda:0 is current must launch on a cuda:1 stream)."""
    return torch.cuda.current_stream(device).cuda_stream


def _check_device(x: torch.Tensor, layer, what: str) -> None:
    """The kernels dereference raw pointers: parameters and input must live on the same GPU."""
    for p in layer.parameters():
        if p.device != x.device:
            raise RuntimeError(f"{what}: parameters are on {p.device} but the input is on {x.device}; move the model with "
                               f".to('{x.device}') (there is no implicit cross-device copy on the HIP path)")
        break


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

    def __init__(self, op_dtype=None, terms: int = 1):
        self.tensors = []
        self.op_dtype = op_dtype
        self.terms = terms

    def op16(self, w: torch.Tensor, dim: int = -1) -> torch.Tensor:
        """fp32 packed weight -> operand type; with 2 weight terms the hi and lo parts are concatenated along ``dim`` (the K
        columns of a row-major array, or dim 0 of a flattened fragment-order image)."""
        hi = w.to(self.op_dtype)
        if self.terms == 1:
            return hi.contiguous()
        lo = (w - hi.float()).to(self.op_dtype)
        if self.terms == 3:   # [hi | lo | hi]: met by activation columns [hi | hi | lo] (GemmCall.ka, csrc/fvit_gemm.hip)
            return torch.cat([hi, lo, hi], dim=dim).contiguous()
        return torch.cat([hi, lo], dim=dim).contiguous()

    def frag16(self, w: torch.Tensor) -> Optional[torch.Tensor]:
        """fragment-order image(s): [hi image | lo image] back to back.  None with 3 terms: the fused kernels that stream
        fragment-order weights take one- and two-term weights only (the x3 modes run the LayerNorm / GEMM / attention chain)."""
        if self.terms == 3:
            return None
        return self.op16(w.reshape(1, -1), dim=0).reshape(-1)

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
        wqf = keep.frag16(frag_pack_qkv(wqkv32, h))
        bqh = bqkv32.view(3, h, 32).permute(1, 0, 2).reshape(h, 96).contiguous()
        wpf = keep.frag16(frag_pack_fc2(_f32(attn.proj.weight)))   # chunks of 32 input columns = heads
    return FvitAttnWeights(keep.ptr(keep.op16(wq), True), keep.ptr(bq), keep.ptr(keep.op16(wp), True), keep.ptr(_f32(attn.proj.bias)),
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
        w1f = keep.frag16(frag_pack_fc1(_f32(mlp.fc1.weight)))
        w2f = keep.frag16(frag_pack_fc2(_f32(mlp.fc2.weight)))
    return FvitMlpWeights(keep.ptr(keep.op16(w1), True), keep.ptr(_f32(mlp.fc1.bias)), keep.ptr(keep.op16(w2), True),
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


def _signature(blocks, device, op_name, replica=False):
    """Identity of the weights the packed copies were made from: (data_ptr, version) of every parameter.  Replicas made by
    ``nn.DataParallel`` are fresh broadcast copies on every forward (version 0, addresses reused by the caching allocator), so for
    them the signature is a content checksum instead (one small reduction + host sync per stage: DataParallel is the reference's
    legacy multi-GPU path, validate.py:243-244; the per-process runner scripts/run_sharded_validate.py has no such cost)."""
    sig = [str(device), op_name]
    if replica:
        ps = [p.detach().float().sum() for blk in blocks for p in blk.parameters()]
        sig.append(tuple(torch.stack(ps).double().cpu().tolist()) if ps else ())
        return tuple(sig)
    for blk in blocks:
        for p in blk.parameters():
            sig.append((p.data_ptr(), p._version))
    return tuple(sig)


# --------------------------------------------------------------------------------------------
# per-layer runtime state
# --------------------------------------------------------------------------------------------
class _Workspace:
    """One zero-initialised scratch allocation of a stage geometry + the bookkeeping that makes sharing it between HIP streams safe."""
    __slots__ = ("desc", "descs", "buf", "last_stream", "event")

    def __init__(self, desc, buf):
        self.desc, self.buf = desc, buf
        self.descs = {}
        self.last_stream = None
        self.event = None


class StageState:
    """Packed weights + tables + workspaces of one FasterViTLayer ON ONE DEVICE (cached on the module, keyed by device)."""

    def __init__(self):
        self.sig = None
        self.keep = None
        self.blocks_c = None
        self.tables = {}      # (Hp, Wp) -> (dict of device tensors, FvitStageTables)
        self.workspaces = {}  # (B, Hp, Wp, H, W, operand dtype, slot) -> _Workspace
        self.lock = threading.RLock()   # packing and the launches of one stage are enqueued under this lock


_STATE_LOCK = threading.Lock()


def _state(layer, device) -> StageState:
    """Per-device runtime state.  ``nn.DataParallel.replicate()`` shallow-copies ``__dict__``, so every replica of a layer sees the SAME
    dict object here; keying it by device gives each replica (one per GPU, driven by its own host thread) its own packed weights,
    tables and workspaces."""
    with _STATE_LOCK:
        states = layer.__dict__.get("_fvit_state")
        if states is None:
            states = {}
            layer.__dict__["_fvit_state"] = states  # plain attribute: not a submodule, not in state_dict
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        st = states.get(key)
        if st is None:
            st = states[key] = StageState()
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


def x3_unsupported_reason(layer):
    """Why a transformer level cannot run the 'f16x3' / 'bf16x3' modes (None: it can).  The x3 modes run the attention kernels with two-term
    q / k / v / P -- any window length since r06 (csrc/fvit_attn.hip up to FVIT_MAX_DENSE_SEQ tokens, csrc/fvit_attnlong.hip beyond) -- up to head_dim 96,
    and have no Dropout on the softmax probabilities."""
    blk0 = layer.blocks[0]
    Cdim, heads = blk0.attn.qkv.in_features, blk0.attn.num_heads
    d = Cdim // heads
    if d > 96:
        return f"head_dim {d} > 96 has no attention kernel instance"
    hier = bool(blk0.do_sr_hat)
    # (r06: every sequence length has a two-term attention kernel -- fvit_attn.hip up to FVIT_MAX_DENSE_SEQ tokens, fvit_attnlong.hip beyond: the 21k 384 /
    #  512 / 768 fine-tunes and large carrier grids take the x3 modes too)
    for blk in layer.blocks:
        for att in (blk.attn,) + ((blk.hat_attn,) if hier and hasattr(blk, "hat_attn") else ()):
            if float(getattr(att.attn_drop, "p", 0.0) or 0.0) > 0.0:
                return "attn_drop_rate > 0 (Dropout on the softmax probabilities) is not implemented for two-term probabilities; use a 16-bit or x2 mode to train with it"
    return None


def _prepare(layer, x_dev, Hp: int, Wp: int):
    st = _state(layer, x_dev)
    op_name = getattr(layer, "hat_operand_dtype", "f16")
    if op_name not in _OP:
        raise ValueError(f"unknown HAT operand mode {op_name!r}; choose from {OPERAND_MODES}")
    op_code, op_dtype, terms = _OP[op_name]
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
    sig = _signature(layer.blocks, x_dev, op_name, bool(getattr(layer, "_is_replica", False))) + (tb["S"], tb["G"])
    if st.sig != sig:
        keep = _Keep(op_dtype, terms)
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
                       qk_scale=float(blk0.attn.scale), weight_terms=terms)
    return st, tb, ctables, desc_common


_TLS = threading.local()   # workspace slot of the calling thread (default 0)


def set_workspace_slot(slot: int) -> None:
    """Scratch slot used by the stage calls of THIS thread from now on.  Forwards that are meant to overlap on different HIP streams
    (the stream shards of the deploy plan) take different slots; forwards that share a slot are serialised on the GPU
    (``_acquire`` inserts an event wait when a slot changes streams)."""
    _TLS.slot = int(slot)


def _slot() -> int:
    return getattr(_TLS, "slot", 0)


@contextlib.contextmanager
def workspace_slot(slot: int):
    prev = _slot()
    set_workspace_slot(slot)
    try:
        yield
    finally:
        set_workspace_slot(prev)


def _workspace(st: StageState, desc_common: dict, B: int, H: int, W: int, device) -> _Workspace:
    # (the workspace LAYOUT is the same for one- and two-term weights -- activations are single-rounded there -- but the descriptor
    # carries the terms: one scratch buffer per key, one descriptor per weight-term count; the x3 modes hold two-term activation
    # rows and get their own buffer)
    terms = desc_common["weight_terms"]
    key = (B, desc_common["Hp"], desc_common["Wp"], H, W, desc_common["operand_dtype"], _slot()) + (("x3",) if terms == 3 else ())
    hit = st.workspaces.get(key)
    if hit is not None:
        d = hit.descs.get(terms)
        if d is None:
            d = hit.descs[terms] = FvitStageDesc(batch=B, H=H, W=W, **desc_common)
        hit.desc = d
        return hit
    desc = FvitStageDesc(batch=B, H=H, W=W, **desc_common)
    lib = _lib.lib()
    nbytes = lib.fvit_stage_workspace_bytes(C.byref(desc))
    if nbytes == 0:
        _lib.check(-1, "fvit_stage_workspace_bytes")
    ws_t = torch.zeros(nbytes, dtype=torch.uint8, device=device)  # zero-filled once, dedicated to this geometry
    ws = st.workspaces[key] = _Workspace(desc, ws_t)
    ws.descs[terms] = desc
    return ws


def _acquire(ws: _Workspace, device) -> int:
    """Make the current stream the owner of ``ws``: if its last user was another stream, wait for that user's stage to finish
    (scratch is reused, not copied).  Returns the raw stream.  Inside a hipGraph capture no cross-stream bookkeeping is done:
    a captured forward uses the slots it was warmed up with, and its fork / join edges are the plan's own events."""
    stream = torch.cuda.current_stream(device)
    ptr = stream.cuda_stream
    if not torch.cuda.is_current_stream_capturing():
        if ws.last_stream is not None and ws.last_stream != ptr and ws.event is not None:
            stream.wait_event(ws.event)
    return ptr


def _release(ws: _Workspace, device) -> None:
    if torch.cuda.is_current_stream_capturing():
        return
    stream = torch.cuda.current_stream(device)
    if ws.event is None:
        ws.event = torch.cuda.Event()
    ws.event.record(stream)
    ws.last_stream = stream.cuda_stream


def token_init(tok, xp: torch.Tensor) -> torch.Tensor:
    """TokenInitializer.forward (AR:745-750) through fvit_token_init: depthwise conv + bias + avg-pool + per-window reorder
    in one HIP kernel, f32 (B, G, C) out.  (MIOpen runs this depthwise conv on its naive path: ~150 us at B = 256.)"""
    _require_gpu(xp, "TokenInitializer")
    with torch.no_grad(), torch.cuda.device(xp.device):
        key = "_fvit_tok"
        cache = tok.__dict__.get(key)
        if not isinstance(cache, dict):
            cache = tok.__dict__[key] = {}
        wt, bs = tok.pos_embed.weight, tok.pos_embed.bias
        if wt.device != xp.device:
            raise RuntimeError(f"TokenInitializer: parameters are on {wt.device} but the input is on {xp.device}")
        replica = bool(getattr(tok, "_is_replica", False))
        sig = (float(wt.detach().float().sum()), float(bs.detach().float().sum())) if replica else \
            (wt.data_ptr(), wt._version, bs.data_ptr(), bs._version)
        st = cache.get(str(xp.device))
        if st is None or st[0] != sig:
            w = wt.detach().float().reshape(-1, 9).contiguous()
            b = bs.detach().float().contiguous()
            st = cache[str(xp.device)] = (sig, w, b)
        _, w, b = st
        pool = tok.to_global_feature.pool
        kh, kw = pool.kernel_size if isinstance(pool.kernel_size, (tuple, list)) else (pool.kernel_size,) * 2
        sh, sw = pool.stride if isinstance(pool.stride, (tuple, list)) else (pool.stride,) * 2
        B, Cc, Hp, Wp = xp.shape
        Ho, Wo = (Hp - kh) // sh + 1, (Wp - kw) // sw + 1
        ct = torch.empty((B, Ho * Wo, Cc), dtype=torch.float32, device=xp.device)
        view = _map_view(xp)
        rc = _lib.lib().fvit_token_init(C.byref(view), w.data_ptr(), b.data_ptr(), ct.data_ptr(), B, Cc, Hp, Wp, kh, kw, sh, sw,
                                        tok.window_size, _stream_ptr(xp.device))
        _lib.check(rc, "fvit_token_init")
        return ct


_WARNED = set()


def _check_mode(layer, x: torch.Tensor, what: str):
    """The fused inference kernels implement EVAL semantics (DropPath / Dropout identity, AR:636-637, 657-658) and have no backward of their own:
      * train mode                               -> RuntimeError here.  ``FasterViTLayer.forward`` does not get here in train mode: it runs the stage
        through ``fastervit_amd.hat_backward`` (unit-kernel chain with stochastic depth + kernel-sequence backward); this guards the low-level entry
        points (``stage_forward`` / the block-level ``HAT.forward``) against silently skipping stochastic depth;
      * grad enabled + a LEAF input that requires grad (the caller differentiates w.r.t. this very tensor)
                                                  -> RuntimeError (outputs are detached, the gradient would silently be zero);
      * grad enabled otherwise                   -> a RuntimeWarning; the stage runs and returns detached outputs.  ``FasterViTLayer.forward`` routes
        eval-mode forwards that carry a graph through ``hat_backward`` wherever the geometry is covered (the gradient then flows, as in PyTorch); the
        warning remains for geometries it does not cover (windows of more than 64 tokens) and for direct calls of this function."""
    if layer.training:
        raise RuntimeError(f"{what}: the fused HIP stage kernels are inference-only (eval semantics: no stochastic depth); call model.eval(), or run the "
                           "stage through the module (FasterViTLayer.forward in train mode uses fastervit_amd.hat_backward: unit-kernel forward with "
                           "DropPath + kernel-sequence backward).")
    if torch.is_grad_enabled():
        if x.requires_grad and x.is_leaf:
            raise RuntimeError(f"{what}: the input requires grad, but this entry point is forward-only: its output would be silently detached. Use "
                               "model.enable_hat_backward() / FasterViTLayer.forward (differentiable where fastervit_amd.hat_backward covers the "
                               "geometry), or run under torch.no_grad().")
        if what not in _WARNED:
            _WARNED.add(what)
            warnings.warn(f"{what}: called with grad enabled; this forward-only HIP entry point returns DETACHED outputs (gradients do not flow "
                          "through it). Wrap inference in torch.no_grad(), or use model.enable_hat_backward().", RuntimeWarning, stacklevel=3)


def check_user_input(x: torch.Tensor, what: str = "FasterViT") -> None:
    """Model-level form of the leaf check above: the caller's own tensor asks for a gradient through a forward-only path."""
    if torch.is_grad_enabled() and x.requires_grad:
        raise RuntimeError(f"{what}: the input requires grad, but this model has a transformer stage whose geometry fastervit_amd.hat_backward does not "
                           "cover (windows or carrier grids of more than 64 tokens): the gradient w.r.t. the input would silently be cut. Run under "
                           "torch.no_grad() (or detach the input).")


def is_prepared(layer, device, batch: Optional[int] = None, hw=None, slots=(0,)) -> bool:
    """True if the packed weights of ``layer`` on ``device`` are current (no packing work will be enqueued by the next call).  With
    ``batch`` and ``hw`` = (H, W) of the stage input also: the index tables of that (padded) geometry and the workspaces of every
    slot in ``slots`` exist -- so that a forked multi-stream forward issues no H2D copy / allocation on a side stream."""
    states = layer.__dict__.get("_fvit_state")
    if not states or getattr(layer, "_is_replica", False):
        return False
    st = states.get((device.type, device.index if device.index is not None else torch.cuda.current_device()))
    if st is None or st.sig is None:
        return False
    op_name = getattr(layer, "hat_operand_dtype", "f16")
    if st.sig[:-2] != _signature(layer.blocks, device, op_name):
        return False
    if batch is None or hw is None:
        return True
    H, W = hw
    ws = layer.window_size
    Hp, Wp = H + (ws - H % ws) % ws, W + (ws - W % ws) % ws
    if (Hp, Wp) not in st.tables:
        return False
    op_code, _, terms = _OP[op_name]
    return all(((batch, Hp, Wp, H, W, op_code, s) + (("x3",) if terms == 3 else ())) in st.workspaces for s in slots)


def stage_forward(layer, x: torch.Tensor, tokenizer=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Transformer branch of FasterViTLayer.forward (AR:848-869) minus the Downsample.  ``x`` and the optional preallocated
    ``out`` (same shape and dtype) may be arbitrary strided views, e.g. the leading channels of a channel-padded map.

    ``tokenizer`` (optional) replaces the layer's TokenInitializer module with an equivalent callable
    returning f32 (B, G, C) carrier tokens (deploy mode passes a 16-bit channels_last version)."""
    _require_gpu(x, "FasterViTLayer")
    _check_mode(layer, x, "FasterViTLayer")
    if len(layer.blocks) == 0:
        return x
    _check_device(x, layer, "FasterViTLayer")
    lib = _lib.lib()
    with torch.no_grad(), torch.cuda.device(x.device):
        B, Cc, H, W = x.shape
        ws = layer.window_size
        pad_r = (ws - W % ws) % ws
        pad_b = (ws - H % ws) % ws
        xp = F.pad(x, (0, pad_r, 0, pad_b)) if (pad_r or pad_b) else x
        Hp, Wp = H + pad_b, W + pad_r
        ct = None
        blk0 = layer.blocks[0]
        if layer.do_gt and blk0.do_sr_hat:
            # TokenInitializer (AR:745-750): fvit_token_init in both modes
            ct = tokenizer(xp) if tokenizer is not None else token_init(layer.global_tokenizer, xp)
        st = _state(layer, x.device)
        with st.lock:
            st, tb, ctables, dc = _prepare(layer, x.device, Hp, Wp)
            wsp = _workspace(st, dc, B, H, W, x.device)
            if out is None:
                out = torch.empty_like(x)  # keeps dtype and memory format (NCHW or channels_last)
            elif out.shape != x.shape or out.dtype != x.dtype or out.device != x.device:
                raise RuntimeError(f"stage_forward: out {tuple(out.shape)} {out.dtype} does not match x {tuple(x.shape)} {x.dtype}")
            vin, vout = _map_view(xp), _map_view(out)
            stream = _acquire(wsp, x.device)
            rc = lib.fvit_hat_stage_forward(C.byref(wsp.desc), st.blocks_c, C.byref(ctables), C.byref(vin),
                                            ct.data_ptr() if ct is not None else None, C.byref(vout), wsp.buf.data_ptr(),
                                            wsp.buf.numel(), stream)
            _lib.check(rc, "fvit_hat_stage_forward")
            _release(wsp, x.device)
        return out


def block_forward(blk, x: torch.Tensor, carrier_tokens: Optional[torch.Tensor]):
    """HAT.forward(x, carrier_tokens) with the reference signature (AR:668-707)."""
    _require_gpu(x, "HAT")
    _check_mode(blk, x, "HAT")
    _check_device(x, blk, "HAT")
    lib = _lib.lib()
    with torch.no_grad(), torch.cuda.device(x.device):
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
        if holder is None or holder.blocks[0] is not blk:   # a DataParallel replica inherits the original's holder through __dict__
            holder = _BlockHolder(blk)
            blk.__dict__["_fvit_holder"] = holder
        st = _state(holder, x.device)
        with st.lock:
            st, tb, ctables, dc = _prepare(holder, x.device, sr0 * ws, sr1 * ws)
            dc = dict(dc, depth=1)
            wsp = _workspace(st, dc, B, sr0 * ws, sr1 * ws, x.device)
            xf = x.float().contiguous().clone()
            ctf = None
            if hier:
                if carrier_tokens is None:
                    raise ValueError("hierarchical HAT block needs carrier tokens")
                ctf = carrier_tokens.float().contiguous().clone()
            stream = _acquire(wsp, x.device)
            rc = lib.fvit_hat_block_forward(C.byref(wsp.desc), st.blocks_c, C.byref(ctables), xf.data_ptr(),
                                            ctf.data_ptr() if ctf is not None else None, wsp.buf.data_ptr(), wsp.buf.numel(), stream)
            _lib.check(rc, "fvit_hat_block_forward")
            _release(wsp, x.device)
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
    def _is_replica(self):
        return bool(getattr(self.blocks[0], "_is_replica", False))

    def parameters(self):
        return self.blocks[0].parameters()

    @property
    def hat_operand_dtype(self):
        return getattr(self.blocks[0], "hat_operand_dtype", "f16")


def workspace_bytes(layer) -> int:
    """Bytes of HIP workspace currently held for this layer (all devices, all cached geometries)."""
    states = layer.__dict__.get("_fvit_state") or {}
    return sum(w.buf.numel() for st in states.values() for w in st.workspaces.values())
