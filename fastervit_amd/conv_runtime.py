"""Deploy-mode plan for the conv side of FasterViT.

In module mode the convolutions are plain PyTorch-ROCm / MIOpen modules (north_star's drop-in default).  The deploy plan
runs the conv side through this library's own kernels instead and removes everything input-independent from the
per-forward path:

  * BatchNorm (eval) is folded into the preceding conv's weights and bias at load:
        PatchEmbed  conv(no bias)+BN(eps 1e-4)          (FV:458-463)
        ConvBlock   conv1+BN1, conv2+BN2 (+gamma)        (FV:490-512)
        final BN + AdaptiveAvgPool2d(1) + head           (FV:925-927, 953-960) -> one fp32 Linear on the pooled map
    exact in real arithmetic; weights are kept as 16-bit channels_last tensors ([Cout][3][3][Cin] matrices), so there is
    no autocast and no per-forward cast of parameters.
  * every 3x3 convolution (+ folded bias + ReLU / GELU + residual) is ONE hand-written HIP kernel (csrc/fvit_conv.hip:
    implicit-GEMM, halo-tiled 64-channel conv, fused two-conv stem); channel counts that are not a multiple of 64 are
    zero-padded to one (``pad_channels``).  MIOpen (``F.conv2d``) + the glue passes of csrc/fvit_glue.hip remain only as the
    fallback for shapes those kernels do not cover (``pad_channels = False`` or ``use_hip_conv = False``).
  * timm's LayerNorm2d (Downsample) is one HBM pass (``ln2d_kernel``).
  * The transformer stages are the same ``fvit_hat_stage_forward`` calls as in module mode.

  * ``plan.precise = True`` (r05; ``compile_inference(..., precise=True)``): the plan that meets north_star's ABSOLUTE logits bar on the deep / wide
    variants.  A replay of the fp32 oracle with single roundings placed one at a time (tests/tools/conv_precision_sim.py, FasterViT-4) shows that a
    16-bit conv OPERAND costs 2-8e-5 but a 16-bit STORED stream 3-4e-4 each (the residual stream of levels 0 / 1, the Downsample outputs, the
    transformer levels' output maps), the LayerNorm2d -> strided-conv operand 3.6e-4 and the K = 27 stem conv's weights / image 2.2e-4 / 1.2e-4.
    So here every stream is a TWO-TERM map (two 16-bit planes, value = hi + lo; the hi plane alone is the next conv's MFMA operand) or fp32
    where the consumer is a transformer level; all conv weights are hi + lo; the three Downsample convs and the first stem conv also take their
    input as two terms (hi.hi + hi.lo + lo.hi).  Kernels: ``conv3x3_kernel<.., PX>``, ``ln2d_kernel<.., PX>``, ``stem_conv_kernel<.., PX>``.

Enabled by ``FasterViT.switch_to_deploy()`` (explicit) or automatically for eval-mode forwards under ``torch.autocast`` with grad
disabled (``FasterViT.auto_deploy``); module mode (plain nn.Module forward, any dtype) remains the default so that the
reference's scripts run unchanged.  Logits of the 16-bit plan differ from the fp32 reference by the conv side's 16-bit rounding:
measured 6.5e-4 .. 9.5e-4 max-abs on FasterViT-0 (asserted < 1e-3 in tests/test_gpu_parity.py).
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from . import _lib, hat_runtime

_CODE = {torch.float16: _lib.FVIT_F16, torch.bfloat16: _lib.FVIT_BF16}


def _stream(device=None):
    return torch.cuda.current_stream(device).cuda_stream


def _bn_scale_shift(bn):
    s = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
    t = bn.bias.float() - bn.running_mean.float() * s
    return s, t


def _fold(conv, bn, extra_scale=None):
    """conv (+ optional bias) followed by eval-mode BN (and an optional per-channel scale) -> (w, b) fp32."""
    s, t = _bn_scale_shift(bn)
    w = conv.weight.float() * s.view(-1, 1, 1, 1)
    b = t if conv.bias is None else conv.bias.float() * s + t
    if extra_scale is not None:
        w = w * extra_scale.float().view(-1, 1, 1, 1)
        b = b * extra_scale.float()
    return w, b


def frag_pack_conv128(w2d: torch.Tensor) -> torch.Tensor:
    """[128][9 * 128] conv weight (row = output channel, column k = tap * 128 + input channel) -> the fragment-order stream of
    fvit_conv3x3_c128_band (include/fvit_hip.h): [wave 4][step 36][ni 2][lane 64][8]; element e of lane 16 g + s of fragment
    (wave, step, ni) = w2d[32 wave + (s >> 2) * 8 + ni * 4 + (s & 3)][step * 32 + 8 g + e]."""
    assert tuple(w2d.shape) == (128, 1152)
    dev = w2d.device
    wave = torch.arange(4, device=dev).view(4, 1, 1)
    ni = torch.arange(2, device=dev).view(1, 2, 1)
    sl = torch.arange(16, device=dev).view(1, 1, 16)
    ch = (32 * wave + (sl >> 2) * 8 + ni * 4 + (sl & 3)).reshape(-1)        # [wave, ni, s]
    t = w2d[ch].view(4, 2, 16, 36, 4, 8)                                      # wave, ni, s, step, g, e
    return t.permute(0, 3, 1, 4, 2, 5).contiguous().view(4, 36, 2, 64, 8)    # wave, step, ni, (g, s), e


class DeployPlan:
    def __init__(self, model, dtype=torch.float16):
        if dtype not in _CODE:
            raise ValueError("deploy dtype must be torch.float16 or torch.bfloat16")
        self.model = model
        self.dtype = dtype
        self.code = _CODE[dtype]
        self.sig = None
        self.t = None
        self.zeros = None
        self.streams = 1      # > 1: run the batch as that many shards on separate HIP streams
        self.side = None
        self.dev = None       # device of the current forward (raw-pointer launches go to torch's current stream on THIS device)
        self.use_hip_conv = True  # fused implicit-GEMM conv kernel where the shape allows; False = MIOpen + glue passes
        # conv-side maps carry their channels padded to a multiple of 64 (zeros), weights / biases / LayerNorm2d parameters are
        # zero-padded to match: every 3x3 conv then runs on the fused HIP kernel (FasterViT-1/2/4: 80 / 96 / 196 / 392 channels
        # would otherwise fall back to MIOpen + glue passes; a 196-channel fp16 pixel is not even 16-byte aligned).  Pad channels
        # stay exactly zero through bias (0), ReLU / GELU (f(0) = 0), residual adds and LayerNorm2d (zero weight / bias there).
        self.pad_channels = True
        # weight terms of the Downsample.reduction convs (FV:435): 2 = hi + lo (r04, opt-in).  These three strided, bias-free convs feed their
        # rounding straight into the next stage; the weight part of it is systematic (the same for every pixel of every image) and makes up
        # 2.4e-4 / 1.7e-4 / 2.1e-4 of FasterViT-0's 4.2e-4 conv-side logits error (per-layer replay on the fp32 oracle).  Measured (A/B x 2 in
        # one box, scripts/r04_calls/call5.sh): logits max-abs 7.5e-4 -> 4.7e-4 (f16), 5.0e-4 -> 3.0e-4 (f16x2), 7.9e-4 -> 6.9e-4 (bf16x2) for
        # 81.0k -> 77.7k images/s (twice the K steps of 3 of the 15 convs).  Default 1 = single rounding: the plan that is timed.
        self.down_weight_terms = int(os.environ.get("FVIT_DOWN_WEIGHT_TERMS", "1"))
        # 2 = two-term weights in EVERY 3x3 conv that runs on the implicit-GEMM kernel (the stem's second conv, the ConvBlock convs, the downsamples;
        # the halo / band / fused-stem kernels take single-term weights, so those shapes fall back to the implicit GEMM): with the x3 HAT modes the
        # "precise deploy" configuration -- 16-bit maps, everything else to ~22 bits
        self.conv_weight_terms = int(os.environ.get("FVIT_CONV_WEIGHT_TERMS", "1"))
        # two-term / fp32 streams + two-term weights everywhere (module docstring): the ABSOLUTE-1e-3 plan of FasterViT-4 / any-res
        self.precise = os.environ.get("FVIT_PRECISE_DEPLOY", "0") == "1"
        # r06: 3x3 convs contract over the REAL input channels only (dense K; include/fvit_hip.h): 29 / 56 K steps instead of 36 / 63 on FasterViT-4's
        # 196- / 392-channel maps (stored as 256 / 448).  Same products in the same order per tap; "0" = the classic [Cout][3][3][Cin padded] matrix
        self.dense_k = os.environ.get("FVIT_CONV_DENSE_K", "1") != "0"
        self.fused_stem = os.environ.get("FVIT_NO_FUSED_STEM", "0") != "1"   # both PatchEmbed convs in one kernel when in_dim == dim == 64 (the 112x112x64 map never reaches HBM)

    # ---- folding -------------------------------------------------------------------------
    def _signature(self):
        m = self.model
        sig = []
        for name, p in list(m.named_parameters()) + list(m.named_buffers()):
            if ".blocks." in name and name.startswith(("levels.2.", "levels.3.")):
                continue  # HAT parameters are tracked by hat_runtime
            sig.append((p.data_ptr(), p._version))
        sig.append(("options", self.precise, self.conv_weight_terms, self.down_weight_terms, self.dense_k))   # what _build depends on besides the parameters
        return tuple(sig)

    def _cp(self, c):
        """Channel count of a conv-side map holding c real channels."""
        return c if (c % 64 == 0 or c <= 3 or not (self.pad_channels and self.use_hip_conv)) else (c + 63) // 64 * 64

    def _padv(self, v, n):
        """1-D parameter zero-padded to n entries."""
        if v is None or v.numel() == n:
            return v
        out = torch.zeros(n, dtype=v.dtype, device=v.device)
        out[:v.numel()] = v
        return out

    def _cw(self, w, terms: int = 1):
        """(MIOpen weight, HIP-kernel weight, band-kernel weight[, weight terms]): the channels_last 16-bit tensor for F.conv2d, -- when the fused
        implicit-GEMM kernel supports the shape (3x3, Cin and Cout multiples of 64) -- its [Cout][3][3][Cin] matrix view, and for
        128 -> 128 channels the fragment-order stream of the row-band kernel.
        Both channel counts are zero-padded to the map layout (``_cp``)."""
        co0, ci0 = w.shape[:2]
        cop, cip = self._cp(co0), self._cp(ci0)
        if (cop, cip) != (co0, ci0):
            wp = torch.zeros((cop, cip) + tuple(w.shape[2:]), dtype=w.dtype, device=w.device)
            wp[:co0, :ci0] = w
            w = wp
        wcl = w.to(self.dtype).contiguous(memory_format=torch.channels_last)
        co, ci, kh, kw = w.shape
        wk = wband = None
        cv = ci   # channels the kernel contracts over per tap (== ci: the classic [Cout][3][3][Cin] matrix)
        if self.use_hip_conv and kh == 3 and kw == 3 and ci % 64 == 0 and co % 64 == 0:
            wk = wcl.permute(0, 2, 3, 1).contiguous()
            lo = (w - wcl.float()).to(self.dtype).permute(0, 2, 3, 1).contiguous() if terms == 2 else None
            cv8 = (ci0 + 7) // 8 * 8
            if self.dense_k and cv8 < ci:
                # r06: the pad channels of the INPUT map leave the contraction (fvit_conv3x3_nhwc_dense): [Cout][terms][kd], column t * cv + c
                cv, kd = cv8, (9 * cv8 + 63) // 64 * 64

                def dense(m):   # m: (co, 3, 3, ci)
                    d = torch.zeros((co, kd), dtype=m.dtype, device=m.device)
                    d[:, :9 * cv] = m[..., :cv].reshape(co, 9 * cv)
                    return d
                # (the classic matrix next to the dense one: the patch form of the kernel -- fvit_conv3x3_patch_form, chosen per call from the map size -- takes it)
                classic = wk.reshape(co, -1) if lo is None else torch.cat([wk.reshape(co, -1), lo.reshape(co, -1)], dim=1).contiguous()
                wk = dense(wk) if lo is None else torch.cat([dense(wk), dense(lo)], dim=1).contiguous()
                return wcl, wk, None, terms, cv, classic
            if terms == 2:   # [Cout][hi (3,3,Cin) | lo (3,3,Cin)]: fvit_conv3x3_nhwc_terms
                wk = torch.cat([wk.reshape(co, -1), lo.reshape(co, -1)], dim=1).contiguous()
                return wcl, wk, None, 2, cv, None
            if (co, ci) == (128, 128):   # the fragment-order image fvit_conv3x3_c128_band streams (level 1 of FasterViT-0)
                wband = frag_pack_conv128(wk.reshape(128, 1152))
        return wcl, wk, wband, 1, cv, None

    def _conv(self, x, w, bias, stride, act, residual=None):
        """act(conv3x3(x, w) + bias) (+ residual): one fused HIP kernel when supported, else MIOpen conv + glue passes."""
        wcl, wk, wband, wterms, cv, wk_classic = w
        B, Ci, Hi, Wi = x.shape
        if wk is not None and x.is_contiguous(memory_format=torch.channels_last):
            Co = wk.shape[0]
            if wk_classic is not None and _lib.lib().fvit_conv3x3_patch_form(B, Hi, Wi, Ci, Co, stride):
                wk, cv = wk_classic, Ci   # 8 x 16 patches + halo tiles (r06): the classic [Cout][terms][3][3][Cin] rows
            Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
            out = residual if residual is not None else torch.empty((B, Co, Ho, Wo), dtype=self.dtype, device=x.device,
                                                                    memory_format=torch.channels_last)
            if self.zeros is None or self.zeros.device != x.device:
                self.zeros = torch.zeros(256, dtype=self.dtype, device=x.device)
            if wband is not None and stride == 1 and _lib.lib().fvit_conv3x3_c128_band_supported(Hi, Wi):   # level 1 of FasterViT-0: one row band of an image per workgroup, weights streamed in fragment order
                rc = _lib.lib().fvit_conv3x3_c128_band(self.code, x.data_ptr(), wband.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                       residual.data_ptr() if residual is not None else None, out.data_ptr(), B, Hi, Wi, act,
                                                       self.zeros.data_ptr(), _stream(self.dev))
                _lib.check(rc, "fvit_conv3x3_c128_band")
                return out
            rc = _lib.lib().fvit_conv3x3_nhwc_dense(self.code, x.data_ptr(), wk.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                    residual.data_ptr() if residual is not None else None, out.data_ptr(), B, Hi, Wi,
                                                    Ci, cv, Co, stride, act, wterms, self.zeros.data_ptr(), _stream(self.dev))
            _lib.check(rc, "fvit_conv3x3_nhwc_dense")
            return out
        y = F.conv2d(x, wcl, None, stride, 1)
        if residual is not None:
            return self._bias_residual(residual, y, bias)
        return self._bias_act(y, bias, act) if bias is not None else y

    @torch.no_grad()
    def _build(self):
        m = self.model
        t = {}
        pe = m.patch_embed.conv_down
        w0, b0 = _fold(pe[0], pe[1])
        w1, b1 = _fold(pe[3], pe[4])
        ct = 2 if (self.conv_weight_terms == 2 or self.precise) else 1
        t["stem"] = (self._cw(w0), self._padv(b0, self._cp(b0.numel())).contiguous(), self._cw(w1, terms=ct),
                     self._padv(b1, self._cp(b1.numel())).contiguous())
        t["stem_k"] = t["stem_k_lo"] = None
        # precise plan with a non-standard stem (in_dim != 64 or in_chans != 3): the first conv runs as an fp32 PyTorch-ROCm conv (channels padded)
        co0 = self._cp(w0.shape[0])
        w0p = torch.zeros((co0,) + tuple(w0.shape[1:]), dtype=torch.float32, device=w0.device)
        w0p[:w0.shape[0]] = w0
        t["stem0_f32"] = (w0p.contiguous(memory_format=torch.channels_last), self._padv(b0, co0).contiguous(), tuple(pe[0].stride))
        if self.use_hip_conv and tuple(w0.shape) == (64, 3, 3, 3) and pe[0].stride == (2, 2):
            wk = torch.zeros(64, 32, device=w0.device, dtype=torch.float32)
            wk[:, :27] = w0.permute(0, 2, 3, 1).reshape(64, 27)       # k = ky*9 + kx*3 + c
            t["stem_k"] = wk.to(self.dtype).contiguous()
            t["stem_k_lo"] = (wk - t["stem_k"].float()).to(self.dtype).contiguous()   # second term of the K = 27 weights (precise plan)
        t["levels"] = []
        for lvl in m.levels:
            e = {}
            if not lvl.transformer_block:
                blocks = []
                for blk in lvl.blocks:
                    wa, ba = _fold(blk.conv1, blk.norm1)
                    wb, bb = _fold(blk.conv2, blk.norm2, blk.gamma if blk.layer_scale else None)
                    cpd = self._cp(ba.numel())
                    blocks.append((self._cw(wa, terms=ct), self._padv(ba, cpd).contiguous(), self._cw(wb, terms=ct), self._padv(bb, cpd).contiguous()))
                e["blocks"] = blocks
            elif getattr(lvl, "do_gt", False):
                tk = lvl.global_tokenizer
                e["tok"] = (self._cw(tk.pos_embed.weight.float())[0], tk.pos_embed.bias.to(self.dtype).contiguous(),
                            tk.to_global_feature.pool.kernel_size, tk.to_global_feature.pool.stride, tk.window_size)
            if lvl.downsample is not None:
                ds = lvl.downsample
                cin = ds.norm.weight.numel()
                e["down"] = (self._padv(ds.norm.weight.float(), self._cp(cin)).contiguous(),
                             self._padv(ds.norm.bias.float(), self._cp(cin)).contiguous(), float(ds.norm.eps),
                             self._cw(ds.reduction[0].weight.float(), terms=2 if (self.down_weight_terms == 2 or ct == 2) else 1), cin)
            t["levels"].append(e)
        if isinstance(m.head, torch.nn.Linear):
            hw = m.head.weight.float()
            hb = m.head.bias.float() if m.head.bias is not None else torch.zeros(hw.shape[0], device=hw.device)
        else:   # num_classes = 0: nn.Identity head, the pooled (normalised) features are the output (FV:927)
            nf = m.norm.weight.numel()
            hw, hb = torch.eye(nf, device=m.norm.weight.device), torch.zeros(nf, device=m.norm.weight.device)
        if isinstance(m.norm, torch.nn.BatchNorm2d):
            s, sh = _bn_scale_shift(m.norm)
            t["head"] = ((hw * s.view(1, -1)).contiguous(), (hb + hw @ sh).contiguous(), None)
        else:  # layer_norm_last: LayerNorm2d kernel, then pool + head
            t["head"] = (hw.contiguous(), hb.contiguous(),
                         (m.norm.weight.float().contiguous(), m.norm.bias.float().contiguous(), float(m.norm.eps)))
        self.t = t

    # ---- kernels -------------------------------------------------------------------------
    def _bias_act(self, x, bias, act):
        B, C, H, W = x.shape
        assert x.is_contiguous(memory_format=torch.channels_last) and x.dtype == self.dtype
        _lib.check(_lib.lib().fvit_bias_act_cl(self.code, x.data_ptr(), bias.data_ptr(), B * H * W, C, act, _stream(self.dev)),
                   "fvit_bias_act_cl")
        return x

    def _bias_residual(self, x, y, bias):
        B, C, H, W = x.shape
        assert x.is_contiguous(memory_format=torch.channels_last) and y.is_contiguous(memory_format=torch.channels_last)
        _lib.check(_lib.lib().fvit_bias_residual_cl(self.code, x.data_ptr(), y.data_ptr(), bias.data_ptr(), B * H * W, C,
                                                    _stream(self.dev)), "fvit_bias_residual_cl")
        return x

    def _pool_head(self, x, hw, hb):
        """AdaptiveAvgPool2d(1) + flatten + head (FV:955-960; final BatchNorm folded into hw / hb) on a channels_last map: fvit_global_avgpool_cl +
        fvit_head_logits (exact-fp32 MFMA) -- no library kernel in the captured graph.  Maps that are not dense channels_last (a strided stage output)
        or whose channel count is not a multiple of 16 take the PyTorch ops."""
        B, C, H, W = x.shape
        if x.dtype in hat_runtime._DT and x.permute(0, 2, 3, 1).is_contiguous() and C % 16 == 0 and hw.shape[1] == C and hw.is_contiguous():
            feat = torch.empty((B, C), dtype=torch.float32, device=x.device)
            _lib.check(_lib.lib().fvit_global_avgpool_cl(hat_runtime._DT[x.dtype], x.data_ptr(), feat.data_ptr(), B, H * W, C, _stream(self.dev)),
                       "fvit_global_avgpool_cl")
            out = torch.empty((B, hw.shape[0]), dtype=torch.float32, device=x.device)
            _lib.check(_lib.lib().fvit_head_logits(feat.data_ptr(), hw.data_ptr(), hb.data_ptr(), out.data_ptr(), B, hw.shape[0], C, _stream(self.dev)),
                       "fvit_head_logits")
            return out
        return F.linear(x.float().mean(dim=(2, 3)), hw, hb)

    def _ln2d(self, x, w, b, eps, c_valid=None):
        """LayerNorm2d over the first c_valid (default: all) channels of a channels_last map; pad channels stay zero."""
        B, C, H, W = x.shape
        cv = C if c_valid is None else c_valid
        if C % 8 or not x.is_contiguous(memory_format=torch.channels_last):
            y = torch.zeros_like(x, dtype=torch.float32)
            y[:, :cv] = F.layer_norm(x[:, :cv].permute(0, 2, 3, 1).float(), (cv,), w[:cv], b[:cv], eps).permute(0, 3, 1, 2)
            return y.to(self.dtype).contiguous(memory_format=torch.channels_last)
        out = torch.empty_like(x)
        _lib.check(_lib.lib().fvit_layernorm2d_cl(self.code, x.data_ptr(), out.data_ptr(), w.data_ptr(), b.data_ptr(), eps,
                                                  B * H * W, C, cv, _stream(self.dev)), "fvit_layernorm2d_cl")
        return out

    def _tokenizer(self, tok):
        w, b, ks, st, cw = tok

        def fn(xp):
            y = F.avg_pool2d(F.conv2d(xp, w, b, padding=1, groups=xp.shape[1]), ks, st)
            B, C, H, W = y.shape
            y = y.reshape(B, C, H // cw, cw, W // cw, cw).permute(0, 2, 4, 3, 5, 1).reshape(B, H * W, C)
            return y.float().contiguous()
        return fn

    # ---- forward -------------------------------------------------------------------------
    @torch.no_grad()
    def _refresh(self):
        sig = self._signature()
        if sig != self.sig:
            with torch.autocast(device_type="cuda", enabled=False):
                self._build()
            self.sig = sig

    def _enter(self, x):
        """Per-call checks: GPU input on the model's device; launches below go to torch's current stream on that device."""
        if not x.is_cuda:
            raise RuntimeError("deploy plan: the input must be on a HIP device (no CPU fallback)")
        p0 = next(self.model.parameters())
        if p0.device != x.device:
            raise RuntimeError(f"deploy plan: model parameters are on {p0.device} but the input is on {x.device}")
        self.dev = x.device

    def _hat_prepared(self, device):
        return self.zeros is not None and self.zeros.device == device and all(
            hat_runtime.is_prepared(lvl, device) for lvl in self.model.levels if lvl.transformer_block and len(lvl.blocks))

    def forward_single(self, x):
        """One shard on the caller's stream and current workspace slot (no sharding)."""
        self._enter(x)
        with torch.cuda.device(x.device):
            self._refresh()
            return self._forward_one(x)

    def forward(self, x):
        self._enter(x)
        with torch.cuda.device(x.device):
            self._refresh()
            return self._forward_sharded(x)

    def _forward_sharded(self, x):
        n = self.streams
        if n <= 1 or x.shape[0] < 2 * n:
            with hat_runtime.workspace_slot(getattr(self, "slot_base", 0)):
                return self._forward_one(x)
        # the batch as n independent shards on n HIP streams (fork / join with events; capturable in a hipGraph): every kernel
        # of this pipeline runs its HBM-bound prologue / epilogue and its MFMA phase in lockstep across workgroups, so two
        # half-size pipelines interleave better than one full-size one
        # slot_base (r06): first workspace slot of this plan; two plans whose forwards are in flight at the same time (inference.PipelinedInference:
        # consecutive steps on alternating streams) must not share the per-(geometry, slot) stage workspaces
        sb = getattr(self, "slot_base", 0)
        sizes = getattr(self, "shard_sizes", None)
        parts = x.split(list(sizes), dim=0) if sizes and sum(sizes) == x.shape[0] and len(sizes) == n else x.chunk(n, dim=0)
        outs = [None] * n
        # weight packing (hat_runtime._prepare) and the zero page are created lazily by the first shard that needs them, on ITS
        # stream; the other streams forked before that work was enqueued and would read half-written packed weights.  Until
        # everything is packed for this device, run the shards one after the other on the caller's stream (first call, or the
        # first call after a weight update).
        # The per-geometry index tables (an H2D copy) and workspaces are created lazily by the first stage call too: the forked form
        # is allowed only for a (device, shard sizes, image size, operand mode) that has completed one serial pass.
        ops = tuple(getattr(lvl, "hat_operand_dtype", "f16") for lvl in self.model.levels if lvl.transformer_block)
        wkey = (str(x.device), tuple(p.shape[0] for p in parts), tuple(x.shape[1:]), ops, getattr(self, "join_from", None), sb)
        warm = self.__dict__.setdefault("_warm_geometries", set())
        serial = getattr(self, "serialize_shards", False) or not self._hat_prepared(x.device) or wkey not in warm
        # join_from = L (r04, ``plan.join_from``; None = off): the shards run levels [0, L) on their own streams, JOIN, and levels
        # [L, end) + head run once on the whole batch on the caller's stream.  The last stage of FasterViT-0 (one 49-token window per
        # image, 512 channels) is 86-workgroup launches per shard that cannot fill the chip; joined it is 196 / 256 workgroups.
        jf = getattr(self, "join_from", None)
        nlev = len(self.model.levels)
        jf = jf if (jf is not None and 0 < jf < nlev) else None
        front = (lambda xi: self._forward_one(xi, 0, jf)) if jf is not None else self._forward_one
        def _cat(xs):   # shards of the precise plan hand on (hi, lo, f32) tuples
            if isinstance(xs[0], tuple):
                return tuple(None if xs[0][k] is None else torch.cat([x_[k] for x_ in xs], dim=0) for k in range(3))
            return torch.cat(xs, dim=0)
        back = (lambda xs: self._forward_one(_cat(xs), jf, None)) if jf is not None else (lambda xs: torch.cat(xs, dim=0))
        if serial:
            # also the measurement aid of bench.py's HIP-event pass: the same shard-sized launches, one after the other on the
            # caller's stream, so that a kernel's event-pair duration is its own and not shared with the other shards' kernels
            for i in range(n):
                with hat_runtime.workspace_slot(sb + i):
                    outs[i] = front(parts[i])
            with hat_runtime.workspace_slot(sb):
                y = back(outs)
            if not torch.cuda.is_current_stream_capturing():
                warm.add(wkey)
            return y
        if self.side is None or len(self.side) != n - 1 or self.side[0].device != x.device:
            self.side = [torch.cuda.Stream(device=x.device) for _ in range(n - 1)]
        main = torch.cuda.current_stream(x.device)
        for i, s in enumerate(self.side):
            s.wait_stream(main)
            with torch.cuda.stream(s), hat_runtime.workspace_slot(sb + i + 1):
                outs[i + 1] = front(parts[i + 1])
        with hat_runtime.workspace_slot(sb):
            outs[0] = front(parts[0])
        for s in self.side:
            main.wait_stream(s)   # join: everything the caller enqueues next (the cat below, its own later work) is ordered after the shards
        with hat_runtime.workspace_slot(sb):
            return back(outs)

    def shard_runner(self, x, n=None):
        """Free-running stream shards for throughput serving: see ``ShardRunner``."""
        return ShardRunner(self, x, n or max(self.streams, 1))

    # ---- the precise plan: two-term / fp32 streams (module docstring) ---------------------------------------------------
    def _conv_px(self, x, x_lo, w, bias, stride, act, res=None, res_lo=None, want="planes"):
        """conv3x3_kernel<.., PX>: x (+ x_lo) -> act(conv + bias) (+ res + res_lo).  ``want``: 'planes' -> (hi, lo) 16-bit planes (in place over the
        residual planes when given), 'single' -> (hi, None), 'f32' -> one fp32 channels_last map."""
        wcl, wk, wband, wterms, cv, wk_classic = w
        if wk is None or not x.is_contiguous(memory_format=torch.channels_last):
            raise RuntimeError("precise deploy plan: this conv shape has no implicit-GEMM kernel (channel counts must pad to multiples of 64)")
        B, Ci, Hi, Wi = x.shape
        Co = wk.shape[0]
        if wk_classic is not None and _lib.lib().fvit_conv3x3_patch_form(B, Hi, Wi, Ci, Co, stride):
            wk, cv = wk_classic, Ci   # the patch form takes the classic rows
        Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
        if self.zeros is None or self.zeros.device != x.device:
            self.zeros = torch.zeros(256, dtype=self.dtype, device=x.device)
        hi = lo = f32 = None
        if want == "f32":
            f32 = torch.empty((B, Co, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        else:
            hi = res if res is not None else torch.empty((B, Co, Ho, Wo), dtype=self.dtype, device=x.device, memory_format=torch.channels_last)
            if want == "planes":
                lo = res_lo if res_lo is not None else torch.empty_like(hi)
        rc = _lib.lib().fvit_conv3x3_nhwc_px_dense(self.code, x.data_ptr(), x_lo.data_ptr() if x_lo is not None else None, wk.data_ptr(),
                                             bias.data_ptr() if bias is not None else None, res.data_ptr() if res is not None else None,
                                             res_lo.data_ptr() if res_lo is not None else None, hi.data_ptr() if hi is not None else None,
                                             lo.data_ptr() if lo is not None else None, f32.data_ptr() if f32 is not None else None,
                                             B, Hi, Wi, Ci, cv, Co, stride, act, wterms, self.zeros.data_ptr(), _stream(self.dev))
        _lib.check(rc, "fvit_conv3x3_nhwc_px_dense")
        return f32 if want == "f32" else (hi, lo)

    def _ln2d_px(self, x, x_lo, x_f32, w, b, eps, c_valid):
        """LayerNorm2d of a two-term (x, x_lo) or fp32 (x_f32) channels_last map -> two 16-bit planes."""
        src = x_f32 if x_f32 is not None else x
        B, C, H, W = src.shape
        if C % 8 or not src.is_contiguous(memory_format=torch.channels_last):
            raise RuntimeError("precise deploy plan: LayerNorm2d needs a channels_last map with C % 8 == 0")
        hi = torch.empty((B, C, H, W), dtype=self.dtype, device=src.device, memory_format=torch.channels_last)
        lo = torch.empty_like(hi)
        _lib.check(_lib.lib().fvit_layernorm2d_px(self.code, x.data_ptr() if x is not None else None, x_lo.data_ptr() if x_lo is not None else None,
                                                  x_f32.data_ptr() if x_f32 is not None else None, hi.data_ptr(), lo.data_ptr(), w.data_ptr(),
                                                  b.data_ptr(), eps, B * H * W, C, c_valid, _stream(self.dev)), "fvit_layernorm2d_px")
        return hi, lo

    def _forward_one_precise(self, x, lv_from=0, lv_to=None):
        """``_forward_one`` with every stream as a two-term map ``(hi, lo)`` or an fp32 map.  Between levels the value handed on is the tuple
        ``(hi, lo, f32)`` (f32 set in front of / behind a transformer level); a partial call (stream shards + join) returns that tuple with the
        planes concatenated by the caller."""
        t = self.t
        levels = self.model.levels
        with torch.autocast(device_type="cuda", enabled=False):
            if lv_from == 0:
                w0, b0, w1, b1 = t["stem"]
                if t["stem_k"] is None or x.shape[1] != 3 or x.dtype not in hat_runtime._DT:
                    # a stem other than the reference's 3 -> 64 (in_dim / in_chans kwargs): its first conv as an fp32 PyTorch-ROCm conv
                    wf, bf, st0 = t["stem0_f32"]
                    y = torch.relu(F.conv2d(x.float(), wf, bf, st0, 1)).to(self.dtype).contiguous(memory_format=torch.channels_last)
                else:
                    B, _, Hi, Wi = x.shape
                    y = torch.empty((B, 64, (Hi - 1) // 2 + 1, (Wi - 1) // 2 + 1), dtype=self.dtype, device=x.device, memory_format=torch.channels_last)
                    view = hat_runtime._map_view(x)
                    _lib.check(_lib.lib().fvit_stem_conv3x3s2_px(self.code, view, t["stem_k"].data_ptr(), t["stem_k_lo"].data_ptr(), b0.data_ptr(),
                                                                 y.data_ptr(), B, Hi, Wi, _stream(self.dev)), "fvit_stem_conv3x3s2_px")
                hi, lo = self._conv_px(y, None, w1, b1, 2, 1)
                f32 = None
            else:
                hi, lo, f32 = x
            for li, (lvl, e) in enumerate(zip(levels, t["levels"])):
                if li < lv_from or (lv_to is not None and li >= lv_to):
                    continue
                if "blocks" in e:
                    if hi is None:   # a conv level behind a transformer level (no reference entrypoint does this): split the fp32 map
                        hi = f32.to(self.dtype)
                        lo = (f32 - hi.float()).to(self.dtype)
                        f32 = None
                    for wa, ba, wb, bb in e["blocks"]:
                        y, _ = self._conv_px(hi, None, wa, ba, 1, 2, want="single")          # conv1 + BN + GELU: an operand, one term
                        hi, lo = self._conv_px(y, None, wb, bb, 1, 0, res=hi, res_lo=lo)     # conv2 + BN (+ gamma) + residual, in place on the stream
                else:
                    if f32 is None:   # a transformer level behind a conv level without a Downsample in between (no reference entrypoint does this)
                        f32 = hi.float() + lo.float()
                    creal = lvl.blocks[0].attn.qkv.in_features if len(lvl.blocks) else f32.shape[1]
                    xin = f32[:, :creal] if f32.shape[1] != creal else f32
                    cpo = self._cp(creal) if "down" in e else creal
                    if cpo != creal:
                        xo = torch.empty((f32.shape[0], cpo, f32.shape[2], f32.shape[3]), dtype=torch.float32, device=f32.device,
                                         memory_format=torch.channels_last)
                        xo[:, creal:] = 0
                        hat_runtime.stage_forward(lvl, xin, out=xo[:, :creal])
                        f32 = xo
                    else:
                        # (an explicit channels_last output: empty_like of the strided channel slice `xin` would be NCHW-contiguous -- uncoalesced
                        # window_reverse stores and no dense [B][HW][C] image for the pool kernel)
                        xo = torch.empty((f32.shape[0], creal, f32.shape[2], f32.shape[3]), dtype=torch.float32, device=f32.device,
                                         memory_format=torch.channels_last)
                        f32 = hat_runtime.stage_forward(lvl, xin, out=xo)
                    hi = lo = None
                if "down" in e:
                    lw, lb, eps, wd, cin = e["down"]
                    if f32 is not None and not f32.is_contiguous(memory_format=torch.channels_last):
                        f32 = f32.contiguous(memory_format=torch.channels_last)
                    nh, nl = self._ln2d_px(hi, lo, f32, lw, lb, eps, cin)
                    nxt_transformer = li + 1 < len(levels) and levels[li + 1].transformer_block
                    if nxt_transformer:
                        f32 = self._conv_px(nh, nl, wd, None, 2, 0, want="f32")
                        hi = lo = None
                    else:
                        hi, lo = self._conv_px(nh, nl, wd, None, 2, 0)
                        f32 = None
            if lv_to is not None:
                return hi, lo, f32
            hw, hb, ln = t["head"]
            if f32 is None:
                f32 = hi.float() + lo.float()
            if ln is not None:
                nh, nl = self._ln2d_px(None, None, f32.contiguous(memory_format=torch.channels_last), *ln, f32.shape[1])
                f32 = nh.float() + nl.float()
            return self._pool_head(f32, hw, hb)

    def _forward_one(self, x, lv_from=0, lv_to=None):
        """Levels [lv_from, lv_to) of the plan; the stem runs in front of level 0, final norm + pool + head after the last level
        (lv_to = None).  A partial call returns the (channels_last, 16-bit) map that the next level takes."""
        if self.precise:
            return self._forward_one_precise(x, lv_from, lv_to)
        t = self.t
        with torch.autocast(device_type="cuda", enabled=False):
            w0, b0, w1, b1 = t["stem"]
            wk1 = w1[1]
            if lv_from > 0:
                pass
            elif (self.fused_stem and t["stem_k"] is not None and x.shape[1] == 3 and x.dtype in hat_runtime._DT and wk1 is not None
                    and w1[3] == 1 and tuple(wk1.shape) == (64, 3, 3, 64)):
                B, _, Hi, Wi = x.shape
                H1, W1 = (Hi - 1) // 2 + 1, (Wi - 1) // 2 + 1
                y = torch.empty((B, 64, (H1 - 1) // 2 + 1, (W1 - 1) // 2 + 1), dtype=self.dtype, device=x.device,
                                memory_format=torch.channels_last)
                view = hat_runtime._map_view(x)
                _lib.check(_lib.lib().fvit_stem_fused(self.code, view, t["stem_k"].data_ptr(), b0.data_ptr(), wk1.data_ptr(),
                                                      b1.data_ptr(), y.data_ptr(), B, Hi, Wi, _stream(self.dev)), "fvit_stem_fused")
                x = y
            elif t["stem_k"] is not None and x.shape[1] == 3 and x.dtype in hat_runtime._DT:
                B, _, Hi, Wi = x.shape   # fused stem kernel reads the caller's image in place (any strides, fp32/16-bit)
                y = torch.empty((B, 64, (Hi - 1) // 2 + 1, (Wi - 1) // 2 + 1), dtype=self.dtype, device=x.device,
                                memory_format=torch.channels_last)
                view = hat_runtime._map_view(x)
                _lib.check(_lib.lib().fvit_stem_conv3x3s2(self.code, view, t["stem_k"].data_ptr(), b0.data_ptr(), y.data_ptr(),
                                                          B, Hi, Wi, _stream(self.dev)), "fvit_stem_conv3x3s2")
                x = self._conv(y, w1, b1, 2, 1)
            else:
                x = x.to(self.dtype).contiguous(memory_format=torch.channels_last)
                x = self._conv(self._conv(x, w0, b0, 2, 1), w1, b1, 2, 1)
            for li, (lvl, e) in enumerate(zip(self.model.levels, t["levels"])):
                if li < lv_from or (lv_to is not None and li >= lv_to):
                    continue
                if "blocks" in e:
                    for wa, ba, wb, bb in e["blocks"]:
                        y = self._conv(x, wa, ba, 1, 2)
                        x = self._conv(y, wb, bb, 1, 0, residual=x)
                else:
                    # the HIP stage reads / writes its maps through strided views: the padded map's first C channels in, and
                    # -- when a Downsample follows -- the first C channels of a zero-initialised padded map out
                    creal = lvl.blocks[0].attn.qkv.in_features if len(lvl.blocks) else x.shape[1]
                    xin = x[:, :creal] if x.shape[1] != creal else x
                    cpo = self._cp(creal) if "down" in e else creal
                    if cpo != creal:
                        xo = torch.empty((x.shape[0], cpo, x.shape[2], x.shape[3]), dtype=self.dtype, device=x.device,
                                         memory_format=torch.channels_last)
                        xo[:, creal:] = 0   # only the pad channels need initialising; the stage writes the first creal
                        hat_runtime.stage_forward(lvl, xin, out=xo[:, :creal])  # TokenInitializer: fvit_token_init in both modes
                        x = xo
                    else:
                        xo = torch.empty((x.shape[0], creal, x.shape[2], x.shape[3]), dtype=self.dtype, device=x.device, memory_format=torch.channels_last)
                        x = hat_runtime.stage_forward(lvl, xin, out=xo)   # (explicit channels_last output: see _forward_one_precise)
                if "down" in e:
                    lw, lb, eps, wd, cin = e["down"]
                    x = self._conv(self._ln2d(x, lw, lb, eps, cin), wd, None, 2, 0)
            if lv_to is not None:
                return x
            hw, hb, ln = t["head"]
            if ln is not None:
                x = self._ln2d(x, *ln)
            return self._pool_head(x, hw, hb)


class ShardRunner:
    """Throughput driver for steady-state inference: the batch is split into n shards, each shard's whole forward is captured in its
    OWN hipGraph on its OWN stream, and ``launch()`` enqueues one replay per stream without any fork / join between steps.

    ``DeployPlan.forward`` (fork / join inside one graph) starts all shards of a step together and ends the step when the slowest
    is done, so the shards walk through the network in lockstep: three stems, then three level-0 conv stacks, ..., then three
    carrier-token branches (22 workgroups each) at the same time.  Independent per-stream graphs let consecutive steps of different
    shards overlap: the streams drift apart and an underfilled phase of one shard (carrier branch, stage 3) runs beside a
    chip-filling phase of another (convs).  Images are independent, so there is no data hazard; ``wait()`` joins all streams.
    Results of the LAST launch are in ``outputs()`` (static buffers, as with any graph replay)."""

    def __init__(self, plan, x, n):
        self.plan = plan
        self.n = n = max(1, min(int(n), x.shape[0]))
        self.streams = [torch.cuda.Stream(device=x.device) for _ in range(n)]
        self.inputs = [p.clone() for p in x.chunk(n, dim=0)]
        self.graphs, self.outs = [], []
        torch.cuda.synchronize()
        for i, (st, xi) in enumerate(zip(self.streams, self.inputs)):
            with torch.cuda.stream(st), torch.no_grad(), hat_runtime.workspace_slot(i):
                for _ in range(2):           # warm-up on this stream: packs weights, sizes this slot's workspaces
                    plan.forward_single(xi)
            st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(g, stream=st), hat_runtime.workspace_slot(i):
                y = plan.forward_single(xi)
            self.graphs.append(g)
            self.outs.append(y)
        torch.cuda.synchronize()

    def set_input(self, x):
        for dst, src in zip(self.inputs, x.chunk(self.n, dim=0)):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()

    def launch(self):
        for st, g in zip(self.streams, self.graphs):
            with torch.cuda.stream(st):
                g.replay()

    def wait(self):
        for st in self.streams:
            st.synchronize()

    def outputs(self):
        self.wait()
        return torch.cat(self.outs, dim=0)
