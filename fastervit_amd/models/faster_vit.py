"""FasterViT module tree for MI355X: same Python API and ``state_dict`` layout as the reference
(``fastervit/models/faster_vit.py`` = FV, ``faster_vit_any_res.py`` = AR), different engine.

* The conv side -- PatchEmbed (FV:443-469), ConvBlock (FV:472-512), Downsample (FV:410-440),
  TokenInitializer (FV:704-738), the final norm / pool / head (FV:925-960) -- stays PyTorch-ROCm
  modules, as BASELINE.json's north_star prescribes.
* The transformer stages -- window_partition, HAT blocks, window_reverse (FV:83-109, 515-701,
  832-841) -- hold only parameters here; their arithmetic runs in hand-written gfx950 kernels
  behind the C ABI of ``include/fvit_hip.h`` (see ``fastervit_amd/hat_runtime.py``).  There is no
  CPU or eager fallback for them: calling a transformer stage without the HIP library or on a
  non-GPU tensor raises RuntimeError.

One implementation covers both reference files: the any-res file is the base file generalised to
list-valued resolutions and is bit-identical at square resolutions.  The
``any_res`` flag selects the two places where the files differ in behaviour (tokenizer rule
FV:821 vs AR:837; optional ``hat_pos_embed`` AR:658).
"""
from __future__ import annotations

import math
import os
from pathlib import Path

import torch
import torch.nn as nn
import torch.nn.functional as F

from .registry import register_pip_model

_TIMM_PENDING = []  # entrypoints to (re-)register with timm, see register_with_timm()


def _timm_register(fn):
    """The reference also registers every entrypoint with timm (FV:975-976: @register_model) so that
    timm.models.create_model -- what validate.py calls -- finds it.  timm is optional here."""
    _TIMM_PENDING.append(fn)
    try:
        from timm.models.registry import register_model
        register_model(fn)
    except Exception:
        pass
    return fn


def register_with_timm():
    """(Re-)register all entrypoints with whatever `timm` is importable now; returns how many."""
    from timm.models.registry import register_model
    for fn in _TIMM_PENDING:
        register_model(fn)
    return len(_TIMM_PENDING)


def _pair(v):
    return [int(v[0]), int(v[1])] if isinstance(v, (tuple, list)) else [int(v), int(v)]


# --------------------------------------------------------------------------------------------
# conv side (PyTorch-ROCm / MIOpen)
# --------------------------------------------------------------------------------------------
class LayerNorm2d(nn.LayerNorm):
    """LayerNorm over the channel dim of an NCHW map; stand-in for timm's LayerNorm2d
    (eps 1e-6, used by Downsample FV:432 and the optional final norm FV:925)."""

    def __init__(self, num_channels, eps=1e-6, affine=True):
        super().__init__(num_channels, eps=eps, elementwise_affine=affine)

    def forward(self, x):
        y = F.layer_norm(x.permute(0, 2, 3, 1), self.normalized_shape, self.weight, self.bias, self.eps)
        return y.permute(0, 3, 1, 2)


class Downsample(nn.Module):
    """FV:410-440: LayerNorm2d then a stride-2 3x3 conv (bias-free) doubling the channels."""

    def __init__(self, dim, keep_dim=False):
        super().__init__()
        self.norm = LayerNorm2d(dim)
        self.reduction = nn.Sequential(nn.Conv2d(dim, dim if keep_dim else 2 * dim, 3, 2, 1, bias=False))

    def forward(self, x):
        return self.reduction(self.norm(x))


class PatchEmbed(nn.Module):
    """FV:443-469: stem, two stride-2 3x3 convs each followed by BatchNorm(eps 1e-4) + ReLU."""

    def __init__(self, in_chans=3, in_dim=64, dim=96):
        super().__init__()
        self.proj = nn.Identity()
        self.conv_down = nn.Sequential(
            nn.Conv2d(in_chans, in_dim, 3, 2, 1, bias=False), nn.BatchNorm2d(in_dim, eps=1e-4), nn.ReLU(),
            nn.Conv2d(in_dim, dim, 3, 2, 1, bias=False), nn.BatchNorm2d(dim, eps=1e-4), nn.ReLU())

    def forward(self, x):
        return self.conv_down(self.proj(x))


class DropPath(nn.Module):
    """Stochastic depth per sample (timm.layers.DropPath as the reference uses it, FV:21, 497, 630, 652): in train mode every sample of dim 0 is kept
    with probability 1 - drop_prob and scaled by 1 / (1 - drop_prob); identity in eval mode.  No parameters, no buffers (the state_dict is unaffected).
    Inside the HAT stages the HIP path does not call this module: it reads ``drop_prob`` and applies the same draw as a per-row factor of the
    residual update (fastervit_amd.hat_backward.drop_path_masks)."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if self.drop_prob <= 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * (mask / keep)

    def extra_repr(self):
        return f"drop_prob={self.drop_prob:.3f}"


def _drop_path(p):
    return DropPath(p) if p and p > 0.0 else nn.Identity()   # FV:497, 630, 652


class ConvBlock(nn.Module):
    """FV:472-512: conv-BN-GELU-conv-BN with an (optionally gamma-scaled) residual."""

    def __init__(self, dim, drop_path=0., layer_scale=None, kernel_size=3):
        super().__init__()
        self.conv1 = nn.Conv2d(dim, dim, kernel_size, 1, 1)
        self.norm1 = nn.BatchNorm2d(dim, eps=1e-5)
        self.act1 = nn.GELU()
        self.conv2 = nn.Conv2d(dim, dim, kernel_size, 1, 1)
        self.norm2 = nn.BatchNorm2d(dim, eps=1e-5)
        self.layer_scale = layer_scale is not None and type(layer_scale) in (int, float)
        if self.layer_scale:
            self.gamma = nn.Parameter(layer_scale * torch.ones(dim))
        self.drop_path = _drop_path(drop_path)

    def forward(self, x, global_feature=None):
        y = self.norm2(self.conv2(self.act1(self.norm1(self.conv1(x)))))
        if self.layer_scale:
            y = y * self.gamma.view(1, -1, 1, 1)
        return x + self.drop_path(y), global_feature   # FV:505-512


class TokenInitializer(nn.Module):
    """FV:704-738 / AR:710-750: carrier-token initialiser (depthwise 3x3 conv + average pool)."""

    def __init__(self, dim, input_resolution, window_size, ct_size=1):
        super().__init__()
        res = _pair(input_resolution)
        ks, ss = [], []
        for r in res:
            out = int(ct_size * r / window_size)
            stride = int(r / out)
            ks.append(r - (out - 1) * stride)
            ss.append(stride)
        self.pos_embed = nn.Conv2d(dim, dim, 3, padding=1, groups=dim)
        seq = nn.Sequential()
        seq.add_module("pos", self.pos_embed)  # same conv under two names, as in the reference
        seq.add_module("pool", nn.AvgPool2d(kernel_size=tuple(ks), stride=tuple(ss)))
        self.to_global_feature = seq
        self.window_size = ct_size

    def forward(self, x):
        y = self.to_global_feature(x)
        B, C, H, W = y.shape
        cw = self.window_size
        y = y.reshape(B, C, H // cw, cw, W // cw, cw)
        return y.permute(0, 2, 4, 3, 5, 1).reshape(B, H * W, C)


# --------------------------------------------------------------------------------------------
# transformer side: parameter holders + constant folding; arithmetic is in libfvit_hip.so
# --------------------------------------------------------------------------------------------
class PosEmbMLPSwinv1D(nn.Module):
    """FV:313-367: absolute position embedding by a 2->512->dim MLP over a normalised grid.

    The reference recomputes it on every forward (deploy is never switched on); it is input
    independent in eval, so ``table()`` is evaluated once per weight version and the add is fused
    into the LayerNorm kernel."""

    def __init__(self, dim, rank=2, seq_length=4, conv=False):
        super().__init__()
        if rank != 2 or conv:
            raise NotImplementedError("only rank=2, conv=False is used by FasterViT")
        self.rank = rank
        self.cpb_mlp = nn.Sequential(nn.Linear(rank, 512, bias=True), nn.ReLU(), nn.Linear(512, dim, bias=False))
        self.register_buffer("relative_bias", torch.zeros(1, seq_length, dim))
        self.seq_length = seq_length
        self.deploy = False

    def switch_to_deploy(self):
        self.deploy = True

    @torch.no_grad()
    def table(self, seq_length=None):
        """(L*L, dim) fp32 on the parameters' device."""
        L = int((seq_length or self.seq_length) ** 0.5)
        w0 = self.cpb_mlp[0].weight
        ax = torch.arange(L, device=w0.device, dtype=torch.float32)
        grid = torch.stack(torch.meshgrid(ax, ax, indexing="ij")).reshape(2, L * L).t()
        grid = (grid - (L // 2)) / (L // 2)
        with torch.autocast(device_type=w0.device.type, enabled=False):  # fp32 by contract (the kernels read f32)
            h = torch.relu(F.linear(grid, w0.float(), self.cpb_mlp[0].bias.float()))
            return F.linear(h, self.cpb_mlp[2].weight.float()).float().contiguous()


class PosEmbMLPSwinv2D(nn.Module):
    """FV:213-310: log-spaced continuous relative position bias (2->512->heads MLP), 16*sigmoid,
    zero-padded top/left for the carrier tokens.  Folded by ``table()``."""

    def __init__(self, window_size, pretrained_window_size, num_heads, seq_length, ct_correct=False, no_log=False):
        super().__init__()
        if ct_correct or no_log:
            raise NotImplementedError("ct_correct / no_log are never enabled by the reference")
        self.window_size = list(window_size)
        self.num_heads = num_heads
        self.seq_length = seq_length
        self.cpb_mlp = nn.Sequential(nn.Linear(2, 512, bias=True), nn.ReLU(inplace=True),
                                     nn.Linear(512, num_heads, bias=False))
        w0, w1 = self.window_size
        p0, p1 = pretrained_window_size
        rh = torch.arange(-(w0 - 1), w0, dtype=torch.float32)
        rw = torch.arange(-(w1 - 1), w1, dtype=torch.float32)
        tab = torch.stack(torch.meshgrid(rh, rw, indexing="ij"), dim=-1).unsqueeze(0).contiguous()
        tab[..., 0] /= ((p0 if p0 > 0 else w0) - 1)
        tab[..., 1] /= ((p1 if p1 > 0 else w1) - 1)
        tab *= 8
        tab = torch.sign(tab) * torch.log2(torch.abs(tab) + 1.0) / math.log2(8)
        self.register_buffer("relative_coords_table", tab)
        yy, xx = torch.meshgrid(torch.arange(w0), torch.arange(w1), indexing="ij")
        flat = torch.stack([yy.reshape(-1), xx.reshape(-1)])
        rel = flat[:, :, None] - flat[:, None, :]
        idx = (rel[0] + w0 - 1) * (2 * w1 - 1) + (rel[1] + w1 - 1)
        self.register_buffer("relative_position_index", idx)
        self.register_buffer("relative_bias", torch.zeros(1, num_heads, seq_length, seq_length))
        self.deploy = False

    def switch_to_deploy(self):
        self.deploy = True

    @torch.no_grad()
    def table(self, S):
        """(heads, S, S) fp32: bias on the trailing window block, zeros on the first S - w0*w1 rows/cols."""
        n = self.window_size[0] * self.window_size[1]
        w0 = self.cpb_mlp[0].weight
        with torch.autocast(device_type=w0.device.type, enabled=False):
            h = torch.relu(F.linear(self.relative_coords_table.float(), w0.float(), self.cpb_mlp[0].bias.float()))
            t = F.linear(h, self.cpb_mlp[2].weight.float()).float().view(-1, self.num_heads)
        b = t[self.relative_position_index.view(-1)].view(n, n, -1).permute(2, 0, 1)
        b = 16 * torch.sigmoid(b)
        pad = S - n
        return F.pad(b, (pad, 0, pad, 0)).contiguous()

    @torch.no_grad()
    def rel_table(self):
        """((heads, (2*w0-1)*(2*w1-1)) fp32, w0): 16*sigmoid(cpb_mlp(relative_coords_table)) BEFORE the relative_position_index
        gather (FV:276-280), head-major -- what the long-window attention kernel indexes arithmetically (square windows only)."""
        w0, w1 = self.window_size
        if w0 != w1:
            raise NotImplementedError("long-window attention expects square windows (the reference only builds those)")
        m0 = self.cpb_mlp[0].weight
        with torch.autocast(device_type=m0.device.type, enabled=False):
            h = torch.relu(F.linear(self.relative_coords_table.float(), m0.float(), self.cpb_mlp[0].bias.float()))
            t = F.linear(h, self.cpb_mlp[2].weight.float()).float().view(-1, self.num_heads)
        return (16 * torch.sigmoid(t)).t().contiguous(), int(w0)


class Mlp(nn.Module):
    """FV:370-407 parameter holder (fc1 -> GELU -> fc2); executed by the fused GEMM kernels."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        if act_layer is not nn.GELU:
            raise NotImplementedError("the HIP MLP epilogue implements exact-erf GELU only")
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features or in_features)
        self.drop = nn.Dropout(drop)


class WindowAttention(nn.Module):
    """FV:515-568 parameter holder (qkv, proj, relative position bias MLP)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., resolution=0,
                 seq_length=0):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = qk_scale or self.head_dim ** -0.5   # FV:538; reaches the kernels through FvitStageDesc.qk_scale
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.pos_emb_funct = PosEmbMLPSwinv2D(window_size=[resolution, resolution],
                                              pretrained_window_size=[resolution, resolution],
                                              num_heads=num_heads, seq_length=seq_length)
        self.resolution = resolution


class HAT(nn.Module):
    """Hierarchical attention block (FV:571-701 / AR:572-707).

    ``forward(x, carrier_tokens)`` keeps the reference signature -- x: (B*nW, ws^2, C) windows,
    carrier_tokens: (B, G, C) or None -- and runs ``fvit_hat_block_forward``.  Inside a
    FasterViTLayer the whole stage is issued by one ``fvit_hat_stage_forward`` call instead."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, sr_ratio=1., window_size=7, last=False, layer_scale=None,
                 ct_size=1, do_propagation=False, any_res=False):
        super().__init__()
        if norm_layer is not nn.LayerNorm:
            raise NotImplementedError("the HIP path implements nn.LayerNorm only")
        sr = _pair(sr_ratio)
        self.sr_ratio = sr
        self.square = sr[0] == sr[1]
        self.do_sr_hat = sr[0] > 1 or sr[1] > 1
        self.cr_window = ct_size
        self.window_size = window_size
        self.last = last
        self.do_propagation = do_propagation
        per_window = ct_size ** 2 if self.do_sr_hat else 0
        total = per_window * sr[0] * sr[1]
        hidden = int(dim * mlp_ratio)
        use_ls = layer_scale is not None and type(layer_scale) in (int, float)

        self.pos_embed = PosEmbMLPSwinv1D(dim, rank=2, seq_length=window_size ** 2)
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                                    proj_drop=drop, resolution=window_size, seq_length=window_size ** 2 + per_window)
        self.drop_path = _drop_path(drop_path)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=hidden, act_layer=act_layer, drop=drop)
        self.gamma3 = nn.Parameter(layer_scale * torch.ones(dim)) if use_ls else 1
        self.gamma4 = nn.Parameter(layer_scale * torch.ones(dim)) if use_ls else 1
        if self.do_sr_hat:
            self.hat_norm1 = norm_layer(dim)
            self.hat_norm2 = norm_layer(dim)
            self.hat_attn = WindowAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                            attn_drop=attn_drop, proj_drop=drop, resolution=int(total ** 0.5),
                                            seq_length=total)
            self.hat_mlp = Mlp(in_features=dim, hidden_features=hidden, act_layer=act_layer, drop=drop)
            self.hat_drop_path = _drop_path(drop_path)
            if self.square or not any_res:
                self.hat_pos_embed = PosEmbMLPSwinv1D(dim, rank=2, seq_length=total)
            self.gamma1 = nn.Parameter(layer_scale * torch.ones(dim)) if use_ls else 1
            self.gamma2 = nn.Parameter(layer_scale * torch.ones(dim)) if use_ls else 1
            self.upsampler = nn.Upsample(size=window_size, mode="nearest")

    def forward(self, x, carrier_tokens):
        from .. import hat_runtime
        return hat_runtime.block_forward(self, x, carrier_tokens)


class FasterViTLayer(nn.Module):
    """One of the four levels (FV:741-843 / AR:753-870): conv blocks, or the HAT transformer stage."""

    def __init__(self, dim, depth, input_resolution, num_heads, window_size, ct_size=1, conv=False, downsample=True,
                 mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0., attn_drop=0., drop_path=0., layer_scale=None,
                 layer_scale_conv=None, only_local=False, hierarchy=True, do_propagation=False, any_res=False):
        super().__init__()
        self.conv = conv
        self.transformer_block = not conv
        self.window_size = window_size
        self.any_res = any_res
        res = _pair(input_resolution)
        base_res = res[0]
        if any_res:  # AR:806-808: blocks are built for the resolution padded to a window multiple
            res = [r + (window_size - r % window_size) % window_size for r in res]
        self.input_resolution = res
        if conv:
            self.blocks = nn.ModuleList([
                ConvBlock(dim=dim, drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                          layer_scale=layer_scale_conv) for i in range(depth)])
            sr = [1, 1]
        else:
            sr = [1, 1] if only_local else [res[0] // window_size, res[1] // window_size]
            self.blocks = nn.ModuleList([
                HAT(dim=dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop,
                    attn_drop=attn_drop, drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                    sr_ratio=sr, window_size=window_size, last=(i == depth - 1), layer_scale=layer_scale,
                    ct_size=ct_size, do_propagation=do_propagation, any_res=any_res) for i in range(depth)])
        self.sr_ratio = sr
        self.downsample = Downsample(dim=dim) if downsample else None
        if any_res:   # AR:837: a non-empty list is always truthy
            gt = len(self.blocks) > 0 and not only_local and hierarchy and not conv
        else:         # FV:821
            gt = len(self.blocks) > 0 and not only_local and base_res // window_size > 1 and hierarchy and not conv
        self.do_gt = bool(gt)
        if self.do_gt:
            self.global_tokenizer = TokenInitializer(dim, res, window_size, ct_size=ct_size)

    def forward(self, x):
        if self.transformer_block:
            from .. import hat_runtime
            want_grad = torch.is_grad_enabled() and x.is_cuda and not self.training
            if self.training and x.is_cuda and len(self.blocks):
                # TRAIN mode: the unit-kernel chain with stochastic depth as one autograd node (the fused inference kernels have eval semantics);
                # raises here if the geometry is not covered (hat_backward.backward_unsupported_reason)
                from .. import hat_backward
                x = hat_backward.stage_forward_with_grad(self, x)
            elif want_grad and self.__dict__.get("hat_backward", False):
                from .. import hat_backward   # the stage as one autograd node: HIP forward, kernel-sequence backward (FasterViT.enable_hat_backward)
                x = hat_backward.stage_forward_with_grad(self, x)   # raises HERE (forward time) if the geometry has no backward
            elif want_grad and len(self.blocks) and ((x.requires_grad and not x.is_leaf) or any(p.requires_grad for p in self.blocks.parameters())):
                # eval-mode forward without torch.no_grad(): the stage input carries a graph (the conv side's parameters require grad) or the stage's
                # OWN parameters do (frozen conv side, trainable HAT blocks: ADVICE r04).  Where the kernel-sequence backward covers the stage the
                # gradient flows (PyTorch's own behaviour); elsewhere hat_runtime warns and detaches (FasterViT._check_grad_request raises first when
                # the caller's own input requires grad).
                from .. import hat_backward
                if hat_backward.backward_unsupported_reason(self, x.shape[2], x.shape[3]) is None:
                    x = hat_backward.stage_forward_with_grad(self, x)
                else:
                    x = hat_runtime.stage_forward(self, x)
            else:
                x = hat_runtime.stage_forward(self, x)
        else:
            for blk in self.blocks:
                x, _ = blk(x, None)
        return x if self.downsample is None else self.downsample(x)


class FasterViT(nn.Module):
    """FV:846-972 / AR:873-1002."""

    def __init__(self, dim, in_dim, depths, window_size, ct_size, mlp_ratio, num_heads, resolution=224,
                 drop_path_rate=0.2, in_chans=3, num_classes=1000, qkv_bias=True, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., layer_scale=None, layer_scale_conv=None, layer_norm_last=False,
                 hat=(False, False, True, False), do_propagation=False, any_res=False, **kwargs):
        super().__init__()
        res = _pair(resolution)
        num_features = int(dim * 2 ** (len(depths) - 1))
        self.num_classes = num_classes
        self.any_res = any_res
        self.patch_embed = PatchEmbed(in_chans=in_chans, in_dim=in_dim, dim=dim)
        n_blocks = sum(depths)  # stochastic-depth schedule (identity in eval); plain Python so meta-device builds work
        dpr = [drop_path_rate * i / max(n_blocks - 1, 1) for i in range(n_blocks)]
        if hat is None:
            hat = [True] * len(depths)
        self.levels = nn.ModuleList()
        for i in range(len(depths)):
            lvl_res = [int(2 ** (-2 - i) * res[0]), int(2 ** (-2 - i) * res[1])]
            self.levels.append(FasterViTLayer(
                dim=int(dim * 2 ** i), depth=depths[i], num_heads=num_heads[i], window_size=window_size[i],
                ct_size=ct_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, conv=(i < 2), drop=drop_rate,
                attn_drop=attn_drop_rate, drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], downsample=(i < 3),
                layer_scale=layer_scale, layer_scale_conv=layer_scale_conv, input_resolution=lvl_res,
                only_local=not hat[i], do_propagation=do_propagation, any_res=any_res))
        self.norm = LayerNorm2d(num_features) if layer_norm_last else nn.BatchNorm2d(num_features)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.head = nn.Linear(num_features, num_classes) if num_classes > 0 else nn.Identity()
        self.apply(self._init_weights)
        self.hat_operand_dtype = "f16"  # MFMA operand type of the HIP path: "f16" (default) or "bf16"

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):  # covers LayerNorm2d
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {"rpb"}

    def set_hat_operand_dtype(self, name: str):
        """Choose the operand mode of the HAT kernels (fp32 accumulate): 'f16' (default) or 'bf16' -- 16-bit operands rounded once --
        or 'f16x2' / 'bf16x2' -- every Linear weight as two 16-bit terms hi + lo (twice the MFMA work on the weights' side; the
        route to logits max-abs < 1e-3 with bf16 operands) -- or 'f16x3' / 'bf16x3' -- weights AND activations
        as two terms (three times the MFMA work, unfused kernel chain): ~22 significant bits through the HAT stages, the route to
        logits max-abs < 1e-3 ABSOLUTE on FasterViT-4 / any-res (measured 2.6e-5 on |7.1| with the fp32 conv side)."""
        from ..hat_runtime import OPERAND_MODES, x3_unsupported_reason
        if name not in OPERAND_MODES:
            raise ValueError(f"operand mode must be one of {OPERAND_MODES}")
        if name.endswith("x3"):
            # geometry the two-term-activation kernels do not cover is refused HERE, by name, not as an error of the first forward
            for li, lvl in enumerate(self.levels):
                why = x3_unsupported_reason(lvl) if lvl.transformer_block and len(lvl.blocks) else None
                if why is not None:
                    raise NotImplementedError(f"operand mode {name!r}, level {li}: {why}")
        self.hat_operand_dtype = name
        for lvl in self.levels:
            lvl.hat_operand_dtype = name
            if lvl.transformer_block:
                for blk in lvl.blocks:   # the block-level API (HAT.forward -> hat_runtime.block_forward) reads the mode from the block
                    blk.hat_operand_dtype = name
        return self

    def switch_to_deploy(self, dtype=torch.float16, streams=1):
        """Opt-in inference plan for the conv side (fastervit_amd/conv_runtime.py): BatchNorm folded into the conv
        weights, 16-bit channels_last activations, fused bias/activation/residual/LayerNorm2d HIP passes.
        ``forward`` then returns fp32 logits for GPU inputs; ``switch_to_deploy(None)`` goes back to module mode."""
        if dtype is None:
            self.__dict__.pop("_deploy_plan", None)
            return self
        from ..conv_runtime import DeployPlan
        plan = DeployPlan(self, dtype)
        plan.streams = int(streams)   # > 1: the batch runs as that many shards on separate HIP streams (see DeployPlan.forward)
        self.__dict__["_deploy_plan"] = plan
        return self

    def _check_grad_request(self, x):
        if x.is_cuda and not self.training and torch.is_grad_enabled() and x.requires_grad:
            from .. import hat_backward
            from ..hat_runtime import check_user_input
            # map size in front of level i: two stride-2 stem convs, one stride-2 Downsample conv per level (all 3x3, pad 1)
            H, W = x.shape[-2], x.shape[-1]
            for _ in range(2):
                H, W = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            for lvl in self.levels:   # d/dx through every HAT stage AT THIS INPUT SIZE, or an error -- never a silently cut gradient
                if lvl.transformer_block and len(lvl.blocks) and hat_backward.backward_unsupported_reason(lvl, H, W) is not None:
                    check_user_input(x)
                if lvl.downsample is not None:
                    H, W = (H - 1) // 2 + 1, (W - 1) // 2 + 1

    def forward_features(self, x):
        self._check_grad_request(x)
        x = self.patch_embed(x)
        for level in self.levels:
            x = level(x)
        return self.norm(x)

    def forward_head(self, x):
        return self.head(torch.flatten(self.avgpool(x), 1))

    #: eval-mode forwards under ``torch.autocast`` (fp16 / bf16) with grad disabled run the conv side through the deploy plan too
    #: (same 16-bit arithmetic class as MIOpen under autocast, BatchNorm folded; logits come back in the autocast dtype as they do
    #: from the module path).  That is the configuration of the reference's ``validate.py --amp``: 2x faster with no call-site
    #: change.  Set ``model.auto_deploy = False`` (or FVIT_AUTO_DEPLOY=0) for the plain nn.Module path.
    auto_deploy = os.environ.get("FVIT_AUTO_DEPLOY", "1") != "0"
    auto_deploy_streams = 3   # stream shards of the automatic plan (batches of fewer than 2 x this many images run unsharded)

    def _has_hooks(self):
        """Forward (pre-)hooks on any submodule: the deploy plan calls kernels, not submodule forwards, and would bypass them."""
        from torch.nn.modules import module as _m
        if _m._global_forward_hooks or _m._global_forward_pre_hooks:
            return True
        return any(m._forward_hooks or m._forward_pre_hooks for m in self.modules())

    def _autocast_plan(self, x):
        if (not self.auto_deploy or self.training or not x.is_cuda or torch.is_grad_enabled() or not torch.is_autocast_enabled()
                or getattr(self, "_is_replica", False) or x.dim() != 4 or x.shape[1] != 3):
            return None
        dt = torch.get_autocast_dtype("cuda") if hasattr(torch, "get_autocast_dtype") else torch.get_autocast_gpu_dtype()
        if dt not in (torch.float16, torch.bfloat16) or next(self.parameters()).device != x.device:
            return None
        if not isinstance(self.head, (nn.Linear, nn.Identity)) or self._has_hooks():
            return None   # a replaced head or registered hooks: stay on the plain nn.Module path
        plans = self.__dict__.setdefault("_auto_plans", {})
        if dt not in plans:
            from ..conv_runtime import DeployPlan
            plans[dt] = DeployPlan(self, dt)
            plans[dt].streams = int(self.auto_deploy_streams)
        return plans[dt], dt

    def compile_inference(self, example: torch.Tensor, dtype=torch.float16, streams: int = 3, graph: bool = True, join_from=None, conv_down_terms=None,
                          precise: bool = False):
        """Throughput entry point: the deploy plan with ``streams`` stream shards captured ONCE into a hipGraph
        with static input / output buffers; returns a callable ``runner(x) -> logits`` that copies ``x`` in and replays the
        graph (any batch size up to the example's; shorter batches are zero-padded and the result sliced).  See
        ``fastervit_amd.inference.CompiledInference``."""
        from ..inference import CompiledInference
        return CompiledInference(self, example, dtype=dtype, streams=streams, graph=graph, join_from=join_from, conv_down_terms=conv_down_terms,
                                 precise=precise)

    def pipelined_inference(self, example: torch.Tensor, depth: int = 2, dtype=torch.float16, streams: int = 1, graph: bool = True, join_from=None,
                            conv_down_terms=None, precise: bool = False):
        """Throughput entry point with ``depth`` whole-batch steps in flight (r06; what ``bench.py`` times): ``depth`` runners of ``compile_inference``'s
        configuration, step k replayed on stream k % depth, so the tail of a step (last stage, head) overlaps the stem / conv levels of the next one.
        ``p.launch()`` enqueues one step and returns its runner (logits in ``runner.static_y`` after ``p.wait()``).  See
        ``fastervit_amd.inference.PipelinedInference``."""
        from ..inference import PipelinedInference
        return PipelinedInference(self, example, depth=depth, streams=streams, dtype=dtype, graph=graph, join_from=join_from, conv_down_terms=conv_down_terms,
                                  precise=precise)

    def enable_hat_backward(self, on: bool = True):
        """Make the transformer stages differentiable in EVAL mode: with grad enabled every HAT stage becomes ONE autograd node whose forward is the HIP
        inference path and whose backward is the kernel sequence of ``fastervit_amd.hat_backward`` (head_dim <= 96, windows and carrier grids of at most
        64 tokens: every entrypoint at 224 x 224; other geometries raise here or -- for a map size that does not fit -- at forward time).  In TRAIN mode
        (``model.train()``) the stages always run as such a node, with stochastic depth (``drop_path_rate``).  The conv stages, norms and head are
        ordinary PyTorch modules and differentiate as usual.  Off by default: eval mode is inference-only."""
        if on:
            from .. import hat_backward
            for i, lvl in enumerate(self.levels):
                if lvl.transformer_block and len(lvl.blocks):
                    why = hat_backward.backward_unsupported_reason(lvl)
                    if why is not None:
                        raise RuntimeError(f"enable_hat_backward: level {i} of this model has no kernel-sequence backward: {why}")
        self.__dict__["hat_backward"] = bool(on)
        for lvl in self.levels:
            if lvl.transformer_block:
                lvl.__dict__["hat_backward"] = bool(on)
        return self

    def forward(self, x):
        self._check_grad_request(x)   # eval on the GPU, forward-only HIP stages: a caller asking for d/dx gets the gradient or an error, not zeros
        plan = self.__dict__.get("_deploy_plan")
        # nn.DataParallel replicas share __dict__ with the original: the plan's folded weights live on the original's device, so
        # replicas run the module path (per-device HAT state in hat_runtime)
        if plan is not None and x.is_cuda and not self.training and not getattr(self, "_is_replica", False):
            if torch.is_grad_enabled() and (self.__dict__.get("hat_backward", False) or x.requires_grad):
                raise RuntimeError("FasterViT: the deploy plan (folded BatchNorm, 16-bit conv kernels) is inference-only and returns detached logits; "
                                   "call it under torch.no_grad(), or switch_to_deploy(None) to differentiate through the module path")
            return plan.forward(x)
        auto = self._autocast_plan(x)
        if auto is not None:
            return auto[0].forward(x).to(auto[1])
        return self.forward_head(self.forward_features(x))

    def _load_state_dict(self, pretrained, strict: bool = False):
        _load_checkpoint(self, pretrained, strict=strict)


def _load_checkpoint(model, filename, map_location="cpu", strict=False, logger=None):
    """Tolerant loader with the reference's key handling (FV:173-210): picks 'state_dict' / 'model',
    strips 'module.' and 'encoder.' prefixes; mismatches are reported, and fatal only if strict."""
    ckpt = torch.load(filename, map_location=map_location)
    if not isinstance(ckpt, dict):
        raise RuntimeError(f"No state_dict found in checkpoint file {filename}")
    sd = ckpt.get("state_dict", ckpt.get("model", ckpt))
    if next(iter(sd)).startswith("module."):
        sd = {k[7:]: v for k, v in sd.items()}
    if sorted(sd)[0].startswith("encoder"):
        sd = {k.replace("encoder.", ""): v for k, v in sd.items() if k.startswith("encoder.")}
    res = model.load_state_dict(sd, strict=False)
    missing = [k for k in res.missing_keys if "num_batches_tracked" not in k]
    if missing or res.unexpected_keys:
        msg = ("The model and loaded state dict do not match exactly\n"
               f"unexpected key in source state_dict: {', '.join(res.unexpected_keys)}\n"
               f"missing keys in source state_dict: {', '.join(missing)}\n")
        if strict:
            raise RuntimeError(msg)
        (logger.warning if logger is not None else print)(msg)
    return ckpt


# --------------------------------------------------------------------------------------------
# entrypoints (FV:975-1418).  Hyper-parameters as data; one factory builds every variant.
# --------------------------------------------------------------------------------------------
_HF = "https://huggingface.co/ahatamiz/FasterViT/resolve/main/"


def _cfg(url="", **kwargs):
    cfg = {"url": url, "num_classes": 1000, "input_size": (3, 224, 224), "pool_size": None, "crop_pct": 0.875,
           "interpolation": "bicubic", "fixed_input_size": True, "mean": (0.485, 0.456, 0.406),
           "std": (0.229, 0.224, 0.225)}
    cfg.update(kwargs)
    return cfg


# name -> (checkpoint file, crop_pct, input side, crop_mode)                      FV:35-80
_CKPT = {
    "0": ("fastervit_0_224_1k.pth.tar", 0.875, 224, "center"), "1": ("fastervit_1_224_1k.pth.tar", 1.0, 224, "center"),
    "2": ("fastervit_2_224_1k.pth.tar", 1.0, 224, "center"), "3": ("fastervit_3_224_1k.pth.tar", 1.0, 224, "center"),
    "4": ("fastervit_4_224_1k.pth.tar", 1.0, 224, "center"), "5": ("fastervit_5_224_1k.pth.tar", 1.0, 224, "center"),
    "6": ("fastervit_6_224_1k.pth.tar", 1.0, 224, "center"),
    "4_21k_224": ("fastervit_4_21k_224_w14.pth.tar", 0.95, 224, "squash"),
    "4_21k_384": ("fastervit_4_21k_384_w24.pth.tar", 1.0, 384, "squash"),
    "4_21k_512": ("fastervit_4_21k_512_w32.pth.tar", 1.0, 512, "squash"),
    "4_21k_768": ("fastervit_4_21k_768_w48.pth.tar", 0.93, 768, "squash"),
}

_H = [False, False, True, False]
_NOH = [False, False, False, False]
# variant -> defaults.  ls = layer_scale / do_propagation family (FasterViT-3 and up)
_ARCH = {
    "0": dict(depths=[2, 3, 6, 5], num_heads=[2, 4, 8, 16], window_size=[7, 7, 7, 7], dim=64, in_dim=64, dpr=0.2, ls=False, hat=_H, path="/tmp/faster_vit_0.pth.tar"),
    "1": dict(depths=[1, 3, 8, 5], num_heads=[2, 4, 8, 16], window_size=[7, 7, 7, 7], dim=80, in_dim=32, dpr=0.2, ls=False, hat=_H, path="/tmp/faster_vit_1.pth.tar"),
    "2": dict(depths=[3, 3, 8, 5], num_heads=[2, 4, 8, 16], window_size=[7, 7, 7, 7], dim=96, in_dim=64, dpr=0.2, ls=False, hat=_H, path="/tmp/faster_vit_2.pth.tar"),
    "3": dict(depths=[3, 3, 12, 5], num_heads=[2, 4, 8, 16], window_size=[7, 7, 7, 7], dim=128, in_dim=64, dpr=0.3, ls=True, hat=_H, path="/tmp/faster_vit_3.pth.tar"),
    "4": dict(depths=[3, 3, 12, 5], num_heads=[4, 8, 16, 32], window_size=[7, 7, 7, 7], dim=196, in_dim=64, dpr=0.3, ls=True, hat=_H, path="/tmp/faster_vit_4.pth.tar"),
    "5": dict(depths=[3, 3, 12, 5], num_heads=[4, 8, 16, 32], window_size=[7, 7, 7, 7], dim=320, in_dim=64, dpr=0.3, ls=True, hat=_H, path="/tmp/faster_vit_5.pth.tar"),
    "6": dict(depths=[3, 3, 16, 8], num_heads=[4, 8, 16, 32], window_size=[7, 7, 7, 7], dim=320, in_dim=64, dpr=0.5, ls=True, hat=_H, path="/tmp/faster_vit_6.pth.tar"),
    "4_21k_224": dict(depths=[3, 3, 12, 5], num_heads=[4, 8, 16, 32], window_size=[7, 7, 14, 7], dim=196, in_dim=64, dpr=0.42, ls=True, hat=_NOH, path="/tmp/fastervit_4_21k_224_w14.pth.tar"),
    "4_21k_384": dict(depths=[3, 3, 12, 5], num_heads=[4, 8, 16, 32], window_size=[7, 7, 24, 12], dim=196, in_dim=64, dpr=0.42, ls=True, hat=_NOH, path="/tmp/fastervit_4_21k_384_w24.pth.tar"),
    "4_21k_512": dict(depths=[3, 3, 12, 5], num_heads=[4, 8, 16, 32], window_size=[7, 7, 32, 16], dim=196, in_dim=64, dpr=0.42, ls=True, hat=_NOH, path="/tmp/fastervit_4_21k_512_w32.pth.tar"),
    "4_21k_768": dict(depths=[3, 3, 12, 5], num_heads=[4, 8, 16, 32], window_size=[7, 7, 48, 24], dim=196, in_dim=64, dpr=0.42, ls=True, hat=_NOH, path="/tmp/fastervit_4_21k_768_w48.pth.tar"),
}
# any-res differences (AR:1007-1440): resolution default, drop-path of the 21k models, cfg input size
_ANYRES_RESOLUTION = {"2": [541, 960]}


def _variant_cfg(variant, any_res):
    fname, crop, side, mode = _CKPT[variant]
    if any_res:
        side = 224  # AR:36-80: every any-res cfg keeps input_size (3, 224, 224)
    return _cfg(url=_HF + fname, crop_pct=crop, input_size=(3, side, side), crop_mode=mode)


default_cfgs = {f"faster_vit_{v}" + ("" if v.startswith("4_21k") else "_224"): _variant_cfg(v, False) for v in _ARCH}


def build_variant(variant: str, any_res: bool, name: str, pretrained=False, **kwargs):
    """Shared body of every entrypoint: pops the reference's kwargs (FV:978-988), builds the model,
    attaches pretrained_cfg / default_cfg, optionally loads ``model_path`` (FV:1002-1008)."""
    a = _ARCH[variant]
    depths = kwargs.pop("depths", a["depths"])
    num_heads = kwargs.pop("num_heads", a["num_heads"])
    window_size = kwargs.pop("window_size", a["window_size"])
    ct_size = kwargs.pop("ct_size", 2)
    dim = kwargs.pop("dim", a["dim"])
    in_dim = kwargs.pop("in_dim", a["in_dim"])
    mlp_ratio = kwargs.pop("mlp_ratio", 4)
    if any_res:
        resolution = kwargs.pop("resolution", _ANYRES_RESOLUTION.get(variant, [576, 960]))
        dpr_default = 0.3 if variant.startswith("4_21k") else a["dpr"]
    else:
        resolution = kwargs.pop("resolution", _CKPT[variant][2] if variant.startswith("4_21k") else 224)
        dpr_default = a["dpr"]
    drop_path_rate = kwargs.pop("drop_path_rate", dpr_default)
    extra = {}
    if a["ls"]:
        extra["layer_scale"] = kwargs.pop("layer_scale", 1e-5)
        extra["do_propagation"] = True
        if variant != "3":
            extra["layer_norm_last"] = False
        kwargs.pop("do_propagation", None)
    model_path = kwargs.pop("model_path", a["path"])
    hat = kwargs.pop("hat", a["hat"])
    for k in ("pretrained_cfg", "pretrained_cfg_overlay"):  # timm's create_model passes these
        kwargs.pop(k, None)
    pretrained_cfg = {k: v for k, v in _variant_cfg(variant, any_res).items() if v is not None}
    kwargs.setdefault("num_classes", pretrained_cfg["num_classes"])
    kwargs.setdefault("in_chans", pretrained_cfg["input_size"][0])
    if pretrained_cfg.get("fixed_input_size", False):
        kwargs.setdefault("img_size", pretrained_cfg["input_size"][-2:])
    model = FasterViT(depths=depths, num_heads=num_heads, window_size=window_size, ct_size=ct_size, dim=dim,
                      in_dim=in_dim, mlp_ratio=mlp_ratio, resolution=resolution, drop_path_rate=drop_path_rate,
                      hat=hat, any_res=any_res, **extra, **kwargs)
    model.pretrained_cfg = pretrained_cfg
    model.default_cfg = model.pretrained_cfg
    if pretrained:
        if not Path(model_path).is_file():
            torch.hub.download_url_to_file(url=model.default_cfg["url"], dst=model_path)
        model._load_state_dict(model_path)
    return model


def _make_entrypoint(variant, any_res, name, module_name):
    def entry(pretrained=False, **kwargs):
        return build_variant(variant, any_res, name, pretrained=pretrained, **kwargs)
    entry.__name__ = name
    entry.__qualname__ = name
    entry.__module__ = module_name
    entry.__doc__ = f"FasterViT entrypoint '{name}' (reference: {'AR' if any_res else 'FV'} @register_model {name})."
    return register_pip_model(_timm_register(entry))


for _v in _ARCH:
    _name = f"faster_vit_{_v}" + ("" if _v.startswith("4_21k") else "_224")
    globals()[_name] = _make_entrypoint(_v, False, _name, __name__)
del _v, _name
