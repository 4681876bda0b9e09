"""ctypes binding of libfvit_hip.so (C ABI declared in include/fvit_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the HIP library is missing or fails to
load, everything that needs it raises RuntimeError.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(_HERE, "csrc")
DIAG = os.environ.get("FVIT_DIAG", "0") == "1"   # diagnosis build (scripts/timeline_*.py, scripts/poison_check.py, the poison test's subprocess)
LIB_PATH = os.path.join(CSRC_DIR, "libfvit_hip_diag.so" if DIAG else "libfvit_hip.so")
# A/B measurements only (scripts/r06_calls): another build of the same ABI, e.g. the previous round's kernels, selected per process
if os.environ.get("FVIT_LIB_PATH"):
    LIB_PATH = os.path.abspath(os.environ["FVIT_LIB_PATH"])

FVIT_ABI_VERSION = 8
FVIT_F32, FVIT_F16, FVIT_BF16 = 0, 1, 2
FVIT_TILE_N, FVIT_TILE_K = 128, 64
FVIT_MASK_BIAS = -30000.0
FVIT_MAX_DENSE_SEQ = 208   # longer window sequences use the online-softmax attention kernel with the compact bias table
FVIT_PROF_KINDS = 11

# every symbol include/fvit_hip.h declares (checked by tests/test_abi.py without a GPU)
EXPORTED_SYMBOLS = (
    "fvit_abi_version", "fvit_last_error", "fvit_attention_spad", "fvit_attention_dense", "fvit_stage_workspace_bytes", "fvit_workspace_init",
    "fvit_hat_stage_forward", "fvit_hat_block_forward", "fvit_token_init", "fvit_window_partition", "fvit_window_reverse", "fvit_gemm_bias_act",
    "fvit_gemm_residual", "fvit_gemm_terms", "fvit_gemm_residual_splitk", "fvit_gemm_terms_lo", "fvit_window_attention_terms", "fvit_window_attention_long_terms",
    "fvit_gather_layernorm_terms", "fvit_win_mlp_fused_terms", "fvit_win_mlp_split_bytes", "fvit_win_mlp_fused_split", "fvit_win_block_fused_split",
    "fvit_win_block_fused_terms", "fvit_attn_block_fused_terms", "fvit_ct_block_fused_terms", "fvit_window_attention", "fvit_window_attention_long",
    "fvit_gather_layernorm", "fvit_ln_gemm_supported", "fvit_ln_gemm", "fvit_attn_block_supported", "fvit_attn_block_fused",
    "fvit_ct_block_supported", "fvit_ct_block_fused", "fvit_win_block_supported", "fvit_win_block_fused", "fvit_win_mlp_supported",
    "fvit_win_mlp_fused", "fvit_mlp_fused_supported", "fvit_mlp_fused", "fvit_bias_act_cl", "fvit_bias_residual_cl", "fvit_layernorm2d_cl",
    "fvit_conv3x3_nhwc", "fvit_conv3x3_nhwc_terms", "fvit_conv3x3_dense_k", "fvit_conv3x3_patch_form", "fvit_conv3x3_nhwc_dense", "fvit_conv3x3_nhwc_px_dense", "fvit_conv3x3_c128_band_supported", "fvit_conv3x3_c128_band", "fvit_stem_conv3x3s2",
    "fvit_stem_fused", "fvit_window_attention_drop", "fvit_bwd_window_attention_drop", "fvit_global_avgpool_cl", "fvit_conv3x3_nhwc_px", "fvit_layernorm2d_px", "fvit_stem_conv3x3s2_px", "fvit_head_logits", "fvit_head_softmax_xent",
    "fvit_head_grad", "fvit_sgd_momentum", "fvit_bwd_blocks", "fvit_bwd_transpose16", "fvit_bwd_scale_cols", "fvit_bwd_gelu", "fvit_bwd_layernorm",
    "fvit_bwd_colsum_finish", "fvit_bwd_colsum16", "fvit_bwd_window_attention", "fvit_tune", "fvit_prof_enable", "fvit_prof_collect",
    "fvit_prof_records", "fvit_prof_kind_name",
)
# only in libfvit_hip_diag.so (the same sources with -DFVIT_DIAG; FVIT_DIAG=1 selects it): diagnosis entry points of include/fvit_hip.h's #ifdef FVIT_DIAG
# section.  The shipped library exports none of them and compiles the ablation knobs out (tests/test_abi.py).
DIAG_SYMBOLS = (
    "fvit_debug_lds_poison", "fvit_debug_regs_poison", "fvit_debug_poison_launches", "fvit_debug_rowhash_begin", "fvit_debug_rowhash_end",
    "fvit_debug_rowhash_dump", "fvit_debug_mlp_trace_begin", "fvit_debug_mlp_trace_end", "fvit_debug_mlp_inputs_begin", "fvit_debug_mlp_inputs_end",
    "fvit_debug_win_mlp_timeline", "fvit_debug_stem_timeline", "fvit_debug_conv_band_timeline", "fvit_debug_attn_block_timeline",
    "fvit_debug_ct_block_timeline",
)


class FvitDebugRowhashRecord(C.Structure):
    _fields_ = [("tag", C.c_char * 24), ("offset", C.c_int64), ("rows", C.c_int64)]


class FvitStageDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "batch", "C", "heads", "dpad", "ws", "H", "W", "Hp", "Wp", "cw", "hier", "square", "hidden",
        "depth", "do_propagation", "operand_dtype", "spad", "gpad")] + [("qk_scale", C.c_float), ("weight_terms", C.c_int32)]


class FvitAttnWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w_qkv", "b_qkv", "w_proj", "b_proj", "bias", "ln_w", "ln_b", "gamma",
                                                  "w_qkv_frag", "b_qkv_heads", "w_proj_frag", "rel_table")] + \
               [("rel_w", C.c_int32), ("rel_ng", C.c_int32)]


class FvitMlpWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w_fc1", "b_fc1", "w_fc2", "b_fc2", "ln_w", "ln_b", "gamma", "w_fc1_frag", "w_fc2_frag")]


class FvitBlockWeights(C.Structure):
    _fields_ = [("attn", FvitAttnWeights), ("mlp", FvitMlpWeights), ("hat_attn", FvitAttnWeights),
                ("hat_mlp", FvitMlpWeights), ("pe_x", C.c_void_p), ("pe_ct", C.c_void_p),
                ("last", C.c_int32), ("_pad", C.c_int32)]


class FvitStageTables(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1_src", "ln1_add", "ct_src", "up_idx")]


class FvitMapView(C.Structure):
    _fields_ = [("data", C.c_void_p), ("stride_b", C.c_int64), ("stride_c", C.c_int64), ("stride_h", C.c_int64),
                ("stride_w", C.c_int64), ("dtype", C.c_int32), ("_pad", C.c_int32)]


class FvitProfEntry(C.Structure):
    _fields_ = [("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


class FvitProfRecord(C.Structure):
    _fields_ = [("kind", C.c_int32), ("grid", C.c_int32), ("ms", C.c_float), ("_pad", C.c_float), ("flops", C.c_double),
                ("bytes", C.c_double), ("name", C.c_char * 40)]


_lib = None


def build(verbose: bool = False) -> str:
    """Compile libfvit_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC_DIR, "-j4"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose:
        print(res.stdout)
    if res.returncode != 0 or not os.path.isfile(LIB_PATH):
        raise RuntimeError("building libfvit_hip.so failed:\n" + res.stdout)
    return LIB_PATH


def _declare(lib):
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    lib.fvit_abi_version.restype = C.c_int
    lib.fvit_last_error.restype = C.c_char_p
    lib.fvit_attention_spad.restype = C.c_int
    lib.fvit_attention_spad.argtypes = [i32]
    lib.fvit_stage_workspace_bytes.restype = C.c_size_t
    lib.fvit_stage_workspace_bytes.argtypes = [C.POINTER(FvitStageDesc)]
    lib.fvit_workspace_init.restype = C.c_int
    lib.fvit_workspace_init.argtypes = [C.POINTER(FvitStageDesc), vp, C.c_size_t, vp]
    lib.fvit_hat_stage_forward.restype = C.c_int
    lib.fvit_hat_stage_forward.argtypes = [C.POINTER(FvitStageDesc), C.POINTER(FvitBlockWeights),
                                           C.POINTER(FvitStageTables), C.POINTER(FvitMapView), vp,
                                           C.POINTER(FvitMapView), vp, C.c_size_t, vp]
    lib.fvit_hat_block_forward.restype = C.c_int
    lib.fvit_hat_block_forward.argtypes = [C.POINTER(FvitStageDesc), C.POINTER(FvitBlockWeights),
                                           C.POINTER(FvitStageTables), vp, vp, vp, C.c_size_t, vp]
    lib.fvit_token_init.restype = C.c_int
    lib.fvit_token_init.argtypes = [C.POINTER(FvitMapView), vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.fvit_window_partition.restype = C.c_int
    lib.fvit_window_partition.argtypes = [C.POINTER(FvitMapView), i32, i32, i32, i32, i32, vp, vp]
    lib.fvit_window_reverse.restype = C.c_int
    lib.fvit_window_reverse.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, C.POINTER(FvitMapView), vp]
    lib.fvit_gemm_bias_act.restype = C.c_int
    lib.fvit_gemm_bias_act.argtypes = [i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.fvit_gemm_residual.restype = C.c_int
    lib.fvit_gemm_residual.argtypes = [i32, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.fvit_gemm_terms.restype = C.c_int
    lib.fvit_gemm_terms.argtypes = [i32, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.fvit_gemm_residual_splitk.restype = C.c_int
    lib.fvit_gemm_residual_splitk.argtypes = [i32, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, vp, C.c_size_t, vp]
    lib.fvit_gemm_terms_lo.restype = C.c_int
    lib.fvit_gemm_terms_lo.argtypes = [i32, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.fvit_window_attention_terms.restype = C.c_int
    lib.fvit_window_attention_terms.argtypes = [i32, vp, i32, i32, vp, i32, i32, vp, i32, i32, i32, i32, f32, vp]
    lib.fvit_window_attention_long_terms.restype = C.c_int
    lib.fvit_window_attention_long_terms.argtypes = [i32, vp, i32, i32, vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, f32, vp]
    lib.fvit_gather_layernorm_terms.restype = C.c_int
    lib.fvit_gather_layernorm_terms.argtypes = [i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, i32, vp, vp, f32, i32, i32, i32, vp]
    lib.fvit_win_mlp_fused_terms.restype = C.c_int
    lib.fvit_win_mlp_fused_terms.argtypes = [i32, vp, i32, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp, i32, vp]
    lib.fvit_win_mlp_split_bytes.restype = C.c_size_t
    lib.fvit_win_mlp_split_bytes.argtypes = [i32, i32, i32]
    lib.fvit_win_mlp_fused_split.restype = C.c_int
    lib.fvit_win_mlp_fused_split.argtypes = [i32, vp, i32, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp]
    lib.fvit_window_attention.restype = C.c_int
    lib.fvit_window_attention.argtypes = [i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, f32, vp]
    lib.fvit_attention_dense.restype = C.c_int
    lib.fvit_attention_dense.argtypes = [i32, i32]
    lib.fvit_window_attention_long.restype = C.c_int
    lib.fvit_window_attention_long.argtypes = [i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, f32, vp]
    lib.fvit_gather_layernorm.restype = C.c_int
    lib.fvit_gather_layernorm.argtypes = [i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, f32, i32, i32, i32, vp]
    lib.fvit_ln_gemm_supported.restype = C.c_int
    lib.fvit_ln_gemm_supported.argtypes = [i32, i32, i32, i32]
    lib.fvit_ln_gemm.restype = C.c_int
    lib.fvit_ln_gemm.argtypes = [i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, f32, i32, i32, i32, vp, i32, vp, vp, i32, i32, i32, vp]
    lib.fvit_attn_block_supported.restype = C.c_int
    lib.fvit_attn_block_supported.argtypes = [i32, i32, i32]
    lib.fvit_attn_block_fused.restype = C.c_int
    lib.fvit_attn_block_fused.argtypes = [i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, f32, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32,
                                          f32, vp]
    lib.fvit_mlp_fused_supported.restype = C.c_int
    lib.fvit_mlp_fused_supported.argtypes = [i32, i32]
    lib.fvit_mlp_fused.restype = C.c_int
    lib.fvit_mlp_fused.argtypes = [i32, vp, i32, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp, vp]
    lib.fvit_bias_act_cl.restype = C.c_int
    lib.fvit_bias_act_cl.argtypes = [i32, vp, vp, C.c_int64, i32, i32, vp]
    lib.fvit_bias_residual_cl.restype = C.c_int
    lib.fvit_bias_residual_cl.argtypes = [i32, vp, vp, vp, C.c_int64, i32, vp]
    lib.fvit_layernorm2d_cl.restype = C.c_int
    lib.fvit_layernorm2d_cl.argtypes = [i32, vp, vp, vp, vp, f32, C.c_int64, i32, i32, vp]
    lib.fvit_conv3x3_nhwc.restype = C.c_int
    lib.fvit_conv3x3_nhwc.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.fvit_conv3x3_nhwc_terms.restype = C.c_int
    lib.fvit_conv3x3_nhwc_terms.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.fvit_window_attention_drop.restype = C.c_int
    lib.fvit_window_attention_drop.argtypes = [i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, f32, vp, vp]
    lib.fvit_bwd_window_attention_drop.restype = C.c_int
    lib.fvit_bwd_window_attention_drop.argtypes = [i32, vp, i32, vp, i32, vp, i32, f32, vp, vp, i32, i32, i32, i32, vp, vp]
    lib.fvit_global_avgpool_cl.restype = C.c_int
    lib.fvit_global_avgpool_cl.argtypes = [i32, vp, vp, i32, i32, i32, vp]
    lib.fvit_conv3x3_patch_form.restype = C.c_int
    lib.fvit_conv3x3_patch_form.argtypes = [i32, i32, i32, i32, i32, i32]
    lib.fvit_conv3x3_dense_k.restype = C.c_int
    lib.fvit_conv3x3_dense_k.argtypes = [i32]
    lib.fvit_conv3x3_nhwc_dense.restype = C.c_int
    lib.fvit_conv3x3_nhwc_dense.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.fvit_conv3x3_nhwc_px_dense.restype = C.c_int
    lib.fvit_conv3x3_nhwc_px_dense.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.fvit_conv3x3_nhwc_px.restype = C.c_int
    lib.fvit_conv3x3_nhwc_px.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.fvit_layernorm2d_px.restype = C.c_int
    lib.fvit_layernorm2d_px.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, f32, C.c_int64, i32, i32, vp]
    lib.fvit_stem_conv3x3s2_px.restype = C.c_int
    lib.fvit_stem_conv3x3s2_px.argtypes = [i32, C.POINTER(FvitMapView), vp, vp, vp, vp, i32, i32, i32, vp]
    lib.fvit_stem_conv3x3s2.restype = C.c_int
    lib.fvit_stem_conv3x3s2.argtypes = [i32, C.POINTER(FvitMapView), vp, vp, vp, i32, i32, i32, vp]
    lib.fvit_stem_fused.restype = C.c_int
    lib.fvit_stem_fused.argtypes = [i32, C.POINTER(FvitMapView), vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.fvit_head_logits.restype = C.c_int
    lib.fvit_head_logits.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
    lib.fvit_head_softmax_xent.restype = C.c_int
    lib.fvit_head_softmax_xent.argtypes = [vp, vp, vp, vp, i32, i32, f32, f32, vp]
    lib.fvit_head_grad.restype = C.c_int
    lib.fvit_head_grad.argtypes = [vp, vp, vp, vp, i32, i32, i32, f32, vp]
    lib.fvit_sgd_momentum.restype = C.c_int
    lib.fvit_sgd_momentum.argtypes = [vp, vp, vp, C.c_int64, f32, f32, f32, vp]
    lib.fvit_win_mlp_supported.restype = C.c_int
    lib.fvit_win_mlp_supported.argtypes = [i32, i32]
    lib.fvit_win_mlp_fused.restype = C.c_int
    lib.fvit_win_mlp_fused.argtypes = list(lib.fvit_mlp_fused.argtypes)
    lib.fvit_win_block_supported.restype = C.c_int
    lib.fvit_win_block_supported.argtypes = [i32, i32, i32]
    lib.fvit_win_block_fused.restype = C.c_int
    lib.fvit_win_block_fused.argtypes = list(lib.fvit_attn_block_fused.argtypes)
    lib.fvit_win_block_fused_split.restype = C.c_int
    lib.fvit_win_block_fused_split.argtypes = list(lib.fvit_attn_block_fused.argtypes)[:-1] + [vp, vp, i32, vp]
    lib.fvit_ct_block_supported.restype = C.c_int
    lib.fvit_ct_block_supported.argtypes = [i32, i32, i32, i32]
    lib.fvit_ct_block_fused.restype = C.c_int
    lib.fvit_ct_block_fused.argtypes = [i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, f32, vp]
    lib.fvit_ct_block_fused_terms.restype = C.c_int
    lib.fvit_ct_block_fused_terms.argtypes = list(lib.fvit_ct_block_fused.argtypes)[:-1] + [i32, vp]
    lib.fvit_win_block_fused_terms.restype = C.c_int
    lib.fvit_win_block_fused_terms.argtypes = list(lib.fvit_attn_block_fused.argtypes)[:-1] + [i32, vp]
    lib.fvit_attn_block_fused_terms.restype = C.c_int
    lib.fvit_attn_block_fused_terms.argtypes = list(lib.fvit_attn_block_fused.argtypes)[:-1] + [i32, vp]
    lib.fvit_conv3x3_c128_band_supported.restype = C.c_int
    lib.fvit_conv3x3_c128_band_supported.argtypes = [i32, i32]
    lib.fvit_conv3x3_c128_band.restype = C.c_int
    lib.fvit_conv3x3_c128_band.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
    lib.fvit_bwd_blocks.restype = C.c_int32
    lib.fvit_bwd_blocks.argtypes = [i32]
    lib.fvit_bwd_transpose16.restype = C.c_int
    lib.fvit_bwd_transpose16.argtypes = [i32, vp, i32, vp, i32, i32, i32, vp]
    lib.fvit_bwd_scale_cols.restype = C.c_int
    lib.fvit_bwd_scale_cols.argtypes = [i32, vp, vp, i32, vp, vp, i32, vp, i32, i32, vp]
    lib.fvit_bwd_gelu.restype = C.c_int
    lib.fvit_bwd_gelu.argtypes = [i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, vp]
    lib.fvit_bwd_layernorm.restype = C.c_int
    lib.fvit_bwd_layernorm.argtypes = [vp, vp, vp, vp, f32, vp, vp, vp, i32, i32, vp]
    lib.fvit_bwd_colsum_finish.restype = C.c_int
    lib.fvit_bwd_colsum_finish.argtypes = [vp, i32, i32, vp, i32, i32, vp]
    lib.fvit_bwd_colsum16.restype = C.c_int
    lib.fvit_bwd_colsum16.argtypes = [i32, vp, i32, vp, i32, i32, vp]
    lib.fvit_bwd_window_attention.restype = C.c_int
    lib.fvit_bwd_window_attention.argtypes = [i32, vp, i32, vp, i32, vp, i32, f32, vp, vp, i32, i32, i32, i32, vp]
    lib.fvit_tune.restype = C.c_int
    lib.fvit_tune.argtypes = [C.c_char_p, i32]
    lib.fvit_prof_enable.restype = C.c_int
    lib.fvit_prof_enable.argtypes = [C.c_int]
    lib.fvit_prof_collect.restype = C.c_int
    lib.fvit_prof_collect.argtypes = [C.POINTER(FvitProfEntry)]
    lib.fvit_prof_records.restype = C.c_int
    lib.fvit_prof_records.argtypes = [C.POINTER(FvitProfRecord), i32]
    lib.fvit_prof_kind_name.restype = C.c_char_p
    lib.fvit_prof_kind_name.argtypes = [C.c_int]


def _declare_diag(lib):
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    lib.fvit_debug_stem_timeline.restype = C.c_int
    lib.fvit_debug_stem_timeline.argtypes = [C.POINTER(FvitMapView), vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]
    lib.fvit_debug_win_mlp_timeline.restype = C.c_int
    lib.fvit_debug_win_mlp_timeline.argtypes = [vp, i32, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp]
    lib.fvit_debug_lds_poison.restype = C.c_int
    lib.fvit_debug_lds_poison.argtypes = [vp, i32, i32, vp]
    lib.fvit_debug_poison_launches.restype = C.c_int
    lib.fvit_debug_poison_launches.argtypes = [vp, C.c_uint32]
    lib.fvit_debug_regs_poison.restype = C.c_int
    lib.fvit_debug_regs_poison.argtypes = [vp, i32, i32, C.c_uint32, vp]
    lib.fvit_debug_rowhash_begin.restype = C.c_int
    lib.fvit_debug_rowhash_begin.argtypes = [vp, C.c_int64]
    lib.fvit_debug_rowhash_end.restype = C.c_int
    lib.fvit_debug_rowhash_end.argtypes = [C.POINTER(FvitDebugRowhashRecord), i32]
    lib.fvit_debug_rowhash_dump.restype = C.c_int
    lib.fvit_debug_rowhash_dump.argtypes = [i32, vp, C.c_int64]
    lib.fvit_debug_mlp_trace_begin.restype = C.c_int
    lib.fvit_debug_mlp_trace_begin.argtypes = [vp, C.c_int64]
    lib.fvit_debug_mlp_trace_end.restype = C.c_int
    lib.fvit_debug_mlp_trace_end.argtypes = [C.POINTER(C.c_int64), C.POINTER(i32), i32]
    lib.fvit_debug_mlp_inputs_begin.restype = C.c_int
    lib.fvit_debug_mlp_inputs_begin.argtypes = [vp, C.c_int64]
    lib.fvit_debug_mlp_inputs_end.restype = C.c_int
    lib.fvit_debug_mlp_inputs_end.argtypes = [C.POINTER(C.c_int64), C.POINTER(i32), i32]
    lib.fvit_debug_conv_band_timeline.restype = C.c_int
    lib.fvit_debug_conv_band_timeline.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]
    lib.fvit_debug_attn_block_timeline.restype = C.c_int
    lib.fvit_debug_attn_block_timeline.argtypes = list(lib.fvit_attn_block_fused.argtypes)[1:-1] + [vp, vp]
    lib.fvit_debug_ct_block_timeline.restype = C.c_int
    lib.fvit_debug_ct_block_timeline.argtypes = list(lib.fvit_ct_block_fused.argtypes)[1:-1] + [vp, vp]


def lib():
    """Load (once) and return the ctypes handle; raises RuntimeError when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the FasterViT HAT path has no CPU/eager fallback. Build it with "
            "`make -C fastervit_amd/csrc -j` (or `python -c 'import __graft_entry__ as g; g.build()'`).")
    try:
        handle = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
    missing = [s for s in EXPORTED_SYMBOLS + (DIAG_SYMBOLS if DIAG else ()) if not hasattr(handle, s)]
    if missing:
        raise RuntimeError(f"{LIB_PATH} lacks symbols {missing}; rebuild it")
    _declare(handle)
    if DIAG:
        _declare_diag(handle)
    if handle.fvit_abi_version() != FVIT_ABI_VERSION:
        raise RuntimeError(f"ABI mismatch: library {handle.fvit_abi_version()} vs binding {FVIT_ABI_VERSION}")
    _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().fvit_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def tune(key: str, value: int) -> None:
    check(lib().fvit_tune(key.encode(), int(value)), "fvit_tune")


def prof_enable(on: bool) -> None:
    check(lib().fvit_prof_enable(1 if on else 0), "fvit_prof_enable")


def prof_collect() -> dict:
    """Return {kind_name: dict(launches, ms, flops, bytes)} for the launches since prof_enable(True)."""
    arr = (FvitProfEntry * FVIT_PROF_KINDS)()
    check(lib().fvit_prof_collect(arr), "fvit_prof_collect")
    out = {}
    for k in range(FVIT_PROF_KINDS):
        e = arr[k]
        out[lib().fvit_prof_kind_name(k).decode()] = dict(launches=int(e.launches), ms=float(e.ms), flops=float(e.flops),
                                                         bytes=float(e.bytes))
    return out


def prof_records(max_records: int = 65536) -> list:
    """Per-launch records since prof_enable(True): dicts(kind, name, grid, ms, flops, bytes) in launch order."""
    arr = (FvitProfRecord * max_records)()
    n = lib().fvit_prof_records(arr, max_records)
    if n < 0:
        check(n, "fvit_prof_records")
    kinds = [lib().fvit_prof_kind_name(k).decode() for k in range(FVIT_PROF_KINDS)]
    return [dict(kind=kinds[arr[i].kind] if 0 <= arr[i].kind < FVIT_PROF_KINDS else "other", name=arr[i].name.decode("utf-8", "replace"),
                 grid=int(arr[i].grid), ms=float(arr[i].ms), flops=float(arr[i].flops), bytes=float(arr[i].bytes)) for i in range(n)]
