/*
 * fvit_hip.h -- C ABI of libfvit_hip.so: the MI355X (gfx950) FasterViT Hierarchical-Attention path.
 *
 * The reference (NVlabs/FasterViT) has no FFI/plugin boundary for this path: it is plain PyTorch
 * (fastervit/models/faster_vit.py, faster_vit_any_res.py).  The drop-in boundary is therefore the
 * Python module API (create_model / FasterViT / FasterViTLayer / HAT) and this
 * C ABI is what those modules bind with ctypes.  Each entry point names the reference code it
 * replaces ("AR:" = fastervit/models/faster_vit_any_res.py, "FV:" = fastervit/models/faster_vit.py).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - The library never allocates or frees device memory.  Inputs, outputs, packed weights, folded
 *     constants, index tables and the workspace are caller-owned device allocations.
 *   - All work is enqueued asynchronously on the caller's hipStream_t; no host synchronisation
 *     (profiling collection excepted), so calls are capturable in a hipGraph.
 *   - Return value: 0 on success, negative FVIT_E* otherwise; message via fvit_last_error()
 *     (thread-local).  No exceptions cross the ABI.
 *   - One process per GPU; hipSetDevice is the caller's job.
 *
 * Numerics: MFMA operands are 16-bit (fp16 by default, bf16 selectable), accumulation fp32, the
 * residual stream, LayerNorm, softmax and all position terms are fp32.
 */
#ifndef FVIT_HIP_H
#define FVIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FVIT_ABI_VERSION 8

/* error codes */
#define FVIT_OK 0
#define FVIT_EINVAL (-1)      /* bad descriptor / unsupported geometry */
#define FVIT_EWORKSPACE (-2)  /* workspace too small */
#define FVIT_ELAUNCH (-3)     /* hip launch error */

/* element types of caller tensors (feature maps in / out) and of MFMA operands */
#define FVIT_F32 0
#define FVIT_F16 1
#define FVIT_BF16 2

/* GEMM tile constants the packed-weight layout depends on */
#define FVIT_TILE_N 128 /* weight rows are zero-padded to a multiple of this */
#define FVIT_TILE_K 64  /* K (columns of weights and activations) is zero-padded to a multiple */
#define FVIT_MASK_BIAS (-30000.0f) /* additive score bias on padded key columns */
#define FVIT_MAX_DENSE_SEQ 208     /* window sequences up to this length use the dense folded bias table and the in-register
                                      attention kernel; longer ones (the 21k 384/512/768 fine-tunes: 576..2304 tokens) use the
                                      online-softmax kernel with the compact relative table (FvitAttnWeights.rel_table) */

typedef void* fvit_stream_t; /* hipStream_t */

/*
 * Geometry of one transformer stage (FasterViTLayer with conv=False; AR:753-870 / FV:741-843).
 * All fields are host values.
 */
typedef struct FvitStageDesc {
    int32_t batch;          /* images B */
    int32_t C;              /* channels */
    int32_t heads;          /* attention heads h (head_dim d = C / h) */
    int32_t dpad;           /* head_dim padded to 32, 64 or 96 (packed q/k/v/proj layout) */
    int32_t ws;             /* window side */
    int32_t H, W;           /* feature-map size before padding */
    int32_t Hp, Wp;         /* padded to a multiple of ws (AR:851-857) */
    int32_t cw;             /* carrier tokens per window side (ct_size); 0 when hier == 0 */
    int32_t hier;           /* 1: carrier-token branch active (do_sr_hat, AR:619) */
    int32_t square;         /* 1: hat_pos_embed present (AR:617,658) */
    int32_t hidden;         /* MLP hidden width (4C) */
    int32_t depth;          /* number of HAT blocks */
    int32_t do_propagation; /* AR:703: last block adds carrier tokens back into the image */
    int32_t operand_dtype;  /* FVIT_F16 or FVIT_BF16: MFMA operand type of packed weights */
    int32_t spad;           /* window sequence (ws^2 + cw^2) padded to a multiple of 16 */
    int32_t gpad;           /* carrier sequence G = cw^2 * nW padded to a multiple of 16 (hier) */
    float qk_scale;         /* score scale of attn and hat_attn; <= 0 selects head_dim^-0.5 (FV:538 `qk_scale or head_dim ** -0.5`) */
    int32_t weight_terms;   /* 1: every Linear weight rounded once to the operand type.  2 ("f16x2" / "bf16x2"): every packed weight array
                               holds TWO 16-bit terms, hi = round(w) and lo = round(w - hi): row-major arrays as [rows][2 * ldk] = [hi | lo]
                               (K-concatenated; the kernels wrap the activation column at ldk), fragment-order arrays as two images back
                               to back [hi image | lo image].  Activations stay single-rounded: the logits error of this path is dominated
                               by the SYSTEMATIC weight rounding (identical for every token, it survives the average pool), not by the
                               per-token activation rounding.
                               3 ("f16x3" / "bf16x3", r04): weights as [hi | lo | hi] (K-concatenated) AND every 16-bit activation
                               (LayerNorm output, q / k / v, attention output, GELU(fc1)) as two terms [hi | lo] in its workspace row:
                               each Linear layer is hi.hi + hi.lo + lo.hi, the attention core runs on two-term q, k, v, P.  Only the
                               unfused kernel chain (LayerNorm, GEMM, attention) takes this mode; three times the MFMA work. */
} FvitStageDesc;

/* One attention sub-block: LayerNorm -> qkv -> softmax(q k^T * scale + bias) v -> proj -> gamma-residual.
 * Replaces nn.LayerNorm + WindowAttention.forward (AR:558-569) + PosEmbMLPSwinv2D (AR:267-311, folded). */
typedef struct FvitAttnWeights {
    const void* w_qkv;   /* op16 [pad128(3*h*dpad)][pad64(C)], rows ordered [q|k|v][head][dpad], zero pad */
    const float* b_qkv;  /* f32  [3*h*dpad], zero on pad rows */
    const void* w_proj;  /* op16 [pad128(C)][pad64(h*dpad)], columns ordered [head][dpad], zero pad */
    const float* b_proj; /* f32  [C] */
    const float* bias;   /* f32  [h][Spad][Spad]: folded 16*sigmoid(cpb_mlp) bias, 0 on carrier rows/cols,
                            FVIT_MASK_BIAS on key columns >= S, 0 on query rows >= S */
    const float* ln_w;   /* f32  [C] */
    const float* ln_b;   /* f32  [C] */
    const float* gamma;  /* f32  [C] or NULL (layer_scale None => 1) */
    /* Optional (all NULL => unfused path): MFMA-fragment-order weights for the fused attention block kernel
     * (csrc/fvit_attnblk.hip; needs C == 256, head_dim == 32).  lane = 16*g + s, e = 0..7:
     *   w_qkv_frag [h][6 (q0 q1 k0 k1 v0 v1)][C/32 (kk)][64][8]: qkv.weight[(ub>>1)*C + head*32 + (ub&1)*16 + s][kch(kk, g, e)]  (kch as in w_fc1_frag)
     *   b_qkv_heads f32 [h][96]: qkv.bias re-ordered per head [q 32 | k 32 | v 32]
     *   w_proj_frag [h][C/16 (cb)][64][8]: proj.weight[ch(cb, s)][head*32 + (e>>2)*16 + 4g + (e&3)], ch as in w_fc2_frag */
    const void* w_qkv_frag;
    const float* b_qkv_heads;
    const void* w_proj_frag;
    /* Sequences longer than FVIT_MAX_DENSE_SEQ (then `bias` may be NULL): the un-gathered bias table
     *   rel_table f32 [h][(2*rel_w-1)^2] = 16*sigmoid(cpb_mlp(relative_coords_table)) transposed to head-major (FV:276-280);
     *   bias(q, k) = rel_table[h][(yq-yk+rel_w-1)*(2*rel_w-1) + (xq-xk+rel_w-1)] for tokens q, k >= rel_ng (token rel_ng + y*rel_w + x),
     *   0 for the rel_ng = S - rel_w^2 leading tokens (carrier tokens / zero padding, FV:282-299). */
    const float* rel_table;
    int32_t rel_w;
    int32_t rel_ng;
} FvitAttnWeights;

/* LayerNorm -> fc1 -> GELU(erf) -> fc2 -> gamma-residual.  Replaces Mlp.forward (AR:399-408). */
typedef struct FvitMlpWeights {
    const void* w_fc1;   /* op16 [pad128(hidden)][pad64(C)] */
    const float* b_fc1;  /* f32  [hidden] */
    const void* w_fc2;   /* op16 [pad128(C)][pad64(hidden)] */
    const float* b_fc2;  /* f32  [C] */
    const float* ln_w;
    const float* ln_b;
    const float* gamma;  /* f32 [C] or NULL */
    /* Optional (both NULL => unfused path): weights pre-packed in MFMA fragment order for the fused MLP kernel
     * (csrc/fvit_mlp.hip).  One fragment = 64 lanes x 8 elements (1 KiB); lane = 16*g + s (g = 0..3, s = 0..15), e = 0..7:
     *   w_fc1_frag [hidden/32][2 (hb)][C/32 (kk)][64][8]:  fc1.weight[j*32 + hb*16 + s][kch(kk, g, e)],
     *                                                       kch(kk, g, e) = (kk>>1)*64 + g*16 + (kk&1)*8 + e  (input channel of k slot kk*32 + 8g + e)
     *   w_fc2_frag [hidden/32][C/16 (cb)][64][8]:          fc2.weight[ch(cb, s)][j*32 + (e>>2)*16 + 4g + (e&3)]
     *                                                       ch(cb, s) = (cb>>2)*64 + (s>>2)*16 + (cb&3)*4 + (s&3)
     * (the k-slot order of w_fc2_frag is the order in which GELU(fc1) leaves the first MFMA's accumulator) */
    const void* w_fc1_frag;
    const void* w_fc2_frag;
} FvitMlpWeights;

/* One HAT block (AR:572-707 / FV:571-701). */
typedef struct FvitBlockWeights {
    FvitAttnWeights attn;     /* norm1 + attn + gamma3 */
    FvitMlpWeights mlp;       /* norm2 + mlp + gamma4 */
    FvitAttnWeights hat_attn; /* hat_norm1 + hat_attn + gamma1 (hier only) */
    FvitMlpWeights hat_mlp;   /* hat_norm2 + hat_mlp + gamma2 (hier only) */
    const float* pe_x;        /* f32 [ws*ws][C]: folded PosEmbMLPSwinv1D of the window tokens (AR:671) */
    const float* pe_ct;       /* f32 [G][C] folded hat_pos_embed (AR:682) or NULL */
    int32_t last;             /* AR:665 */
    int32_t _pad;
} FvitBlockWeights;

/* Index tables (device int32), built once per geometry by the host by running the reference's own
 * view/permute chains on an arange (reproduces the non-square ct_window quirk). */
typedef struct FvitStageTables {
    const int32_t* ln1_src;  /* [nW*S]  source row of each window-tensor row for the norm1 pass:
                                >= 0: row of X (relative to the image), < 0: -(r+1) = row r of the
                                carrier buffer R (carrier rows after ct_window) */
    const int32_t* ln1_add;  /* [nW*S]  row of pe_x to add, or -1 */
    const int32_t* ct_src;   /* [G]     X row (relative to the image) feeding raster carrier r (ct_dewindow) */
    const int32_t* up_idx;   /* [ws*ws] carrier slot (0..cw^2-1) each window token receives in propagation */
} FvitStageTables;

/* Strided view of a caller feature map (B, C, H, W) in element units; any memory format. */
typedef struct FvitMapView {
    void* data;
    int64_t stride_b, stride_c, stride_h, stride_w;
    int32_t dtype; /* FVIT_F32 / FVIT_F16 / FVIT_BF16 */
    int32_t _pad;
} FvitMapView;

int fvit_abi_version(void);
const char* fvit_last_error(void);

/* Padded sequence length (multiple of 16) the attention kernel uses for S tokens; the folded bias
 * tables must be laid out [heads][spad][spad] with this value (FvitStageDesc.spad / .gpad). */
int fvit_attention_spad(int32_t S);
/* 1 if a (window sequence, padded head dim) pair runs on the in-register kernel with the dense bias table
 * (S <= FVIT_MAX_DENSE_SEQ, and S <= 128 for dpad 96), 0 if it needs FvitAttnWeights.rel_table and the online-softmax kernel. */
int fvit_attention_dense(int32_t S, int32_t dpad);

/* Bytes of workspace fvit_hat_stage_forward needs for this geometry.  The workspace must be
 * zero-filled once (fvit_workspace_init or any memset) before its first use with a given
 * descriptor and must not be shared between different descriptors without re-initialising. */
size_t fvit_stage_workspace_bytes(const FvitStageDesc* desc);
int fvit_workspace_init(const FvitStageDesc* desc, void* workspace, size_t bytes, fvit_stream_t stream);

/* Transformer branch of FasterViTLayer.forward without the Downsample (AR:848-869 / FV:832-841):
 * window_partition (+ carrier tokens in front) -> depth x HAT.forward -> window_reverse (+ crop).
 *   in       : (B, C, Hp, Wp) padded feature map (the caller applies F.pad, AR:853-854)
 *   ct_init  : f32 (B, G, C) TokenInitializer output in windowed order (AR:745-750), or NULL
 *   out      : (B, C, H, W)
 */
int fvit_hat_stage_forward(const FvitStageDesc* desc, const FvitBlockWeights* blocks,
                           const FvitStageTables* tables, const FvitMapView* in,
                           const float* ct_init, const FvitMapView* out, void* workspace,
                           size_t workspace_bytes, fvit_stream_t stream);

/* HAT.forward on already-partitioned windows (AR:668-707), for callers that drive blocks
 * themselves: x f32 (B*nW, ws^2, C) in/out, ct f32 (B, G, C) in/out (NULL when hier == 0).
 * desc->depth is ignored (one block). */
int fvit_hat_block_forward(const FvitStageDesc* desc, const FvitBlockWeights* block,
                           const FvitStageTables* tables, float* x, float* ct, void* workspace,
                           size_t workspace_bytes, fvit_stream_t stream);

/* TokenInitializer.forward (AR:745-750 with AR:737-741 / FV:733-738): depthwise 3x3 conv (pad 1, f32 weight [C][3][3] + bias [C]),
 * AvgPool2d((pool_kh, pool_kw), (pool_sh, pool_sw)) and the view/permute into per-window carrier order, in one kernel.
 * in: (B, C, Hp, Wp) padded map, any strides / fp32, fp16, bf16; ct_out: f32 (B, G, C), G = pooled H * pooled W. */
int fvit_token_init(const FvitMapView* in, const float* weight, const float* bias, float* ct_out, int32_t batch, int32_t C,
                    int32_t Hp, int32_t Wp, int32_t pool_kh, int32_t pool_kw, int32_t pool_sh, int32_t pool_sw, int32_t cw,
                    fvit_stream_t stream);

/* window_partition (AR:84-88) / window_reverse (AR:91-94) as standalone ops on f32 token tensors. */
int fvit_window_partition(const FvitMapView* in, int32_t batch, int32_t C, int32_t Hp, int32_t Wp,
                          int32_t ws, float* windows, fvit_stream_t stream);
int fvit_window_reverse(const float* windows, int32_t batch, int32_t C, int32_t Hp, int32_t Wp,
                        int32_t H, int32_t W, int32_t ws, const FvitMapView* out, fvit_stream_t stream);

/* ---- unit entry points (used by the parity tests to localise failures) ---- */

/* out[m][n] (op16, ld ldo) = epilogue(sum_k A[m][k] * Wt[n][k] + bias[n]); epilogue: 0 none, 1 GELU(erf).
 * A: op16 [pad128(M)][lda], Wt: op16 [pad128(N)][ldw], K a multiple of 64. */
int fvit_gemm_bias_act(int32_t operand_dtype, const void* A, int32_t lda, const void* Wt, int32_t ldw,
                       const float* bias, void* out, int32_t ldo, int32_t M, int32_t N, int32_t K,
                       int32_t act, fvit_stream_t stream);
/* x[m][n] (f32, ld ldx) += gamma[n] * (sum_k A[m][k] * Wt[n][k] + bias[n]); gamma may be NULL (=1). */
int fvit_gemm_residual(int32_t operand_dtype, const void* A, int32_t lda, const void* Wt, int32_t ldw,
                       const float* bias, const float* gamma, float* x, int32_t ldx, int32_t M,
                       int32_t N, int32_t K, fvit_stream_t stream);

/* fvit_gemm_residual with an fp32 scratch for deterministic split-K (r04): when the 128 x 128 tile grid is small (<= 230 workgroups) and K is long
 * (>= 8 K tiles of 64), `splits` (2 .. 8) x the workgroups each accumulate 1 / splits of K into slab [splits][M][N], and a second launch adds the
 * partials in split order and applies bias / gamma / the residual update: no atomics, bitwise repeatable; the result differs from the unsplit
 * GEMM only by fp32 summation order.  slab_bytes >= splits * M * N * 4, else (or slab NULL) the call is fvit_gemm_residual.  The stage path gives
 * the carrier-token branch's proj / fc2 GEMMs (FasterViT-4: 42 workgroups x 49 K tiles) this scratch from its workspace. */
int fvit_gemm_residual_splitk(int32_t operand_dtype, const void* A, int32_t lda, const void* Wt, int32_t ldw, const float* bias, const float* gamma,
                              float* x, int32_t ldx, int32_t M, int32_t N, int32_t K, float* slab, size_t slab_bytes, fvit_stream_t stream);

/* The same GEMM with K-concatenated weight terms (FvitStageDesc.weight_terms): Wt is [pad128(N)][ldw >= K], K = ka or 2 * ka, the
 * contraction index k reads activation column k mod ka (A is [pad128(M)][lda >= ka]).  epilogue: 0 bias, 1 bias + GELU (out op16),
 * 2 gamma-residual into f32 out (gamma may be NULL = 1).  With K == ka this is fvit_gemm_bias_act / fvit_gemm_residual. */
int fvit_gemm_terms(int32_t operand_dtype, const void* A, int32_t lda, const void* Wt, int32_t ldw, const float* bias, const float* gamma,
                    void* out, int32_t ldo, int32_t M, int32_t N, int32_t K, int32_t ka, int32_t epilogue, fvit_stream_t stream);
/* Two-term ACTIVATIONS (FvitStageDesc.weight_terms = 3, the "x3" operand modes; r04).  A 16-bit activation row holds two images
 * [hi | lo], lo = round(y - hi) at column offset *_lo_off of the same row, so that hi + lo carries ~22 significant bits:
 *   fvit_gather_layernorm_terms   fvit_gather_layernorm whose n_out rows are [hi | lo] (ldn >= 2 * lo_off, lo_off >= pad(C));
 *   fvit_gemm_terms_lo            fvit_gemm_terms with K = 3 * ka allowed: weights [hi | lo | hi] against activation columns
 *                                 [hi | hi | lo] (A is [..][lda >= 2 * ka]) = hi.hi + hi_a.lo_w + lo_a.hi_w, the lo.lo product dropped;
 *                                 out_lo_off > 0 (epilogues 0 / 1): the output is stored as two terms too; GELU then uses the
 *                                 1.5e-7-accurate erf instead of the 5e-5 polynomial;
 *   fvit_window_attention_terms   fvit_window_attention on two-term q / k / v (scores qh.kh + qh.kl + ql.kh; P and V as two terms in
 *                                 registers / LDS), output rows [hi | lo].  Dense windows (fvit_attention_dense);
 *   fvit_window_attention_long_terms (r06) the same on windows beyond the dense kernel: fvit_window_attention_long with two-term q / k / v, the tile's
 *                                 probabilities split in registers, both V^T images in LDS, output rows [hi | lo] (the 21k 384 / 512 / 768 fine-tunes).
 * The reference computes these in fp32 (FV:557-568, 398-407); this is the route to logits max-abs < 1e-3 ABSOLUTE on the deep /
 * wide variants whose logits reach |7|. */
int fvit_gemm_terms_lo(int32_t operand_dtype, const void* A, int32_t lda, const void* Wt, int32_t ldw, const float* bias, const float* gamma,
                       void* out, int32_t ldo, int32_t out_lo_off, int32_t M, int32_t N, int32_t K, int32_t ka, int32_t epilogue,
                       fvit_stream_t stream);
int fvit_window_attention_terms(int32_t operand_dtype, const void* qkv, int32_t ldq, int32_t q_lo_off, void* out, int32_t ldo,
                                int32_t o_lo_off, const float* bias, int32_t nwin, int32_t S, int32_t heads, int32_t dpad, float scale,
                                fvit_stream_t stream);
int fvit_window_attention_long_terms(int32_t operand_dtype, const void* qkv, int32_t ldq, int32_t q_lo_off, void* out, int32_t ldo, int32_t o_lo_off,
                                     const float* rel_table, int32_t rel_w, int32_t rel_ng, int32_t nwin, int32_t S, int32_t heads, int32_t dpad, float scale,
                                     fvit_stream_t stream);
int fvit_gather_layernorm_terms(int32_t operand_dtype, const float* srcA, int32_t rowsA, const float* srcB, int32_t rowsB,
                                const int32_t* src_idx, const int32_t* add_idx, const float* add, float* x_out, void* n_out, int32_t ldn,
                                int32_t lo_off, const float* ln_w, const float* ln_b, float eps, int32_t rows, int32_t rows_per_image,
                                int32_t C, fvit_stream_t stream);
/* Windowed multi-head attention core on packed qkv (op16 [rows][ldq], columns [q|k|v][head][dpad]):
 * out (op16 [rows][ldo], columns [head][dpad]) = softmax(q k^T * scale + bias) v per (window, head).
 * S tokens per window (rows w*S .. w*S+S-1), bias f32 [heads][Spad][Spad] as in FvitAttnWeights. */
int fvit_window_attention(int32_t operand_dtype, const void* qkv, int32_t ldq, void* out, int32_t ldo,
                          const float* bias, int32_t nwin, int32_t S, int32_t heads, int32_t dpad,
                          float scale, fvit_stream_t stream);
/* fvit_window_attention with Dropout on the probabilities (TRAIN mode, r05; WindowAttention.attn_drop, FV:563-564: softmax, then Dropout, then . v):
 * drop_mask op16 [nwin * heads][S][spad] (spad = fvit_attention_spad(S)), entries 0 or 1 / keep, applied after the softmax normalisation; NULL = none. */
int fvit_window_attention_drop(int32_t operand_dtype, const void* qkv, int32_t ldq, void* out, int32_t ldo, const float* bias, int32_t nwin, int32_t S,
                               int32_t heads, int32_t dpad, float scale, const void* drop_mask, fvit_stream_t stream);
/* Same contract for windows of more than FVIT_MAX_DENSE_SEQ tokens (any S >= 1 is accepted): online softmax over key tiles,
 * bias from the compact table rel_table f32 [heads][(2*rel_w-1)^2] (NULL = no bias) with rel_ng + rel_w^2 == S, see FvitAttnWeights. */
int fvit_window_attention_long(int32_t operand_dtype, const void* qkv, int32_t ldq, void* out, int32_t ldo,
                               const float* rel_table, int32_t rel_w, int32_t rel_ng, int32_t nwin, int32_t S,
                               int32_t heads, int32_t dpad, float scale, fvit_stream_t stream);
/* Row gather + optional add + LayerNorm: for row i (image b = i / rows_per_image, p = i % rows_per_image)
 *   v = (src_idx ? (src_idx[p] >= 0 ? srcA[b*rowsA + src_idx[p]] : srcB[b*rowsB - src_idx[p] - 1]) : srcA[i])
 *       + (add_idx && add_idx[p] >= 0 ? add[add_idx[p]] : 0)
 *   if x_out: x_out[i] = v (f32);  n_out[i] (op16, ld ldn, zero-padded to ldn) = LN(v) * ln_w + ln_b  */
int fvit_gather_layernorm(int32_t operand_dtype, const float* srcA, int32_t rowsA, const float* srcB,
                          int32_t rowsB, const int32_t* src_idx, const int32_t* add_idx,
                          const float* add, float* x_out, void* n_out, int32_t ldn, const float* ln_w,
                          const float* ln_b, float eps, int32_t rows, int32_t rows_per_image, int32_t C,
                          fvit_stream_t stream);

/* LayerNorm folded into the Linear layer that consumes it: out[i][n] = act( LN(v[i]) . Wt[n][:] + bias[n] ) with v[i], the optional f32
 * copy x_out[i] = v[i] and the LayerNorm exactly as in fvit_gather_layernorm (AR:616, 648-649 + AR:560 / 402-403); the normalised rows
 * never reach HBM.  C (= K) must be 256 or 512, N a multiple of 16, Wt op16 [pad128(N)][ldw] (fvit_ln_gemm_supported); x_out must not
 * alias srcA / srcB (several column groups read every source row).  act 0: bias, 1: bias + exact-erf GELU.  out op16, ld ldo. */
int fvit_ln_gemm_supported(int32_t C, int32_t N, int32_t ldw, int32_t ldo);
int fvit_ln_gemm(int32_t operand_dtype, const float* srcA, int32_t rowsA, const float* srcB, int32_t rowsB, const int32_t* src_idx,
                 const int32_t* add_idx, const float* add, float* x_out, const float* ln_w, const float* ln_b, float eps, int32_t rows,
                 int32_t rows_per_image, int32_t C, const void* Wt, int32_t ldw, const float* bias, void* out, int32_t ldo, int32_t N,
                 int32_t act, fvit_stream_t stream);

/* fvit_mlp_fused's contract for C == 512, hidden == 2048 (stage 3 of FasterViT-0) with a different work split: 64-row workgroups whose 8
 * waves split hidden units / output channels and stream their weight slices from L2 into registers (fvit_winmlp.hip); also C == 256,
 * hidden == 1024 with 128-row workgroups. */
int fvit_win_mlp_supported(int32_t C, int32_t hidden);
int fvit_win_mlp_fused(int32_t operand_dtype, float* x, int32_t M, int32_t C, int32_t hidden, const float* ln_w, const float* ln_b,
                       float eps, const void* w_fc1_frag, const float* b_fc1, const void* w_fc2_frag, const float* b_fc2,
                       const float* gamma, fvit_stream_t stream);
/* ... with `terms` weight images back to back in w_fc1_frag / w_fc2_frag (1, or 2 = [hi image | lo image]). */
int fvit_win_mlp_fused_terms(int32_t operand_dtype, float* x, int32_t M, int32_t C, int32_t hidden, const float* ln_w, const float* ln_b,
                             float eps, const void* w_fc1_frag, const float* b_fc1, const void* w_fc2_frag, const float* b_fc2,
                             const float* gamma, int32_t terms, fvit_stream_t stream);
/* C = 512 only: the hidden units of every 64-row group split over nsplit (1, 2) workgroups that meet in L2 -- each stores its fp32
 * partial output, the last arriver of a group adds the partials in split order (bitwise repeatable) and applies the residual.
 *   slab     f32 scratch of fvit_win_mlp_split_bytes(M, C, nsplit) bytes
 *   counters int32 [ceil(M / 64)], ZERO before the first launch; the kernel leaves them zero. */
size_t fvit_win_mlp_split_bytes(int32_t M, int32_t C, int32_t nsplit);
int fvit_win_mlp_fused_split(int32_t operand_dtype, float* x, int32_t M, int32_t C, int32_t hidden, const float* ln_w, const float* ln_b,
                             float eps, const void* w_fc1_frag, const float* b_fc1, const void* w_fc2_frag, const float* b_fc2,
                             const float* gamma, int32_t terms, float* slab, int32_t* counters, int32_t nsplit, fvit_stream_t stream);

/* fvit_attn_block_fused's contract for C == 512, heads == 16, 48 < S <= 64 (stage 3 of FasterViT-0) with a different work split: one
 * workgroup per window, its 8 waves split heads / output channels and stream their weight slices from L2 into registers (fvit_winblk.hip);
 * also C == 256, heads == 8 with 4 waves. */
int fvit_win_block_supported(int32_t C, int32_t heads, int32_t S);
int fvit_win_block_fused(int32_t operand_dtype, const float* srcA, int32_t rowsA, const float* srcB, int32_t rowsB,
                         const int32_t* src_idx, const int32_t* add_idx, const float* add, const float* ln_w,
                         const float* ln_b, float eps, int32_t rows_per_image, const void* w_qkv_frag,
                         const float* b_qkv_heads, const void* w_proj_frag, const float* b_proj, const float* gamma,
                         const float* bias, float* x_out, int32_t nwin, int32_t S, int32_t heads, int32_t C, float scale,
                         fvit_stream_t stream);
/* C = 512 only: the 16 heads of every window split over nsplit (1, 2) workgroups that meet in L2 (as fvit_win_mlp_fused_split).
 *   slab     f32 scratch, nwin * nsplit * 64 * C * 4 bytes;  counters int32 [nwin], ZERO before the first launch (left zero). */
int fvit_win_block_fused_split(int32_t operand_dtype, const float* srcA, int32_t rowsA, const float* srcB, int32_t rowsB, const int32_t* src_idx,
                               const int32_t* add_idx, const float* add, const float* ln_w, const float* ln_b, float eps,
                               int32_t rows_per_image, const void* w_qkv_frag, const float* b_qkv_heads, const void* w_proj_frag,
                               const float* b_proj, const float* gamma, const float* bias, float* x_out, int32_t nwin, int32_t S,
                               int32_t heads, int32_t C, float scale, float* slab, int32_t* counters, int32_t nsplit, fvit_stream_t stream);

/* C = 512 only, two-term weights (terms 1 / 2): w_qkv_frag / w_proj_frag hold `terms` images back to back (hi image, then lo image of
 * w - hi, each in the single-term fragment order); the k loops run once per image into the same fp32 accumulators. */
int fvit_win_block_fused_terms(int32_t operand_dtype, const float* srcA, int32_t rowsA, const float* srcB, int32_t rowsB, const int32_t* src_idx,
                               const int32_t* add_idx, const float* add, const float* ln_w, const float* ln_b, float eps,
                               int32_t rows_per_image, const void* w_qkv_frag, const float* b_qkv_heads, const void* w_proj_frag,
                               const float* b_proj, const float* gamma, const float* bias, float* x_out, int32_t nwin, int32_t S,
                               int32_t heads, int32_t C, float scale, int32_t terms, fvit_stream_t stream);
/* fvit_attn_block_fused with the weight terms of the fragment arrays (r04): terms = 2 reads [hi image | lo image] of w_qkv_frag / w_proj_frag
 * (C = 256, windows of 49 .. 64 tokens: the stage-2 window attention of FasterViT-0 in the x2 operand modes) on the double-buffered 8-wave form:
 * per head the lo slice of the qkv weights lands in the second buffer while the first P1 pass runs, a second pass adds into the same q / k / v
 * accumulators, and the proj MFMAs run once per term on the same attention output. */
int fvit_attn_block_fused_terms(int32_t operand_dtype, const float* srcA, int32_t rowsA, const float* srcB, int32_t rowsB, const int32_t* src_idx,
                                const int32_t* add_idx, const float* add, const float* ln_w, const float* ln_b, float eps,
                                int32_t rows_per_image, const void* w_qkv_frag, const float* b_qkv_heads, const void* w_proj_frag,
                                const float* b_proj, const float* gamma, const float* bias, float* x_out, int32_t nwin, int32_t S,
                                int32_t heads, int32_t C, float scale, int32_t terms, fvit_stream_t stream);

/* The whole carrier-token branch of one HAT block in one kernel (AR:679-686), one workgroup per image:
 *   ct[b][i] = X[b * rowsA + src_idx[i]] (+ add[i]);  ct += gamma1 * attn(LayerNorm1(ct));  ct += gamma2 * mlp(LayerNorm2(ct))  -> R [batch * G][C]
 * attention over the G <= 16 carrier tokens of an image (bias f32 [heads][16][16], mask on padded keys), weights in the fragment-major
 * packings of fvit_attn_block_fused / fvit_mlp_fused.  Needs C == 256, heads == 8, hidden == 1024 (fvit_ct_block_supported).  gamma1 /
 * gamma2 / add may be null; every src_idx entry must be >= 0 (carrier tokens are rows of X; there is no second source here). */
int fvit_ct_block_supported(int32_t C, int32_t heads, int32_t G, int32_t hidden);
int fvit_ct_block_fused(int32_t operand_dtype, const float* X, int32_t rowsA, const int32_t* src_idx, const float* add, float* R,
                        int32_t batch, int32_t G, int32_t heads, int32_t C, int32_t hidden, const float* ln1_w, const float* ln1_b,
                        const void* w_qkv_frag, const float* b_qkv_heads, const void* w_proj_frag, const float* b_proj, const float* gamma1,
                        const float* bias, float scale, const float* ln2_w, const float* ln2_b, const void* w_fc1_frag, const float* b_fc1,
                        const void* w_fc2_frag, const float* b_fc2, const float* gamma2, float eps, fvit_stream_t stream);

/* fvit_ct_block_fused with `terms` (1 / 2) weight images back to back in each of the four fragment arrays (see fvit_win_block_fused_terms). */
int fvit_ct_block_fused_terms(int32_t operand_dtype, const float* X, int32_t rowsA, const int32_t* src_idx, const float* add, float* R,
                              int32_t batch, int32_t G, int32_t heads, int32_t C, int32_t hidden, const float* ln1_w, const float* ln1_b,
                              const void* w_qkv_frag, const float* b_qkv_heads, const void* w_proj_frag, const float* b_proj, const float* gamma1,
                              const float* bias, float scale, const float* ln2_w, const float* ln2_b, const void* w_fc1_frag, const float* b_fc1,
                              const void* w_fc2_frag, const float* b_fc2, const float* gamma2, float eps, int32_t terms, fvit_stream_t stream);

/* Fused attention sub-block: x_out[i] = x_in[i] + gamma * proj(softmax(q k^T * scale + bias) v), [q|k|v] = qkv(LayerNorm(x_in)),
 * x_in[i] = gathered source row + optional add row (exactly the row selection of fvit_gather_layernorm), per window of S rows
 * (AR:671-696).  One kernel; needs C == 256, heads == 8 (head_dim 32), S <= 16 or 48 < S <= 64 (fvit_attn_block_supported).
 * bias f32 [heads][spad][spad] with spad = fvit_attention_spad(S); x_out may alias srcA (rows are read before they are written
 * by the workgroup that owns them). */
int fvit_attn_block_supported(int32_t C, int32_t heads, int32_t S);
int fvit_attn_block_fused(int32_t operand_dtype, const float* srcA, int32_t rowsA, const float* srcB, int32_t rowsB,
                          const int32_t* src_idx, const int32_t* add_idx, const float* add, const float* ln_w,
                          const float* ln_b, float eps, int32_t rows_per_image, const void* w_qkv_frag,
                          const float* b_qkv_heads, const void* w_proj_frag, const float* b_proj, const float* gamma,
                          const float* bias, float* x_out, int32_t nwin, int32_t S, int32_t heads, int32_t C, float scale,
                          fvit_stream_t stream);

/* Fused MLP sub-block: x[m][:] += gamma * fc2(GELU(fc1(LayerNorm(x[m][:])))) in one kernel (AR:697, AR:399-408).
 * x f32 [M][C] in place; w_fc1_frag / w_fc2_frag as in FvitMlpWeights.
 * Supported: C == 256, hidden % 32 == 0, hidden <= 4C (fvit_mlp_fused_supported); otherwise FVIT_EINVAL. */
int fvit_mlp_fused_supported(int32_t C, int32_t hidden);
int fvit_mlp_fused(int32_t operand_dtype, float* x, int32_t M, int32_t C, int32_t hidden, const float* ln_w,
                   const float* ln_b, float eps, const void* w_fc1_frag, const float* b_fc1, const void* w_fc2_frag,
                   const float* b_fc2, const float* gamma, fvit_stream_t stream);

/* ---- conv-side glue (deploy mode), channels-last 16-bit maps: x[pixel][c], C contiguous ----
 * With BatchNorm folded into the conv weights at load, the PyTorch-ROCm convolutions of PatchEmbed /
 * ConvBlock / Downsample run bias-free and these three HBM-bound passes do the rest. */
/* x = act(x + bias[c]) in place; act 0 none, 1 ReLU (PatchEmbed FV:460-464), 2 GELU-erf (ConvBlock FV:507). */
int fvit_bias_act_cl(int32_t dtype, void* x, const float* bias, int64_t n_pixels, int32_t C, int32_t act,
                     fvit_stream_t stream);
/* x = x + y + bias[c] in place (ConvBlock residual, FV:512, with conv2's bias and norm2/gamma folded). */
int fvit_bias_residual_cl(int32_t dtype, void* x, const void* y, const float* bias, int64_t n_pixels, int32_t C,
                          fvit_stream_t stream);
/* Per-pixel LayerNorm (timm LayerNorm2d of Downsample, FV:432,438; eps 1e-6), fp32 statistics.  Pixels are C channels apart;
 * the statistics run over the first C_valid of them (C_valid <= 0: all C).  With C_valid < C the trailing channels must hold zeros
 * on input (channel-padded deploy maps) and weight / bias must be zero there, so they stay zero on output. */
int fvit_layernorm2d_cl(int32_t dtype, const void* in, void* out, const float* weight, const float* bias, float eps,
                        int64_t n_pixels, int32_t C, int32_t C_valid, fvit_stream_t stream);

/* AdaptiveAvgPool2d(1) + flatten (FV:926, 955-956) of a channels-last map: out f32 [B][C] = mean over the HW pixels of in [B][HW][C] (fp32 / fp16 / bf16),
 * fp32 sums in a fixed order (bitwise repeatable).  With fvit_head_logits the tail of the deploy plan: no library kernel in the timed graph. */
int fvit_global_avgpool_cl(int32_t dtype, const void* in, float* out, int32_t B, int32_t HW, int32_t C, fvit_stream_t stream);

/* 3x3 convolution, pad 1, stride 1 or 2, on channels-last 16-bit maps as an implicit GEMM on the MFMA cores with the
 * epilogue fused: out = act(conv(in, weight) + bias) (+ residual).  Replaces (deploy mode, BatchNorm folded into
 * weight/bias) conv + BN + ReLU of PatchEmbed (FV:462-464), conv-BN-GELU / conv-BN-gamma-residual of ConvBlock
 * (FV:502-512) and Downsample.reduction (FV:435).
 *   in [B][Hi][Wi][Cin], weight [Cout][3][3][Cin] (= channels_last storage of the PyTorch weight), bias f32 [Cout] or
 *   NULL, residual [B][Ho][Wo][Cout] or NULL (may alias out), out [B][Ho][Wo][Cout]; act 0 none / 1 ReLU / 2 GELU;
 *   zeros: >= 128 bytes of zeros (padding taps read it).  Needs Cin % 64 == 0 and Cout % 64 == 0. */
int fvit_conv3x3_nhwc(int32_t dtype, const void* in, const void* weight, const float* bias, const void* residual,
                      void* out, int32_t B, int32_t Hi, int32_t Wi, int32_t Cin, int32_t Cout, int32_t stride,
                      int32_t act, const void* zeros, fvit_stream_t stream);
/* The same convolution with the weights as TWO 16-bit terms (weight_terms = 2): weight is [Cout][2 * 9 * Cin] = [hi | lo] per output channel
 * (hi = round(w), lo = round(w - hi), each in [3][3][Cin] order); the lo image's K steps re-read the activation tile of the same
 * (tap, channel) step.  The conv's weights then carry ~22 bits while its maps stay 16-bit.  Used by the deploy plan for the three
 * Downsample.reduction convs (FV:435), whose weight rounding -- systematic, identical for every pixel of every image -- is 2.4e-4 / 1.7e-4 /
 * 2.1e-4 of FasterViT-0's 4.2e-4 conv-side logits error.  weight_terms = 1: fvit_conv3x3_nhwc. */
int fvit_conv3x3_nhwc_terms(int32_t dtype, const void* in, const void* weight, const float* bias, const void* residual,
                            void* out, int32_t B, int32_t Hi, int32_t Wi, int32_t Cin, int32_t Cout, int32_t stride,
                            int32_t act, int32_t weight_terms, const void* zeros, fvit_stream_t stream);

/* ---- two-term MAPS (r05; the "precise" deploy plan, fastervit_amd/conv_runtime.py) -------------------------------------------------------
 * A conv-side stream is a pair of 16-bit channels-last planes (value = hi + lo, lo = round(v - hi): ~22 significant bits) or, in front of /
 * behind a transformer level, one fp32 map.  Why: replaying the fp32 reference with ONE 16-bit rounding at a time (tests/tools/
 * conv_precision_sim.py, faster_vit_4_224) a rounded conv OPERAND costs 2-8e-5 of logits error, a rounded STORED stream (the residual stream of
 * the ConvBlocks FV:502-512, the Downsample outputs FV:437-440, a transformer level's output map) 3-4e-4 each, the LayerNorm2d -> strided conv
 * operand 3.6e-4.  The reference keeps all of these in fp32.
 *
 * fvit_conv3x3_nhwc_px: fvit_conv3x3_nhwc_terms on such maps.
 *   in / in_lo          input planes; in_lo NULL = the input as ONE term (the hi plane is the MFMA operand).  With in_lo (needs weight_terms 2)
 *                       the contraction runs in.w_hi + in.w_lo + in_lo.w_hi (the lo.lo product dropped, relative 2^-22)
 *   residual / _lo      residual planes (lo optional), added in fp32; may alias out / out_lo (in-place update of the stream)
 *   out / out_lo        output planes: hi = round(y); lo = round(y - hi) when out_lo is given
 *   out_f32             instead of out / out_lo: the output as one fp32 map [B][Ho][Wo][Cout]
 *   act 2 uses the 1.5e-7-accurate erf GELU (the 16-bit entry points use a 5e-5 polynomial). */
int fvit_conv3x3_nhwc_px(int32_t dtype, const void* in, const void* in_lo, const void* weight, const float* bias, const void* residual,
                         const void* residual_lo, void* out, void* out_lo, float* out_f32, int32_t B, int32_t Hi, int32_t Wi, int32_t Cin,
                         int32_t Cout, int32_t stride, int32_t act, int32_t weight_terms, const void* zeros, fvit_stream_t stream);
/* ---- dense K (r06): maps whose channel count is padded (FasterViT-4: 196 real channels in a 256-channel map, 392 in 448) -------------------
 * The classic entry points contract over 9 * Cin columns, pad channels included (1.31 x / 1.14 x the K steps at 196 / 392).  The *_dense forms
 * contract over the cin_valid real channels only: weight is [Cout][weight_terms][kd] with kd = fvit_conv3x3_dense_k(cin_valid) =
 * 9 * cin_valid rounded up to 64; column t * cin_valid + c holds w[co][tap t][channel c], the tail of the row is zero.  Cin stays the channel
 * STRIDE of the input map; cin_valid % 8 == 0 (a 16-byte chunk never straddles two taps).  cin_valid == Cin is the classic layout and kernel.
 * Everything else as fvit_conv3x3_nhwc_terms / fvit_conv3x3_nhwc_px.  fvit_conv3x3_dense_k returns -1 for an unsupported cin_valid. */
int fvit_conv3x3_dense_k(int32_t cin_valid);
/* r06: 1 when fvit_conv3x3_nhwc / _terms / _px run this shape in the PATCH form (8 x 16 output patches, the 10 x 18 halo of a 64-channel chunk staged once
 * for all nine taps and both weight terms): stride 1, Cout >= 128, a patch grid that wastes <= fvit_tune("conv_patch_max_waste_pct") of the pixels, and
 * fvit_tune("conv_patch") on (the default; max waste 10 %).  The patch form takes the CLASSIC weight layout (cin_valid == Cin): a caller that packs dense-K rows asks here first. */
int fvit_conv3x3_patch_form(int32_t B, int32_t Hi, int32_t Wi, int32_t Cin, int32_t Cout, int32_t stride);
int fvit_conv3x3_nhwc_dense(int32_t dtype, const void* in, const void* weight, const float* bias, const void* residual,
                            void* out, int32_t B, int32_t Hi, int32_t Wi, int32_t Cin, int32_t cin_valid, int32_t Cout, int32_t stride,
                            int32_t act, int32_t weight_terms, const void* zeros, fvit_stream_t stream);
int fvit_conv3x3_nhwc_px_dense(int32_t dtype, const void* in, const void* in_lo, const void* weight, const float* bias, const void* residual,
                               const void* residual_lo, void* out, void* out_lo, float* out_f32, int32_t B, int32_t Hi, int32_t Wi, int32_t Cin,
                               int32_t cin_valid, int32_t Cout, int32_t stride, int32_t act, int32_t weight_terms, const void* zeros,
                               fvit_stream_t stream);
/* fvit_layernorm2d_cl on a two-term map (in + in_lo; in_lo may be NULL) or an fp32 map (in_f32, then in = in_lo = NULL); the result as two
 * planes (out_lo may be NULL).  Statistics and affine in fp32 (timm LayerNorm2d, FV:432,438). */
int fvit_layernorm2d_px(int32_t dtype, const void* in, const void* in_lo, const float* in_f32, void* out, void* out_lo, const float* weight,
                        const float* bias, float eps, int64_t n_pixels, int32_t C, int32_t C_valid, fvit_stream_t stream);
/* fvit_stem_conv3x3s2 with the K = 27 weights as two terms (weight_lo: same [64][32] layout; NULL = fvit_stem_conv3x3s2) and the image split
 * hi + lo in registers: w_hi.x_hi + w_lo.x_hi + w_hi.x_lo.  The first conv's weight rounding is systematic (2.2e-4 of logits error on
 * faster_vit_4_224), its image rounding 1.2e-4. */
int fvit_stem_conv3x3s2_px(int32_t dtype, const FvitMapView* in, const void* weight, const void* weight_lo, const float* bias, void* out,
                           int32_t B, int32_t Hi, int32_t Wi, fvit_stream_t stream);

/* The same convolution for Cin = Cout = 128, stride 1, maps up to 30 pixels wide (level 1 of FasterViT-0: 28 x 28), one ROW BAND of an
 * image per workgroup: the band's input rows + halo go to LDS once, the weights stream from L2 into registers in MFMA fragment order.
 *   w_frag  op16 [4][36][2][64][8]: element e of lane 16 g + s of fragment (wave, step, ni) =
 *           weight[32 wave + (s >> 2) * 8 + ni * 4 + (s & 3)][step * 32 + 8 g + e], weight = the [128][3][3][128] matrix of fvit_conv3x3_nhwc
 *   zeros   >= 256 bytes of zeros.  Other arguments as fvit_conv3x3_nhwc.  fvit_conv3x3_c128_band_supported: 1 when W <= 30 (and the
 *   "conv_band" tuning knob is not 0). */
int fvit_conv3x3_c128_band_supported(int32_t H, int32_t W);
int fvit_conv3x3_c128_band(int32_t dtype, const void* in, const void* w_frag, const float* bias, const void* residual, void* out,
                           int32_t B, int32_t H, int32_t W, int32_t act, const void* zeros, fvit_stream_t stream);

/* Stem convolution of PatchEmbed (FV:458-460): 3x3, stride 2, pad 1, 3 -> 64 channels, + bias (folded BatchNorm) + ReLU.
 * in: strided view of the (B, 3, Hi, Wi) image in fp32 / fp16 / bf16 (the model's NCHW fp32 input needs no conversion);
 * weight: op16 [64][32], column k = ky*9 + kx*3 + c, zero for k >= 27; out: op16 [B][Ho][Wo][64] channels-last. */
int fvit_stem_conv3x3s2(int32_t dtype, const FvitMapView* in, const void* weight, const float* bias, void* out,
                        int32_t B, int32_t Hi, int32_t Wi, fvit_stream_t stream);

/* Both convolutions of PatchEmbed in one kernel for in_dim = dim = 64 (FV:458-464): conv1 as above (w1, b1), then 3x3 stride 2 pad 1
 * 64 -> 64 (w2 op16 [64][3][3][64], b2) + ReLU; the 112x112x64 intermediate lives in LDS only.  out: op16 [B][H2][W2][64] with
 * H1 = (Hi-1)/2+1, H2 = (H1-1)/2+1 (same for W). */
int fvit_stem_fused(int32_t dtype, const FvitMapView* in, const void* w1, const float* b1, const void* w2, const float* b2,
                    void* out, int32_t B, int32_t Hi, int32_t Wi, fvit_stream_t stream);

/* ---- backward of the MLP sub-block  y = x + gamma * fc2(GELU(fc1(LayerNorm(x))))  (Mlp.forward FV:398-407 inside HAT.forward FV:691; the reference
 * differentiates it with autograd, train.py:820-951) -- SURVEY.md section 8 row f-4, the slice VERDICT r02 item 9 scopes (csrc/fvit_bwd.hip).
 * The four GEMMs of the backward run through fvit_gemm_bias_act / fvit_gemm_residual (fp32 results through the residual epilogue into zeroed or
 * accumulating buffers); these are the memory-bound pieces between them.  Host sequence: fastervit_amd/hat_backward.py.  No atomics: column sums
 * are per-block partials (fvit_bwd_blocks(M) blocks of 64 rows) added in block order by fvit_bwd_colsum_finish.  op16 = operand_dtype. */
int32_t fvit_bwd_blocks(int32_t M);
/* out[n][m] = in[m][n] for m < M, n < N; columns M .. ld_out - 1 of the N written rows are zeroed (ld_out: a multiple of 64 >= M). */
int fvit_bwd_transpose16(int32_t operand_dtype, const void* in, int32_t ld_in, void* out, int32_t ld_out, int32_t M, int32_t N, fvit_stream_t stream);
/* dz[m][c] (op16) = gamma[c] * dy[m][c] (gamma NULL = 1); part f32 [blocks][2][C]: [0] column sums of dy * z (-> dgamma), [1] of gamma * dy (-> db2). */
int fvit_bwd_scale_cols(int32_t operand_dtype, const float* dy, const void* z, int32_t ldz, const float* gamma, void* dz, int32_t lddz, float* part,
                        int32_t M, int32_t C, fvit_stream_t stream);
/* dh NULL: out = GELU(a) (erf form).  dh given: out = dh * GELU'(a), part f32 [blocks][H] = column sums of out (-> db1). */
int fvit_bwd_gelu(int32_t operand_dtype, const void* a, int32_t lda, const void* dh, int32_t lddh, void* out, int32_t ldo, float* part, int32_t M,
                  int32_t H, fvit_stream_t stream);
/* LayerNorm backward: dx[m] = (dy ? dy[m] : 0) + rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dxn[m] * ln_w; stats f32 [M][2] = (mean, rstd);
 * part f32 [blocks][2][C]: [0] column sums of dxn * xhat (-> d ln_w), [1] of dxn (-> d ln_b).  All tensors f32 [M][C]. */
int fvit_bwd_layernorm(const float* x, const float* dxn, const float* dy, const float* ln_w, float eps, float* dx, float* stats, float* part,
                       int32_t M, int32_t C, fvit_stream_t stream);
/* part f32 [blocks][N] = column sums of an op16 [M][ld] matrix over the block's rows (-> the qkv bias gradient). */
int fvit_bwd_colsum16(int32_t operand_dtype, const void* in, int32_t ld, float* part, int32_t M, int32_t N, fvit_stream_t stream);
/* Backward of the windowed attention core (WindowAttention.forward FV:557-568 between the two Linears; fvit_window_attention's layouts): per
 * (window, head)  dv = P^T dO, dS = P * (dO v^T - rowsum(dO v^T * P)), dq = scale * dS k, dk = scale * dS^T q  -> dqkv (op16, qkv's layout);
 * dbias_part f32 [nwin][heads][S][S] = dS (sum over windows with fvit_bwd_colsum_finish) or NULL.  S <= 64; D = the padded head_dim of the
 * qkv layout, 32 / 64 / 96 (pad channels are zero and receive zero gradients). */
int fvit_bwd_window_attention(int32_t operand_dtype, const void* qkv, int32_t ld, const void* dO, int32_t ldo, const float* bias, int32_t spad,
                              float scale, void* dqkv, float* dbias_part, int32_t nwin, int32_t S, int32_t heads, int32_t D, fvit_stream_t stream);
/* ... with the attn_drop mask of the forward (fvit_window_attention_drop): dP = (dO V^T) . mask, dV = (P . mask)^T dO. */
int fvit_bwd_window_attention_drop(int32_t operand_dtype, const void* qkv, int32_t ld, const void* dO, int32_t ldo, const float* bias, int32_t spad,
                                   float scale, void* dqkv, float* dbias_part, int32_t nwin, int32_t S, int32_t heads, int32_t D, const void* drop_mask,
                                   fvit_stream_t stream);
/* out[i] (+)= sum over b < blocks of part[b * stride + i], i < n, in block order. */
int fvit_bwd_colsum_finish(const float* part, int32_t blocks, int32_t stride, float* out, int32_t n, int32_t accumulate, fvit_stream_t stream);

/* ---- head-only training step (north_star's training clause; csrc/fvit_head.hip) ----
 * The classifier FasterViT.head = nn.Linear(F, N) (FV:927, 959) is trained on the pooled features of the frozen HIP backbone; the
 * reference wraps the WHOLE model in DistributedDataParallel (train.py:542-551) and reduces the loss for logging (train.py:910).
 * All tensors fp32, row-major, caller-owned; no atomics (bit-reproducible gradients).  F must be a multiple of 16. */
/* logits[b][n] = feat[b][:] . W[n][:] + bias[n]   (exact-fp32 MFMA v_mfma_f32_16x16x4_f32) */
int fvit_head_logits(const float* feat, const float* W, const float* bias, float* logits, int32_t B, int32_t N, int32_t F,
                     fvit_stream_t stream);
/* Label-smoothed cross entropy (timm LabelSmoothingCrossEntropy, selected at train.py:685; smoothing 0 = nn.CrossEntropyLoss,
 * train.py:687): loss_rows[b] = (1 - eps) * (-log p[b][t_b]) + eps * mean_n(-log p[b][n]); then IN PLACE
 * logits[b][n] <- (p[b][n] - q[b][n]) * inv_global_batch with q = (1 - eps) * onehot(t_b) + eps / N, i.e. d(mean loss)/d(logits).
 * row_stats: scratch f32 [2 * B]. */
int fvit_head_softmax_xent(float* logits_inout, const int64_t* target, float* loss_rows, float* row_stats, int32_t B, int32_t N,
                           float smoothing, float inv_global_batch, fvit_stream_t stream);
/* grad_flat = [ dW (N x F) = dlogits^T . feat | db (N) = column sums of dlogits | loss (1) = sum(loss_rows) * inv_global_batch ]:
 * ONE contiguous buffer of N*F + N + 1 floats, so that a single all-reduce (SUM over ranks) yields the global-batch gradient AND
 * the global mean loss. */
int fvit_head_grad(const float* dlogits, const float* feat, const float* loss_rows, float* grad_flat, int32_t B, int32_t N, int32_t F,
                   float inv_global_batch, fvit_stream_t stream);
/* SGD with momentum and (coupled) weight decay on a flat buffer: m = mu * m + (g + wd * p);  p -= lr * m. */
int fvit_sgd_momentum(float* param, float* momentum, const float* grad, int64_t n, float lr, float mu, float weight_decay,
                      fvit_stream_t stream);

/* Performance-experiment knobs for A/B runs inside one process; the same keys can be preset through the environment as
 * FVIT_TUNE_<key>=<int> (read once per key).  Kernel-selection knobs never change results beyond fp32 summation order:
 *   "mlp_fused", "attn_fused" 0/1; "mlp_fused_min_rows", "attn_fused_min_rows", "mlp_fused512_min_rows", "attn_fused512_min_rows";
 *   "mlp_variant" (-1 auto), "ab_variant", "ct_fused" 0/1 (carrier branch in one kernel), "win_fused" 0/1 (stage-3 attention sub-block in one kernel), "win_fused256" 0/1 (its 4-wave C = 256 instance for stage 2, off), "win_mlp" 0/1 (stage-3 MLP sub-block in one kernel), "win_mlp256" 0/1 (its 4-wave 64-row C = 256 instance for stage 2 (default) or fvit_mlp_fused's kernel), "ct_variant", "ct_touch", "ct8_depth" 2/3/4, "ln_gemm" 0/1, "ln_gemm_max_rows", "lngemm_nt", "gemm_stagger", "ab_stagger", "mlp_stagger" (0 / 1 / 2), "gemm_bm64_max_grid", "gemm_nw8_max_grid", "conv_halo" 0/1,
 *   "conv_halo_grid", "conv64_variant", "conv128_narrow", "stem_fused_grid";
 *   r05: "gemm_x3_dual" 1/0 (dual K tiles of the x3 GEMMs vs the K-concatenated walk), "conv_n128_ragged" 1/0 (128 x 128 conv tiles with a ragged last N tile
 *   for Cout % 128 == 64 vs 128 x 64 tiles), "gemm_splitk" 0/1;
 *   r06: "win_mlp_pipe" 1/0 (software-pipelined super-chunk loop of the C = 512 MLP kernel vs the plain loop; bitwise the same result).
 *   Removed in r06 with the kernel instances they selected (all measured no better than the defaults): "gemm_ring", "gemm_3stage_max_grid", "mlp_ring4_max_grid",
 *   "win_mlp256_depth", "win_stage3", "win_mlp256" = 1 / 3, "mlp_variant" = 2 / 4 / 5 / 6.
 * Guarded by a mutex; launches read the values at launch time.
 * DIAGNOSIS BUILD ONLY (libfvit_hip_diag.so, compiled with -DFVIT_DIAG; selected by FVIT_DIAG=1 in the Python binding): the "*_ablate" keys
 * ("mlp_ablate", "ab_ablate", "conv_ablate", "conv_halo_ablate") and "ablate_skip" switch off parts of a kernel / whole kernels for timing and
 * DO produce wrong results; in the shipped library they are compiled out (the keys are accepted and ignored). */
int fvit_tune(const char* key, int32_t value);

#ifdef FVIT_DIAG   /* ---- diagnosis entry points: libfvit_hip_diag.so only (r05); the shipped libfvit_hip.so does not export them ---- */
/* Diagnosis aid (tests / scripts only): `blocks` workgroups that fill 64 KiB of LDS each with a NaN pattern, spin `spin` iterations and
 * exit; run beside a forward on another stream, it exposes reads of uninitialised LDS.  sink: >= 4 bytes of device memory. */
int fvit_debug_lds_poison(void* sink, int32_t blocks, int32_t spin, fvit_stream_t stream);
/* Diagnosis aid (tests / scripts only): `blocks` single-wave workgroups that each write `pattern` into all 512 vector registers of the
 * wave and into 40 KiB of LDS, spin and exit (one wave fills a SIMD's register file: >= 1024 blocks cover the chip).  Run BEFORE a kernel
 * on the same stream it exposes reads of registers / LDS the kernel never wrote (they keep the previous occupant's content). */
int fvit_debug_regs_poison(void* sink, int32_t blocks, int32_t spin, uint32_t pattern, fvit_stream_t stream);
/* sink != null: that poison kernel (2048 blocks) is launched in front of EVERY kernel of this library, on the kernel's stream, until the
 * call is repeated with sink = null.  Results must not depend on it, nor on the pattern. */
int fvit_debug_poison_launches(void* sink, uint32_t pattern);

/* Diagnosis aid (tests / scripts only): between _begin and _end, fvit_hat_stage_forward appends one 32-bit hash per row of its fp32
 * streams (window tensor X, carrier stream R) to `buf` after every launch that writes them.  Single host thread only.  _end returns the
 * number of records and copies up to max_records of them. */
#define FVIT_DEBUG_MAX_RECORDS 512
typedef struct { char tag[24]; int64_t offset; int64_t rows; } FvitDebugRowhashRecord;
int fvit_debug_rowhash_begin(void* buf, int64_t capacity_words);
int fvit_debug_rowhash_end(FvitDebugRowhashRecord* out, int32_t max_records);
/* also copy the buffer of record number `record` (in trace order) to dst (device memory); up to 64 records; record < 0 clears the list */
int fvit_debug_rowhash_dump(int32_t record, void* dst, int64_t capacity_bytes);
/* Diagnosis aid: while active, every C = 256 fused-MLP launch writes per-lane hashes of its intermediate state (normalised input
 * fragments; per hidden chunk: the W1 fragments as read from LDS, the pre-GELU accumulators, the GELU output fragment, the W2 fragments as
 * read, the output accumulators) to consecutive slices of buf: [workgroup * 4 + wave][1 + 5 * hidden / 32][64 lanes] words.  _end returns
 * the number of traced launches (<= 64) with their slice offsets (words) and row counts.  Single host thread only. */
/* The 8-wave form of fvit_ct_block_fused (fp16) with s_memtime stamps per wave: stamps u64 [batch][8][16]: 0 entry, 1 constants in LDS + first
 * weight steps and rows requested, 2 LayerNorm 1, 3 barrier, 4 attention, 5 barrier, 6 proj + residual, 7 barrier, 8 LayerNorm 2, 9 fc1 + GELU,
 * 10 barrier, 11 fc2, 12 end. */
int fvit_debug_ct_block_timeline(const float* X, int32_t rowsA, const int32_t* src_idx, const float* add, float* R,
                                 int32_t batch, int32_t G, int32_t heads, int32_t C, int32_t hidden, const float* ln1_w, const float* ln1_b,
                                 const void* w_qkv_frag, const float* b_qkv_heads, const void* w_proj_frag, const float* b_proj, const float* gamma1,
                                 const float* bias, float scale, const float* ln2_w, const float* ln2_b, const void* w_fc1_frag, const float* b_fc1,
                                 const void* w_fc2_frag, const float* b_fc2, const float* gamma2, float eps, void* stamps, fvit_stream_t stream);
/* fvit_attn_block_fused (fp16, C = 256, 48 < S <= 64: the default 4-wave form) with s_memtime stamps per wave: stamps u64 [nwin][4][16]:
 * 0 entry, 1 first weight slice requested + rows gathered, 2 LayerNorm done, 3 .. 10 end of head 0 .. 7, 14 head loop done, 15 end. */
int fvit_debug_attn_block_timeline(const float* srcA, int32_t rowsA, const float* srcB, int32_t rowsB, const int32_t* src_idx,
                                   const int32_t* add_idx, const float* add, const float* ln_w, const float* ln_b, float eps,
                                   int32_t rows_per_image, const void* w_qkv_frag, const float* b_qkv_heads, const void* w_proj_frag,
                                   const float* b_proj, const float* gamma, const float* bias, float* x_out, int32_t nwin, int32_t S,
                                   int32_t heads, int32_t C, float scale, void* stamps, fvit_stream_t stream);
/* fvit_conv3x3_c128_band (fp16) with s_memtime stamps per wave: stamps u64 [B * bands][4][8], bands = ceil(H / (224 / (W + 2))):
 * 0 start, 1 band DMA + first weight steps requested, 2 band landed, 3 K loop done, 4 first half of the epilogue done, 5 end. */
int fvit_debug_conv_band_timeline(const void* in, const void* w_frag, const float* bias, const void* residual, void* out, int32_t B,
                                  int32_t H, int32_t W, int32_t act, const void* zeros, void* stamps, fvit_stream_t stream);
/* Phase accounting of the fused stem kernel (fp16): the same launch as fvit_stem_fused with every wave accumulating s_memtime ticks per phase,
 * u64 [workgroups (<= 512)][4 waves][8]: 0 phase A (gathers + conv1 + LDS writes), 1 barrier after A, 2 phase B (conv2), 3 epilogue,
 * 4 barrier before A, 5 tiles processed, 6 kernel entry -> exit. */
int fvit_debug_stem_timeline(const FvitMapView* in, const void* w1, const float* b1, const void* w2, const float* b2, void* out,
                             int32_t B, int32_t Hi, int32_t Wi, void* stamps, fvit_stream_t stream);
/* Phase timeline of the N-split MLP kernel (fp16, one weight term; C = 512, or C = 256 in its 4-wave form): the same launch as
 * fvit_win_mlp_fused with lane 0 of every wave writing s_memtime stamps, u64 [workgroups][waves][16]:
 * 0 entry, 1 first ring steps issued, 2 rows loaded, 3 LayerNorm published, 4..13 end of super-chunk, 14 before the epilogue, 15 end. */
int fvit_debug_win_mlp_timeline(float* x, int32_t M, int32_t C, int32_t hidden, const float* ln_w, const float* ln_b, float eps,
                                const void* w_fc1_frag, const float* b_fc1, const void* w_fc2_frag, const float* b_fc2, const float* gamma,
                                void* stamps, fvit_stream_t stream);
int fvit_debug_mlp_trace_begin(void* buf, int64_t capacity_words);
int fvit_debug_mlp_trace_end(int64_t* offsets, int32_t* rows, int32_t max_launches);
/* same protocol: every C = 256 fused-MLP launch stores its input rows exactly as its own loads returned them ([M][256] floats per launch) */
int fvit_debug_mlp_inputs_begin(void* buf, int64_t capacity_floats);
int fvit_debug_mlp_inputs_end(int64_t* offsets, int32_t* rows, int32_t max_launches);
#endif /* FVIT_DIAG */

/* ---- built-in kernel timer (HIP events around every launch, on the launch stream) ---- */
#define FVIT_PROF_KINDS 11
/* kind ids */
#define FVIT_K_PARTITION 0
#define FVIT_K_LAYERNORM 1
#define FVIT_K_GEMM_BIAS 2
#define FVIT_K_GEMM_GELU 3
#define FVIT_K_GEMM_RESID 4
#define FVIT_K_ATTENTION 5
#define FVIT_K_REVERSE 6
#define FVIT_K_OTHER 7
#define FVIT_K_MLP_FUSED 8
#define FVIT_K_CONV 9
#define FVIT_K_ATTN_FUSED 10
typedef struct FvitProfEntry {
    int64_t launches;
    double ms;     /* summed event-to-event time */
    double flops;  /* algorithmic FLOPs of those launches (2*M*N*K for GEMMs, 4*S*S*d per head for attention) */
    double bytes;  /* algorithmic HBM bytes (compulsory reads + writes) */
} FvitProfEntry;
/* one record per launch, in launch order: what bench.py builds its per-(kernel, launch shape) roofline rows from */
typedef struct FvitProfRecord {
    int32_t kind;   /* FVIT_K_* */
    int32_t grid;   /* workgroups of the launch (0 when the launcher does not report it) */
    float ms;       /* event-to-event time of THIS launch */
    float _pad;
    double flops;   /* algorithmic FLOPs of this launch */
    double bytes;   /* algorithmic (compulsory) HBM bytes of this launch */
    char name[40];  /* kernel family + variant, e.g. "gemm_residual bm64" */
} FvitProfRecord;
int fvit_prof_enable(int on);             /* starts/stops recording; enabling resets the counters */
int fvit_prof_collect(FvitProfEntry* out); /* synchronises the recorded events; out[FVIT_PROF_KINDS] */
int fvit_prof_records(FvitProfRecord* out, int32_t max_records); /* per-launch records since fvit_prof_enable(1); returns the count (<= max) or < 0 */
const char* fvit_prof_kind_name(int kind);

#ifdef __cplusplus
}
#endif
#endif /* FVIT_HIP_H */
