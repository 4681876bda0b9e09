// Internal header shared by the HIP translation units of libfvit_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <mutex>
#include <stdio.h>

#include "../../include/fvit_hip.h"

namespace fvit {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// 16-bit MFMA operand traits: fp16 (default) or bf16, fp32 accumulate.
template <typename T> struct Op16;
template <> struct Op16<_Float16> {
    typedef h8 v8;
    typedef h4 v4;
    static __device__ __forceinline__ f4 mfma(v8 a, v8 b, f4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Op16<__bf16> {
    typedef b8 v8;
    typedef b4 v4;
    static __device__ __forceinline__ f4 mfma(v8 a, v8 b, f4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

// Narrowing of an fp32 activation to the MFMA operand type.  fp16 SATURATES at +-65504 (one v_med3_f32): an activation beyond the
// fp16 range becomes the largest finite value instead of inf, so that no inf - inf / 0 * inf NaN can appear downstream (softmax of
// +-inf scores, a zero-padded weight column times inf) and a checkpoint with out-of-range activations degrades instead of
// poisoning the batch (tests/test_gpu_kernels.py::test_f16_operands_saturate).  bf16 has the fp32 exponent range: plain cast.
template <typename T> __device__ __forceinline__ T sat16(float x);
template <> __device__ __forceinline__ _Float16 sat16<_Float16>(float x) { return (_Float16)__builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f); }
template <> __device__ __forceinline__ __bf16 sat16<__bf16>(float x) { return (__bf16)x; }

// Cross-row lane exchanges on the VALU (gfx950 v_permlane16_swap / v_permlane32_swap) instead of __shfl_xor, which is
// ds_bpermute_b32 -- an LDS-pipeline instruction.  r02 finding (profiles/r02_repeatability_hunt.log): in kernels that have LDS-DMA
// (global_load_lds) traffic landing in the workgroup's LDS, a ds_bpermute issued by a wave whose sibling waves' DMA is still in
// flight occasionally returned a wrong lane value when kernels of another stream shared the CU (the LayerNorm row mean of one wave
// off by a partial sum: all 16 rows of that wave shifted, ~1e-3 relative in the kernel's output, different run to run).  The swaps
// touch no LDS hardware, and are two VALU instructions instead of an LDS round trip.
//   v_permlane16_swap v, s: odd 16-lane rows of v <-> even rows of s; with v = s = x, {v, s} afterwards = {own or partner, partner or own}
//   of lane ^ 16 in a fixed order per lane pair, so op(v, s) is bitwise the same in both lanes of a pair for commutative op.
__device__ __forceinline__ float sum_xor16(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_xor32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float max_xor16(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float max_xor32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// all-reduce over groups of N consecutive lanes (N a power of two <= 64; every lane of a group gets the result): DPP inside a 16-lane
// row, the swaps above across rows -- no LDS-pipeline instruction.  quad_perm [1,0,3,2] / [2,3,0,1] are lane ^ 1 / lane ^ 2; after them
// every quad is uniform, so row_half_mirror (lane 7 - i of the 8-group) and row_mirror (lane 15 - i) pair each quad / half row with
// the other one exactly like lane ^ 4 / lane ^ 8 would.
template <int CTRL>
__device__ __forceinline__ float dpp_lane(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
template <int N>
__device__ __forceinline__ float group_sum(float x) {
    static_assert(N >= 1 && N <= 64 && (N & (N - 1)) == 0, "group size");
    if constexpr (N >= 2) x += dpp_lane<0xB1>(x);
    if constexpr (N >= 4) x += dpp_lane<0x4E>(x);
    if constexpr (N >= 8) x += dpp_lane<0x141>(x);
    if constexpr (N >= 16) x += dpp_lane<0x140>(x);
    if constexpr (N >= 32) x = sum_xor16(x);
    if constexpr (N >= 64) x = sum_xor32(x);
    return x;
}
template <int N>
__device__ __forceinline__ float group_max(float x) {
    static_assert(N >= 1 && N <= 64 && (N & (N - 1)) == 0, "group size");
    if constexpr (N >= 2) x = fmaxf(x, dpp_lane<0xB1>(x));
    if constexpr (N >= 4) x = fmaxf(x, dpp_lane<0x4E>(x));
    if constexpr (N >= 8) x = fmaxf(x, dpp_lane<0x141>(x));
    if constexpr (N >= 16) x = fmaxf(x, dpp_lane<0x140>(x));
    if constexpr (N >= 32) x = max_xor16(x);
    if constexpr (N >= 64) x = max_xor32(x);
    return x;
}

// erf with |error| <= 1.5e-7 (Abramowitz & Stegun 7.1.26): far below the 16-bit rounding of any GELU
// output here, branch-free and much cheaper than libm erff inside fused epilogues.
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(1.0f + 0.3275911f * ax);
    float y = 1.061405429f;
    y = y * t - 1.453152027f;
    y = y * t + 1.421413741f;
    y = y * t - 0.284496736f;
    y = y * t + 0.254829592f;
    y = 1.0f - y * t * __expf(-ax * ax);
    return copysignf(y, x);
}
// nn.GELU() (exact-erf form, AR:403) without transcendentals: erf(z) ~ z * Q(z^2) on |z| <= 3 (degree-8 minimax fit,
// |erf error| <= 2.6e-5 incl. the clamp tail 1 - erf(3) = 2.2e-5), GELU absolute error <= 5.4e-5 for all x -- an order of
// magnitude below the 16-bit rounding of the value it feeds (scripts/fit_gelu.py reproduces the fit and the bound).
// 14 plain VALU ops (SLP-packable into v_pk_fma_f32) instead of ~23 issue-cycle equivalents with rcp + exp: the fused
// epilogues are VALU-bound, not MFMA-bound, on this term.
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = __builtin_amdgcn_fmed3f(x * 0.70710678118654752f, -3.0f, 3.0f);
    const float u = z * z;
    float q = 4.075095461e-08f;
    q = q * u - 1.945139275e-06f;
    q = q * u + 4.106515917e-05f;
    q = q * u - 5.110726343e-04f;
    q = q * u + 4.235583358e-03f;
    q = q * u - 2.510324307e-02f;
    q = q * u + 1.110798195e-01f;
    q = q * u - 3.753151596e-01f;
    q = q * u + 1.128268480e+00f;
    const float hx = 0.5f * x;
    return hx * (z * q) + hx;
}

// gelu_fast over N values at once, every Horner step issued for all N values before the next step (r06).  The per-value form above compiles to
// ONE dependent chain of 15 VALU instructions per value and hipcc emits the values one after the other: the fused MLP kernels ran 32 such chains
// back to back per super-chunk (r05 ISA: 224 v_fmaak + 32 v_fmamk in strict dependence, ~3.5 issue cycles each instead of 2).  Source order alone
// does not survive (the DAG scheduler re-serialises the chains into the register-minimal order, and __builtin_amdgcn_sched_barrier has no data
// edge to them: r06 ISA check): an empty asm that takes all N partial results as read-write operands pins every step.  Same operations in the
// same order per value => bitwise the same results as gelu_fast.
template <int N>
__device__ __forceinline__ void pin_values(float (&q)[N]) {
    static_assert(N == 4 || N == 8, "pin_values: 4 or 8 values");
    // (not volatile: the data dependence alone orders the steps, and the statement stays free to move between independent instructions, e.g. MFMAs)
    if constexpr (N == 4) asm("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
    else asm("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]));
}
template <int N>
__device__ __forceinline__ void gelu_fast_n(float (&x)[N]) {
    float z[N], u[N], q[N];
#pragma unroll
    for (int i = 0; i < N; ++i) z[i] = __builtin_amdgcn_fmed3f(x[i] * 0.70710678118654752f, -3.0f, 3.0f);
#pragma unroll
    for (int i = 0; i < N; ++i) u[i] = z[i] * z[i];
    pin_values(u);
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = 4.075095461e-08f * u[i] - 1.945139275e-06f;
#define FVIT_GELU_STEP(c) \
    pin_values(q);        \
    _Pragma("unroll") for (int i = 0; i < N; ++i) q[i] = q[i] * u[i] + (c);
    FVIT_GELU_STEP(4.106515917e-05f)
    FVIT_GELU_STEP(-5.110726343e-04f)
    FVIT_GELU_STEP(4.235583358e-03f)
    FVIT_GELU_STEP(-2.510324307e-02f)
    FVIT_GELU_STEP(1.110798195e-01f)
    FVIT_GELU_STEP(-3.753151596e-01f)
    FVIT_GELU_STEP(1.128268480e+00f)
#undef FVIT_GELU_STEP
    pin_values(q);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float hx = 0.5f * x[i];
        x[i] = hx * (z[i] * q[i]) + hx;
    }
}

// M values (a multiple of 4) in groups of 8 (+ one group of 4)
template <int M>
__device__ __forceinline__ void gelu_fast_each(float (&x)[M]) {
    static_assert(M % 4 == 0, "gelu_fast_each: a multiple of 4 values");
#pragma unroll
    for (int j = 0; j + 8 <= M; j += 8) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = x[j + i];
        gelu_fast_n<8>(t);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[j + i] = t[i];
    }
    if constexpr (M % 8 == 4) {
        float t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = x[M - 4 + i];
        gelu_fast_n<4>(t);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[M - 4 + i] = t[i];
    }
}

// the same function through fast_erf (|erf error| <= 1.5e-7): the two-term-activation mode (weight_terms 3) carries ~22 significant
// bits through every Linear layer, the polynomial above would be its error floor
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f)); }
// gelu_erf over N values in lockstep (the two-term-activation epilogues): the same operations per value, the steps pinned like gelu_fast_n
template <int N>
__device__ __forceinline__ void gelu_erf_n(float (&x)[N]) {
    float ax[N], t[N], y[N], e[N];
#pragma unroll
    for (int i = 0; i < N; ++i) ax[i] = fabsf(x[i] * 0.70710678118654752f);
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = __frcp_rn(1.0f + 0.3275911f * ax[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = __expf(-ax[i] * ax[i]);
    pin_values(t);
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = 1.061405429f * t[i] - 1.453152027f;
    pin_values(y);
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = y[i] * t[i] + 1.421413741f;
    pin_values(y);
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = y[i] * t[i] - 0.284496736f;
    pin_values(y);
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = y[i] * t[i] + 0.254829592f;
    pin_values(y);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float r = copysignf(1.0f - y[i] * t[i] * e[i], x[i] * 0.70710678118654752f);
        x[i] = 0.5f * x[i] * (1.0f + r);
    }
}
template <int M>
__device__ __forceinline__ void gelu_erf_each(float (&x)[M]) {
    static_assert(M % 4 == 0, "gelu_erf_each: a multiple of 4 values");
#pragma unroll
    for (int j = 0; j + 8 <= M; j += 8) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = x[j + i];
        gelu_erf_n<8>(t);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[j + i] = t[i];
    }
    if constexpr (M % 8 == 4) {
        float t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = x[M - 4 + i];
        gelu_erf_n<4>(t);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[M - 4 + i] = t[i];
    }
}

__host__ __device__ constexpr int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: remember which devices a kernel instance was opted in on
// (one static DeviceOnce per kernel instance; one process may drive several GPUs, e.g. nn.DataParallel in the reference's validate.py)
struct DeviceOnce {
    std::atomic<uint64_t> mask{0};
    bool first_on_current_device() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;   // unknown: just set the attribute again
        const uint64_t bit = 1ull << dev;
        return (mask.fetch_or(bit) & bit) == 0;
    }
};

// ---- error plumbing (thread-local message, negative return codes) ----
void set_error(const char* fmt, ...);
int check_launch(const char* what);

// ---- tuning knobs (A/B experiments inside one process; see fvit_tune in fvit_hip.h) ----
int tune_get(const char* key, int dflt);
// timing ablation (results are WRONG when non-zero; bench A/B only): skip the launches of a kernel family to measure its marginal cost
// inside the concurrent stream shards.  bits: 1 winmlp<256>, 2 winmlp<512>, 4 winblk, 8 attnblk, 16 ctblk, 32 conv3x3 implicit GEMM,
// 64 halo conv, 128 fused stem
// Measurement hazards live only in the DIAGNOSIS build (libfvit_hip_diag.so, -DFVIT_DIAG): the shipped library has no knob that
// can make a kernel skip work and no fvit_debug_* entry point.
#ifdef FVIT_DIAG
inline bool ablate_skip(int bit) { return (tune_get("ablate_skip", 0) & bit) != 0; }
inline int diag_knob(const char* key) { return tune_get(key, 0); }   // per-kernel ablation masks ("conv_ablate", "mlp_ablate", ...): results are WRONG when non-zero
#else
inline bool ablate_skip(int) { return false; }
inline int diag_knob(const char*) { return 0; }
#endif

// ---- diagnostics state (kernel timer records, row-hash / MLP traces, poison sink) is process-global: every access -- including
// the ones inside ProfScope, i.e. on every launch -- takes this mutex, so that host threads driving different devices / streams
// (nn.DataParallel replicas) may launch while a profile or a trace is being recorded.  Uncontended cost: one atomic per launch.
std::recursive_mutex& diag_mutex();

// ---- built-in kernel timer ----
struct ProfScope {
    ProfScope(int kind, double flops, double bytes, hipStream_t stream);
    ~ProfScope();
    int slot_;
    hipStream_t stream_;
};
// names the launch the innermost live ProfScope brackets: kernel family + launch shape (workgroups), so that the per-launch records
// (fvit_prof_records) can be matched with a rocprofv3 kernel trace / PMC row of the same (kernel, grid).  No-op when the timer is off.
void prof_note(const char* kernel, int grid);
#ifdef FVIT_DIAG
void dbg_poison_before_launch(hipStream_t st);   // diagnosis: register / LDS poison kernel in front of every launch (fvit_debug_poison_launches)
void dbg_rowhash(const char* tag, const void* ptr, long long rows, int row_bytes, hipStream_t st);   // no-op unless fvit_debug_rowhash_begin is active
#else
inline void dbg_poison_before_launch(hipStream_t) {}
inline void dbg_rowhash(const char*, const void*, long long, int, hipStream_t) {}
#endif

// ---- launchers (defined in the .hip files) ----
struct GemmCall {
    int dtype;           // FVIT_F16 / FVIT_BF16
    const void* A;       // activations op16 [pad128(M)][lda]
    int lda;
    const void* W;       // weights op16 [pad128(N)][ldw]
    int ldw;
    const float* bias;   // [N] or null
    const float* gamma;  // [N] or null (residual epilogue only)
    void* out;           // op16 [..][ldo] (bias / gelu) or f32 x [..][ldo] (residual)
    int ldo;
    int M, N, K;
    int epilogue;        // 0 bias, 1 bias+gelu, 2 gamma-residual into f32
    // residual epilogue only (optional): after the residual update, x[m][:] += add[add_idx ? add_idx[m % rows_per_image] : m % rows_per_image]
    // (rows of length N; negative index = none).  Used to apply the NEXT block's position embedding (AR:671: x = pos_embed(x)) in
    // the fc2 epilogue of the current block, so that the next block's norm1 has no in-place add and folds into its qkv GEMM.
    const float* add = nullptr;
    const int32_t* add_idx = nullptr;
    int rows_per_image = 1;
    // K-concatenated weight terms (FvitStageDesc.weight_terms): the weight rows hold K columns = terms x ka, the activation rows ka
    // columns that are re-used for every term (column k of the contraction reads A column k >= ka ? k - ka : k).  0 = K.
    // K = 3 ka (weight_terms 3, "x3"): weights [hi | lo | hi], activation rows hold TWO terms [hi | lo] (2 ka columns): the three
    // segments are hi.hi + hi_a.lo_w + lo_a.hi_w, i.e. both operands carried as two 16-bit terms, the lo.lo product dropped.
    int ka = 0;
    // epilogues 0 / 1: > 0 = the output is stored as two terms, hi at column n and lo = round(y - hi) at column out_lo_off + n
    int out_lo_off = 0;
    // residual epilogue: optional fp32 scratch for deterministic split-K (small grids with long K, see fvit_gemm.hip); nullptr = never split
    float* splitk_slab = nullptr;
    size_t splitk_bytes = 0;
};
int launch_gemm(const GemmCall& c, hipStream_t stream);

struct MlpFusedCall {
    int dtype;
    float* x;  // [M][C] f32 in place
    int M, C, hidden;
    const float* ln_w;
    const float* ln_b;
    float eps;
    const void* w1f;  // fragment-major fc1 weight
    const float* b1;
    const void* w2f;  // fragment-major fc2 weight
    const float* b2;
    const float* gamma;
    int terms = 1;    // weight terms (1, or 2 = [hi image | lo image]); fvit_winmlp.hip only
    // fvit_winmlp.hip, C = 512: split the hidden units of every 64-row group over nsplit (2 / 4) workgroups that meet in L2
    float* slab = nullptr;     // f32 [ceil(M / 64)][nsplit][64 x C], scratch
    int* counters = nullptr;   // int32 [ceil(M / 64)], zero before the first launch (the kernel leaves them zero)
    int nsplit = 1;
    void* ts = nullptr;        // diagnosis (fvit_debug_win_mlp_timeline): u64 [workgroups][waves][16] phase stamps
};
size_t winmlp_split_slab_bytes(int64_t M, int C, int nsplit);
bool mlp_fused_supported(int C, int hidden);
int launch_mlp_fused(const MlpFusedCall& c, hipStream_t stream);
// same contract for C = 512 / hidden 2048: 64-row workgroups whose 8 waves split hidden units / output channels (fvit_winmlp.hip)
bool winmlp_supported(int C, int hidden);
int launch_winmlp(const MlpFusedCall& c, hipStream_t stream);

struct AttnBlkCall {
    int dtype;
    // gather + LayerNorm source (fvit_gather_layernorm semantics)
    const float* srcA;
    int rowsA;
    const float* srcB;
    int rowsB;
    const int32_t* src_idx;
    const int32_t* add_idx;
    const float* add;
    const float* ln_w;
    const float* ln_b;
    float eps;
    int rows_per_image;
    // fragment-major weights
    const void* wqkv_f;
    const float* bqkv;
    const void* wproj_f;
    const float* bproj;
    const float* gamma;
    const float* bias;
    float* x_out;
    int nwin, S, heads, C;
    float scale;
    // fvit_winblk.hip, C = 512: split the heads of every window over two workgroups that meet in L2
    float* slab = nullptr;     // f32 [nwin][nsplit][64 x C], scratch
    int* counters = nullptr;   // int32 [nwin], zero before the first launch (the kernel leaves them zero)
    int nsplit = 1;
    int terms = 1;             // weight terms of the fragment arrays (fvit_winblk.hip, C = 512: 1 or 2; fvit_attnblk.hip: 1)
    void* ts = nullptr;        // fvit_attnblk.hip: stamp buffer of the timeline instance (fvit_debug_attn_block_timeline)
};
bool attnblk_supported(int C, int heads, int S);
int launch_attnblk(const AttnBlkCall& c, hipStream_t stream);
// r06: the same contract for C = 256 / 8 heads / 49..64-token windows with a wave per (window, head) (fvit_attnblk2.hip); launch_attnblk dispatches to it
bool attnblk2_supported(int C, int heads, int S);
int launch_attnblk2(const AttnBlkCall& c, hipStream_t stream);
// same contract for C = 512 / 16 heads, one 49..64-token window per workgroup, waves split heads / output channels (fvit_winblk.hip)
bool winblk_supported(int C, int heads, int S);
int launch_winblk(const AttnBlkCall& c, hipStream_t stream);
struct AttnCall {
    int dtype;
    const void* qkv;  // op16 [rows][ldq], columns [q|k|v][head][dpad]
    int ldq;
    void* out;        // op16 [rows][ldo], columns [head][dpad]
    int ldo;
    const float* bias;  // f32 [heads][Spad][Spad] (S <= FVIT_MAX_DENSE_SEQ)
    int nwin, S, heads, dpad;
    float scale;
    // S > FVIT_MAX_DENSE_SEQ: compact bias table f32 [heads][(2*rel_w-1)^2] (or null), n_g = rel_ng leading tokens without bias
    const float* rel_table;
    int rel_w, rel_ng;
    int d;  // real head_dim (<= dpad) for the FLOP count of the kernel timer; 0 = unknown (dpad is used)
    // two-term activations (weight_terms 3): qkv rows hold [q|k|v] hi at column 0 and lo at column q_lo_off; scores = qh.kh + qh.kl + ql.kh,
    // P and V as two terms too (P in registers), the output written as hi at column 0 and lo at column o_lo_off.  0 = single terms.
    int q_lo_off = 0, o_lo_off = 0;
    // train mode (r05): Dropout on the softmax probabilities (WindowAttention.attn_drop, FV:564): op16 [nwin * heads][S][Spad], 0 or 1 / keep; null = none
    const void* drop_mask = nullptr;
};
bool attention_dense(int S, int dpad);                               // in-register kernel + dense bias table, else the long kernel
int launch_attention(const AttnCall& c, hipStream_t stream);        // dispatches on attention_dense(S, dpad)
int launch_attention_long(const AttnCall& c, hipStream_t stream);   // fvit_attnlong.hip

struct LnCall {
    int dtype;
    const float* srcA;
    int rowsA;
    const float* srcB;
    int rowsB;
    const int32_t* src_idx;
    const int32_t* add_idx;
    const float* add;
    float* x_out;
    void* n_out;
    int ldn;
    const float* ln_w;
    const float* ln_b;
    float eps;
    int rows, rows_per_image, C;
    int lo_off = 0;   // > 0: n_out holds two terms per row, hi at column c and lo = round(y - hi) at column lo_off + c (ldn >= lo_off + pad64(C))
};
int launch_gather_layernorm(const LnCall& c, hipStream_t stream);

// LayerNorm folded into the A-operand staging of the Linear layer that consumes it (fvit_lngemm.hip); ln.n_out / ln.ldn are unused
// whole carrier-token branch of one HAT block (fvit_ctblk.hip)
struct CtBlkCall {
    int dtype;
    const float* X; int rowsA;        // window tensor, rows per image
    const int32_t* src_idx;           // [G]
    const float* add;                 // hat_pos_embed rows [G][C] or null
    float* R;                         // out [batch * G][C]
    int batch, G, heads, C, hidden;
    const float* ln1_w; const float* ln1_b; const void* wqkv_f; const float* bqkv; const void* wproj_f; const float* bproj; const float* gamma1;
    const float* bias; float scale;
    const float* ln2_w; const float* ln2_b; const void* w1f; const float* b1; const void* w2f; const float* b2; const float* gamma2;
    float eps;
    int terms = 1;   // weight terms of the four fragment arrays (1 or 2)
    void* ts = nullptr;   // stamp buffer of the timeline instance (fvit_debug_ct_block_timeline)
};
bool ctblk_supported(int C, int heads, int G, int hidden);
int launch_ctblk(const CtBlkCall& c, hipStream_t stream);

struct LnGemmCall {
    LnCall ln;           // row selection + LayerNorm parameters (x_out optional, must not alias the sources)
    const void* W;       // weights op16 [pad128(N)][ldw]
    int ldw;
    const float* bias;   // [N] or null
    void* out;           // op16 [..][ldo]
    int ldo;
    int N;
    int epilogue;        // 0 bias, 1 bias + GELU
};
bool ln_gemm_supported(int C, int N, int ldw, int ldo);
int launch_ln_gemm(const LnGemmCall& c, hipStream_t stream);

struct PartitionCall {
    FvitMapView in;  // (B, C, Hp, Wp)
    int batch, C, Hp, Wp, ws;
    float* x;          // token rows, f32, row length C
    int rows_per_win;  // S: rows reserved per window in x (ws*ws + ncw)
    int row_off;       // ncw: first local-token row inside a window
    const float* ct;   // optional (B, G, C) windowed carrier tokens copied into rows [0, ncw) (or null)
    int ncw;
};
int launch_partition(const PartitionCall& c, hipStream_t stream);

struct ReverseCall {
    const float* x;
    int rows_per_win, row_off;
    int batch, C, Hp, Wp, H, W, ws;
    FvitMapView out;       // (B, C, H, W)
    const float* gamma;    // propagation: out = x + gamma * x[carrier up_idx] (gamma null => 1)
    const int32_t* up_idx; // null => no propagation
};
int launch_reverse(const ReverseCall& c, hipStream_t stream);

// copy f32 rows [rows][C] out of / into the windowed tensor's carrier slots (block-level API)
int launch_ct_copy(float* x, int rows_per_win, int row_off, int ncw, float* ct, int nwin_total, int C, int to_x,
                   hipStream_t stream);
// TokenInitializer: depthwise conv3x3 + bias + AvgPool2d + per-window reorder -> f32 (B, G, C)
int launch_token_init(const FvitMapView& in, const float* w, const float* bias, float* out, int B, int C, int Hp, int Wp, int kh, int kw,
                      int sh, int sw, int cw, hipStream_t stream);
// x[win][ncw + t] += gamma * x[win][up_idx[t]] (block-level API; the stage API fuses this into window_reverse)
int launch_propagate(float* x, const float* gamma, const int32_t* up_idx, int rows_per_win, int ncw, int nloc,
                     int nwin_total, int C, hipStream_t stream);

}  // namespace fvit
