// fvit_bwd.hip -- the memory-bound pieces of the backward pass of the MLP sub-block of a HAT block
//     y = x + gamma * fc2(GELU(fc1(LayerNorm(x))))        (Mlp.forward FV:398-407, HAT.forward FV:691 / AR:697; train.py:820-951 runs autograd over it)
// SURVEY.md section 8 row f-4 (training path), the part VERDICT r02 item 9 scopes: dX, dW1, dW2, db1, db2, dgamma and the LayerNorm gradients of
// ONE sub-block, checked against torch.autograd (tests/test_gpu_backward.py).  The four GEMMs of the backward (dh = dz W2, dxn = da W1,
// dW2 = dz^T h, dW1 = da^T xn) run on gemm_kernel / gemm_pp_kernel through the existing entry points (16-bit operands, fp32 accumulation,
// fp32 outputs through the residual epilogue into zeroed buffers -- which is also gradient accumulation); this file holds what is left:
//   transpose16        op16 [M][N] -> [N][pad64(M)] (zero padded): the "A^T B" GEMMs read both operands K-contiguous
//   scale_cols         dz = gamma * dy (op16) and the column partial sums of dy * z (-> dgamma) and gamma * dy (-> db2)
//   gelu_fwd / gelu_bwd  h = GELU(a);  da = dh * GELU'(a) and the column partial sums of da (-> db1)      (erf form, libm erff / expf)
//   layernorm_bwd_rows dx = dy + rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dxn * w; per-row (mean, rstd) saved
//   layernorm_bwd_cols column partial sums of dxn * xhat (-> dw) and dxn (-> db)
//   colsum_finish      partial sums added in block order: no atomics, bit-reproducible gradients
#include "fvit_common.h"

namespace fvit {
namespace {

constexpr int BWD_ROWS = 64;   // rows per block of the column-sum kernels

template <typename T>
__global__ __launch_bounds__(256) void transpose16_kernel(const T* __restrict__ in, int ld_in, T* __restrict__ out, int ld_out, int M, int N) {
    __shared__ T tile[64][66];
    const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int m = m0 + r, n = n0 + tx;
        tile[r][tx] = (m < M && n < N) ? in[(size_t)m * ld_in + n] : (T)0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int n = n0 + r, m = m0 + tx;
        if (n < N && m < ld_out) out[(size_t)n * ld_out + m] = tile[tx][r];
    }
}

// part: f32 [blocks][2][C]: [0] = sum_m dy * z, [1] = sum_m gamma * dy over the block's rows
template <typename T>
__global__ __launch_bounds__(256) void scale_cols_kernel(const float* __restrict__ dy, const T* __restrict__ z, int ldz, const float* __restrict__ gamma,
                                                          T* __restrict__ dz, int lddz, float* __restrict__ part, int M, int C) {
    const int m0 = blockIdx.x * BWD_ROWS, m1 = min(m0 + BWD_ROWS, M);
    for (int c = threadIdx.x; c < C; c += 256) {
        const float g = gamma ? gamma[c] : 1.0f;
        float s0 = 0.f, s1 = 0.f;
        for (int m = m0; m < m1; ++m) {
            const float d = dy[(size_t)m * C + c];
            const float v = g * d;
            s0 += d * (float)z[(size_t)m * ldz + c];
            s1 += v;
            dz[(size_t)m * lddz + c] = (T)v;
        }
        part[((size_t)blockIdx.x * 2 + 0) * C + c] = s0;
        part[((size_t)blockIdx.x * 2 + 1) * C + c] = s1;
    }
}

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

// MODE 0: out = GELU(a).  MODE 1: out = dh * GELU'(a), part f32 [blocks][H] = column sums of out (unrounded) over the block's rows
template <typename T, int MODE>
__global__ __launch_bounds__(256) void gelu_kernel(const T* __restrict__ a, int lda, const T* __restrict__ dh, int lddh, T* __restrict__ out, int ldo,
                                                    float* __restrict__ part, int M, int H) {
    const int m0 = blockIdx.x * BWD_ROWS, m1 = min(m0 + BWD_ROWS, M);
    for (int j = threadIdx.x; j < H; j += 256) {
        float s = 0.f;
        for (int m = m0; m < m1; ++m) {
            const float x = (float)a[(size_t)m * lda + j];
            float v;
            if (MODE == 0) v = gelu_exact(x);
            else {
                v = (float)dh[(size_t)m * lddh + j] * gelu_grad(x);
                s += v;
            }
            out[(size_t)m * ldo + j] = (T)v;
        }
        if (MODE == 1) part[(size_t)blockIdx.x * H + j] = s;
    }
}

// one wave per row: stats[m] = (mean, rstd); dx[m] = dy[m] + rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dxn * w
__global__ __launch_bounds__(256) void layernorm_bwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ dxn, const float* __restrict__ dy,
                                                                  const float* __restrict__ w, float eps, float* __restrict__ dx, float* __restrict__ stats,
                                                                  int M, int C) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* xr = x + (size_t)m * C;
    const float* gr = dxn + (size_t)m * C;
    auto wsum = [](float v) { return group_sum<64>(v); };   // VALU lane exchanges (fvit_common.h), no LDS-pipeline shuffles
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mean = wsum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; q += d * d; }
    const float rstd = rsqrtf(wsum(q) / (float)C + eps);
    float a1 = 0.f, a2 = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float g = gr[c] * w[c];
        a1 += g;
        a2 += g * (xr[c] - mean) * rstd;
    }
    const float m1 = wsum(a1) / (float)C, m2 = wsum(a2) / (float)C;
    for (int c = lane; c < C; c += 64) {
        const float xh = (xr[c] - mean) * rstd;
        const float g = gr[c] * w[c];
        dx[(size_t)m * C + c] = (dy ? dy[(size_t)m * C + c] : 0.f) + rstd * (g - m1 - xh * m2);
    }
    if (lane == 0) { stats[2 * m] = mean; stats[2 * m + 1] = rstd; }
}

// part f32 [blocks][2][C]: [0] = sum_m dxn * xhat, [1] = sum_m dxn
__global__ __launch_bounds__(256) void layernorm_bwd_cols_kernel(const float* __restrict__ x, const float* __restrict__ dxn, const float* __restrict__ stats,
                                                                  float* __restrict__ part, int M, int C) {
    const int m0 = blockIdx.x * BWD_ROWS, m1 = min(m0 + BWD_ROWS, M);
    for (int c = threadIdx.x; c < C; c += 256) {
        float s0 = 0.f, s1 = 0.f;
        for (int m = m0; m < m1; ++m) {
            const float d = dxn[(size_t)m * C + c];
            s0 += d * (x[(size_t)m * C + c] - stats[2 * m]) * stats[2 * m + 1];
            s1 += d;
        }
        part[((size_t)blockIdx.x * 2 + 0) * C + c] = s0;
        part[((size_t)blockIdx.x * 2 + 1) * C + c] = s1;
    }
}

// column partial sums of an op16 matrix: part f32 [blocks][N] (-> the qkv bias gradient)
template <typename T>
__global__ __launch_bounds__(256) void colsum16_kernel(const T* __restrict__ in, int ld, float* __restrict__ part, int M, int N) {
    const int m0 = blockIdx.x * BWD_ROWS, m1 = min(m0 + BWD_ROWS, M);
    for (int n = threadIdx.x; n < N; n += 256) {
        float s = 0.f;
        for (int m = m0; m < m1; ++m) s += (float)in[(size_t)m * ld + n];
        part[(size_t)blockIdx.x * N + n] = s;
    }
}

// Backward of the windowed attention core (WindowAttention.forward FV:557-568 between the qkv and proj Linears), one workgroup per (window, head):
//   S = q k^T * scale + bias, P = softmax(S), O = P v;  given dO:
//   dV = P^T dO;  dP = dO v^T;  dS = P * (dP - rowsum(dP * P));  dq = scale * dS k;  dk = scale * dS^T q;  dbias = dS
// qkv / dqkv: op16 [rows][ld], columns [q|k|v][head][D] (D = the PADDED head_dim 32 / 64 / 96: zero pad channels give zero gradients); dO: op16 [rows][ldo], columns [head][D]; bias f32 [heads][spad][spad];
// dbias_part f32 [nwin][heads][S][S] (summed over windows afterwards, in window order).  fp32 arithmetic on values staged in LDS: this is the
// training path's correctness reference, not a tuned kernel (2 MFLOP per workgroup).
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const T* __restrict__ qkv, int ld, const T* __restrict__ dO, int ldo, const float* __restrict__ bias,
                                                        int spad, float scale, T* __restrict__ dqkv, float* __restrict__ dbias_part, int S, int heads,
                                                        const T* __restrict__ drop = nullptr) {
    // drop (train mode, r05): attn_drop mask op16 [win * heads + h][S][spad] (0 or 1 / keep).  O = (P . M) V:  dP = (dO V^T) . M,  dV = (P . M)^T dO
    constexpr int SM = 64;
    __shared__ float q[SM][D + 1], k[SM][D + 1], v[SM][D + 1], g[SM][D + 1];   // g = dO
    __shared__ float P[SM][SM + 1], dS[SM][SM + 1];
    __shared__ float rsum[SM];
    const int win = blockIdx.x / heads, h = blockIdx.x - win * heads;
    const int tid = threadIdx.x;
    const size_t row0 = (size_t)win * S;
    const int HD = heads * D;
    for (int i = tid; i < S * D; i += 256) {
        const int r = i / D, d = i - r * D;
        const T* pr = qkv + (row0 + r) * ld + h * D + d;
        q[r][d] = (float)pr[0];
        k[r][d] = (float)pr[HD];
        v[r][d] = (float)pr[2 * HD];
        g[r][d] = (float)dO[(row0 + r) * ldo + h * D + d];
    }
    __syncthreads();
    // scores and dP
    for (int e = tid; e < S * S; e += 256) {
        const int i = e / S, j = e - i * S;
        float sc = 0.f, dp = 0.f;
#pragma unroll 8
        for (int d = 0; d < D; ++d) {
            sc += q[i][d] * k[j][d];
            dp += g[i][d] * v[j][d];
        }
        P[i][j] = sc * scale + (bias ? bias[((size_t)h * spad + i) * spad + j] : 0.f);
        dS[i][j] = drop ? dp * (float)drop[((size_t)blockIdx.x * S + i) * spad + j] : dp;
    }
    __syncthreads();
    // softmax per row (thread per row), then rowsum(dP * P)
    if (tid < S) {
        float mx = -3.0e38f;
        for (int j = 0; j < S; ++j) mx = fmaxf(mx, P[tid][j]);
        float sum = 0.f;
        for (int j = 0; j < S; ++j) { const float ev = expf(P[tid][j] - mx); P[tid][j] = ev; sum += ev; }
        const float inv = 1.0f / sum;
        float r = 0.f;
        for (int j = 0; j < S; ++j) { const float pv = P[tid][j] * inv; P[tid][j] = pv; r += pv * dS[tid][j]; }
        rsum[tid] = r;
    }
    __syncthreads();
    for (int e = tid; e < S * S; e += 256) {
        const int i = e / S, j = e - i * S;
        const float ds = P[i][j] * (dS[i][j] - rsum[i]);
        dS[i][j] = ds;
        if (dbias_part) dbias_part[(((size_t)win * heads + h) * S + i) * S + j] = ds;
    }
    __syncthreads();
    for (int e = tid; e < S * D; e += 256) {
        const int r = e / D, d = e - r * D;
        float dq = 0.f, dk = 0.f, dv = 0.f;
        for (int j = 0; j < S; ++j) {
            dq += dS[r][j] * k[j][d];
            dk += dS[j][r] * q[j][d];
            dv += (drop ? P[j][r] * (float)drop[((size_t)blockIdx.x * S + j) * spad + r] : P[j][r]) * g[j][d];
        }
        T* pw = dqkv + (row0 + r) * ld + h * D + d;
        pw[0] = (T)(dq * scale);
        pw[HD] = (T)(dk * scale);
        pw[2 * HD] = (T)dv;
    }
}

// out[i] (+)= sum_b part[b * stride + i], b in block order (fixed order: bit-reproducible)
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int blocks, int stride, float* __restrict__ out, int n, int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int b = 0; b < blocks; ++b) s += part[(size_t)b * stride + i];
    out[i] = accumulate ? out[i] + s : s;
}

}  // namespace
}  // namespace fvit

using namespace fvit;

extern "C" {

int32_t fvit_bwd_blocks(int32_t M) { return (M + BWD_ROWS - 1) / BWD_ROWS; }

int fvit_bwd_transpose16(int32_t dtype, const void* in, int32_t ld_in, void* out, int32_t ld_out, int32_t M, int32_t N, fvit_stream_t stream) {
    if (!in || !out || M <= 0 || N <= 0 || ld_in < N || ld_out < M || (ld_out % 64) != 0) {
        set_error("bwd_transpose16: bad arguments M=%d N=%d ld_in=%d ld_out=%d (ld_out must be a multiple of 64 >= M)", M, N, ld_in, ld_out);
        return FVIT_EINVAL;
    }
    const dim3 grid((N + 63) / 64, ld_out / 64);
    if (dtype == FVIT_F16) hipLaunchKernelGGL((transpose16_kernel<_Float16>), grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)in, ld_in, (_Float16*)out, ld_out, M, N);
    else if (dtype == FVIT_BF16) hipLaunchKernelGGL((transpose16_kernel<__bf16>), grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)in, ld_in, (__bf16*)out, ld_out, M, N);
    else { set_error("bwd_transpose16: dtype %d", dtype); return FVIT_EINVAL; }
    return check_launch("transpose16_kernel");
}

int fvit_bwd_scale_cols(int32_t dtype, const float* dy, const void* z, int32_t ldz, const float* gamma, void* dz, int32_t lddz, float* part,
                        int32_t M, int32_t C, fvit_stream_t stream) {
    if (!dy || !z || !dz || !part || M <= 0 || C <= 0 || ldz < C || lddz < C) { set_error("bwd_scale_cols: bad arguments"); return FVIT_EINVAL; }
    const int blocks = (M + BWD_ROWS - 1) / BWD_ROWS;
    if (dtype == FVIT_F16) hipLaunchKernelGGL((scale_cols_kernel<_Float16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, (const _Float16*)z, ldz, gamma, (_Float16*)dz, lddz, part, M, C);
    else if (dtype == FVIT_BF16) hipLaunchKernelGGL((scale_cols_kernel<__bf16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, (const __bf16*)z, ldz, gamma, (__bf16*)dz, lddz, part, M, C);
    else { set_error("bwd_scale_cols: dtype %d", dtype); return FVIT_EINVAL; }
    return check_launch("scale_cols_kernel");
}

int fvit_bwd_gelu(int32_t dtype, const void* a, int32_t lda, const void* dh, int32_t lddh, void* out, int32_t ldo, float* part, int32_t M, int32_t H,
                  fvit_stream_t stream) {
    if (!a || !out || M <= 0 || H <= 0 || lda < H || ldo < H || (dh && (!part || lddh < H))) { set_error("bwd_gelu: bad arguments"); return FVIT_EINVAL; }
    const int blocks = (M + BWD_ROWS - 1) / BWD_ROWS;
#define FVIT_GELU(T_) do { \
        if (dh) hipLaunchKernelGGL((gelu_kernel<T_, 1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const T_*)a, lda, (const T_*)dh, lddh, (T_*)out, ldo, part, M, H); \
        else hipLaunchKernelGGL((gelu_kernel<T_, 0>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const T_*)a, lda, (const T_*)nullptr, 0, (T_*)out, ldo, part, M, H); } while (0)
    if (dtype == FVIT_F16) FVIT_GELU(_Float16);
    else if (dtype == FVIT_BF16) FVIT_GELU(__bf16);
    else { set_error("bwd_gelu: dtype %d", dtype); return FVIT_EINVAL; }
#undef FVIT_GELU
    return check_launch("gelu_kernel");
}

int fvit_bwd_layernorm(const float* x, const float* dxn, const float* dy, const float* ln_w, float eps, float* dx, float* stats, float* part,
                       int32_t M, int32_t C, fvit_stream_t stream) {
    if (!x || !dxn || !ln_w || !dx || !stats || !part || M <= 0 || C <= 0) { set_error("bwd_layernorm: bad arguments"); return FVIT_EINVAL; }
    hipLaunchKernelGGL(layernorm_bwd_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, dxn, dy, ln_w, eps, dx, stats, M, C);
    hipLaunchKernelGGL(layernorm_bwd_cols_kernel, dim3((M + BWD_ROWS - 1) / BWD_ROWS), dim3(256), 0, (hipStream_t)stream, x, dxn, (const float*)stats, part, M, C);
    return check_launch("layernorm_bwd_kernel");
}

int fvit_bwd_colsum16(int32_t dtype, const void* in, int32_t ld, float* part, int32_t M, int32_t N, fvit_stream_t stream) {
    if (!in || !part || M <= 0 || N <= 0 || ld < N) { set_error("bwd_colsum16: bad arguments"); return FVIT_EINVAL; }
    const int blocks = (M + BWD_ROWS - 1) / BWD_ROWS;
    if (dtype == FVIT_F16) hipLaunchKernelGGL((colsum16_kernel<_Float16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)in, ld, part, M, N);
    else if (dtype == FVIT_BF16) hipLaunchKernelGGL((colsum16_kernel<__bf16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16*)in, ld, part, M, N);
    else { set_error("bwd_colsum16: dtype %d", dtype); return FVIT_EINVAL; }
    return check_launch("colsum16_kernel");
}

int fvit_bwd_window_attention_drop(int32_t dtype, const void* qkv, int32_t ld, const void* dO, int32_t ldo, const float* bias, int32_t spad, float scale,
                                   void* dqkv, float* dbias_part, int32_t nwin, int32_t S, int32_t heads, int32_t D, const void* drop_mask,
                                   fvit_stream_t stream);

int fvit_bwd_window_attention(int32_t dtype, const void* qkv, int32_t ld, const void* dO, int32_t ldo, const float* bias, int32_t spad, float scale,
                              void* dqkv, float* dbias_part, int32_t nwin, int32_t S, int32_t heads, int32_t D, fvit_stream_t stream) {
    return fvit_bwd_window_attention_drop(dtype, qkv, ld, dO, ldo, bias, spad, scale, dqkv, dbias_part, nwin, S, heads, D, nullptr, stream);
}

int fvit_bwd_window_attention_drop(int32_t dtype, const void* qkv, int32_t ld, const void* dO, int32_t ldo, const float* bias, int32_t spad, float scale,
                                   void* dqkv, float* dbias_part, int32_t nwin, int32_t S, int32_t heads, int32_t D, const void* drop_mask,
                                   fvit_stream_t stream) {
    if (drop_mask && spad < S) { set_error("bwd_window_attention: the attn_drop mask is [nwin * heads][S][spad], spad >= S"); return FVIT_EINVAL; }
    if (!qkv || !dO || !dqkv || nwin <= 0 || S < 1 || S > 64 || heads <= 0 || (D != 32 && D != 64 && D != 96) || ld < 3 * heads * D || ldo < heads * D || (bias && spad < S)) {
        set_error("bwd_window_attention: unsupported arguments nwin=%d S=%d heads=%d D=%d (need S <= 64, padded head_dim 32 / 64 / 96)", nwin, S, heads, D);
        return FVIT_EINVAL;
    }
    const dim3 grid(nwin * heads);
#define FVIT_ATTN_BWD(T_, D_) hipLaunchKernelGGL((attn_bwd_kernel<T_, D_>), grid, dim3(256), 0, (hipStream_t)stream, (const T_*)qkv, ld, (const T_*)dO, ldo, bias, spad, scale, (T_*)dqkv, dbias_part, S, heads, (const T_*)drop_mask)
#define FVIT_ATTN_BWD_D(T_) do { if (D == 32) FVIT_ATTN_BWD(T_, 32); else if (D == 64) FVIT_ATTN_BWD(T_, 64); else FVIT_ATTN_BWD(T_, 96); } while (0)
    if (dtype == FVIT_F16) FVIT_ATTN_BWD_D(_Float16);
    else if (dtype == FVIT_BF16) FVIT_ATTN_BWD_D(__bf16);
    else { set_error("bwd_window_attention: dtype %d", dtype); return FVIT_EINVAL; }
#undef FVIT_ATTN_BWD_D
#undef FVIT_ATTN_BWD
    return check_launch("attn_bwd_kernel");
}

int fvit_bwd_colsum_finish(const float* part, int32_t blocks, int32_t stride, float* out, int32_t n, int32_t accumulate, fvit_stream_t stream) {
    if (!part || !out || blocks <= 0 || n <= 0 || stride < n) { set_error("bwd_colsum_finish: bad arguments"); return FVIT_EINVAL; }
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, part, blocks, stride, out, n, accumulate);
    return check_launch("colsum_finish_kernel");
}

}  // extern "C"
