// fvit_head.hip -- head-only training step of the classifier (gfx950), fp32 throughout.
//
// north_star's training clause: data-parallel image batches, the HAT backbone frozen (forward kernels only), ONE all-reduce on the
// classifier gradient + loss (reference: DDP over every parameter, train.py:542-551; loss reduce_tensor, train.py:910, 991-993).
// The classifier is FasterViT.head = nn.Linear(num_features, num_classes) (FV:927) on the pooled, normalised features (FV:953-960).
//
//   logits = feat . W^T + b                                   head_logits_kernel   (forward of FV:959)
//   loss_b = (1-eps) * nll_b + eps * mean_n(-log p_bn)          head_xent_loss_kernel  (timm LabelSmoothingCrossEntropy, train.py:653-659)
//   dlogits = (softmax(logits) - q) / global_batch              head_xent_grad_kernel  (in place)
//   dW = dlogits^T . feat,  db = sum_b dlogits,  loss = sum_b loss_b / global_batch      head_grad_kernel
//   m = mu * m + (g + wd * p);  p -= lr * m                     sgd_kernel
//
// The two contractions run on the exact-fp32 matrix instruction v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain): one wave64 per
// 16x16 output tile, operands read straight from L2-resident rows as 16-byte (logits) / 4-byte (grad) loads -- the tensors are tiny
// (256 x 512 features, 1000 x 512 weights) and the step is launch-latency-, not throughput-bound.  No atomics: every output has ONE
// writer and a fixed summation order, so the gradient is bit-reproducible (the gloo / RCCL all-reduce then decides the cross-rank order).
#include "fvit_common.h"

namespace fvit {

namespace {

__device__ __forceinline__ f4 mfma_f32(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// logits[b][n] = sum_k feat[b][k] * W[n][k] + bias[n];  tile = 16 rows (b) x 16 classes (n) per wave
__global__ __launch_bounds__(256) void head_logits_kernel(const float* __restrict__ feat, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ logits, int B, int N, int F,
                                                          int tiles_n, int tiles) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= tiles) return;
    const int tb = tile / tiles_n, tn = tile - tb * tiles_n;
    const int g = lane >> 4, s = lane & 15;
    const int rb = min(tb * 16 + s, B - 1), rn = min(tn * 16 + s, N - 1);   // clamped rows: tail tiles recompute the last row, never stored
    const float* pa = feat + (size_t)rb * F + 4 * g;
    const float* pw = W + (size_t)rn * F + 4 * g;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    // k slot g of MFMA step t <-> feature k0 + 4g + t for BOTH operands (any k permutation applied to both is the same sum)
    for (int k0 = 0; k0 < F; k0 += 16) {
        const f4 a = *(const f4*)(pa + k0);
        const f4 w = *(const f4*)(pw + k0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = mfma_f32(a[t], w[t], acc);
    }
    const int n = tn * 16 + s;
    if (n < N) {
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = tb * 16 + g * 4 + r;   // C/D: row = 4 * (lane >> 4) + r, col = lane & 15
            if (b < B) logits[(size_t)b * N + n] = acc[r] + bv;
        }
    }
}

__device__ __forceinline__ float wave_max(float v) {
    v = group_max<64>(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = group_sum<64>(v);
    return v;
}

// Label-smoothed cross entropy in two passes: (1) one wave per row -> loss_b and the row's (max, 1 / sum exp), from the ORIGINAL logits;
// (2) elementwise, in place: logits <- (softmax - q) / global_batch.
__global__ __launch_bounds__(256) void head_xent_loss_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                             float* __restrict__ loss_rows, float* __restrict__ row_stats, int B, int N,
                                                             float smoothing) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* x = logits + (size_t)b * N;
    float m = -3.0e38f, sx = 0.f;
    for (int n = lane; n < N; n += 64) {
        const float v = x[n];
        m = fmaxf(m, v);
        sx += v;
    }
    m = wave_max(m);
    sx = wave_sum(sx);
    float se = 0.f;
    for (int n = lane; n < N; n += 64) se += __expf(x[n] - m);
    se = wave_sum(se);
    if (lane == 0) {
        const float lse = m + __logf(se);
        const int64_t t = target[b];
        // a label outside [0, N) (torch's cross_entropy asserts on it): no out-of-bounds read, and the row's loss -- hence the
        // step's loss and the whole [dW | db | loss] buffer's last entry -- becomes NaN instead of a silently wrong number
        const bool t_ok = t >= 0 && t < (int64_t)N;
        const float nll = t_ok ? lse - x[t] : __builtin_nanf("");   // -log p_t
        const float smooth = lse - sx / (float)N;     // mean_n(-log p_n)
        loss_rows[b] = (1.0f - smoothing) * nll + smoothing * smooth;
        row_stats[2 * b] = m;
        row_stats[2 * b + 1] = 1.0f / se;
    }
}

__global__ __launch_bounds__(256) void head_xent_grad_kernel(float* __restrict__ logits, const int64_t* __restrict__ target,
                                                             const float* __restrict__ row_stats, int B, int N, float smoothing,
                                                             float inv_batch) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * N) return;
    const int b = (int)(i / N), n = (int)(i - (size_t)b * N);
    const float p = __expf(logits[i] - row_stats[2 * b]) * row_stats[2 * b + 1];
    const float q = smoothing / (float)N + (n == target[b] ? 1.0f - smoothing : 0.f);
    logits[i] = (p - q) * inv_batch;
}

// grad = [dW (N x F) | db (N) | loss (1)];  tile = 16 classes x 16 features per wave, contraction over the B samples
__global__ __launch_bounds__(256) void head_grad_kernel(const float* __restrict__ dl, const float* __restrict__ feat,
                                                        const float* __restrict__ loss_rows, float* __restrict__ grad, int B, int N, int F,
                                                        int tiles_f, int tiles, float inv_batch) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= tiles) return;
    const int tn = tile / tiles_f, tf = tile - tn * tiles_f;
    const int g = lane >> 4, s = lane & 15;
    const int n = min(tn * 16 + s, N - 1);      // class of the A-operand row this lane feeds (clamped; tail rows are not stored)
    const int f = tf * 16 + s;                  // feature column of the B operand (F % 16 == 0)
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    float dsum = 0.f;
    for (int b0 = 0; b0 < B; b0 += 16) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int b = b0 + 4 * g + t;       // k slot g of step t <-> sample b0 + 4g + t, for both operands
            const float a = b < B ? dl[(size_t)b * N + n] : 0.f;
            const float v = b < B ? feat[(size_t)b * F + f] : 0.f;
            acc = mfma_f32(a, v, acc);
            dsum += a;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int nn = tn * 16 + g * 4 + r;
        if (nn < N) grad[(size_t)nn * F + tf * 16 + s] = acc[r];
    }
    if (tf == 0) {   // db: column sums of dlogits; lane (g, s) holds the samples {b0 + 4g + t} of class tn*16 + s
        dsum = sum_xor32(sum_xor16(dsum));
        if (g == 0 && tn * 16 + s < N) grad[(size_t)N * F + tn * 16 + s] = dsum;
        if (tn == 0) {   // loss: fixed-order sum of the per-row losses
            float ls = 0.f;
            for (int b = lane; b < B; b += 64) ls += loss_rows[b];
            ls = wave_sum(ls);
            if (lane == 0) grad[(size_t)N * F + N] = ls * inv_batch;
        }
    }
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, float* __restrict__ m, const float* __restrict__ g, int64_t n,
                                                  float lr, float mu, float wd) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float mv = mu * m[i] + (g[i] + wd * p[i]);
    m[i] = mv;
    p[i] -= lr * mv;
}

}  // namespace

}  // namespace fvit

using namespace fvit;

extern "C" {

int fvit_head_logits(const float* feat, const float* W, const float* bias, float* logits, int32_t B, int32_t N, int32_t F,
                     fvit_stream_t stream) {
    if (!feat || !W || !logits || B <= 0 || N <= 0 || F <= 0 || (F % 16) != 0) {
        set_error("head_logits: bad arguments B=%d N=%d F=%d (F must be a multiple of 16)", B, N, F);
        return FVIT_EINVAL;
    }
    const int tiles_n = (N + 15) / 16, tiles = ((B + 15) / 16) * tiles_n;
    ProfScope prof(FVIT_K_OTHER, 2.0 * B * (double)N * F, 4.0 * ((double)B * F + (double)N * F + (double)B * N), (hipStream_t)stream);
    hipLaunchKernelGGL(head_logits_kernel, dim3((tiles + 3) / 4), dim3(256), 0, (hipStream_t)stream, feat, W, bias, logits, B, N, F, tiles_n,
                       tiles);
    return check_launch("head_logits_kernel");
}

int fvit_head_softmax_xent(float* logits_inout, const int64_t* target, float* loss_rows, float* row_stats, int32_t B, int32_t N,
                           float smoothing, float inv_global_batch, fvit_stream_t stream) {
    if (!logits_inout || !target || !loss_rows || !row_stats || B <= 0 || N <= 0 || smoothing < 0.f || smoothing >= 1.f) {
        set_error("head_softmax_xent: bad arguments B=%d N=%d smoothing=%f", B, N, smoothing);
        return FVIT_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(FVIT_K_OTHER, 0.0, 12.0 * B * (double)N, st);
    hipLaunchKernelGGL(head_xent_loss_kernel, dim3((B + 3) / 4), dim3(256), 0, st, logits_inout, target, loss_rows, row_stats, B, N, smoothing);
    const size_t total = (size_t)B * N;
    hipLaunchKernelGGL(head_xent_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, logits_inout, target, row_stats, B, N,
                       smoothing, inv_global_batch);
    return check_launch("head_xent kernels");
}

int fvit_head_grad(const float* dlogits, const float* feat, const float* loss_rows, float* grad_flat, int32_t B, int32_t N, int32_t F,
                   float inv_global_batch, fvit_stream_t stream) {
    if (!dlogits || !feat || !loss_rows || !grad_flat || B <= 0 || N <= 0 || F <= 0 || (F % 16) != 0) {
        set_error("head_grad: bad arguments B=%d N=%d F=%d (F must be a multiple of 16)", B, N, F);
        return FVIT_EINVAL;
    }
    const int tiles_f = F / 16, tiles = ((N + 15) / 16) * tiles_f;
    ProfScope prof(FVIT_K_OTHER, 2.0 * B * (double)N * F, 4.0 * ((double)B * F + (double)N * F + (double)B * N), (hipStream_t)stream);
    hipLaunchKernelGGL(head_grad_kernel, dim3((tiles + 3) / 4), dim3(256), 0, (hipStream_t)stream, dlogits, feat, loss_rows, grad_flat, B, N, F,
                       tiles_f, tiles, inv_global_batch);
    return check_launch("head_grad_kernel");
}

int fvit_sgd_momentum(float* param, float* momentum, const float* grad, int64_t n, float lr, float mu, float weight_decay,
                      fvit_stream_t stream) {
    if (!param || !momentum || !grad || n <= 0) {
        set_error("sgd_momentum: bad arguments n=%lld", (long long)n);
        return FVIT_EINVAL;
    }
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, momentum, grad, n, lr, mu,
                       weight_decay);
    return check_launch("sgd_kernel");
}

}  // extern "C"
