// fvit_glue.hip -- HBM-bound glue kernels for the conv side in deploy mode (gfx950), channels-last
// 16-bit feature maps (C contiguous).  They replace the separate passes PyTorch-ROCm issues around the
// MIOpen convolutions of PatchEmbed / ConvBlock / Downsample once BatchNorm is folded into the conv
// weights:
//   bias_act      : y = act(x + bias[c])                in place  (conv bias + ReLU FV:460-464 / GELU FV:507)
//   bias_residual : x = x + y + bias[c]                 in place  (conv bias + residual FV:512)
//   layernorm2d   : per-pixel LayerNorm over C, eps 1e-6           (timm LayerNorm2d in Downsample FV:432,438)
// Each moves every byte once; all are priced against the HBM roof.
#include "fvit_common.h"

namespace fvit {

namespace {


// VEC elements of T per thread (VEC * sizeof(T) = 16 or 8 bytes); C % VEC == 0 so a vector never straddles pixels
template <typename T, int VEC, int MODE>  // MODE 0: bias+act (act in p.act), 1: bias+residual
__global__ __launch_bounds__(256) void bias_kernel(T* __restrict__ x, const T* __restrict__ y, const float* __restrict__ bias,
                                                   int64_t nvec, int C, int act) {
    typedef T vt __attribute__((ext_vector_type(VEC)));
    constexpr int U = 4;  // vectors per thread per iteration: 4 x 16 B loads in flight before the first use
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < nvec; i0 += stride * U) {
        vt v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < nvec) {
                v[u] = *((const vt*)x + i);
                if (MODE == 1) w[u] = *((const vt*)y + i);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < nvec) {
                const int c = (int)((i * VEC) % C);
                vt o;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    float f = (float)v[u][j] + bias[c + j];
                    if (MODE == 1) f += (float)w[u][j];
                    else if (act == 1) f = fmaxf(f, 0.f);
                    else if (act == 2) f = gelu_fast(f);
                    o[j] = (T)f;
                }
                *((vt*)x + i) = o;
            }
        }
    }
}

// LPP lanes cooperate on one pixel (64 / LPP pixels per wave); the pixel's C channels live in registers,
// 8 per lane and per step (MAXV steps).  LPP = 8 for C = 64 so that narrow maps still use every lane.
template <typename T, int LPP, int MAXV>
__global__ __launch_bounds__(256) void ln2d_kernel(const T* __restrict__ in, T* __restrict__ out, const float* __restrict__ w,
                                                   const float* __restrict__ b, float eps, int64_t npix, int C, int Cv) {
    typedef T v8 __attribute__((ext_vector_type(8)));
    constexpr int PPW = 64 / LPP;  // pixels per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPP;
    const int64_t pix = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * PPW + lane / LPP;
    const bool ok = pix < npix;
    const int C8 = C >> 3;
    const T* src = in + (ok ? pix : 0) * C;
    float v[MAXV][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c8 = sub + i * LPP;
        if (c8 < C8) {
            const v8 t = *((const v8*)src + c8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[i][j] = (float)t[j]; sum += v[i][j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
#pragma unroll
    for (int o = LPP / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    // Cv <= C real channels; the C - Cv trailing pad channels hold zeros (channel-padded deploy maps): they add nothing to
    // the sum and (0 - mean)^2 each to the squared deviations, which is taken out again below
    const float mean = sum / (float)Cv;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (sub + i * LPP < C8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; sq += d * d; }
        }
    }
#pragma unroll
    for (int o = LPP / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    sq -= (float)(C - Cv) * mean * mean;
    const float rstd = rsqrtf(fmaxf(sq, 0.f) / (float)Cv + eps);
    if (!ok) return;
    T* dst = out + pix * C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c8 = sub + i * LPP;
        if (c8 < C8) {
            v8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (T)((v[i][j] - mean) * rstd * w[c8 * 8 + j] + b[c8 * 8 + j]);
            *((v8*)dst + c8) = o;
        }
    }
}

template <typename T>
int bias_launch(T* x, const T* y, const float* bias, int64_t n, int C, int act, int mode, hipStream_t stream) {
    const int vec = (C % 8 == 0) ? 8 : 4;
    const int64_t nvec = n / vec;
    const int64_t want = (nvec + 256 * 4 - 1) / (256 * 4);
    const int grid = (int)(want < 8192 ? (want > 0 ? want : 1) : 8192);
    if (vec == 8) {
        if (mode) hipLaunchKernelGGL((bias_kernel<T, 8, 1>), dim3(grid), dim3(256), 0, stream, x, y, bias, nvec, C, act);
        else hipLaunchKernelGGL((bias_kernel<T, 8, 0>), dim3(grid), dim3(256), 0, stream, x, y, bias, nvec, C, act);
    } else {
        if (mode) hipLaunchKernelGGL((bias_kernel<T, 4, 1>), dim3(grid), dim3(256), 0, stream, x, y, bias, nvec, C, act);
        else hipLaunchKernelGGL((bias_kernel<T, 4, 0>), dim3(grid), dim3(256), 0, stream, x, y, bias, nvec, C, act);
    }
    return check_launch(mode ? "bias_residual" : "bias_act");
}

template <typename T>
int ln2d_launch(const T* in, T* out, const float* w, const float* b, float eps, int64_t npix, int C, int Cv, hipStream_t stream) {
    const int c8 = C / 8;
#define FVIT_LN2D(LPP, MAXV)                                                                                          \
    hipLaunchKernelGGL((ln2d_kernel<T, LPP, MAXV>), dim3((unsigned)((npix + 4 * (64 / LPP) - 1) / (4 * (64 / LPP)))), \
                       dim3(256), 0, stream, in, out, w, b, eps, npix, C, Cv)
    if (c8 <= 8) FVIT_LN2D(8, 1);
    else if (c8 <= 16) FVIT_LN2D(16, 1);
    else if (c8 <= 32) FVIT_LN2D(32, 1);
    else if (c8 <= 64) FVIT_LN2D(64, 1);
    else if (c8 <= 128) FVIT_LN2D(64, 2);
    else if (c8 <= 256) FVIT_LN2D(64, 4);
    else {
        set_error("layernorm2d: C=%d too wide (max 2048)", C);
        return FVIT_EINVAL;
    }
#undef FVIT_LN2D
    return check_launch("layernorm2d");
}

}  // namespace
}  // namespace fvit

using namespace fvit;

extern "C" {

int fvit_bias_act_cl(int32_t dtype, void* x, const float* bias, int64_t n_pixels, int32_t C, int32_t act, fvit_stream_t stream) {
    if (!x || !bias || n_pixels <= 0 || C <= 0 || (C % 4) || act < 0 || act > 2) {
        set_error("bias_act: bad arguments (C=%d must be a multiple of 4, act in 0..2)", C);
        return FVIT_EINVAL;
    }
    const int64_t n = n_pixels * C;
    ProfScope prof(FVIT_K_OTHER, 0.0, 4.0 * n, (hipStream_t)stream);
    if (dtype == FVIT_F16) return bias_launch<_Float16>((_Float16*)x, nullptr, bias, n, C, act, 0, (hipStream_t)stream);
    if (dtype == FVIT_BF16) return bias_launch<__bf16>((__bf16*)x, nullptr, bias, n, C, act, 0, (hipStream_t)stream);
    set_error("bias_act: dtype %d not supported (16-bit maps only)", dtype);
    return FVIT_EINVAL;
}

int fvit_bias_residual_cl(int32_t dtype, void* x, const void* y, const float* bias, int64_t n_pixels, int32_t C,
                          fvit_stream_t stream) {
    if (!x || !y || !bias || n_pixels <= 0 || C <= 0 || (C % 4)) {
        set_error("bias_residual: bad arguments (C=%d must be a multiple of 4)", C);
        return FVIT_EINVAL;
    }
    const int64_t n = n_pixels * C;
    ProfScope prof(FVIT_K_OTHER, 0.0, 6.0 * n, (hipStream_t)stream);
    if (dtype == FVIT_F16) return bias_launch<_Float16>((_Float16*)x, (const _Float16*)y, bias, n, C, 0, 1, (hipStream_t)stream);
    if (dtype == FVIT_BF16) return bias_launch<__bf16>((__bf16*)x, (const __bf16*)y, bias, n, C, 0, 1, (hipStream_t)stream);
    set_error("bias_residual: dtype %d not supported (16-bit maps only)", dtype);
    return FVIT_EINVAL;
}

int fvit_layernorm2d_cl(int32_t dtype, const void* in, void* out, const float* weight, const float* bias, float eps,
                        int64_t n_pixels, int32_t C, int32_t C_valid, fvit_stream_t stream) {
    if (C_valid <= 0) C_valid = C;
    if (!in || !out || !weight || !bias || n_pixels <= 0 || C <= 0 || (C % 8) || C_valid > C) {
        set_error("layernorm2d: bad arguments (C=%d must be a multiple of 8, C_valid=%d <= C)", C, C_valid);
        return FVIT_EINVAL;
    }
    ProfScope prof(FVIT_K_OTHER, 0.0, 4.0 * n_pixels * C, (hipStream_t)stream);
    if (dtype == FVIT_F16) return ln2d_launch<_Float16>((const _Float16*)in, (_Float16*)out, weight, bias, eps, n_pixels, C, C_valid, (hipStream_t)stream);
    if (dtype == FVIT_BF16) return ln2d_launch<__bf16>((const __bf16*)in, (__bf16*)out, weight, bias, eps, n_pixels, C, C_valid, (hipStream_t)stream);
    set_error("layernorm2d: dtype %d not supported (16-bit maps only)", dtype);
    return FVIT_EINVAL;
}

}  // extern "C"
