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
// PX (fvit_layernorm2d_px): the input is a two-term map (in + in_lo, 16-bit planes) or ONE fp32 map (in_f32), the output two planes out / out_lo
template <typename T, int LPP, int MAXV, bool PX = false>
__global__ __launch_bounds__(256) void ln2d_kernel(const T* __restrict__ in, T* __restrict__ out, const float* __restrict__ w,
                                                   const float* __restrict__ b, float eps, int64_t npix, int C, int Cv,
                                                   const T* __restrict__ in_lo = nullptr, const float* __restrict__ in_f32 = nullptr,
                                                   T* __restrict__ out_lo = nullptr) {
    typedef T v8 __attribute__((ext_vector_type(8)));
    constexpr int PPW = 64 / LPP;  // pixels per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPP;
    const int64_t pix = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * PPW + lane / LPP;
    const bool ok = pix < npix;
    const int C8 = C >> 3;
    const T* src = (PX && in_f32) ? nullptr : in + (ok ? pix : 0) * C;
    float v[MAXV][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c8 = sub + i * LPP;
        if (c8 < C8) {
            if (PX && in_f32) {
                const float* sf = in_f32 + (ok ? pix : 0) * C + c8 * 8;
                const f4 t0 = *(const f4*)sf, t1 = *(const f4*)(sf + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[i][j] = t0[j]; v[i][4 + j] = t1[j]; }
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += v[i][j];
            } else {
                const v8 t = *((const v8*)src + c8);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = (float)t[j];
                if (PX && in_lo) {
                    const v8 tl = *((const v8*)(in_lo + (ok ? pix : 0) * C) + c8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[i][j] += (float)tl[j];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += v[i][j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
    sum = group_sum<LPP>(sum);
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
    sq = group_sum<LPP>(sq);
    sq -= (float)(C - Cv) * mean * mean;
    const float rstd = rsqrtf(fmaxf(sq, 0.f) / (float)Cv + eps);
    if (!ok) return;
    T* dst = out + pix * C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c8 = sub + i * LPP;
        if (c8 < C8) {
            v8 o, ol;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float y = (v[i][j] - mean) * rstd * w[c8 * 8 + j] + b[c8 * 8 + j];
                o[j] = (T)y;
                if (PX) ol[j] = (T)(y - (float)o[j]);
            }
            *((v8*)dst + c8) = o;
            if (PX && out_lo) *((v8*)(out_lo + pix * C) + c8) = ol;
        }
    }
}

// AdaptiveAvgPool2d(1) of a channels-last map (FV:926, 955): feat[b][c] = mean over the HW pixels of in[b][.][c], fp32 sums in a fixed order
// (lane-serial over pixels, then a fixed tree over the 4 waves of the workgroup): bitwise repeatable.  One workgroup per (image, 64-channel group);
// wave w takes pixels w, w + 4, ...; lane = channel.
template <typename TIN>
__global__ __launch_bounds__(256) void avgpool_kernel(const TIN* __restrict__ in, float* __restrict__ out, int HW, int C) {
    __shared__ float part[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C) {
        const TIN* src = in + (size_t)b * HW * C + c;
        for (int p = w; p < HW; p += 4) s += (float)src[(size_t)p * C];
    }
    part[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < C) out[(size_t)b * C + c] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x])) / (float)HW;
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
int ln2d_launch(const T* in, T* out, const float* w, const float* b, float eps, int64_t npix, int C, int Cv, hipStream_t stream, bool px = false,
                const T* in_lo = nullptr, const float* in_f32 = nullptr, T* out_lo = nullptr) {
    const int c8 = C / 8;
#define FVIT_LN2D(LPP, MAXV)                                                                                                       \
    if (px) hipLaunchKernelGGL((ln2d_kernel<T, LPP, MAXV, true>), dim3((unsigned)((npix + 4 * (64 / LPP) - 1) / (4 * (64 / LPP)))), \
                               dim3(256), 0, stream, in, out, w, b, eps, npix, C, Cv, in_lo, in_f32, out_lo);                        \
    else hipLaunchKernelGGL((ln2d_kernel<T, LPP, MAXV>), dim3((unsigned)((npix + 4 * (64 / LPP) - 1) / (4 * (64 / LPP)))),           \
                            dim3(256), 0, stream, in, out, w, b, eps, npix, C, Cv, nullptr, nullptr, nullptr)
    if (c8 <= 8) { FVIT_LN2D(8, 1); }
    else if (c8 <= 16) { FVIT_LN2D(16, 1); }
    else if (c8 <= 32) { FVIT_LN2D(32, 1); }
    else if (c8 <= 64) { FVIT_LN2D(64, 1); }
    else if (c8 <= 128) { FVIT_LN2D(64, 2); }
    else if (c8 <= 256) { FVIT_LN2D(64, 4); }
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

int fvit_global_avgpool_cl(int32_t dtype, const void* in, float* out, int32_t B, int32_t HW, int32_t C, fvit_stream_t stream) {
    if (!in || !out || B <= 0 || HW <= 0 || C <= 0) { set_error("global_avgpool: bad arguments"); return FVIT_EINVAL; }
    ProfScope prof(FVIT_K_OTHER, 0.0, (double)B * HW * C * (dtype == FVIT_F32 ? 4.0 : 2.0), (hipStream_t)stream);
    const dim3 grid((C + 63) / 64, B);
    if (dtype == FVIT_F32) hipLaunchKernelGGL((avgpool_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)in, out, HW, C);
    else if (dtype == FVIT_F16) hipLaunchKernelGGL((avgpool_kernel<_Float16>), grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)in, out, HW, C);
    else if (dtype == FVIT_BF16) hipLaunchKernelGGL((avgpool_kernel<__bf16>), grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)in, out, HW, C);
    else { set_error("global_avgpool: dtype %d", dtype); return FVIT_EINVAL; }
    return check_launch("avgpool_kernel");
}

int fvit_layernorm2d_px(int32_t dtype, const void* in, const void* in_lo, const float* in_f32, void* out, void* out_lo, const float* weight,
                        const float* bias, float eps, int64_t n_pixels, int32_t C, int32_t C_valid, fvit_stream_t stream) {
    if (C_valid <= 0) C_valid = C;
    if ((!in && !in_f32) || (in && in_f32) || (in_lo && !in) || !out || !weight || !bias || n_pixels <= 0 || C <= 0 || (C % 8) || C_valid > C) {
        set_error("layernorm2d_px: bad arguments (one of in / in_f32, C=%d a multiple of 8, C_valid=%d <= C)", C, C_valid);
        return FVIT_EINVAL;
    }
    ProfScope prof(FVIT_K_OTHER, 0.0, (double)n_pixels * C * ((in_f32 ? 4.0 : (in_lo ? 4.0 : 2.0)) + (out_lo ? 4.0 : 2.0)), (hipStream_t)stream);
    if (dtype == FVIT_F16) return ln2d_launch<_Float16>((const _Float16*)in, (_Float16*)out, weight, bias, eps, n_pixels, C, C_valid, (hipStream_t)stream, true,
                                                        (const _Float16*)in_lo, in_f32, (_Float16*)out_lo);
    if (dtype == FVIT_BF16) return ln2d_launch<__bf16>((const __bf16*)in, (__bf16*)out, weight, bias, eps, n_pixels, C, C_valid, (hipStream_t)stream, true,
                                                       (const __bf16*)in_lo, in_f32, (__bf16*)out_lo);
    set_error("layernorm2d_px: dtype %d not supported (16-bit planes only)", dtype);
    return FVIT_EINVAL;
}

#ifdef FVIT_DIAG
/* Test / diagnosis aid: fills the LDS and the registers of every CU it lands on with a NaN pattern and exits.  Launched on a side
 * stream beside a forward, it turns any read of uninitialised LDS (whose content is otherwise whatever the previous workgroup left --
 * repeatable on one stream, timing-dependent with concurrent streams) into NaNs / large errors. */
__global__ __launch_bounds__(256) void lds_poison_kernel(unsigned* sink, int spin) {
    extern __shared__ __attribute__((aligned(16))) unsigned pl[];
    for (int i = threadIdx.x; i < 16 * 1024; i += 256) pl[i] = 0x7fc07fc0u;   // 64 KiB: f32 NaN = two f16 NaNs
    __syncthreads();
    unsigned acc = 0;
    for (int k = 0; k < spin; ++k) acc += pl[(threadIdx.x * 17 + k * 31) & (16 * 1024 - 1)];
    if (acc == 1u) sink[0] = acc;
}

/* Test / diagnosis aid: one wave per SIMD that writes a NaN pattern into ALL 512 of its vector registers (256 VGPRs + 256 AGPRs) and
 * into its share of the LDS, spins, and exits.  Registers and LDS are not cleared between waves, so a kernel that reads a register or an
 * LDS byte it never wrote returns whatever the previous occupant left: the same value run after run on one stream (which is how such a
 * read passes parity and repeatability tests) and something else when another stream's kernels share the CU.  Run before a kernel,
 * this makes that read visible on ONE stream. */
__global__ __launch_bounds__(64) void regs_poison_kernel(unsigned* sink, int spin, unsigned pattern) {
    extern __shared__ __attribute__((aligned(16))) unsigned pl[];
    for (int i = threadIdx.x; i < 10 * 1024; i += 64) pl[i] = pattern;   // 40 KiB per wave, 4 waves per CU
    __syncthreads();
    unsigned acc = 0;
    for (int k = 0; k < spin; ++k) acc += pl[(threadIdx.x * 17 + k * 31) % (10 * 1024)];
    asm volatile(
        ".set fvit_i, 1\n"
        ".rept 255\n"
        "v_mov_b32 v[fvit_i], %0\n"
        ".set fvit_i, fvit_i + 1\n"
        ".endr\n"
        ".set fvit_i, 0\n"
        ".rept 256\n"
        "v_accvgpr_write_b32 a[fvit_i], %0\n"
        ".set fvit_i, fvit_i + 1\n"
        ".endr\n"
        :
        : "s"(pattern)
        : "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255",
          "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
    if (acc == 1u) sink[0] = acc;
}
int fvit_debug_regs_poison(void* sink, int32_t blocks, int32_t spin, uint32_t pattern, fvit_stream_t stream) {
    if (!sink || blocks <= 0) { set_error("debug_regs_poison: bad arguments"); return FVIT_EINVAL; }
    hipLaunchKernelGGL(regs_poison_kernel, dim3(blocks), dim3(64), 40 * 1024, (hipStream_t)stream, (unsigned*)sink, spin, pattern);
    return check_launch("regs_poison_kernel");
}

int fvit_debug_lds_poison(void* sink, int32_t blocks, int32_t spin, fvit_stream_t stream) {
    if (!sink || blocks <= 0) { set_error("debug_lds_poison: bad arguments"); return FVIT_EINVAL; }
    hipLaunchKernelGGL(lds_poison_kernel, dim3(blocks), dim3(256), 64 * 1024, (hipStream_t)stream, (unsigned*)sink, spin);
    return check_launch("lds_poison_kernel");
}

/* Diagnosis aid: one 32-bit hash per row of a device buffer, appended to a caller-provided trace buffer after selected launches of the
 * stage driver (fvit_debug_rowhash_begin / _end).  Comparing the traces of two identical calls names the first kernel and the rows
 * whose output differs. */
__global__ __launch_bounds__(256) void rowhash_kernel(const unsigned* __restrict__ x, long long rows, int words, unsigned* __restrict__ out) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const unsigned* r = x + row * words;
    unsigned h = 0x811C9DC5u + (unsigned)lane;
    for (int i = lane; i < words; i += 64) h = (h ^ r[i]) * 0x01000193u;
    h *= (unsigned)(2 * lane + 1);
    for (int o = 32; o; o >>= 1) h ^= __shfl_xor(h, o);
    if (lane == 0) out[row] = h;
}
#endif  // FVIT_DIAG

}  // extern "C"

#ifdef FVIT_DIAG

namespace fvit {
struct DbgTrace { unsigned* buf = nullptr; long long cap = 0, used = 0; int nrec = 0; int ndump = 0; int dump_rec[64]; void* dump_dst[64]; long long dump_cap[64]; FvitDebugRowhashRecord rec[FVIT_DEBUG_MAX_RECORDS]; };
static DbgTrace g_dbg;
void dbg_rowhash(const char* tag, const void* ptr, long long rows, int row_bytes, hipStream_t st) {
    std::lock_guard<std::recursive_mutex> lock(diag_mutex());
    if (!g_dbg.buf || !ptr || rows <= 0) return;
    if (g_dbg.nrec >= FVIT_DEBUG_MAX_RECORDS || g_dbg.used + rows > g_dbg.cap) return;
    FvitDebugRowhashRecord& r = g_dbg.rec[g_dbg.nrec++];
    snprintf(r.tag, sizeof(r.tag), "%s", tag);
    r.offset = g_dbg.used; r.rows = rows;
    hipLaunchKernelGGL(rowhash_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const unsigned*)ptr, rows, row_bytes / 4, g_dbg.buf + g_dbg.used);
    g_dbg.used += rows;
    for (int i = 0; i < g_dbg.ndump; ++i)
        if (g_dbg.nrec - 1 == g_dbg.dump_rec[i] && g_dbg.dump_dst[i] && rows * row_bytes <= g_dbg.dump_cap[i])
            (void)hipMemcpyAsync(g_dbg.dump_dst[i], ptr, (size_t)rows * row_bytes, hipMemcpyDeviceToDevice, st);
}
}  // namespace fvit

namespace fvit {
static unsigned* g_poison_sink = nullptr;
static unsigned g_poison_pattern = 0;
void dbg_poison_before_launch(hipStream_t st) {
    std::lock_guard<std::recursive_mutex> lock(diag_mutex());
    if (!g_poison_sink) return;
    hipLaunchKernelGGL(regs_poison_kernel, dim3(2048), dim3(64), 40 * 1024, st, g_poison_sink, 64, g_poison_pattern);
}
}  // namespace fvit

extern "C" {
int fvit_debug_poison_launches(void* sink, uint32_t pattern) {
    std::lock_guard<std::recursive_mutex> lock(fvit::diag_mutex());
    fvit::g_poison_sink = (unsigned*)sink; fvit::g_poison_pattern = pattern;
    return FVIT_OK;
}
int fvit_debug_rowhash_begin(void* buf, int64_t capacity_words) {
    std::lock_guard<std::recursive_mutex> lock(fvit::diag_mutex());
    fvit::g_dbg.buf = (unsigned*)buf; fvit::g_dbg.cap = capacity_words; fvit::g_dbg.used = 0; fvit::g_dbg.nrec = 0;
    return FVIT_OK;
}
int fvit_debug_rowhash_dump(int32_t record, void* dst, int64_t capacity_bytes) {
    std::lock_guard<std::recursive_mutex> lock(fvit::diag_mutex());
    if (record < 0) { fvit::g_dbg.ndump = 0; return FVIT_OK; }
    if (fvit::g_dbg.ndump >= 64) { set_error("debug_rowhash_dump: at most 64 records"); return FVIT_EINVAL; }
    const int i = fvit::g_dbg.ndump++;
    fvit::g_dbg.dump_rec[i] = record; fvit::g_dbg.dump_dst[i] = dst; fvit::g_dbg.dump_cap[i] = capacity_bytes;
    return FVIT_OK;
}
int fvit_debug_rowhash_end(FvitDebugRowhashRecord* out, int32_t max_records) {
    std::lock_guard<std::recursive_mutex> lock(fvit::diag_mutex());
    int n = fvit::g_dbg.nrec < max_records ? fvit::g_dbg.nrec : max_records;
    for (int i = 0; i < n && out; ++i) out[i] = fvit::g_dbg.rec[i];
    fvit::g_dbg.buf = nullptr;
    return n;
}
}  // extern "C"
#endif  // FVIT_DIAG
