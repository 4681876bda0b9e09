// fvit_lngemm.hip -- LayerNorm folded into the A-operand staging of the following Linear layer (gfx950):
//
//   out[m][n] = epilogue( sum_k LN(v[m])[k] * Wt[n][k] + bias[n] ),   v[m] = gathered source row (+ add row)   [fvit_gather_layernorm]
//
// Replaces  nn.LayerNorm (AR:616 / 648-649)  +  the nn.Linear that consumes it: hat_norm1 -> hat_attn.qkv (AR:682-683, with the
// ct_dewindow gather and hat_pos_embed add), hat_norm2 -> hat_mlp.fc1 (AR:684) and norm2 -> mlp.fc1 (AR:697) on the small-grid
// launches of the carrier-token branch and of stage 3, where a separate LayerNorm kernel is one more launch at the dispatch floor
// plus a write + re-read of the normalised rows (22 LayerNorm launches per stream shard).
//
// Design (C = K = 256 or 512, i.e. KC = 4 or 8 chunks of 64 channels):
//   * a workgroup (4 wave64) owns 64 rows and NT consecutive 128-column weight tiles.  Prologue: each wave normalises 16 of the
//     rows in registers (fp32, two-pass mean / variance, the arithmetic of ln_kernel) and writes them ONCE, as 16-bit MFMA operands,
//     into an LDS panel [KC][64 rows][64 k] that stays resident for the whole workgroup -- in the swizzled layout gemm_kernel's
//     fragment reads expect, so the main loop is gemm_kernel's with the activation tile already in place.
//   * main loop over the NT * KC weight tiles (16 KiB each, 16-byte global_load_lds, 4-deep ring behind a COUNTED s_waitcnt vmcnt + raw
//     s_barrier: three tiles in flight; a 2-deep ring made every tile one exposed LDS-DMA round trip, 2-2.5 us per tile on the
//     264-workgroup stage-3 launches, r02): only the WEIGHTS stream; the activation operand costs one fp32 row read per workgroup
//     instead of one 16-bit tile read per K step.
//   * epilogue per 128-column tile: bias (+ exact-erf GELU), packed 16-bit stores (gemm_kernel's swapped-operand layout).
//   * the optional fp32 copy of the gathered rows (x_out: the carrier-token stream R) is written by the workgroups of column
//     group 0 only; x_out must not alias the gather sources (the in-place `x += pos_embed` of the window branch keeps the
//     separate LayerNorm kernel: other column groups would read rows a neighbour has already rewritten).
#include "fvit_common.h"

namespace fvit {

namespace {

constexpr int BK = 64, BN = 128, BMT = 64;   // 4 wave64 per workgroup
constexpr int W_TILE_BYTES = BN * BK * 2;   // 16 KiB
constexpr int A_CHUNK_BYTES = BMT * BK * 2; // 8 KiB: 64 rows x 64 k

struct LnGemmParams {
    // LayerNorm source (LnCall semantics)
    const float* srcA;
    const float* srcB;
    const int32_t* src_idx;
    const int32_t* add_idx;
    const float* add;
    float* x_out;
    const float* ln_w;
    const float* ln_b;
    float eps;
    int rowsA, rowsB, rows_per_image;
    // GEMM
    const void* W;
    const float* bias;
    void* out;
    int ldw, ldo;
    int M, N;
    int tiles_n, groups_n, nt;   // 128-column tiles, column groups of `nt` tiles
};

__device__ __forceinline__ int swz_x(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ int swz_w(int r) { return ((0x78 >> (2 * ((r >> 4) & 3))) & 3) | ((r & 2) << 1); }

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <typename T>
__device__ __forceinline__ void stage_w(const T* __restrict__ g, int ld, int row0, int k0, char* tile, int wave, int lane) {
    // 16 pieces of 1 KiB (8 rows x 128 B); wave w issues pieces 4w .. 4w + 3 (same image as gemm_kernel's W tile)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave * 4 + i;
        const int r = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz_w(r);
        glds16(g + (size_t)(row0 + r) * ld + k0 + c * 8, tile + piece * 1024);
    }
}

template <typename T, int EPI, int KC>
__global__ __launch_bounds__(256, 1) void lngemm_kernel(LnGemmParams p) {
    typedef typename Op16<T>::v8 v8;
    constexpr int K = KC * 64;
    constexpr int LPR = K / 8;        // lanes per row: a lane owns 8 consecutive channels = one 16-byte operand chunk
    constexpr int RPS = 64 / LPR;     // rows a wave normalises per step (1 at K = 512, 2 at K = 256)
    constexpr int STEPS = 16 / RPS;   // 16 rows per wave
    constexpr int NS = 4;             // weight ring depth
    __shared__ __attribute__((aligned(16))) char smem[KC * A_CHUNK_BYTES + NS * W_TILE_BYTES];
    char* const apanel = smem;
    char* const wring = smem + KC * A_CHUNK_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile id (bijective for any grid size): column groups of one row block land on the same XCD
    const int nblk = gridDim.x, b = blockIdx.x;
    const int q = nblk >> 3, rr = nblk & 7, xcd = b & 7, idx = b >> 3;
    const int v = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    const int tm = v / p.groups_n, tg = v - tm * p.groups_n;
    const int m0 = tm * BMT;
    const int tn0 = tg * p.nt;
    const int ntiles = min(p.nt, p.tiles_n - tn0);
    const int total = ntiles * KC;   // weight tiles this workgroup streams

    const T* __restrict__ W = (const T*)p.W;
    auto stage_tile_t = [&](int t) {   // weight tile t of this workgroup's sequence -> ring slot t % NS
        const int n1 = t / KC, k1 = t - n1 * KC;
        stage_w<T>(W, p.ldw, (tn0 + n1) * BN, k1 * BK, wring + (t % NS) * W_TILE_BYTES, wave, lane);
    };
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < total) stage_tile_t(t);

    // ---- prologue: gather + add + LayerNorm of 16 rows per wave, straight into the LDS operand panel ----
    {
        const int sub = lane / LPR, l = lane - sub * LPR;   // row inside the step, chunk owner
        const f4 w0 = *(const f4*)(p.ln_w + l * 8), w1 = *(const f4*)(p.ln_w + l * 8 + 4);
        const f4 b0 = *(const f4*)(p.ln_b + l * 8), b1 = *(const f4*)(p.ln_b + l * 8 + 4);
        const int kt = l >> 3, c = l & 7;                   // 64-channel chunk, 16-byte piece inside the chunk's 128-byte row
#pragma unroll 4
        for (int st = 0; st < STEPS; ++st) {
            const int r = wave * 16 + st * RPS + sub;       // row inside the 64-row block
            const int row = min(m0 + r, p.M - 1);           // tail rows recompute the last row; their outputs are never stored
            const int bi = row / p.rows_per_image, pr = row - bi * p.rows_per_image;
            const float* src;
            if (p.src_idx) {
                const int si = p.src_idx[pr];
                src = si >= 0 ? p.srcA + ((size_t)bi * p.rowsA + si) * K : p.srcB + ((size_t)bi * p.rowsB + (-si - 1)) * K;
            } else {
                src = p.srcA + (size_t)row * K;
            }
            f4 x0 = *(const f4*)(src + l * 8), x1 = *(const f4*)(src + l * 8 + 4);
            if (p.add) {
                const int ai = p.add_idx ? p.add_idx[pr] : pr;
                if (ai >= 0) {
                    x0 += *(const f4*)(p.add + (size_t)ai * K + l * 8);
                    x1 += *(const f4*)(p.add + (size_t)ai * K + l * 8 + 4);
                }
            }
            float sum = ((x0[0] + x0[1]) + (x0[2] + x0[3])) + ((x1[0] + x1[1]) + (x1[2] + x1[3]));
            sum = group_sum<LPR>(sum);
            const float mean = sum / (float)K;
            const f4 d0 = x0 - mean, d1 = x1 - mean;
            float sq = ((d0[0] * d0[0] + d0[1] * d0[1]) + (d0[2] * d0[2] + d0[3] * d0[3])) +
                       ((d1[0] * d1[0] + d1[1] * d1[1]) + (d1[2] * d1[2] + d1[3] * d1[3]));
            sq = group_sum<LPR>(sq);
            const float rstd = rsqrtf(sq / (float)K + p.eps);
            if (p.x_out && tg == 0 && m0 + r < p.M) {
                float* xo = p.x_out + (size_t)(m0 + r) * K + l * 8;
                *(f4*)xo = x0;
                *(f4*)(xo + 4) = x1;
            }
            v8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = sat16<T>(d0[e] * rstd * w0[e] + b0[e]);
                o[4 + e] = sat16<T>(d1[e] * rstd * w1[e] + b1[e]);
            }
            *(v8*)(apanel + kt * A_CHUNK_BYTES + r * 128 + ((c ^ swz_x(r)) << 4)) = o;
        }
    }

    // per-lane fragment addressing (bytes inside a tile), as in gemm_kernel with MI = 2
    const int g = lane >> 4, s = lane & 15;
    int xrow[2], wrow[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) xrow[i] = wm * 32 + i * 16 + s;
#pragma unroll
    for (int i = 0; i < 4; ++i) wrow[i] = wn * 64 + (s >> 2) * 16 + i * 4 + (s & 3);

    f4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < total; ++t) {
        // weight tile t landed once at most the 4 LDS-DMA instructions per wave of each later tile are outstanding (and, at t = 0, the
        // operand panel is complete: lgkmcnt(0)); every wave is past tile t - 1.  Raw barrier: __syncthreads() would drain the queue.
        const int ahead = min(NS - 2, total - 1 - t);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int nti = t / KC, kt = t - nti * KC;
        if (t + NS - 1 < total) stage_tile_t(t + NS - 1);   // refills the slot tile t - 1 was read from
        const char* xt = apanel + kt * A_CHUNK_BYTES;
        const char* wt = wring + (t % NS) * W_TILE_BYTES;
        v8 xf[2][2], wf[2][4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int c = kk * 4 + g;
#pragma unroll
            for (int i = 0; i < 2; ++i) xf[kk][i] = *(const v8*)(xt + xrow[i] * 128 + ((c ^ swz_x(xrow[i])) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[kk][i] = *(const v8*)(wt + wrow[i] * 128 + ((c ^ swz_w(wrow[i])) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = Op16<T>::mfma(wf[kk][ni], xf[kk][mi], acc[ni][mi]);
        if (kt == KC - 1) {
            // ---- epilogue of column tile tn0 + nti: lane holds out[m][nb .. nb + 15] for 2 rows m ----
            const int nb = (tn0 + nti) * BN + wn * 64 + g * 16;
            if (nb < p.N) {   // N is a multiple of 16
                float bias[16];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f4 tb = p.bias ? *(const f4*)(p.bias + nb + j * 4) : (f4){0.f, 0.f, 0.f, 0.f};
                    bias[j * 4 + 0] = tb[0]; bias[j * 4 + 1] = tb[1]; bias[j * 4 + 2] = tb[2]; bias[j * 4 + 3] = tb[3];
                }
                T* O = (T*)p.out;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const int m = m0 + wm * 32 + mi * 16 + s;
                    if (m < p.M) {
                        v8 o0, o1;
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) {
                            const f4 a = acc[ni][mi];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float y = a[r] + bias[ni * 4 + r];
                                if (EPI == 1) y = gelu_fast(y);
                                if (ni < 2) o0[ni * 4 + r] = sat16<T>(y); else o1[(ni - 2) * 4 + r] = sat16<T>(y);
                            }
                        }
                        T* po = O + (size_t)m * p.ldo + nb;
                        *(v8*)po = o0;
                        *(v8*)(po + 8) = o1;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
        }
    }
}

template <typename T>
int launch_t(const LnGemmCall& c, hipStream_t stream) {
    LnGemmParams p;
    p.srcA = c.ln.srcA; p.srcB = c.ln.srcB; p.src_idx = c.ln.src_idx; p.add_idx = c.ln.add_idx; p.add = c.ln.add; p.x_out = c.ln.x_out;
    p.ln_w = c.ln.ln_w; p.ln_b = c.ln.ln_b; p.eps = c.ln.eps; p.rowsA = c.ln.rowsA; p.rowsB = c.ln.rowsB;
    p.rows_per_image = c.ln.rows_per_image > 0 ? c.ln.rows_per_image : 1;
    p.W = c.W; p.bias = c.bias; p.out = c.out; p.ldw = c.ldw; p.ldo = c.ldo; p.M = c.ln.rows; p.N = c.N;
    p.tiles_n = (c.N + BN - 1) / BN;
    const int tiles_m = (p.M + BMT - 1) / BMT;
    const int K = c.ln.C;
    // column tiles per workgroup: as few as keep the launch within one round of workgroups (K = 512: one 96-KiB workgroup per CU;
    // K = 256: two 64-KiB workgroups per CU), so that the LayerNorm prologue is repeated for as few column groups as possible
    const int cap = 256 + tune_get("lngemm_extra_wgs", 32);   // 96 / 128 KiB of LDS: one workgroup per CU
    int nt = tune_get("lngemm_nt", 0);
    if (nt <= 0) {
        nt = 1;
        while (nt < p.tiles_n && tiles_m * ((p.tiles_n + nt - 1) / nt) > cap) ++nt;
    }
    p.nt = nt;
    p.groups_n = (p.tiles_n + nt - 1) / nt;
    const int grid = tiles_m * p.groups_n;
    const double flops = 2.0 * p.M * (double)c.N * K + 8.0 * p.M * (double)K;
    const double bytes = 4.0 * p.M * (double)K + 2.0 * c.N * (double)K + 2.0 * p.M * (double)c.N + (c.ln.x_out ? 4.0 * p.M * (double)K : 0.0);
    ProfScope prof(c.epilogue == 1 ? FVIT_K_GEMM_GELU : FVIT_K_GEMM_BIAS, flops, bytes, stream);
    prof_note(c.epilogue == 1 ? "lngemm_kernel<1>" : "lngemm_kernel<0>", grid);
#define FVIT_LNG(E, KC_) hipLaunchKernelGGL((lngemm_kernel<T, E, KC_>), dim3(grid), dim3(256), 0, stream, p)
    if (K == 256) { if (c.epilogue == 1) FVIT_LNG(1, 4); else FVIT_LNG(0, 4); }
    else { if (c.epilogue == 1) FVIT_LNG(1, 8); else FVIT_LNG(0, 8); }
#undef FVIT_LNG
    return check_launch("lngemm_kernel");
}

}  // namespace

bool ln_gemm_supported(int C, int N, int ldw, int ldo) {
    return (C == 256 || C == 512) && N > 0 && (N % 16) == 0 && ldw >= C && (ldw % 8) == 0 && (ldo % 8) == 0;
}

int launch_ln_gemm(const LnGemmCall& c, hipStream_t stream) {
    if (!ln_gemm_supported(c.ln.C, c.N, c.ldw, c.ldo) || c.ln.rows <= 0 || !c.ln.srcA || !c.W || !c.out || !c.ln.ln_w || !c.ln.ln_b ||
        (c.ln.x_out && (c.ln.x_out == c.ln.srcA || c.ln.x_out == c.ln.srcB))) {
        set_error("ln_gemm: unsupported arguments C=%d N=%d rows=%d ldw=%d ldo=%d (C must be 256 or 512; x_out must not alias the sources)",
                  c.ln.C, c.N, c.ln.rows, c.ldw, c.ldo);
        return FVIT_EINVAL;
    }
    if (c.ln.dtype == FVIT_F16) return launch_t<_Float16>(c, stream);
    if (c.ln.dtype == FVIT_BF16) return launch_t<__bf16>(c, stream);
    set_error("ln_gemm: operand dtype %d not supported", c.ln.dtype);
    return FVIT_EINVAL;
}

}  // namespace fvit
