// fvit_gemm.hip -- 16-bit MFMA GEMM with fused epilogues for the HAT Linear layers (gfx950).
//
//   out[m][n] = epilogue( sum_k X[m][k] * Wt[n][k] + bias[n] )
//
// Replaces the nn.Linear calls of WindowAttention (qkv, proj; AR:560,567) and Mlp (fc1, fc2;
// AR:402-405) together with the ops the reference runs after them: bias add, exact-erf GELU
// (AR:403), and the gamma-scaled residual add (AR:696-697).
//
// Design (CDNA4):
//   * 128x128 output tile, BK = 64, 256 threads = 4 wave64 (2x2), each wave a 64x64 sub-tile as
//     4x4 fragments of v_mfma_f32_16x16x32_{f16,bf16}; fp32 accumulators (64 VGPR).
//   * Operands are staged HBM -> LDS with 16-byte global_load_lds (no VGPR round trip), double
//     buffered (2 x 2 x 16 KiB), one barrier per K step.  The LDS image is lane-linear, so the
//     bank-conflict swizzle is applied on the per-lane SOURCE address and undone on the ds_read.
//   * The MFMA is issued "swapped": the weight tile is the A operand and the activation tile the
//     B operand, and weight rows are assigned to A-row slots so that every lane ends up holding
//     16 CONSECUTIVE output columns of one output row: the epilogue is 16-byte vector traffic
//     (bias / gamma loads, fp32 residual read-modify-write, packed 16-bit stores), 128 B
//     contiguous per row and store instruction.
//   * blockIdx -> tile mapping is XCD aware: each XCD (blockIdx % 8) owns a contiguous range of
//     tiles ordered n-fastest, so the activation tile is fetched from HBM once per XCD L2 and the
//     (small) weight matrix stays L2 resident.
#include <algorithm>

#include "fvit_common.h"

namespace fvit {

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;  // 16 KiB per operand tile

struct GemmParams {
    const void* A;
    const void* W;
    const float* bias;
    const float* gamma;
    void* out;
    int lda, ldw, ldo;
    int M, N, K;
    int arow_max, wrow_max;   // last valid row of the 128-row padded operands
    int ka;       // activation columns of one term; contraction index k reads A column (k >= ka ? k - ka : k) (K = 1, 2 or 3 x ka, see GemmCall)
    int lo_off;   // epilogues 0 / 1: > 0 = also store the second term lo = round(y - hi) at column offset lo_off (elements) of the same row
    int splits;   // > 1 (EPI 3): deterministic split-K -- the grid is splits x tiles, workgroup (sp, tile) accumulates K tiles [sp nk / splits, (sp + 1) nk / splits)
    float* slab;  // EPI 3: f32 [splits][M][N] partial sums (no bias); splitk_reduce_kernel adds them in split order and applies the residual epilogue
    int tiles_m, tiles_n;
    int stagger;  // 1: workgroup (tm, tn) walks the K tiles starting at a tile-dependent offset (see gemm_kernel)
    int x3;       // 1 (X3 instances, K = 3 x ka): "dual" K loop -- a K tile is 32 contraction indices of BOTH terms of both operands: LDS rows hold
                  // [hi k0..k0+31 | lo k0..k0+31] (the lo image at column ka of the same global row), and the tile contributes w_hi.a_hi + w_lo.a_hi + w_hi.a_lo
    const float* add;        // residual epilogue: optional row table added after the update (see GemmCall)
    const int32_t* add_idx;
    int rpi;
};

template <bool V> struct BoolTag { static constexpr bool value = V; };

// swizzle of the 16-byte chunk index inside a 128-byte LDS row.
//  X tile: fragments read 16 consecutive rows           -> f = (r >> 1) & 7
//  W tile: fragments read rows {g*16 + ni*4 + r'}        -> f = perm(g) | (r' >> 1) << 2
__device__ __forceinline__ int swz_x(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ int swz_w(int r) { return ((0x78 >> (2 * ((r >> 4) & 3))) & 3) | ((r & 2) << 1); }

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <typename T, bool IS_W, int ROWS, int NW>
__device__ __forceinline__ void stage_tile(const T* __restrict__ g, int ld, int row0, int k0, char* tile,
                                           int wave, int lane, int rmax, int lo_col = 0) {
    // ROWS/8 pieces of 1 KiB (8 rows x 128 B); wave w issues pieces PW*w .. PW*w + PW-1.
    constexpr int PW = ROWS / 8 / NW;
    static_assert(PW >= 1, "tile too small for this many waves");
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int piece = wave * PW + i;
        const int r = piece * 8 + (lane >> 3);
        const int f = IS_W ? swz_w(r) : swz_x(r);
        const int c = (lane & 7) ^ f;
        // rows are clamped to the last row of the (128-row padded) operand: the 256-row tiles reach up to 128 rows further
        // lo_col > 0 (dual x3 tiles): chunks 0-3 of the LDS row = 32 columns of the hi image at k0, chunks 4-7 = the same 32 columns of the lo image
        const int col = lo_col > 0 ? k0 + (c & 3) * 8 + ((c >> 2) ? lo_col : 0) : k0 + c * 8;
        const T* src = g + (size_t)min(row0 + r, rmax) * ld + col;
        glds16(src, tile + piece * 1024);
    }
}


// NSTAGE = 2: double buffer, one drained barrier per K step, two workgroups per CU -- for grids that fill the chip.
// NSTAGE = 3: three-deep ring with a COUNTED s_waitcnt vmcnt (the tile after the current one stays in flight across
//             the raw s_barrier), one workgroup per CU -- for small grids (<= ~1.5 workgroups per CU), where a K step is
//             otherwise one full LDS-DMA round trip (~1.3 us) because nothing else on the CU hides it.
// MI: 16-row fragments per wave along M (4: 128-row tile; 2: 64-row tile, 48 KiB LDS, three workgroups per CU -- for
//     small grids, where twice the workgroups at higher occupancy hide the per-K-step DMA latency better)
// NW: waves per workgroup, NW/2 along M x 2 along N (8 = two waves per SIMD even with one workgroup per CU).
// Measured r01 (scripts/bench_gemm.py, profiles/r01_gemm_small_grid_variants.log): on the shard-sized stage-3 GEMMs (132-264
// workgroups, K = 512..2048) the time per K step is 0.58 us WHATEVER the tile height (64 / 128 rows), the ring depth (2 / 3), the
// LDS read schedule or the waves per workgroup (4 / 8): it is set by the memory system, not by the instruction stream -- a CU
// sustains ~16 KiB of loads in flight (scripts/probes/dma_probe.hip: 125 GB/s per CU from L2, 30-40 GB/s per CU from HBM / MALL,
// independent of ring depth), and the A operand of these GEMMs was just written by the previous kernel and comes from the
// memory side.  NW = 8 and NSTAGE = 3 are therefore opt-in knobs only.
// NI: 16-column fragments per wave along N (4: 128-column tile; 8: 256-column tile).  NW = 8, MI = 4, NI = 8 is the 256 x 256 x 64 tile
//     (r03): 4 waves along M x 2 along N, each wave 64 rows x 128 columns = 128 accumulator registers, two waves per SIMD, 128 KiB of
//     LDS (2 x (32 + 32) KiB), one workgroup per CU.  Per K step a wave reads 12 fragments per 32-deep half and issues 32 MFMAs on
//     them (2.7 MFMAs per ds_read_b128 against 2.0 for the 128 x 128 tile), and a byte staged into LDS is used by 4 (activations) / 2
//     (weights) waves: the shape for the large Linear layers of FasterViT-4 (K = 832 ... 6272), where the 128 x 128 tile plateaus
//     at 0.22-0.25 of the MFMA peak.
template <typename T, int EPI, int NSTAGE, int MI, int NW, int NI = 4, bool X3 = false>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void gemm_kernel(GemmParams p) {
    typedef typename Op16<T>::v8 v8;
    constexpr int BMT = (NW / 2) * 16 * MI;            // rows of the workgroup tile (NW/2 waves along M)
    constexpr int BNT = 2 * 16 * NI;                   // columns of the workgroup tile (2 waves along N)
    constexpr int XT_BYTES = BMT * BK * 2;             // activation tile bytes
    constexpr int WT_BYTES = BNT * BK * 2;             // weight tile bytes
    __shared__ __attribute__((aligned(16))) char smem[NSTAGE * (XT_BYTES + WT_BYTES)];  // X ring, then W ring

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile id (bijective for any grid size)
    const int nblk = gridDim.x, b = blockIdx.x;
    const int q = nblk >> 3, rr = nblk & 7, xcd = b & 7, idx = b >> 3;
    const int v = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    int sp = 0, vt = v;
    if (EPI == 3) {   // split-K: consecutive ids = the tiles of one K range
        const int tiles = p.tiles_m * p.tiles_n;
        sp = v / tiles;
        vt = v - sp * tiles;
    }
    const int tm = vt / p.tiles_n, tn = vt - tm * p.tiles_n;
    const int m0 = tm * BMT, n0 = tn * BNT;

    const T* __restrict__ A = (const T*)p.A;
    const T* __restrict__ W = (const T*)p.W;

    f4 acc[NI][MI];  // [ni][mi]
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

    const int nk_all = X3 ? p.ka / 32 : p.K / BK;   // X3: one tile per 32 contraction indices (all three products)
    const int k_lo = EPI == 3 ? sp * nk_all / p.splits : 0;
    const int nk = EPI == 3 ? (sp + 1) * nk_all / p.splits - k_lo : nk_all;
    // K-order stagger: the workgroups of an XCD that share an activation panel (same tm) or a weight panel (same tn) start at
    // different K tiles, so a panel chunk is fetched from the memory side by ONE of them and found in L2 by the others when
    // they get there; in lockstep every sharer waits on the same outstanding miss (hit-under-miss = full memory latency for
    // all of them, every K step).  The sum order changes per tile but is fixed per (tm, tn): results stay bit-reproducible.
    int rot = 0;
    if (p.stagger) {
        const int per = p.tiles_n < nk ? nk / p.tiles_n : 1;
        rot = (tn * per + tm) % nk;
    }
    auto ktile = [&](int kt) { const int k = kt + rot; return k_lo + (k >= nk ? k - nk : k); };
    auto acol = [&](int kt) { if (X3) return ktile(kt) * 32; const int k = ktile(kt) * BK; return k >= p.ka ? k - p.ka : k; };   // K = 1, 2 or 3 x ka: terms [hi w | lo w | hi w] meet A columns [hi | hi | lo]
    auto wcol = [&](int kt) { return ktile(kt) * (X3 ? 32 : BK); };
    const int lo_col = X3 ? p.ka : 0;
    char* const xring = smem;
    char* const wring = smem + NSTAGE * XT_BYTES;
#pragma unroll
    for (int st = 0; st < NSTAGE - 1; ++st) {
        if (st < nk) {
            stage_tile<T, false, BMT, NW>(A, p.lda, m0, acol(st), xring + st * XT_BYTES, wave, lane, p.arow_max, lo_col);
            stage_tile<T, true, BNT, NW>(W, p.ldw, n0, wcol(st), wring + st * WT_BYTES, wave, lane, p.wrow_max, lo_col);
        }
    }

    // per-lane fragment addressing (bytes inside a tile)
    const int g = lane >> 4, s = lane & 15;
    int xrow[MI], wrow[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) xrow[i] = wm * (16 * MI) + i * 16 + s;        // activation row (B operand column)
#pragma unroll
    for (int i = 0; i < NI; ++i) wrow[i] = wn * (16 * NI) + (i >> 2) * 64 + (s >> 2) * 16 + (i & 3) * 4 + (s & 3);  // weight row for A-row slot s of fragment i

    int cur = 0;  // ring slot of tile kt
    for (int kt = 0; kt < nk; ++kt) {
        if (NSTAGE == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
            // each wave issues PER LDS-DMA instructions per stage (X pieces + W pieces): tile kt has landed once at most the
            // instructions of the (up to NSTAGE - 2) tiles after it are still outstanding.  Raw barrier: __syncthreads() would
            // drain the queue (vmcnt(0)).
            constexpr int PER = BMT / 8 / NW + BNT / 8 / NW;
            const int ahead = min(NSTAGE - 2, nk - 1 - kt);
            if (NSTAGE >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * PER) : "memory");
            else if (ahead >= 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        {
            // refill the slot that was read in iteration kt-1 (every wave is past that: it arrived at this barrier)
            const int nxt = kt + NSTAGE - 1;
            int slot = cur + NSTAGE - 1;
            if (slot >= NSTAGE) slot -= NSTAGE;
            if (nxt < nk) {
                stage_tile<T, false, BMT, NW>(A, p.lda, m0, acol(nxt), xring + slot * XT_BYTES, wave, lane, p.arow_max, lo_col);
                stage_tile<T, true, BNT, NW>(W, p.ldw, n0, wcol(nxt), wring + slot * WT_BYTES, wave, lane, p.wrow_max, lo_col);
            }
        }
        const char* xt = xring + cur * XT_BYTES;
        const char* wt = wring + cur * WT_BYTES;
        if constexpr (NI == 8) {
            static_assert(!X3, "the dual x3 K loop needs both chunk halves of a tile at once: 128-column tiles or gemm_pp_kernel");
            // 256-column tile: one 32-deep half at a time (12 fragments = 48 registers in flight beside the 128 accumulators); the
            // SIMD's second wave covers the LDS round trip
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int c = kk * 4 + g;
                v8 xf[MI], wf[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) xf[i] = *(const v8*)(xt + xrow[i] * 128 + ((c ^ swz_x(xrow[i])) << 4));
#pragma unroll
                for (int i = 0; i < NI; ++i) wf[i] = *(const v8*)(wt + wrow[i] * 128 + ((c ^ swz_w(wrow[i])) << 4));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = Op16<T>::mfma(wf[ni], xf[mi], acc[ni][mi]);
            }
        } else {
        // all fragment reads of the K step are issued up front (both 32-deep halves): left to itself the compiler reads one
        // half, drains lgkmcnt, multiplies, reads the next half, drains again -- three exposed LDS round trips per K step, which
        // is what bounds small grids (one wave per SIMD, nothing else to hide them)
        v8 xf[2][MI], wf[2][4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int c = kk * 4 + g;
#pragma unroll
            for (int i = 0; i < MI; ++i) xf[kk][i] = *(const v8*)(xt + xrow[i] * 128 + ((c ^ swz_x(xrow[i])) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[kk][i] = *(const v8*)(wt + wrow[i] * 128 + ((c ^ swz_w(wrow[i])) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (X3) {   // chunk half 0 = hi, half 1 = lo of the same 32 contraction indices: w_hi.a_hi + w_lo.a_hi + w_hi.a_lo
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = Op16<T>::mfma(wf[pr == 1][ni], xf[pr == 2][mi], acc[ni][mi]);
        } else {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = Op16<T>::mfma(wf[kk][ni], xf[kk][mi], acc[ni][mi]);
        }
        }
        if (++cur == NSTAGE) cur = 0;
    }

    // ---- epilogue: per 64-column group cg the lane holds out[m][nb .. nb+15] for MI rows m (one per mi) ----
#pragma unroll
    for (int cg = 0; cg < NI / 4; ++cg) {
    const int nb = n0 + wn * (16 * NI) + cg * 64 + g * 16;
    if (nb >= p.N) continue;  // N is a multiple of 16
    float bias[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f4 t = p.bias ? *(const f4*)(p.bias + nb + j * 4) : (f4){0.f, 0.f, 0.f, 0.f};
        bias[j * 4 + 0] = t[0]; bias[j * 4 + 1] = t[1]; bias[j * 4 + 2] = t[2]; bias[j * 4 + 3] = t[3];
    }
    if (EPI == 3) {   // split-K partial: raw fp32 accumulators into this split's slab image
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m0 + wm * (16 * MI) + mi * 16 + s;
            if (m < p.M) {
                float* P = p.slab + ((size_t)sp * p.M + m) * p.N + nb;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) *(f4*)(P + ni * 4) = acc[cg * 4 + ni][mi];
            }
        }
    } else if (EPI == 2) {
        float gam[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f4 t = p.gamma ? *(const f4*)(p.gamma + nb + j * 4) : (f4){1.f, 1.f, 1.f, 1.f};
            gam[j * 4 + 0] = t[0]; gam[j * 4 + 1] = t[1]; gam[j * 4 + 2] = t[2]; gam[j * 4 + 3] = t[3];
        }
        float* X = (float*)p.out;
        if (!p.add) {
            // two passes (r03): all residual rows first, then the updates and stores -- in one load-update-store loop every store
            // had to stay ahead of the next load (possible alias): 4 MI dependent round trips per lane
            f4 xr[MI][4];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int m = min(m0 + wm * (16 * MI) + mi * 16 + s, p.M - 1);   // clamped for the load
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) xr[mi][ni] = *(const f4*)(X + (size_t)m * p.ldo + nb + ni * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int m = m0 + wm * (16 * MI) + mi * 16 + s;
                if (m < p.M) {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        f4 x = xr[mi][ni];
                        const f4 a = acc[cg * 4 + ni][mi];
#pragma unroll
                        for (int r = 0; r < 4; ++r) x[r] += gam[ni * 4 + r] * (a[r] + bias[ni * 4 + r]);
                        *(f4*)(X + (size_t)m * p.ldo + nb + ni * 4) = x;
                    }
                }
            }
        } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = m0 + wm * (16 * MI) + mi * 16 + s;
            if (m < p.M) {
                float* px = X + (size_t)m * p.ldo + nb;
                const float* pa = nullptr;
                if (p.add) {
                    const int pr = m % p.rpi;
                    const int ai = p.add_idx ? p.add_idx[pr] : pr;
                    if (ai >= 0) pa = p.add + (size_t)ai * p.N + nb;
                }
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    f4 x = *(f4*)(px + ni * 4);
                    f4 a = acc[cg * 4 + ni][mi];
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[r] += gam[ni * 4 + r] * (a[r] + bias[ni * 4 + r]);
                    if (pa) x += *(const f4*)(pa + ni * 4);
                    *(f4*)(px + ni * 4) = x;
                }
            }
        }
        }
    } else {
        // one dispatch on "two-term output" per column group (r05): the per-element form `lo_off > 0 ? gelu_erf : gelu_fast` inside the unrolled loops
        // kept BOTH bodies (and a second evaluation for the lo term) in the instruction stream: 21k instructions, ~2 erf per value
        T* O = (T*)p.out;
        auto store16 = [&](auto lo_tag) {
            constexpr bool LO = decltype(lo_tag)::value;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int m = m0 + wm * (16 * MI) + mi * 16 + s;
                if (m < p.M) {
                    float y[16];
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        const f4 a = acc[cg * 4 + ni][mi];
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[ni * 4 + r] = a[r] + bias[ni * 4 + r];
                    }
                    if constexpr (EPI == 1) {   // 16 values as two groups of 8 interleaved chains (fvit_common.h), bitwise the per-value functions
                        if constexpr (LO) gelu_erf_each<16>(y); else gelu_fast_each<16>(y);
                    }
                    v8 o0, o1;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { o0[j] = sat16<T>(y[j]); o1[j] = sat16<T>(y[8 + j]); }
                    T* po = O + (size_t)m * p.ldo + nb;
                    *(v8*)po = o0;
                    *(v8*)(po + 8) = o1;
                    if constexpr (LO) {   // two-term activations (weight_terms 3): the rounding remainder as a second 16-bit term
                        v8 l0, l1;
#pragma unroll
                        for (int j = 0; j < 8; ++j) { l0[j] = sat16<T>(y[j] - (float)o0[j]); l1[j] = sat16<T>(y[8 + j] - (float)o1[j]); }
                        *(v8*)(po + p.lo_off) = l0;
                        *(v8*)(po + p.lo_off + 8) = l1;
                    }
                }
            }
        };
        if (p.lo_off > 0) store16(BoolTag<true>{}); else store16(BoolTag<false>{});
    }
    }
}


// ------------------------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, two wave groups in PING-PONG (r03): while the four waves of one group issue the 16 MFMAs of a phase, the four waves
// of the other group read their next fragments from LDS and request the next operand half-tile, and the barriers swap the roles.
//
//   waves   8 = 2 (M halves: the GROUP) x 4 (N quarters); wave tile 128 rows x 64 columns = 8 x 4 accumulator fragments (128 registers)
//   LDS     2 buffers x {X0, X1, W0, W1} half-tiles of 128 rows x 64 k (16 KiB each, the piece / swizzle layout of stage_tile) = 128 KiB
//   K tile  4 phases of 16 MFMAs: (a0, b0), (a0, b1), (a1, b1), (a1, b0) -- a = 64-row half of the wave's rows (8 fragment reads), b = 32-column
//           half of its columns (4 reads, both halves stay in registers); reads per phase 12 / 4 / 8 / 0
//   a phase reads + one half-tile request (2 LDS-DMA instructions per wave) | s_barrier | lgkmcnt(0), 16 MFMAs at raised priority | s_barrier
//           group 1 runs one barrier behind group 0, so a group's MFMA section coincides with the other group's read section
//   requests in phase 1 .. 4 of K tile t: X0(t+1), X1(t+1) into the other buffer, W0(t+2), W1(t+2) into THIS buffer (its W halves were read for
//           the last time in phase 2, and phase 2 retires its reads BEFORE its barrier).  Every destination was last read, and those reads
//           retired, at least one barrier before the first wave requests into it.
//   waits   ONE counted vmcnt per K tile, at the end of phase 4's read section: everything but the two W requests just issued has landed
//           (own pieces; the youngest needed request, X1(t+1), is two phases old); group 1 executes that wait one barrier after group 0 and
//           still one barrier before group 0 reads K tile t+1.
// The loads are never drained inside the loop and there is no point where all eight waves wait for memory at once.
// ------------------------------------------------------------------------------------------------------------
// XR (r06): depth of the ACTIVATION ring.  2 = the layout above.  3 = a third pair of X half-tiles (160 KiB = the whole LDS of the CU): X0(t+2), X1(t+2) are
// requested in phases 1 / 2 of K tile t into the slot X(t-1) was read from (the same hazard distance as X(t+1) into "the other buffer"), the W requests stay
// where they are, and the counted wait at the end of phase 4 leaves FOUR requests in flight (X(t+2), W(t+2): vmcnt(8)).  The activation panel was written by
// the previous kernel and comes from the memory side; the weights come from L2: the longer lookahead goes to the operand with the longer latency.
template <typename T, int EPI, bool X3 = false, int XR = 2>
__global__ __launch_bounds__(512, 1) void gemm_pp_kernel(GemmParams p) {
    typedef typename Op16<T>::v8 v8;
    constexpr int HT = 128 * BK * 2;            // half-tile bytes
    constexpr int XB = 2 * HT;                  // X0 X1 of one K tile / W0 W1 of one K tile
    constexpr int XRING = XR * XB;
    static_assert(XR == 2 || XR == 3, "activation ring: 2 or 3 K tiles");
    __shared__ __attribute__((aligned(1024))) char smem[XRING + 2 * XB];   // X ring, then the two W buffers

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;    // wm = group
    const int g = lane >> 4, s = lane & 15;

    const int nblk = gridDim.x, b = blockIdx.x;
    const int q = nblk >> 3, rr = nblk & 7, xcd = b & 7, idx = b >> 3;
    const int v = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    const int tm = v / p.tiles_n, tn = v - tm * p.tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;

    const T* __restrict__ A = (const T*)p.A;
    const T* __restrict__ W = (const T*)p.W;
    const int nk = X3 ? p.ka / 32 : p.K / BK;   // X3: dual tiles of 32 contraction indices x {hi, lo} (see GemmParams.x3)
    const int lo_col = X3 ? p.ka : 0;
    auto acol = [&](int kt) { if (X3) return kt * 32; const int k = kt * BK; return k >= p.ka ? k - p.ka : k; };   // K = 1, 2 or 3 x ka (see gemm_kernel)
    char* const wring = smem + XRING;
    auto xslot = [&](int kt) { return smem + (XR == 3 ? kt % 3 : (kt & 1)) * XB; };
    auto wslot = [&](int kt) { return wring + (kt & 1) * XB; };
    auto req_x = [&](int h, int kt) { stage_tile<T, false, 128, 8>(A, p.lda, m0 + 128 * h, acol(kt), xslot(kt) + h * HT, wave, lane, p.arow_max, lo_col); };
    auto req_w = [&](int h, int kt) { stage_tile<T, true, 128, 8>(W, p.ldw, n0 + 128 * h, kt * (X3 ? 32 : BK), wslot(kt) + h * HT, wave, lane, p.wrow_max, lo_col); };

    f4 acc[4][8];  // [ni][mi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

    // fragment byte offsets inside this wave's X half-tile / W half-tile (kk = 0; kk = 1 flips chunk bit 2): fragment i adds 2048 (X) / 512 (W)
    const int xo = wm * HT + s * 128, xs = (s >> 1) & 7;
    const int wr0 = (wn & 1) * 64 + (s >> 2) * 16 + (s & 3);
    const int wo = (wn >> 1) * HT + wr0 * 128, ws = swz_w(wr0);   // (relative to the W buffer of the K tile)
    const int xoff0 = xo + ((g ^ xs) << 4), xoff1 = xo + (((4 + g) ^ xs) << 4);
    const int woff0 = wo + ((g ^ ws) << 4), woff1 = wo + (((4 + g) ^ ws) << 4);

    // prologue: K tile 0 complete, the W halves (XR = 3: and the X halves) of K tile 1 in flight
    req_x(0, 0); req_x(1, 0); req_w(0, 0); req_w(1, 0);
    if (nk > 1) {
        if (XR == 3) { req_x(0, 1); req_x(1, 1); }
        req_w(0, 1); req_w(1, 1);
        if (XR == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");
    if (wm == 1) asm volatile("s_barrier" ::: "memory");   // group 1 runs one barrier behind

    v8 xf[2][4], wb0[2][2], wb1[2][2];
#define FVIT_PP_LOADX(a_)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                \
        xf[0][i] = *(const v8*)(cx + xoff0 + ((a_) * 4 + i) * 2048);                                               \
        xf[1][i] = *(const v8*)(cx + xoff1 + ((a_) * 4 + i) * 2048);                                               \
    }
#define FVIT_PP_LOADW(dst_, b_)                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
        dst_[0][i] = *(const v8*)(cw + woff0 + ((b_) * 2 + i) * 512);                                              \
        dst_[1][i] = *(const v8*)(cw + woff1 + ((b_) * 2 + i) * 512);                                              \
    }
#define FVIT_PP_MFMA(a_, b_, wsrc_)                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    if ((a_) == 0 && (b_) == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   /* phase 2: the W reads retire before the barrier */ \
    else asm volatile("s_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");                                           \
    __builtin_amdgcn_s_setprio(1);                                                                                 \
    _Pragma("unroll") for (int kk = 0; kk < (X3 ? 3 : 2); ++kk)                                                    \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                           \
            _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                       \
                acc[(b_) * 2 + ni][(a_) * 4 + mi] = Op16<T>::mfma(wsrc_[X3 ? (kk == 1) : kk][ni], xf[X3 ? (kk == 2) : kk][mi], acc[(b_) * 2 + ni][(a_) * 4 + mi]); \
    __builtin_amdgcn_s_setprio(0);                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    asm volatile("s_barrier" ::: "memory");                                                                       \
    __builtin_amdgcn_sched_barrier(0);

    for (int kt = 0; kt < nk; ++kt) {
        const char* const cx = xslot(kt);                // this K tile
        const char* const cw = wslot(kt);
        const bool more = kt + 1 < nk, more2 = kt + 2 < nk;
        const bool morex = XR == 3 ? more2 : more;       // the X tile requested during this K tile: t + 2 (three-deep ring) or t + 1
        const int ktx = kt + (XR == 3 ? 2 : 1);
        // phase 1
        FVIT_PP_LOADW(wb0, 0)
        FVIT_PP_LOADX(0)
        if (morex) req_x(0, ktx);
        FVIT_PP_MFMA(0, 0, wb0)
        // phase 2
        FVIT_PP_LOADW(wb1, 1)
        if (morex) req_x(1, ktx);
        FVIT_PP_MFMA(0, 1, wb1)
        // phase 3
        FVIT_PP_LOADX(1)
        if (more2) req_w(0, kt + 2);
        FVIT_PP_MFMA(1, 1, wb1)
        // phase 4
        if (more2) {
            req_w(1, kt + 2);
            if (XR == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        FVIT_PP_MFMA(1, 0, wb0)
    }
#undef FVIT_PP_LOADX
#undef FVIT_PP_LOADW
#undef FVIT_PP_MFMA
    if (wm == 0) asm volatile("s_barrier" ::: "memory");   // group 0 meets group 1's last barrier

    // ---- epilogue (the layout of gemm_kernel: lane holds out[m][nb .. nb + 15] for one row m per mi), four row fragments at a time.
    //      Wave tiles that lie inside M take a branch-free path: behind a per-lane `if (m < M)` the compiler cannot count the stores in
    //      flight and drains them (vmcnt(0)) before every row fragment -- eight write round trips per wave. ----
    const int nb = n0 + wn * 64 + g * 16;
    if (nb >= p.N) return;  // N is a multiple of 16
    float bias[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f4 t = p.bias ? *(const f4*)(p.bias + nb + j * 4) : (f4){0.f, 0.f, 0.f, 0.f};
        bias[j * 4 + 0] = t[0]; bias[j * 4 + 1] = t[1]; bias[j * 4 + 2] = t[2]; bias[j * 4 + 3] = t[3];
    }
    const int mw = m0 + wm * 128 + s;
    auto finish = [&](auto full_tag, auto add_tag, auto lo_tag) {
        constexpr bool FULL = decltype(full_tag)::value, ADD = decltype(add_tag)::value;
        if constexpr (EPI == 2) {
            float gam[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f4 t = p.gamma ? *(const f4*)(p.gamma + nb + j * 4) : (f4){1.f, 1.f, 1.f, 1.f};
                gam[j * 4 + 0] = t[0]; gam[j * 4 + 1] = t[1]; gam[j * 4 + 2] = t[2]; gam[j * 4 + 3] = t[3];
            }
            float* X = (float*)p.out;
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                f4 xr[4][4];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    const int m = FULL ? mw + (mh * 4 + mi) * 16 : min(mw + (mh * 4 + mi) * 16, p.M - 1);   // clamped for the load
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) xr[mi][ni] = *(const f4*)(X + (size_t)m * p.ldo + nb + ni * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    const int m = mw + (mh * 4 + mi) * 16;
                    if (FULL || m < p.M) {
                        const float* pa = nullptr;
                        if constexpr (ADD) {
                            const int pr = m % p.rpi;
                            const int ai = p.add_idx ? p.add_idx[pr] : pr;
                            if (ai >= 0) pa = p.add + (size_t)ai * p.N + nb;
                        }
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) {
                            f4 x = xr[mi][ni];
                            const f4 a = acc[ni][mh * 4 + mi];
#pragma unroll
                            for (int r = 0; r < 4; ++r) x[r] += gam[ni * 4 + r] * (a[r] + bias[ni * 4 + r]);
                            if constexpr (ADD) { if (pa) x += *(const f4*)(pa + ni * 4); }
                            *(f4*)(X + (size_t)m * p.ldo + nb + ni * 4) = x;
                        }
                    }
                }
            }
        } else {
            T* O = (T*)p.out;
            constexpr bool LO = decltype(lo_tag)::value;   // two-term output: dispatched ONCE (see gemm_kernel)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const int m = mw + mi * 16;
                if (FULL || m < p.M) {
                    float y[16];
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        const f4 a = acc[ni][mi];
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[ni * 4 + r] = a[r] + bias[ni * 4 + r];
                    }
                    if constexpr (EPI == 1) {   // 16 values as two groups of 8 interleaved chains (fvit_common.h), bitwise the per-value functions
                        if constexpr (LO) gelu_erf_each<16>(y); else gelu_fast_each<16>(y);
                    }
                    v8 o0, o1;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { o0[j] = sat16<T>(y[j]); o1[j] = sat16<T>(y[8 + j]); }
                    T* po = O + (size_t)m * p.ldo + nb;
                    *(v8*)po = o0;
                    *(v8*)(po + 8) = o1;
                    if constexpr (LO) {
                        v8 l0, l1;
#pragma unroll
                        for (int j = 0; j < 8; ++j) { l0[j] = sat16<T>(y[j] - (float)o0[j]); l1[j] = sat16<T>(y[8 + j] - (float)o1[j]); }
                        *(v8*)(po + p.lo_off) = l0;
                        *(v8*)(po + p.lo_off + 8) = l1;
                    }
                }
            }
        }
    };
    const bool full = m0 + wm * 128 + 128 <= p.M;
    if (EPI == 2 && p.add) { if (full) finish(BoolTag<true>{}, BoolTag<true>{}, BoolTag<false>{}); else finish(BoolTag<false>{}, BoolTag<true>{}, BoolTag<false>{}); }
    else if (EPI != 2 && p.lo_off > 0) { if (full) finish(BoolTag<true>{}, BoolTag<false>{}, BoolTag<true>{}); else finish(BoolTag<false>{}, BoolTag<false>{}, BoolTag<true>{}); }
    else if (full) finish(BoolTag<true>{}, BoolTag<false>{}, BoolTag<false>{});
    else finish(BoolTag<false>{}, BoolTag<false>{}, BoolTag<false>{});
}

// split-K second pass: x[m][n] += gamma[n] * (sum_s slab[s][m][n] + bias[n]), partial sums added in split order (fixed: bitwise repeatable)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, int splits, float* __restrict__ X, int ldo,
                                                             const float* __restrict__ bias, const float* __restrict__ gamma, int M, int N) {
    const int n4 = N >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)M * n4) return;
    const int m = (int)(idx / n4), c = (int)(idx - (int64_t)m * n4) * 4;
    f4 sum = *(const f4*)(slab + (size_t)m * N + c);
    for (int sp = 1; sp < splits; ++sp) sum += *(const f4*)(slab + ((size_t)sp * M + m) * N + c);
    const f4 b = bias ? *(const f4*)(bias + c) : (f4){0.f, 0.f, 0.f, 0.f};
    const f4 g = gamma ? *(const f4*)(gamma + c) : (f4){1.f, 1.f, 1.f, 1.f};
    f4 x = *(const f4*)(X + (size_t)m * ldo + c);
    x += g * (sum + b);
    *(f4*)(X + (size_t)m * ldo + c) = x;
}

template <typename T>
int launch_t(const GemmCall& c, hipStream_t stream) {
    GemmParams p;
    p.A = c.A; p.W = c.W; p.bias = c.bias; p.gamma = c.gamma; p.out = c.out;
    p.lda = c.lda; p.ldw = c.ldw; p.ldo = c.ldo;
    p.M = c.M; p.N = c.N; p.K = c.K;
    p.ka = c.ka > 0 ? c.ka : c.K;
    p.lo_off = c.epilogue == 2 ? 0 : c.out_lo_off;
    p.splits = 1;
    p.slab = nullptr;
    p.arow_max = (c.M + 127) / 128 * 128 - 1;
    p.wrow_max = (c.N + 127) / 128 * 128 - 1;
    // 256 x 256 tiles once there are enough of them (fvit_tune "gemm256_min_tiles"; 0 = never): the large Linear layers of FasterViT-4.  192 (r03-r05: a chip-filling
    // launch beside two other stream shards) -> 96 in r06: with whole-batch launches and two steps in flight a half-filling 256-tile launch beats the 128-tile one
    // (scripts/r06_calls/call19.sh, two interleaved rounds: FasterViT-4 7.05k -> 7.15k 16-bit / 3.37k -> 3.43k precise, any-res 293 -> 298 precise; 64 = 96, 32 slightly worse).
    // Bitwise the 128 x 128 tile's result (tests/test_gpu_kernels.py).
    const int t256 = ((c.M + 255) / 256) * ((c.N + 255) / 256);
    const int min256 = tune_get("gemm256_min_tiles", 96);
    const bool big = min256 > 0 && t256 >= min256 && p.K / BK >= 4;
    p.tiles_n = big ? (c.N + 255) / 256 : (c.N + BN - 1) / BN;
    p.stagger = tune_get("gemm_stagger", 0);
    p.add = c.epilogue == 2 ? c.add : nullptr;
    p.add_idx = c.add_idx;
    p.rpi = c.rows_per_image > 0 ? c.rows_per_image : 1;
    const int grid128 = ((c.M + 127) / 128) * p.tiles_n;
    // tile / workgroup shape by grid size (fvit_tune knobs for A/B):
    //   grid128 <= bm64_max : 64-row tiles (twice the workgroups; +2 % images/s on shard-sized launches), else 128-row tiles
    //   grid128 <= nw8_max  : 8 waves per workgroup (default off: no gain, see gemm_kernel)
    const int nw8_max = tune_get("gemm_nw8_max_grid", 0), bm64_max = tune_get("gemm_bm64_max_grid", 400);
    const bool nw8 = !big && grid128 <= nw8_max;
    // 64-row tiles only for short K (r03 sweep, profiles/r03_gemm_and_shard_launch_knob_sweeps.log): the long-K residual GEMMs of FasterViT-4
    // (K = 2048 .. 6272, cold weights: every K step is a memory-side round trip) want more MFMA work per step: +2.6 % images/s
    const bool small = !big && grid128 <= bm64_max && c.K <= tune_get("gemm_bm64_max_k", 1024);   // 64-row tiles
    p.tiles_m = big ? (c.M + 255) / 256 : small ? (c.M + 63) / 64 : (c.M + 127) / 128;
    int grid = p.tiles_m * p.tiles_n;
    // deterministic split-K (r04) for the residual GEMMs of small grids with long K (the carrier-token branch of FasterViT-4: 42 / 14 workgroups x 49 K
    // tiles = 48 us at 0.02-0.035 of the MFMA peak; stage 3: 221 workgroups x 98 K tiles): `splits` x the workgroups, each over 1 / splits of K, fp32
    // partials into the caller's slab, then splitk_reduce_kernel adds them in split order and applies bias / gamma / residual.  No atomics.
    // OPT-IN (fvit_tune "gemm_splitk" = 1).  Measured r04 (scripts/r04_calls/call6.sh, A/B x 2 in one box): the launches get 3 x shorter (46 -> ~15 us)
    // and the STEP gets slower -- FasterViT-4 batch 128: 6 371 / 6 390 -> 6 261 / 6 265 images/s; any-res 571 -> 576 (noise): with three stream shards
    // a 42-workgroup launch holding a sixth of the chip for 46 us costs the other shards almost nothing, 336 workgroups for 15 us do.  The measure
    // of a kernel inside the shard pipeline is CU x time, not latency (the r03 sibling-split result again, profiles/HISTORY.md).
    int splits = 1;
    if (c.epilogue == 2 && !c.add && c.splitk_slab && !big && tune_get("gemm_splitk", 0)) {
        const int nkt = p.K / BK;
        splits = std::min(std::min(8, tune_get("gemm_splitk_slots", 460) / std::max(grid, 1)), nkt / std::max(tune_get("gemm_splitk_min_ktiles", 4), 1));
        if (splits < 2 || (size_t)splits * c.M * (size_t)c.N * 4 > c.splitk_bytes || p.stagger) splits = 1;
    }
    if (splits > 1) { p.splits = splits; p.slab = c.splitk_slab; grid *= splits; }
    const double flops = 2.0 * c.M * (double)c.N * p.ka;   // algorithmic: the extra weight terms are a precision cost, not work
    double bytes = 2.0 * c.M * (double)p.ka + 2.0 * c.N * (double)c.K;  // operands once
    int kind;
    if (c.epilogue == 2) {
        bytes += 8.0 * c.M * (double)c.N;  // fp32 residual read + write
        kind = FVIT_K_GEMM_RESID;
    } else {
        bytes += 2.0 * c.M * (double)c.N;
        kind = c.epilogue == 1 ? FVIT_K_GEMM_GELU : FVIT_K_GEMM_BIAS;
    }
    ProfScope prof(kind, flops, bytes, stream);
    const bool pp = big && tune_get("gemm_pp", 1);
    // three-deep activation ring of the ping-pong tile (160 KiB of LDS, gemm_pp_kernel<.., XR = 3>; fvit_tune "gemm_pp_xring" = 2 restores the double buffer).
    // r06, scripts/r06_calls/call24.sh, two interleaved rounds on FasterViT-4 batch 128: 16-bit plan 7 119 / 7 131 -> 7 180 / 7 169 images/s (+0.7 %); the dual
    // x3 K loop of the precise plan measured equal (3 436 / 3 432 -> 3 437 / 3 442) and keeps the two-deep ring.  Bitwise the same result (same K order).
    const bool xr3 = tune_get("gemm_pp_xring", 3) == 3;
    // "dual" K loop of the x3 operand modes (r05, fvit_tune "gemm_x3_dual" = 0 restores the K-concatenated walk): [hi | hi | lo] x [hi | lo | hi] stages the
    // hi activation tile twice and the hi weight tile twice -- six operand tiles from L2 for three products; the dual tile holds 32 contraction indices of
    // both terms of both operands -- four tiles for the same three products, and 24 instead of 16 MFMAs behind every barrier
    const bool x3 = c.K == 3 * p.ka && c.lda >= 2 * p.ka && (pp || !big) && splits == 1 && !nw8 && !p.stagger && tune_get("gemm_x3_dual", 1);
    p.x3 = x3 ? 1 : 0;
    prof_note(c.epilogue == 2 ? (pp ? "gemm_pp_kernel<2> 256x256" : big ? "gemm_kernel<2> 256x256" : small ? "gemm_kernel<2> 64-row" : "gemm_kernel<2> 128-row")
                              : c.epilogue == 1 ? (pp ? "gemm_pp_kernel<1> 256x256" : big ? "gemm_kernel<1> 256x256" : small ? "gemm_kernel<1> 64-row" : "gemm_kernel<1> 128-row")
                                                : (pp ? "gemm_pp_kernel<0> 256x256" : big ? "gemm_kernel<0> 256x256" : small ? "gemm_kernel<0> 64-row" : "gemm_kernel<0> 128-row"), grid);
    if (splits > 1) prof_note(small ? "gemm_kernel<2> 64-row split-K" : "gemm_kernel<2> 128-row split-K", grid);
    // (measured r01 / r03: 3- and 4-deep rings with counted vmcnt gave no gain over 2 stages at 64..392 workgroups, and their instances missed the 2-waves-per-SIMD
    // register target; the opt-in knobs "gemm_3stage_max_grid" / "gemm_ring" and those instances were removed in r06: git history, profiles/HISTORY.md)
#define FVIT_GEMM(E, NS, MI_, NW_) hipLaunchKernelGGL((gemm_kernel<T, E, NS, MI_, NW_>), dim3(grid), dim3(64 * NW_), 0, stream, p)
#define FVIT_GEMM256(E) hipLaunchKernelGGL((gemm_kernel<T, E, 2, 4, 8, 8>), dim3(grid), dim3(512), 0, stream, p)
#define FVIT_GEMM_E(NS, MI_, NW_) \
    switch (splits > 1 ? 3 : c.epilogue) { case 0: FVIT_GEMM(0, NS, MI_, NW_); break; case 1: FVIT_GEMM(1, NS, MI_, NW_); break; case 3: FVIT_GEMM(3, NS, MI_, NW_); break; \
                                           default: FVIT_GEMM(2, NS, MI_, NW_); break; }
    // the ping-pong form of the 256 x 256 tile (default since r03: 8-25 % faster than the 2-stage form on every shape that selects the tile,
    // bitwise the same result; FasterViT-4 batch 128 + 0.7 %, any-res + 0.6 % end to end -- profiles/r03_gemm_ping_pong_256_tile.log)
    if (x3) {
#define FVIT_GEMM_X3(E, MI_) hipLaunchKernelGGL((gemm_kernel<T, E, 2, MI_, 4, 4, true>), dim3(grid), dim3(256), 0, stream, p)
        if (pp) {
            switch (c.epilogue) {
                case 0: hipLaunchKernelGGL((gemm_pp_kernel<T, 0, true>), dim3(grid), dim3(512), 0, stream, p); break;
                case 1: hipLaunchKernelGGL((gemm_pp_kernel<T, 1, true>), dim3(grid), dim3(512), 0, stream, p); break;
                default: hipLaunchKernelGGL((gemm_pp_kernel<T, 2, true>), dim3(grid), dim3(512), 0, stream, p); break;
            }
        } else if (small) {
            switch (c.epilogue) { case 0: FVIT_GEMM_X3(0, 2); break; case 1: FVIT_GEMM_X3(1, 2); break; default: FVIT_GEMM_X3(2, 2); break; }
        } else {
            switch (c.epilogue) { case 0: FVIT_GEMM_X3(0, 4); break; case 1: FVIT_GEMM_X3(1, 4); break; default: FVIT_GEMM_X3(2, 4); break; }
        }
#undef FVIT_GEMM_X3
    } else if (pp && xr3) {
        switch (c.epilogue) {
            case 0: hipLaunchKernelGGL((gemm_pp_kernel<T, 0, false, 3>), dim3(grid), dim3(512), 0, stream, p); break;
            case 1: hipLaunchKernelGGL((gemm_pp_kernel<T, 1, false, 3>), dim3(grid), dim3(512), 0, stream, p); break;
            default: hipLaunchKernelGGL((gemm_pp_kernel<T, 2, false, 3>), dim3(grid), dim3(512), 0, stream, p); break;
        }
    } else if (pp) {
        switch (c.epilogue) {
            case 0: hipLaunchKernelGGL((gemm_pp_kernel<T, 0>), dim3(grid), dim3(512), 0, stream, p); break;
            case 1: hipLaunchKernelGGL((gemm_pp_kernel<T, 1>), dim3(grid), dim3(512), 0, stream, p); break;
            default: hipLaunchKernelGGL((gemm_pp_kernel<T, 2>), dim3(grid), dim3(512), 0, stream, p); break;
        }
    }
    else if (big) { switch (c.epilogue) { case 0: FVIT_GEMM256(0); break; case 1: FVIT_GEMM256(1); break; default: FVIT_GEMM256(2); break; } }
    else if (nw8 && small) { FVIT_GEMM_E(2, 1, 8) }        // 64 rows = 4 waves along M x 16 rows
    else if (nw8) { FVIT_GEMM_E(2, 2, 8) }            // 128 rows = 4 waves x 32 rows
    else if (small) { FVIT_GEMM_E(2, 2, 4) }          // 64 rows = 2 waves x 32 rows
    else { FVIT_GEMM_E(2, 4, 4) }
#undef FVIT_GEMM_E
#undef FVIT_GEMM256
#undef FVIT_GEMM
    if (splits > 1) {
        const int64_t n = (int64_t)c.M * (c.N / 4);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float*)c.splitk_slab, splits, (float*)c.out, c.ldo,
                           c.bias, c.gamma, c.M, c.N);
    }
    return check_launch("gemm_kernel");
}

}  // namespace

int launch_gemm(const GemmCall& c, hipStream_t stream) {
    if (c.M <= 0 || c.N <= 0 || c.K <= 0 || (c.K % BK) != 0 || (c.N % 16) != 0 || (c.lda % 8) != 0 ||
        (c.ldw % 8) != 0 || (c.ldo % (c.epilogue == 2 ? 4 : 8)) != 0 || c.ldw < c.K ||
        (c.ka > 0 ? (c.ka % BK != 0 || (c.K != c.ka && c.K != 2 * c.ka && c.K != 3 * c.ka) || c.lda < (c.K == 3 * c.ka ? 2 * c.ka : c.ka)) : c.lda < c.K) ||
        c.out_lo_off < 0 || (c.out_lo_off % 8) != 0 || (c.out_lo_off > 0 && (c.epilogue == 2 || c.out_lo_off < c.N || c.ldo < c.out_lo_off + c.N))) {
        set_error("gemm: unsupported shape M=%d N=%d K=%d lda=%d ldw=%d ldo=%d (need K%%64==0, N%%16==0)", c.M, c.N,
                  c.K, c.lda, c.ldw, c.ldo);
        return FVIT_EINVAL;
    }
    if (c.dtype == FVIT_F16) return launch_t<_Float16>(c, stream);
    if (c.dtype == FVIT_BF16) return launch_t<__bf16>(c, stream);
    set_error("gemm: operand dtype %d not supported", c.dtype);
    return FVIT_EINVAL;
}

}  // namespace fvit
