// fvit_attnblk2.hip -- the fused attention sub-block of HAT for C = 256 / 8 heads / head_dim 32, windows of 49..64 tokens (stage 2 of
// FasterViT-0), re-cut in r06 so that a WAVE owns a (window, head) instead of 16 rows (gfx950):
//
//   x_out = x_in + gamma * proj( softmax( q k^T * scale + bias ) v ),   [q|k|v] = qkv( LayerNorm(x_in) )      (AR:671-696 / FV:665-690)
//
// Same contract, same packed weights (w_qkv_frag / w_proj_frag, include/fvit_hip.h) and the same rounding points as attnblk_kernel
// (fvit_attnblk.hip).  Why a second cut: in attnblk_kernel a wave owns 16 rows, so EVERY 16x16x32 MFMA of qkv / proj needs its own 1-KiB
// weight fragment from LDS (1 KiB of LDS read per MFMA, at the 256 B/clk ceiling of the CU before any DMA write), the per-head weight
// slices arrive by LDS-DMA behind two workgroup barriers per head, and k / v cross waves through LDS: the launch spent 3.2 us per head for
// 0.48 us of MFMA (profiles/r06_sq_counters_by_kernel.json: MFMA busy 0.10, wait 0.47).  Here:
//
//   workgroup = 4 waves = NWIN windows.  NWIN = 2 (128 padded rows): one workgroup per CU, one wave per SIMD (up to 512 registers), every weight
//             fragment feeds 8 MFMAs.  NWIN = 1 (64 rows): <= 256 registers and 72 KiB of LDS, TWO workgroups per CU whose phases interleave
//             (and leave room for the other stream shard's kernels), every weight fragment feeds 4 MFMAs.  Described for NWIN = 2:
//   prologue  wave w gathers channels [64w, 64w + 64) of all 128 rows (the layout of its proj accumulator), LayerNorm statistics meet
//             across the four waves through LDS (fixed order), the normalised rows are parked in LDS in MFMA fragment order (64 KiB).
//   phase 1   wave w runs heads 2w, 2w + 1 of BOTH windows: the head's q / k / v weights stream from L2 straight into registers (a 4-deep
//             ring, no LDS, no barrier), every weight fragment feeds 8 MFMAs (8 row blocks), every activation fragment from LDS feeds 6;
//             q^T, k^T, v accumulators of a whole window are in ONE wave, so scores / softmax / P.V need no exchange at all
//             (transposed chaining as in attnblk_kernel: the accumulators are the next MFMA's operands); the normalised O^T fragment
//             goes to LDS in B-fragment order (64 KiB).
//   phase 2   ONE barrier, then wave w computes output channels [64w, 64w + 64) of all 128 rows: proj weights (32 KiB per wave) straight
//             from L2, O fragments from LDS (each feeds 4 MFMAs), the residual rows re-read from L2 / MALL under the MFMAs.
//   No barrier and no LDS-DMA inside the head loop; 1152 MFMAs per wave between three workgroup barriers.
// In place (x_out == srcA, as the stage runner calls it): phase 2 gathers the residual rows a SECOND time, so a row's source must be the row itself or a row of srcB
// (true for the stage tables: local rows map to themselves, carrier rows come from the carrier buffer) -- the same contract as the C = 512 instances of
// attnblk_kernel / winblk_kernel, which re-gather in their epilogues too.
#include "fvit_common.h"
#include <type_traits>

namespace fvit {

namespace {

struct AttnBlk2Params {
    const float* srcA;
    const float* srcB;
    const int32_t* src_idx;
    const int32_t* add_idx;
    const float* add;
    const float* ln_w;
    const float* ln_b;
    float eps;
    int rowsA, rowsB, rows_per_image;
    const void* wqkv_f;   // op16 [8][6][8][64][8]
    const float* bqkv;    // f32  [8][96]
    const void* wproj_f;  // op16 [8][16][64][8]
    const float* bproj;   // f32  [256]
    const float* gamma;   // f32  [256] or null
    const float* bias;    // f32  [8][64][64]
    float* x_out;         // f32  [rows][256]
    int nwin, S;
    float scale;
    unsigned long long* ts;   // TS instance only (fvit_debug_attn_block_timeline): s_memtime stamps [workgroup][wave][16]: 0 entry, 1 row table + small tables,
                              // 2 rows gathered, 3 LayerNorm fragments published, 4 / 6 q k v of head 0 / 1, 5 / 7 attention of head 0 / 1, 8 O fragments of every
                              // wave visible, 9 proj done, 10 end (stores drained)
};

template <typename T, int NWIN, bool TS = false>
__global__ __launch_bounds__(256, NWIN == 1 ? 2 : 1) void attnblk2_kernel(AttnBlk2Params p) {
#define FVIT_AB2_STAMP(k) if constexpr (TS) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if ((threadIdx.x & 63) == 0) p.ts[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (k)] = __builtin_amdgcn_s_memtime(); }
    FVIT_AB2_STAMP(0)
    typedef typename Op16<T>::v8 v8;
    constexpr int C = 256, KK = 8, HEADS = 8, SP = 64;
    constexpr int NRB = 4 * NWIN;                           // 16-row blocks per workgroup
    constexpr int ROWS = 16 * NRB;
    constexpr int QKV_BYTES = 6 * KK * 1024, PROJ_BYTES = 16 * 1024;
    constexpr int OFF_XN = 0;                               // [rb][kk] fragments of LayerNorm(x): 32 KiB per window
    constexpr int OFF_AO = OFF_XN + NRB * KK * 1024;        // [rb][head] fragments of the attention output: 32 KiB per window
    constexpr int OFF_ST = OFF_AO + NRB * HEADS * 1024;     // LayerNorm partials f32 [2][rows][4 waves]
    constexpr int OFF_RI = OFF_ST + 2 * ROWS * 4 * 4;       // per row: source offset (floats; bit 31 = srcB), add offset (floats; -1 = none)
    constexpr int OFF_BQ = OFF_RI + ROWS * 2 * 4;           // qkv bias [8][96]
    constexpr int OFF_BP = OFF_BQ + HEADS * 96 * 4;         // proj bias, gamma
    __shared__ __attribute__((aligned(16))) char smem[OFF_BP + 2 * C * 4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, s = lane & 15;
    const int lane16 = lane * 16;
    float* st0 = (float*)(smem + OFF_ST);
    float* st1 = st0 + ROWS * 4;
    uint32_t* rinfo = (uint32_t*)(smem + OFF_RI);
    float* bqs = (float*)(smem + OFF_BQ);
    float* bps = (float*)(smem + OFF_BP);
    float* gms = bps + C;

    // ---- small tables; row bookkeeping: every lane resolves its own rows (32-bit arithmetic: the launcher checks the range), the gather below
    // starts without a barrier; wave 0 parks the offsets in LDS for the second gather of phase 2 ----
    for (int i = tid; i < HEADS * 96; i += 256) bqs[i] = p.bqkv[i];
    {
        bps[tid] = p.bproj[tid];
        gms[tid] = p.gamma ? p.gamma[tid] : 1.0f;
    }
    const int ch0 = wave * 64 + g * 16;   // this lane's 16 channels: ch0 .. ch0 + 15 (prologue rows and proj accumulator alike)
    auto load_rows = [&](f4 (&xv)[NRB][4], const uint32_t (&so)[NRB], const uint32_t (&ao)[NRB]) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const float* src = ((so[rb] & 0x80000000u) ? p.srcB : p.srcA) + (so[rb] & 0x7fffffffu) + ch0;
#pragma unroll
            for (int q = 0; q < 4; ++q) xv[rb][q] = *(const f4*)(src + q * 4);
        }
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            if (ao[rb] != 0xffffffffu) {
                const float* ap = p.add + ao[rb] + ch0;
#pragma unroll
                for (int q = 0; q < 4; ++q) xv[rb][q] += *(const f4*)(ap + q * 4);
            }
        }
    };

    // ---- prologue: gather, LayerNorm (statistics over the four waves' channel quarters), fragments to LDS ----
    {
        uint32_t so[NRB], ao[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const int wi = rb >> 2, tok = (rb & 3) * 16 + s;
            const int win = blockIdx.x * NWIN + wi;
            const uint32_t row = (uint32_t)(win < p.nwin ? win : p.nwin - 1) * (uint32_t)p.S + (uint32_t)(tok < p.S ? tok : p.S - 1);   // clamped: always a real row
            const uint32_t b = row / (uint32_t)p.rows_per_image, pr = row - b * (uint32_t)p.rows_per_image;
            ao[rb] = 0xffffffffu;
            if (p.src_idx) {
                const int si = p.src_idx[pr];
                so[rb] = si >= 0 ? (b * (uint32_t)p.rowsA + (uint32_t)si) * C : (0x80000000u | ((b * (uint32_t)p.rowsB + (uint32_t)(-si - 1)) * C));
            } else {
                so[rb] = row * C;
            }
            if (p.add) {
                const int ai = p.add_idx ? p.add_idx[pr] : (int)pr;
                if (ai >= 0) ao[rb] = (uint32_t)ai * C;
            }
        }
        f4 xv[NRB][4];
        load_rows(xv, so, ao);
        if (wave == 0 && g == 0) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                rinfo[(rb * 16 + s) * 2] = so[rb];
                rinfo[(rb * 16 + s) * 2 + 1] = ao[rb];
            }
        }
        FVIT_AB2_STAMP(2)
        // LayerNorm statistics: every wave reduces its 64 channels of a row to (mean, M2) locally, the four quarters meet in LDS and are combined
        // by Chan's formula in a fixed order: ONE barrier, and no E[x^2] - mean^2 cancellation
        float mloc[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) sum += (xv[rb][q][0] + xv[rb][q][1]) + (xv[rb][q][2] + xv[rb][q][3]);
            sum = sum_xor32(sum_xor16(sum));
            mloc[rb] = sum * (1.0f / 64.0f);
        }
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            float sq = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f4 d = xv[rb][q] - mloc[rb];
                sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
            sq = sum_xor32(sum_xor16(sq));
            if (g == 0) {
                st0[(rb * 16 + s) * 4 + wave] = mloc[rb];
                st1[(rb * 16 + s) * 4 + wave] = sq;
            }
        }
        __syncthreads();
        float mean[NRB], rstd[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const f4 pm = *(const f4*)(st0 + (rb * 16 + s) * 4);
            const f4 pq = *(const f4*)(st1 + (rb * 16 + s) * 4);
            const float m = ((pm[0] + pm[1]) + (pm[2] + pm[3])) * 0.25f;
            const f4 dm = pm - m;
            const float m2 = ((pq[0] + pq[1]) + (pq[2] + pq[3])) + 64.0f * ((dm[0] * dm[0] + dm[1] * dm[1]) + (dm[2] * dm[2] + dm[3] * dm[3]));
            mean[rb] = m;
            rstd[rb] = rsqrtf(m2 * (1.0f / C) + p.eps);
        }
        f4 lw[4], lb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            lw[q] = *(const f4*)(p.ln_w + ch0 + q * 4);
            lb[q] = *(const f4*)(p.ln_b + ch0 + q * 4);
        }
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            // k slot kk * 32 + 8g + e carries channel (kk >> 1) * 64 + 16g + (kk & 1) * 8 + e (hat_runtime.kslot_channels): kk = 2 wave + j
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                v8 o;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        o[h2 * 4 + r] = sat16<T>((xv[rb][2 * j + h2][r] - mean[rb]) * rstd[rb] * lw[2 * j + h2][r] + lb[2 * j + h2][r]);
                *(v8*)(smem + OFF_XN + (rb * KK + 2 * wave + j) * 1024 + lane16) = o;
            }
        }
    }
    __syncthreads();
    FVIT_AB2_STAMP(3)

    // ---- phase 1: heads 2 wave, 2 wave + 1 of the workgroup's windows ----
    const char* __restrict__ Wq = (const char*)p.wqkv_f;
    const char* xn_l = smem + OFF_XN + lane16;
    char* ao_l = smem + OFF_AO + lane16;
    // the q / k / v weight fragments of a head are requested RING - 1 k steps ahead (L2 -> registers), across the head boundary too; the
    // sched_barriers keep hipcc from sinking the requests down to their first use (r06 ISA check: without them every k step began with
    // "global_load, s_waitcnt vmcnt(1), v_mfma": one exposed L2 round trip per step).  NWIN = 2: 4 slots and a rolled head loop (8 % 4 == 0: the
    // slots repeat per head); NWIN = 1: 3 slots (256-register budget), both heads unrolled (slot of step t = t % 3 over the 16 steps).
    constexpr int RING = NWIN == 2 ? 4 : 3;
    v8 wr[RING][6];
    {
        const char* wq0 = Wq + (size_t)(2 * wave) * QKV_BYTES + lane16;
#pragma unroll
        for (int t = 0; t < RING - 1; ++t)
#pragma unroll
            for (int ub = 0; ub < 6; ++ub) wr[t][ub] = *(const v8*)(wq0 + (ub * KK + t) * 1024);
    }
    v8 xfr[2][NRB];   // activation fragments of the current / next k step (LDS reads one step ahead)
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) xfr[0][rb] = *(const v8*)(xn_l + (rb * KK + 0) * 1024);

    auto do_head = [&](const int hh, auto SLOT0) {
        constexpr int slot0 = decltype(SLOT0)::value;           // ring slot of this head's k step 0
        const int h = 2 * wave + hh;
        const char* wq = Wq + (size_t)h * QKV_BYTES + lane16;   // fragment (ub, kk) at (ub * KK + kk) * 1024
        const char* wq_next = hh == 0 ? wq + QKV_BYTES : wq;    // (last head: a harmless re-read of its own first fragments)
        // bias rows of the head's 64 x 64 table, requested one attention step ahead; the first ones here, landing under the k loop
        constexpr int NBS = NWIN == 2 ? 1 : 2;                  // query blocks per attention step (NWIN = 2: one query block of both windows)
        const float* bias_h = p.bias + (size_t)h * SP * SP + (size_t)s * SP + g * 4;
        f4 bzn[NBS][4];
#pragma unroll
        for (int j = 0; j < NBS; ++j)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) bzn[j][kb] = *(const f4*)(bias_h + (j * 16) * SP + kb * 16);
        f4 aq[2][NRB], ak[2][NRB], av[NRB][2];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            aq[0][rb] = aq[1][rb] = ak[0][rb] = ak[1][rb] = av[rb][0] = av[rb][1] = (f4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            __builtin_amdgcn_sched_barrier(0);
            {
                const int tk = kk + RING - 1;
                const char* src = tk < KK ? wq + tk * 1024 : wq_next + (tk - KK) * 1024;
#pragma unroll
                for (int ub = 0; ub < 6; ++ub) wr[(slot0 + tk) % RING][ub] = *(const v8*)(src + ub * KK * 1024);
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) xfr[(kk + 1) & 1][rb] = *(const v8*)(xn_l + (rb * KK + ((kk + 1) & (KK - 1))) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            const v8* w = wr[(slot0 + kk) % RING];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                const v8 xf = xfr[kk & 1][rb];
                aq[0][rb] = Op16<T>::mfma(w[0], xf, aq[0][rb]);      // q^T, k^T: weights are the A operand => D[dim][token]
                aq[1][rb] = Op16<T>::mfma(w[1], xf, aq[1][rb]);
                ak[0][rb] = Op16<T>::mfma(w[2], xf, ak[0][rb]);
                ak[1][rb] = Op16<T>::mfma(w[3], xf, ak[1][rb]);
                av[rb][0] = Op16<T>::mfma(xf, w[4], av[rb][0]);      // v: activations are the A operand => D[token][dim]
                av[rb][1] = Op16<T>::mfma(xf, w[5], av[rb][1]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TS) { asm volatile("s_nop 0" ::"v"(aq[0][0]), "v"(av[NRB - 1][1]) : "memory"); }
        FVIT_AB2_STAMP(4 + 2 * hh)
        const float* bq = bqs + h * 96;
        const f4 bq0 = *(const f4*)(bq + g * 4), bq1 = *(const f4*)(bq + 16 + g * 4);
        const f4 bk0 = *(const f4*)(bq + 32 + g * 4), bk1 = *(const f4*)(bq + 48 + g * 4);
        const float bv0 = bq[64 + s], bv1 = bq[80 + s];
        // k and v of every window as MFMA operands (transposed chaining: the accumulators ARE the fragments)
        v8 kf[NWIN][4], vf[NWIN][2][2];
#pragma unroll
        for (int wi = 0; wi < NWIN; ++wi) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    kf[wi][kb][r] = sat16<T>(ak[0][wi * 4 + kb][r] + bk0[r]);
                    kf[wi][kb][4 + r] = sat16<T>(ak[1][wi * 4 + kb][r] + bk1[r]);
                }
#pragma unroll
            for (int k32 = 0; k32 < 2; ++k32)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vf[wi][0][k32][r] = sat16<T>(av[wi * 4 + 2 * k32][0][r] + bv0);
                    vf[wi][0][k32][4 + r] = sat16<T>(av[wi * 4 + 2 * k32 + 1][0][r] + bv0);
                    vf[wi][1][k32][r] = sat16<T>(av[wi * 4 + 2 * k32][1][r] + bv1);
                    vf[wi][1][k32][4 + r] = sat16<T>(av[wi * 4 + 2 * k32 + 1][1][r] + bv1);
                }
        }
        // attention, TWO (window, query block) items side by side, stage by stage: the only thing that can fill the latency of an item's dependent chain
        // (MFMA -> max -> exp -> sum -> MFMA) inside a wave is the other item's chain.  NWIN = 2: query block `it` of both windows; NWIN = 1: query blocks 2 it, 2 it + 1.
        constexpr int NIT = NWIN == 2 ? 4 : 2;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            f4 bz[NBS][4];
#pragma unroll
            for (int j = 0; j < NBS; ++j)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) bz[j][kb] = bzn[j][kb];
            {   // next step's bias rows (the last request of a head is unused)
                const int nit = (it + 1) % NIT;
#pragma unroll
                for (int j = 0; j < NBS; ++j)
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) bzn[j][kb] = *(const f4*)(bias_h + ((NWIN == 2 ? nit : 2 * nit + j) * 16) * SP + kb * 16);
            }
            f4 sc[2][4];
            float mx[2], sum[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int wi = NWIN == 2 ? j : 0, qb = NWIN == 2 ? it : 2 * it + j;
                v8 qf;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    qf[r] = sat16<T>(aq[0][wi * 4 + qb][r] + bq0[r]);
                    qf[4 + r] = sat16<T>(aq[1][wi * 4 + qb][r] + bq1[r]);
                }
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) sc[j][kb] = Op16<T>::mfma(kf[wi][kb], qf, (f4){0.f, 0.f, 0.f, 0.f});   // S^T[key 4g + r of block kb][query s]
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float m4[4];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[j][kb][r] = sc[j][kb][r] * p.scale + bz[NWIN == 2 ? 0 : j][kb][r];
                    m4[kb] = fmaxf(fmaxf(sc[j][kb][0], sc[j][kb][1]), fmaxf(sc[j][kb][2], sc[j][kb][3]));
                }
                mx[j] = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) mx[j] = max_xor32(max_xor16(mx[j]));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float s4[4];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[j][kb][r] = __expf(sc[j][kb][r] - mx[j]);
                    s4[kb] = (sc[j][kb][0] + sc[j][kb][1]) + (sc[j][kb][2] + sc[j][kb][3]);
                }
                sum[j] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) sum[j] = sum_xor32(sum_xor16(sum[j]));
            f4 o0[2], o1[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int wi = NWIN == 2 ? j : 0;
                o0[j] = o1[j] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k32 = 0; k32 < 2; ++k32) {
                    v8 pf;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pf[r] = (T)sc[j][2 * k32][r];
                        pf[4 + r] = (T)sc[j][2 * k32 + 1][r];
                    }
                    o0[j] = Op16<T>::mfma(vf[wi][0][k32], pf, o0[j]);   // O^T[dim 4g + r][query s]
                    o1[j] = Op16<T>::mfma(vf[wi][1][k32], pf, o1[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int wi = NWIN == 2 ? j : 0, qb = NWIN == 2 ? it : 2 * it + j;
                const float inv = 1.0f / sum[j];
                v8 of;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    of[r] = sat16<T>(o0[j][r] * inv);
                    of[4 + r] = sat16<T>(o1[j][r] * inv);
                }
                *(v8*)(ao_l + ((wi * 4 + qb) * HEADS + h) * 1024) = of;
            }
        }
        FVIT_AB2_STAMP(5 + 2 * hh)
    };
    if constexpr (NWIN == 2) {
#pragma unroll 1
        for (int hh = 0; hh < 2; ++hh) do_head(hh, std::integral_constant<int, 0>());
    } else {
        do_head(0, std::integral_constant<int, 0>());
        do_head(1, std::integral_constant<int, KK % RING>());
    }

    // ---- phase 2: out^T[channels 64 wave ..][rows] = sum over heads of Wproj[:, head] . O^T ----
    const char* wp = (const char*)p.wproj_f + (size_t)(4 * wave) * 1024 + lane16;   // fragment (head, cb) at head * PROJ_BYTES + cb * 1024
    constexpr int PH = NWIN == 2 ? HEADS : HEADS / 2;   // heads per batch of proj weight fragments (NWIN = 1: two batches of 64 registers)
    v8 pw[PH][4];
#pragma unroll
    for (int h = 0; h < PH; ++h)
#pragma unroll
        for (int cbl = 0; cbl < 4; ++cbl) pw[h][cbl] = *(const v8*)(wp + (size_t)h * PROJ_BYTES + cbl * 1024);
    __syncthreads();   // every wave's O fragments are in LDS
    FVIT_AB2_STAMP(8)
    f4 xv[NRB][4];
    {
        uint32_t so[NRB], ao[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            so[rb] = rinfo[(rb * 16 + s) * 2];
            ao[rb] = rinfo[(rb * 16 + s) * 2 + 1];
        }
        load_rows(xv, so, ao);
    }   // the residual rows again (L2 / MALL), landing under the MFMAs below
    f4 oacc[4][NRB];
#pragma unroll
    for (int cbl = 0; cbl < 4; ++cbl)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) oacc[cbl][rb] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int hb = 0; hb < HEADS; hb += PH) {
        if (hb > 0) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < PH; ++h)
#pragma unroll
                for (int cbl = 0; cbl < 4; ++cbl) pw[h][cbl] = *(const v8*)(wp + (size_t)(hb + h) * PROJ_BYTES + cbl * 1024);
        }
#pragma unroll
        for (int h = 0; h < PH; ++h) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                const v8 of = *(const v8*)(ao_l + (rb * HEADS + hb + h) * 1024);
#pragma unroll
                for (int cbl = 0; cbl < 4; ++cbl) oacc[cbl][rb] = Op16<T>::mfma(pw[h][cbl], of, oacc[cbl][rb]);
            }
        }
    }

    if constexpr (TS) { asm volatile("s_nop 0" ::"v"(oacc[0][0]), "v"(oacc[3][NRB - 1]) : "memory"); }
    FVIT_AB2_STAMP(9)
    // ---- epilogue: x_out[row][ch0 + 4 cbl + r] = x_in + gamma * (out + bproj) ----
    f4 bv[4], gv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bv[q] = *(const f4*)(bps + ch0 + q * 4);
        gv[q] = *(const f4*)(gms + ch0 + q * 4);
    }
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
        const int wi = rb >> 2, tok = (rb & 3) * 16 + s;
        const int win = blockIdx.x * NWIN + wi;
        if (win < p.nwin && tok < p.S) {
            float* px = p.x_out + ((size_t)win * p.S + tok) * C + ch0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f4 x = xv[rb][q];
                const f4 a = oacc[q][rb];
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] += gv[q][r] * (a[r] + bv[q][r]);
                *(f4*)(px + q * 4) = x;
            }
        }
    }
    FVIT_AB2_STAMP(10)
#undef FVIT_AB2_STAMP
}

}  // namespace

bool attnblk2_supported(int C, int heads, int S) { return C == 256 && heads == 8 && S > 48 && S <= 64; }

int launch_attnblk2(const AttnBlkCall& c, hipStream_t stream) {
    if (!attnblk2_supported(c.C, c.heads, c.S) || c.terms != 1 || c.nwin <= 0 || !c.wqkv_f || !c.wproj_f || !c.x_out || !c.bias) {
        set_error("attn_block (wave-per-head form): unsupported arguments C=%d heads=%d S=%d nwin=%d terms=%d", c.C, c.heads, c.S, c.nwin, c.terms);
        return FVIT_EINVAL;
    }
    const int rpi = c.rows_per_image > 0 ? c.rows_per_image : 1;
    // 32-bit float offsets inside the kernel (bit 31 = "from srcB")
    const double maxoff = ((double)c.nwin * c.S / rpi + 1.0) * (double)(c.rowsA > c.rowsB ? c.rowsA : c.rowsB) * c.C;
    if (maxoff >= 2147483648.0 || (double)c.nwin * c.S * c.C >= 2147483648.0) {
        set_error("attn_block (wave-per-head form): %d windows exceed the 32-bit row offsets", c.nwin);
        return FVIT_EINVAL;
    }
    AttnBlk2Params p;
    p.srcA = c.srcA; p.srcB = c.srcB; p.src_idx = c.src_idx; p.add_idx = c.add_idx; p.add = c.add; p.ln_w = c.ln_w; p.ln_b = c.ln_b;
    p.eps = c.eps; p.rowsA = c.rowsA; p.rowsB = c.rowsB; p.rows_per_image = rpi;
    p.wqkv_f = c.wqkv_f; p.bqkv = c.bqkv; p.wproj_f = c.wproj_f; p.bproj = c.bproj; p.gamma = c.gamma; p.bias = c.bias;
    p.x_out = c.x_out; p.nwin = c.nwin; p.S = c.S; p.scale = c.scale; p.ts = (unsigned long long*)c.ts;
    // windows per workgroup: 1 (default: 256 registers / 72 KiB of LDS, two workgroups per CU) or 2 (fvit_tune "ab2_nwin": one 138-KiB workgroup per CU)
    const int nwin_wg = tune_get("ab2_nwin", 1) == 2 ? 2 : 1;
    const int grid = (c.nwin + nwin_wg - 1) / nwin_wg;
    if (c.ts) {   // timeline instance (diagnosis entry point): fp16 only
        if (c.dtype != FVIT_F16) { set_error("attn_block timeline: fp16 only"); return FVIT_EINVAL; }
        if (nwin_wg == 2) hipLaunchKernelGGL((attnblk2_kernel<_Float16, 2, true>), dim3(grid), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((attnblk2_kernel<_Float16, 1, true>), dim3(grid), dim3(256), 0, stream, p);
        return check_launch("attnblk2_kernel");
    }
    prof_note(nwin_wg == 2 ? "attnblk2_kernel<256,S64,2 windows>" : "attnblk2_kernel<256,S64>", grid);
    if (c.dtype == FVIT_F16) {
        if (nwin_wg == 2) hipLaunchKernelGGL((attnblk2_kernel<_Float16, 2>), dim3(grid), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((attnblk2_kernel<_Float16, 1>), dim3(grid), dim3(256), 0, stream, p);
    } else if (c.dtype == FVIT_BF16) {
        if (nwin_wg == 2) hipLaunchKernelGGL((attnblk2_kernel<__bf16, 2>), dim3(grid), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((attnblk2_kernel<__bf16, 1>), dim3(grid), dim3(256), 0, stream, p);
    } else { set_error("attn_block: operand dtype %d not supported", c.dtype); return FVIT_EINVAL; }
    return check_launch("attnblk2_kernel");
}

}  // namespace fvit
