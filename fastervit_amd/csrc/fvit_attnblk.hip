// fvit_attnblk.hip -- fused attention sub-block of HAT for head_dim 32, C = 256 / 8 heads or C = 512 / 16 heads (gfx950):
//
//   x_out = x_in + gamma * proj( softmax( q k^T * scale + bias ) v ),   [q|k|v] = qkv( LayerNorm(x_in) )
//   x_in  = gathered source row (+ position embedding row)              (AR:671-696 / FV:665-690)
//
// ONE kernel instead of gather-LayerNorm + qkv GEMM + attention + proj GEMM: the unfused path moves the
// normalised activations, the 3C-wide qkv tensor and the attention output through HBM and reads / writes the
// fp32 residual stream twice; here a workgroup reads its rows once and writes them once.
//
// Work split: a workgroup of 8 wave64 owns 8 row blocks of 16 tokens = 8 / NRB windows (NRB = 4: windows of up
// to 64 tokens, the 53-token HAT windows; NRB = 1: windows of up to 16 tokens, the carrier-token attention).
// Each wave keeps LayerNorm(x) of its 16 rows as MFMA fragments (32 VGPR) and the proj accumulator of its 16
// rows x 256 channels (64 VGPR) for the whole kernel, and walks the heads:
//   P1  q^T, k^T (A = weight fragments, B = x fragments) and v (A = x fragments, B = weight fragments): 48 MFMAs.
//       In these orientations the accumulators ARE the operands the next MFMAs need ("transposed chaining"):
//       q^T -> B operand of S^T = K.Q^T, k^T -> A operand of S^T, v -> half an A operand of O^T = V^T.P^T.
//   X   k and v fragments are exchanged between the NRB waves of a window through LDS (2 x 1 KiB per wave).
//   P2  S^T = K.Q^T * scale + bias (folded table, staged in LDS), softmax down the key axis (in-lane + 2 xor
//       shuffles), O^T = V^T.P^T: 8 MFMAs; the normalised O^T accumulator is the B operand of
//   P3  out^T += Wproj[:, head] . O^T: 16 MFMAs.
// Per head a workgroup needs 48 KiB of qkv weights, 16 KiB of proj weights and the head's bias table; all are
// pre-packed in MFMA fragment order (HBM image = LDS image) and arrive by 16-byte global_load_lds: the proj /
// bias slices of head h while P1(h) runs, the qkv slice of head h+1 while P2/P3(h) run.  Two barriers per head,
// no ordinary global loads inside the head loop (they would queue behind the DMA in the in-order vmcnt counter).
#include "fvit_common.h"

namespace fvit {

namespace {

struct AttnBlkParams {
    // gather + LayerNorm (same semantics as fvit_gather_layernorm)
    const float* srcA;
    const float* srcB;
    const int32_t* src_idx;
    const int32_t* add_idx;
    const float* add;
    const float* ln_w;
    const float* ln_b;
    float eps;
    int rowsA, rowsB, rows_per_image;
    // weights (fragment-major) and folded bias
    const void* wqkv_f;   // op16 [heads][6][C/32][64][8]
    const float* bqkv;    // f32  [heads][96]  (q 32, k 32, v 32)
    const void* wproj_f;  // op16 [heads][C/16][64][8]
    const float* bproj;   // f32  [C]
    const float* gamma;   // f32  [C] or null
    const float* bias;    // f32  [heads][SP][SP]
    float* x_out;         // f32  [rows][C]
    int nwin, S, heads;
    float scale;
    int ablate;  // timing experiments only (wrong results): 1 = no weight DMA inside the head loop (bits 2 / 4 / 8 -- skip P1 / P2 / P3 --
                 // were removed in r02: runtime branches around the phases kept the compiler from scheduling across them)
    unsigned long long* ts;   // TS instance only (fvit_debug_attn_block_timeline): s_memtime stamps [workgroup][wave][16]: 0 entry, 1 first weight
                              // slice requested + rows gathered, 2 LayerNorm done, 3 .. 10 end of head 0 .. 7, 14 head loop done, 15 end
    int stagger; // 1: workgroup b walks the heads starting at head (b / 8) % heads (b % 8 = XCD, observed): the workgroups of an XCD
                 // stream different weight slices at any time, so a slice is fetched from the memory side once per XCD and found in
                 // L2 by the other workgroups (in lockstep they all wait on the same outstanding miss)
};

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// NRB: row blocks (of 16 tokens) per window: 4 (S <= 64) or 1 (S <= 16); NW: waves per workgroup (8 or 4);
// BIAS_LDS: stage the head's bias table in LDS (else read it from L2 with ordinary loads issued ahead of the DMA)
// CC: channels (256: stage 2 of FasterViT-0, 8 heads; 512: stage 3, 16 heads -- one 142-KiB workgroup per CU, one wave per SIMD,
//     so the 128 + 128 + 64 VGPRs of input rows, proj accumulator and LayerNorm fragments fit without spilling)
// DBQ: the per-head qkv weight slice is double buffered: the slice of head h + 1 is requested at the TOP of head h (behind this head's
//      proj / bias pieces) and has the whole head to land; barrier B then waits with a COUNTED vmcnt for the proj / bias pieces only.
//      Without it the next slice is requested after barrier B and needed ~0.3 us later (P2 + P3), i.e. one exposed LDS-DMA round trip
//      per head.  Costs 48 KiB of LDS => one workgroup per CU: used with 8-wave workgroups (two windows, two waves per SIMD).
// WT (r04): weight terms.  2 = wqkv_f / wproj_f hold [hi image | lo image] (FvitStageDesc.weight_terms = 2, the x2 operand modes).  Built on the DBQ form
//      (8 waves, two qkv buffers, bias from L2): buffer 0 takes the hi slice of a head, buffer 1 its lo slice -- requested at the top of the head,
//      consumed after a third barrier by a second pass of P1 into the SAME q / k / v accumulators --, the proj region holds both proj slices and
//      P3 runs once per term on the same O^T fragment.  Activations are rounded once, as everywhere in the x2 modes.
template <typename T, int CC, int NRB, int NW, bool BIAS_LDS, bool DBQ = false, bool TS = false, int WT = 1>
__global__ __launch_bounds__(64 * NW, (CC == 256 && WT == 1) ? 2 : 1) void attnblk_kernel(AttnBlkParams p) {
    static_assert(WT == 1 || (WT == 2 && DBQ && !BIAS_LDS && !TS), "two weight terms: DBQ form, bias from L2");
    typedef typename Op16<T>::v8 v8;
#define FVIT_AB_STAMP(k) if constexpr (TS) { if ((threadIdx.x & 63) == 0) p.ts[((size_t)blockIdx.x * NW + (threadIdx.x >> 6)) * 16 + (k)] = __builtin_amdgcn_s_memtime(); }
    FVIT_AB_STAMP(0)
    typedef typename Op16<T>::v4 v4;
    constexpr int C = CC, KK = C / 32, CB = C / 16;
    constexpr int SP = NRB * 16;              // padded window length
    constexpr int WPW = NW / NRB;             // windows per workgroup
    constexpr int NKB32 = (NRB + 1) / 2;      // 32-key groups per window
    constexpr int QKV_FRAGS = 6 * KK;         // 48 KiB
    constexpr int QKV_BYTES = QKV_FRAGS * 1024, PROJ_BYTES = CB * 1024, BIAS_BYTES = BIAS_LDS ? SP * SP * 4 : 0;
    constexpr int KX_BYTES = NW * 1024;       // one 1-KiB k fragment per wave
    constexpr int VX_BYTES = WPW * 2 * NKB32 * 1024;
    constexpr int OFF_PROJ = (DBQ ? 2 : 1) * QKV_BYTES, OFF_BIAS = OFF_PROJ + WT * PROJ_BYTES, OFF_KX = OFF_BIAS + ((BIAS_BYTES + 1023) / 1024) * 1024;
    constexpr int OFF_VX = OFF_KX + KX_BYTES, OFF_BQ = OFF_VX + VX_BYTES;
    constexpr int MAX_HEADS = CC / 32;
    constexpr int OFF_BP = OFF_BQ + MAX_HEADS * 96 * 4;   // proj bias and gamma (2 x C floats): no global loads in the epilogue
    __shared__ __attribute__((aligned(16))) char smem[OFF_BP + 2 * CC * 4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, s = lane & 15;
    const int lane16 = lane * 16;
    const int wi = wave / NRB, qb = wave - wi * NRB;           // window inside the workgroup, row block inside the window
    const int win = blockIdx.x * WPW + wi;
    const bool win_ok = win < p.nwin;
    const int tok = qb * 16 + s;                                // token inside the window
    const bool row_ok = win_ok && tok < p.S;
    const int64_t row = (int64_t)(win_ok ? win : p.nwin - 1) * p.S + (tok < p.S ? tok : p.S - 1);   // clamped: always a real row
    float* bqs = (float*)(smem + OFF_BQ);
    float* bps = (float*)(smem + OFF_BP);
    float* gms = bps + CC;

    const char* __restrict__ Wq = (const char*)p.wqkv_f;
    const char* __restrict__ Wp = (const char*)p.wproj_f;

    auto dma_qkv = [&](int h, int slot, int term = 0) {   // 48 fragments into qkv buffer `slot` (always 0 without DBQ); term 1 = the lo image
        const char* src = Wq + ((size_t)term * p.heads + h) * QKV_BYTES + lane16;
        char* dst = smem + slot * QKV_BYTES;
#pragma unroll
        for (int i = 0; i < QKV_FRAGS / NW; ++i) glds16(src + (wave + NW * i) * 1024, dst + (wave + NW * i) * 1024);
    };
    auto dma_proj_bias = [&](int h) {   // 16 proj fragments + (BIAS_LDS) the head's bias table
        const char* src = Wp + (size_t)h * PROJ_BYTES + lane16;
#pragma unroll
        for (int i = 0; i < CB / NW; ++i) glds16(src + (wave + NW * i) * 1024, smem + OFF_PROJ + (wave + NW * i) * 1024);
        if constexpr (WT == 2) {
            const char* src2 = Wp + ((size_t)p.heads + h) * PROJ_BYTES + lane16;
#pragma unroll
            for (int i = 0; i < CB / NW; ++i) glds16(src2 + (wave + NW * i) * 1024, smem + OFF_PROJ + PROJ_BYTES + (wave + NW * i) * 1024);
        }
        if (BIAS_LDS) {
            const char* bsrc = (const char*)(p.bias + (size_t)h * SP * SP) + lane16;
            constexpr int BP = (BIAS_BYTES + 1023) / 1024;   // 16 (SP = 64) or 1 (SP = 16)
#pragma unroll
            for (int i = 0; i < (BP + NW - 1) / NW; ++i) {
                const int piece = wave + NW * i;
                if (piece < BP) glds16(bsrc + piece * 1024, smem + OFF_BIAS + piece * 1024);
            }
        }
    };

    // ---- prologue: qkv bias to LDS, first weight slice in flight, gather + LayerNorm into B/A fragments ----
    for (int i = tid; i < p.heads * 96; i += 64 * NW) bqs[i] = p.bqkv[i];
    for (int i = tid; i < CC; i += 64 * NW) {
        bps[i] = p.bproj[i];
        gms[i] = p.gamma ? p.gamma[i] : 1.0f;
    }
    if (NRB == 1) {   // the upper half of every 32-key group never gets written: keep it finite
        for (int i = tid; i < VX_BYTES / 4; i += 64 * NW) ((float*)(smem + OFF_VX))[i] = 0.f;
    }
    const int hrot = p.stagger ? (int)((blockIdx.x >> 3) % (unsigned)p.heads) : 0;
    auto head_of = [&](int it) { const int hh = it + hrot; return hh >= p.heads ? hh - p.heads : hh; };
    dma_qkv(head_of(0), 0);
    if constexpr (WT == 2) dma_qkv(head_of(0), 1, 1);   // both slices of the first head

    const float* src;
    const float* addp = nullptr;
    {
        const int b = (int)(row / p.rows_per_image), pr = (int)(row - (int64_t)b * p.rows_per_image);
        if (p.src_idx) {
            const int si = p.src_idx[pr];
            src = si >= 0 ? p.srcA + ((size_t)b * p.rowsA + si) * C : p.srcB + ((size_t)b * p.rowsB + (-si - 1)) * C;
        } else {
            src = p.srcA + (size_t)row * C;
        }
        if (p.add) {
            const int ai = p.add_idx ? p.add_idx[pr] : pr;
            if (ai >= 0) addp = p.add + (size_t)ai * C;
        }
    }
    // GEMM k slot kk*32 + 8g + e carries input channel kch(kk, g, e) = (kk>>1)*64 + g*16 + (kk&1)*8 + e (w_qkv_frag is packed in
    // that order): the 64 values a lane gathers are then exactly the 64 output channels it owns in the proj accumulator, and
    // the residual epilogue uses them from registers instead of gathering the row a second time
    constexpr bool KEEPX = CC == 256;   // C = 512: 128 more VGPRs next to the 128 of the proj accumulator would spill: re-gather instead
    v8 xf[KK];
    f4 v[KK][2];
    {
        float sum = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int co = (kk >> 1) * 64 + g * 16 + (kk & 1) * 8 + h2 * 4;
                f4 t = *(const f4*)(src + co);
                if (addp) t += *(const f4*)(addp + co);
                v[kk][h2] = t;
                sum += (t[0] + t[1]) + (t[2] + t[3]);
            }
        sum = sum_xor32(sum_xor16(sum));
        if constexpr (TS) { asm volatile("s_nop 0" ::"v"(sum) : "memory"); }
        FVIT_AB_STAMP(1)
        const float mean = sum / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const f4 d = v[kk][h2] - mean;
                sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
        sq = sum_xor32(sum_xor16(sq));
        const float rstd = rsqrtf(sq / (float)C + p.eps);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            v8 o;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int co = (kk >> 1) * 64 + g * 16 + (kk & 1) * 8 + h2 * 4;
                const f4 w = *(const f4*)(p.ln_w + co);
                const f4 b = *(const f4*)(p.ln_b + co);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[h2 * 4 + r] = sat16<T>((v[kk][h2][r] - mean) * rstd * w[r] + b[r]);
            }
            xf[kk] = o;
        }
    }

    f4 oacc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) oacc[cb] = (f4){0.f, 0.f, 0.f, 0.f};
    if constexpr (TS) { asm volatile("s_nop 0" ::"v"(xf[KK - 1]) : "memory"); }
    FVIT_AB_STAMP(2)

    const char* wq_base = smem + lane16;              // qkv fragments: + (ub * KK + kk) * 1024 (+ the buffer of this head with DBQ)
    const char* wp_l = smem + OFF_PROJ + lane16;      // proj fragments: + cb * 1024
    const float* bias_l = (const float*)(smem + OFF_BIAS);
    char* kx = smem + OFF_KX;
    char* vx = smem + OFF_VX;

    for (int hit = 0; hit < p.heads; ++hit) {
        const int h = head_of(hit);
        // ---- barrier A: qkv slice of head h landed; every wave is done with P2/P3 of head h-1 ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f4 bzg[NRB];
        if (!BIAS_LDS) {   // ordinary loads are issued AHEAD of this head's DMA, so waiting for them never drains the DMA queue
            const float* bg = p.bias + ((size_t)h * SP + tok) * SP + g * 4;
#pragma unroll
            for (int kb = 0; kb < NRB; ++kb) bzg[kb] = *(const f4*)(bg + kb * 16);
        }
        if (!(p.ablate & 1)) dma_proj_bias(h);
        const bool next_q = DBQ && hit + 1 < p.heads && !(p.ablate & 1);
        if constexpr (WT == 1) {
            if (next_q) dma_qkv(head_of(hit + 1), (hit + 1) & 1);
        }   // WT = 2: both slices of this head landed at barrier A (hi requested after the previous head's barrier A2, lo after its barrier B)
        const char* wq_l = wq_base + ((DBQ && WT == 1) ? (hit & 1) * QKV_BYTES : 0);   // (WT = 2: buffer 0 = hi slice, buffer 1 = lo slice)

        // ---- P1: q^T, k^T, v ----
        // software-pipelined over the k steps with two fragment register sets: the 6 fragments of step kk + 1 are requested before
        // the 6 MFMAs of step kk issue, so a wave that is alone on its SIMD sees ONE exposed LDS round trip per head here instead
        // of one per MFMA pair (r02 ISA audit: the compiler's own order was "2 ds_read, s_waitcnt lgkmcnt(0), 1-2 MFMA" x 24)
        f4 acc[6];
#pragma unroll
        for (int ub = 0; ub < 6; ++ub) acc[ub] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int term = 0; term < WT; ++term) {
            if (WT == 2 && term == 1) {
                // ---- barrier A2: every wave is done reading buffer 0 (no memory wait: the lo slice landed before barrier A) ----
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (next_q) dma_qkv(head_of(hit + 1), 0, 0);   // next head's hi slice: in flight until the next barrier A
                wq_l = wq_base + QKV_BYTES;
            }
            v8 wa[6], wb[6];
#pragma unroll
            for (int ub = 0; ub < 6; ++ub) wa[ub] = *(const v8*)(wq_l + (ub * KK + 0) * 1024);
#pragma unroll
            for (int kk = 0; kk < KK; kk += 2) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ub = 0; ub < 6; ++ub) wb[ub] = *(const v8*)(wq_l + (ub * KK + kk + 1) * 1024);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ub = 0; ub < 4; ++ub) acc[ub] = Op16<T>::mfma(wa[ub], xf[kk], acc[ub]);       // q0 q1 k0 k1: weights are the A operand
#pragma unroll
                for (int ub = 4; ub < 6; ++ub) acc[ub] = Op16<T>::mfma(xf[kk], wa[ub], acc[ub]);       // v0 v1: activations are A => D[key][dim]
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 2 < KK) {
#pragma unroll
                    for (int ub = 0; ub < 6; ++ub) wa[ub] = *(const v8*)(wq_l + (ub * KK + kk + 2) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int ub = 0; ub < 4; ++ub) acc[ub] = Op16<T>::mfma(wb[ub], xf[kk + 1], acc[ub]);
#pragma unroll
                for (int ub = 4; ub < 6; ++ub) acc[ub] = Op16<T>::mfma(xf[kk + 1], wb[ub], acc[ub]);
            }
        }
        const float* bq = bqs + h * 96;
        v8 qf, kf;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const f4 bqv = *(const f4*)(bq + blk * 16 + g * 4);
            const f4 bkv = *(const f4*)(bq + 32 + blk * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                qf[blk * 4 + r] = sat16<T>(acc[blk][r] + bqv[r]);
                kf[blk * 4 + r] = sat16<T>(acc[2 + blk][r] + bkv[r]);
            }
        }
        // exchange: this wave's k fragment and its half of the v fragments
        *(v8*)(kx + wave * 1024 + lane16) = kf;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const float bv = bq[64 + db * 16 + s];
            v4 vh;
#pragma unroll
            for (int r = 0; r < 4; ++r) vh[r] = sat16<T>(acc[4 + db][r] + bv);
            *(v4*)(vx + ((wi * 2 + db) * NKB32 + (qb >> 1)) * 1024 + lane16 + (qb & 1) * 8) = vh;
        }

        // ---- barrier B: proj / bias slices landed, k / v visible, qkv buffer free ----
        if (DBQ) {
            // the next head's qkv pieces (QKV_FRAGS / NW per wave, issued AFTER this head's proj / bias pieces) stay in flight; raw
            // barrier: __syncthreads() would drain them
            if (next_q) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(QKV_FRAGS / NW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if constexpr (WT == 2) {
                if (next_q) dma_qkv(head_of(hit + 1), 1, 1);   // buffer 1 is free (every wave is past its second P1 pass): next head's lo slice
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (hit + 1 < p.heads && !(p.ablate & 1)) dma_qkv(head_of(hit + 1), 0);
        }

        // ---- P2: scores^T, softmax over keys, O^T ----
        f4 sc[NRB];
        float mx = -3.0e38f;
#pragma unroll
        for (int kb = 0; kb < NRB; ++kb) {
            const v8 kfa = *(const v8*)(kx + (wi * NRB + kb) * 1024 + lane16);
            f4 a = Op16<T>::mfma(kfa, qf, (f4){0.f, 0.f, 0.f, 0.f});
            const f4 bz = BIAS_LDS ? *(const f4*)(bias_l + tok * SP + kb * 16 + g * 4) : bzg[kb];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a[r] = a[r] * p.scale + bz[r];
                mx = fmaxf(mx, a[r]);
            }
            sc[kb] = a;
        }
        mx = max_xor32(max_xor16(mx));
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NRB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(sc[kb][r] - mx);
                sc[kb][r] = e;
                sum += e;
            }
        sum = sum_xor32(sum_xor16(sum));
        const float inv = 1.0f / sum;
        f4 o[2] = {(f4){0.f, 0.f, 0.f, 0.f}, (f4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int k32 = 0; k32 < NKB32; ++k32) {
            v8 pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = (T)sc[2 * k32][r];
                pf[4 + r] = (2 * k32 + 1 < NRB) ? (T)sc[(2 * k32 + 1 < NRB) ? 2 * k32 + 1 : 0][r] : (T)0.f;
            }
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const v8 vf = *(const v8*)(vx + ((wi * 2 + db) * NKB32 + k32) * 1024 + lane16);
                o[db] = Op16<T>::mfma(vf, pf, o[db]);
            }
        }
        v8 of;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            of[r] = sat16<T>(o[0][r] * inv);
            of[4 + r] = sat16<T>(o[1][r] * inv);
        }
        // ---- P3: out^T += Wproj[:, head h] . O^T (fragment batches of 4, batch c + 1 requested before the MFMAs of batch c) ----
#pragma unroll
        for (int term = 0; term < WT; ++term) {
            constexpr int PB = 4;
            const char* wp_t = wp_l + term * PROJ_BYTES;
            v8 pa[PB], pb[PB];
#pragma unroll
            for (int i = 0; i < PB; ++i) pa[i] = *(const v8*)(wp_t + i * 1024);
#pragma unroll
            for (int c0 = 0; c0 < CB; c0 += 2 * PB) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < PB; ++i) pb[i] = *(const v8*)(wp_t + (c0 + PB + i) * 1024);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < PB; ++i) oacc[c0 + i] = Op16<T>::mfma(pa[i], of, oacc[c0 + i]);
                __builtin_amdgcn_sched_barrier(0);
                if (c0 + 2 * PB < CB) {
#pragma unroll
                    for (int i = 0; i < PB; ++i) pa[i] = *(const v8*)(wp_t + (c0 + 2 * PB + i) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < PB; ++i) oacc[c0 + PB + i] = Op16<T>::mfma(pb[i], of, oacc[c0 + PB + i]);
            }
        }
        if constexpr (TS) {
            asm volatile("s_nop 0" ::"v"(oacc[CB - 1]) : "memory");
            if (hit < 8) { FVIT_AB_STAMP(3 + hit) }
        }
    }
    FVIT_AB_STAMP(14)

    // ---- epilogue: x_out[row] = x_in + gamma * (out + bproj); fragment cb, slot 4g + r <-> channel (cb>>2)*64 + 16g + (cb&3)*4 + r ----
    if (row_ok) {
        float* px = p.x_out + (size_t)row * C;
#pragma unroll
        for (int cg = 0; cg < CB / 4; ++cg) {
            const int c0 = cg * 64 + g * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f4 xv;
                if (KEEPX) {
                    xv = v[2 * cg + (q >> 1)][q & 1];   // channel c0 + 4q = kch(2cg + (q>>1), g, 4(q&1)): the gathered row, still in registers
                } else {
                    xv = *(const f4*)(src + c0 + q * 4);
                    if (addp) xv += *(const f4*)(addp + c0 + q * 4);
                }
                const f4 bv = *(const f4*)(bps + c0 + q * 4);
                const f4 gv = *(const f4*)(gms + c0 + q * 4);
                const f4 a = oacc[cg * 4 + q];
#pragma unroll
                for (int r = 0; r < 4; ++r) xv[r] += gv[r] * (a[r] + bv[r]);
                *(f4*)(px + c0 + q * 4) = xv;
            }
        }
    }
    if constexpr (TS) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    FVIT_AB_STAMP(15)
#undef FVIT_AB_STAMP
}

}  // namespace

// S <= 16 uses the 16-token instance (bias tables padded to 16), 48 < S <= 64 the 64-token one (padded to 64)
bool attnblk_supported(int C, int heads, int S) {
    if (C == 256 && heads == 8) return (S >= 1 && S <= 16) || (S > 48 && S <= 64);
    return C == 512 && heads == 16 && S > 48 && S <= 64;   // stage 3 of FasterViT-0: 7x7 windows, no carrier tokens
}

int launch_attnblk(const AttnBlkCall& c, hipStream_t stream) {
    if (ablate_skip(8)) return FVIT_OK;
    if (!attnblk_supported(c.C, c.heads, c.S) || c.nwin <= 0 || !c.wqkv_f || !c.wproj_f || !c.x_out) {
        set_error("attn_block: unsupported arguments C=%d heads=%d S=%d nwin=%d", c.C, c.heads, c.S, c.nwin);
        return FVIT_EINVAL;
    }
    AttnBlkParams p;
    p.srcA = c.srcA; p.srcB = c.srcB; p.src_idx = c.src_idx; p.add_idx = c.add_idx; p.add = c.add; p.ln_w = c.ln_w; p.ln_b = c.ln_b;
    p.eps = c.eps; p.rowsA = c.rowsA; p.rowsB = c.rowsB; p.rows_per_image = c.rows_per_image > 0 ? c.rows_per_image : 1;
    p.wqkv_f = c.wqkv_f; p.bqkv = c.bqkv; p.wproj_f = c.wproj_f; p.bproj = c.bproj; p.gamma = c.gamma; p.bias = c.bias;
    p.x_out = c.x_out; p.nwin = c.nwin; p.S = c.S; p.heads = c.heads; p.scale = c.scale;
    p.ablate = diag_knob("ab_ablate");
    p.stagger = tune_get("ab_stagger", 0);
    p.ts = (unsigned long long*)c.ts;
    const double rows = (double)c.nwin * c.S;
    const double flops = 2.0 * rows * c.C * 4.0 * c.C + 4.0 * c.nwin * (double)c.heads * c.S * (double)c.S * 32.0;
    const double bytes = 8.0 * rows * c.C + 8.0 * c.C * c.C;
    ProfScope prof(FVIT_K_ATTN_FUSED, flops, bytes, stream);
    prof_note(c.C == 256 ? (c.S <= 16 ? "attnblk_kernel<256,S16>" : "attnblk_kernel<256,S64>") : "attnblk_kernel<512,S64>", c.nwin);
    const bool small = c.S <= 16;
    // r06: stage-2 windows, one weight term: the wave-per-(window, head) cut (fvit_attnblk2.hip); fvit_tune("ab_variant", 0 / 1 / 2) restores the forms below
    if (attnblk2_supported(c.C, c.heads, c.S) && c.terms == 1 && tune_get("ab_variant", 0) == 3) return launch_attnblk2(c, stream);
    if (c.terms != 1 && (c.terms != 2 || c.C != 256 || small)) {
        set_error("attn_block: weight terms %d at C=%d S=%d (two terms: C = 256, 48 < S <= 64 only)", c.terms, c.C, c.S);
        return FVIT_EINVAL;
    }
    if (c.terms == 2) {   // [hi image | lo image] weights: the double-buffered 8-wave form with a second P1 / P3 pass per head
        prof_note("attnblk_kernel<256,S64,2 terms>", (c.nwin + 1) / 2);
        if (c.ts) { set_error("attn_block timeline: one weight term only"); return FVIT_EINVAL; }
        if (c.dtype == FVIT_F16) hipLaunchKernelGGL((attnblk_kernel<_Float16, 256, 4, 8, false, true, false, 2>), dim3((c.nwin + 1) / 2), dim3(512), 0, stream, p);
        else if (c.dtype == FVIT_BF16) hipLaunchKernelGGL((attnblk_kernel<__bf16, 256, 4, 8, false, true, false, 2>), dim3((c.nwin + 1) / 2), dim3(512), 0, stream, p);
        else { set_error("attn_block: operand dtype %d not supported", c.dtype); return FVIT_EINVAL; }
        return check_launch("attnblk_kernel");
    }
    // 0 (default): 4 waves / 1 window per workgroup, bias from L2, two workgroups per CU;
    // 1: 8 waves / 2 windows, bias table in LDS, one workgroup per CU -- equal on whole-batch launches (99.9 vs 96.7 us: the
    //    sub-block is bound by its three fp32 passes over X), but the smaller workgroups of variant 0 fill the chip better on
    //    the shard-sized launches of the stream-sharded deploy plan (+2..3 % images/s, r01 sweep r20)
    const int variant = tune_get("ab_variant", 0);
#define FVIT_AB(T, NRB, NW, BL) do { \
        if (c.C == 256) hipLaunchKernelGGL((attnblk_kernel<T, 256, NRB, NW, BL>), dim3((c.nwin + (NW / NRB) - 1) / (NW / NRB)), dim3(64 * NW), 0, stream, p); \
        else hipLaunchKernelGGL((attnblk_kernel<T, 512, 4, 4, false>), dim3(c.nwin), dim3(256), 0, stream, p); \
    } while (0)
#define FVIT_AB_DBQ(T) hipLaunchKernelGGL((attnblk_kernel<T, 256, 4, 8, true, true>), dim3((c.nwin + 1) / 2), dim3(512), 0, stream, p)
    // variant 2: 8 waves / 2 windows, bias in LDS, double-buffered qkv slices (one 149-KiB workgroup per CU)
    const bool dbq = variant == 2 && c.C == 256 && !small;
    if (c.ts) {   // timeline instance: the default fp16 form only
        if (c.dtype != FVIT_F16 || c.C != 256 || small) { set_error("attn_block timeline: fp16, C = 256, 48 < S <= 64 only"); return FVIT_EINVAL; }
        hipLaunchKernelGGL((attnblk_kernel<_Float16, 256, 4, 4, false, false, true>), dim3(c.nwin), dim3(256), 0, stream, p);
        return check_launch("attnblk_kernel");
    }
    if (c.dtype == FVIT_F16) {
        if (small) FVIT_AB(_Float16, 1, 8, true);
        else if (dbq) FVIT_AB_DBQ(_Float16);
        else if (variant == 1) FVIT_AB(_Float16, 4, 8, true);
        else FVIT_AB(_Float16, 4, 4, false);
    } else if (c.dtype == FVIT_BF16) {
        if (small) FVIT_AB(__bf16, 1, 8, true);
        else if (dbq) FVIT_AB_DBQ(__bf16);
        else if (variant == 1) FVIT_AB(__bf16, 4, 8, true);
        else FVIT_AB(__bf16, 4, 4, false);
    } else {
        set_error("attn_block: operand dtype %d not supported", c.dtype);
        return FVIT_EINVAL;
    }
#undef FVIT_AB
#undef FVIT_AB_DBQ
    return check_launch("attnblk_kernel");
}

}  // namespace fvit
