// fvit_winmlp.hip -- MLP sub-block of a HAT block for C = 512, hidden 2048 (stage 3 of FasterViT-0) with the waves of a 64-row workgroup
// splitting the N dimension (gfx950):
//
//   x += gamma * fc2( GELU( fc1( LayerNorm(x) ) ) )                                   (AR:697 with AR:398-407)
//
// Same contract as fvit_mlp_fused.  The per-row-block form of that kernel (fvit_mlp.hip: every wave 16 rows, ALL 4 MiB of weights through
// LDS per 64 rows in lockstep) loses at this shape; unfused it is LayerNorm + fc1 GEMM + fc2 GEMM, three chip-wide launches per block and
// stream shard.  Here (the work split of fvit_ctblk.hip / fvit_winblk.hip): a workgroup of 8 waves owns 64 rows = 4 row blocks.
//   A  LayerNorm of the 64 rows into MFMA B-operand fragments in LDS (XN[rb][kk], 64 KiB).
//   per super-chunk of 256 hidden units (8 of them):
//   B  wave w computes the 32 units of chunk 8 sc + w for ALL four row blocks (W1 slice straight from L2 through a register ring),
//      applies bias + GELU and publishes the four H^T fragments in LDS (H[buf][w][rb], double-buffered: one barrier per super-chunk);
//   C  wave w accumulates output channels 64w .. 64w + 63 for all rows over the 8 chunks of the super-chunk (its W2 slice through the ring,
//      the H^T fragments of all waves from LDS).
// Every weight fragment is read from L2 exactly once per workgroup and feeds four MFMAs; the hidden activation never leaves the CU.
//
// Measured (FasterViT-0 stage 3, 4 214 rows = 66 workgroups, r02 call r4a): the stage-3 forward ALONE gets slower (547 -> 622 us: 4 MiB of
// cold weights per workgroup at the ~65 GB/s a CU pulls from the memory side), but the step gets faster: 71.9k -> 76.3k images/s (+6.1 %).
// 66 CUs for ~75 us instead of the whole chip for LayerNorm + two GEMMs leave the other two stream shards the rest of the GPU.
// On by default (fvit_tune "win_mlp").
// The C = 256 / hidden 1024 instance (128-row workgroups, stage 2: 143 workgroups per shard instead of the 285 of mlp_fused_kernel) is
// correct and tested but buys nothing there (774 vs 773 us per stage-2 forward, 78.7k vs 79.9k images/s, call r4c): 143 workgroups are not
// narrow, and the 64-row kernel it would replace is not weight-bound (fvit_tune "win_mlp256" = 1).  What does pay in stage 2 is the
// 4-wave form ("win_mlp256" = 2, the default): 64-row workgroups of 4 waves, 210 registers and 68 KiB of LDS, so TWO workgroups share a
// CU and their LayerNorm / fc1 / exchange / fc2 phases interleave instead of running in lockstep: 47 vs 50 us per launch against
// mlp_fused_kernel<256> (LDS reads per MFMA a quarter of that kernel's), 774 -> 756 us per stage-2 forward, 79.8k -> 80.8k images/s (call r4h).
#include "fvit_common.h"

namespace fvit {

namespace {

struct WinMlpParams {
    float* x;            // [M][C] fp32 residual stream, updated in place
    const float* ln_w;
    const float* ln_b;
    const void* w1f;     // op16 [hidden/32][2][C/32][64][8]
    const float* b1;
    const void* w2f;     // op16 [hidden/32][C/16][64][8]
    const float* b2;
    const float* gamma;  // [C] or null
    float eps;
    int M;
    float* slab;     // NSPLIT > 1: f32 partial outputs [row group][NSPLIT][waves][CBW * NRB][64 lanes][4], 64 x C x 4 bytes per (row group, split)
    int* counters;   // NSPLIT > 1: one arrival counter per row group, zero before the launch, zero again after it
    unsigned long long* ts;   // TS instances only (fvit_debug_win_mlp_timeline): s_memtime stamps [workgroup][wave][16]
    int ablate;   // DIAGNOSIS build only (fvit_tune "wm_ablate", results are wrong): 1 = no weight loads inside the main loop (the ring keeps the first steps'
                  // fragments), 2 = no barrier inside the main loop, 4 = GELU -> identity (bias + narrowing stay).  Always 0 in the shipped library.
};

// phase stamps of the TS (timeline) instances: 0 kernel entry, 1 first ring steps issued, 2 rows loaded, 3 LayerNorm written + barrier,
// 4 .. 4 + NSC - 1 end of super-chunk sc (capped at slot 13), 14 before the epilogue, 15 end
template <bool TS>
__device__ __forceinline__ void stamp(const WinMlpParams& p, int wave, int lane, int nw, int k) {
    if constexpr (TS) {
        if (p.ts && lane == 0) p.ts[((size_t)blockIdx.x * nw + wave) * 16 + (k < 15 ? k : 15)] = __builtin_amdgcn_s_memtime();
    }
}

// CC / HID: channels / hidden units; NRB: row blocks of 16 per workgroup (4: 64 rows, C = 512; 8: 128 rows, C = 256)
// NWV: waves per workgroup (8: one workgroup per CU; 4: 256 registers per wave and <= 70 KiB of LDS, two workgroups per CU whose phases interleave)
// SP: weight terms (FvitStageDesc.weight_terms).  SP = 2: w1f / w2f are two images back to back (hi, lo); every weight step runs once
// per term against the SAME activation fragments (hi steps first, then lo), so the hidden activation and the output accumulate
// X . (W_hi + W_lo) with X rounded once -- twice the weight stream and twice the MFMAs, nothing else changes.
// NSPLIT (r03, C = 512): the hidden units of a 64-row group are split over NSPLIT workgroups (each: LayerNorm of the 64 rows, fc1 + GELU
// of ITS HID / NSPLIT units, the fc2 partial sum over those units for all C channels).  One workgroup per 64 rows streams all 4 MiB of
// weights through one CU's L2 port (65-105 GB/s: >= 40 us whatever the instruction stream does); NSPLIT = 4 streams 1 MiB per CU on four
// CUs.  The partial outputs meet in L2: every workgroup stores its fp32 partial (128 KiB, lane-linear), releases it at agent scope and
// takes a ticket on the row group's counter; the LAST arriver acquires, re-reads all NSPLIT partials and adds them in the FIXED order
// split 0, 1, .. (bitwise repeatable whoever arrives last), applies bias / gamma / residual and resets the counter.  Nobody waits for
// anybody (no spin: placement- and residency-independent).  Sibling workgroups get block ids that differ by a multiple of 8 (same XCD
// under the observed round-robin dispatch): their partials and the rows they all read stay in one L2 (speed only, never correctness).
// LDS bytes of a workgroup: XN (NRB x C / 32 KiB), H (single or double buffered, see HBUF below), the fc1 bias
// one workgroup per CU (all of its LDS, one or two waves per SIMD): the 8-wave forms, and (r06 experiment) 4 waves x 128 rows with up to 512 registers per wave
constexpr bool winmlp_one_per_cu(int NWV, int NRB) { return NWV == 8 || NRB == 8; }
template <int CC, int HID, int NRB, int NWV>
constexpr int winmlp_lds_bytes() {
    constexpr int hbuf = (NRB * (CC / 32) + 2 * NWV * NRB) * 1024 + HID * 4 <= (winmlp_one_per_cu(NWV, NRB) ? 150 : 72) * 1024 ? 2 : 1;
    return NRB * (CC / 32) * 1024 + hbuf * NWV * NRB * 1024 + HID * 4;
}

// The kernel body as a device function (blk = blockIdx.x of a stand-alone launch): fvit_stage3.hip runs it as one phase of a persistent workgroup
// (the timeline stamps index by blockIdx.x: stand-alone launches only).
template <typename T, int CC, int HID, int NRB, int DEPTH, int NWV = 8, int SP = 1, int NSPLIT = 1, bool TS = false, bool PIPE = false>
__device__ __forceinline__ void winmlp_body(const WinMlpParams& p, char* const smem, const int blk) {
    typedef typename Op16<T>::v8 v8;
    constexpr int C = CC, KK = C / 32, CB = C / 16, NW = NWV;
    constexpr int CBW = CB / NW;                   // output channel blocks per wave (4 / 2)
    constexpr int NSC = HID / 32 / NW / NSPLIT;    // super-chunks of this workgroup: NW chunks of 32 units each, one chunk per wave
    static_assert(HID % (32 * NW * NSPLIT) == 0, "hidden units must split evenly");
    constexpr int F1S = 2 * KK / 8;                // fc1 steps of 8 fragments per chunk: 4 k steps x 2 unit blocks each
    constexpr int CPS = 8 / CBW;                   // chunks per fc2 step of 8 fragments
    constexpr int F2S = NW / CPS;                  // fc2 steps per super-chunk
    constexpr int F1T = SP * F1S, F2T = SP * F2S;  // ... times the weight terms
    constexpr int SPS = F1T + F2T;                 // steps per super-chunk (8 / 4 per term)
    constexpr size_t WBYTES = (size_t)HID * C * 2; // one fragment-order image of fc1 (= of fc2)
    static_assert(SPS % DEPTH == 0, "ring slots must be static inside the super-chunk loop");
    constexpr bool LN_EVEN = NW % NRB == 0;        // else (NRB = 5 / 6 with 8 waves): waves 0 .. NRB - 1 take one whole row block each in phase A
    constexpr int WPR = LN_EVEN ? NW / NRB : 1;    // waves sharing a row block in the LayerNorm phase (2 / 1)
    constexpr int HBUF = (NRB * KK + 2 * NW * NRB) * 1024 + HID * 4 <= (winmlp_one_per_cu(NW, NRB) ? 150 : 72) * 1024 ? 2 : 1;   // H double-buffered when it fits
    constexpr int OFF_H = NRB * KK * 1024;         // XN: 64 KiB; H: HBUF x NW x NRB KiB; fc1 bias
    constexpr int OFF_B1 = OFF_H + HBUF * NW * NRB * 1024;
    static_assert(OFF_B1 + HID * 4 == winmlp_lds_bytes<CC, HID, NRB, NWV>(), "LDS layout");
    float* b1s = (float*)(smem + OFF_B1);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, s = lane & 15;
    const int lane16 = lane * 16;
    int rg = blk, sp = 0;
    if constexpr (NSPLIT > 1) {
        const int xcd = blk & 7, j = blk >> 3;
        rg = (j / NSPLIT) * 8 + xcd;
        sp = j % NSPLIT;
        if (rg * (16 * NRB) >= p.M) return;   // the whole sibling group is out of range (grid padded to 8 x NSPLIT)
    }
    const int row0 = rg * (16 * NRB);
    const int sc0 = sp * NSC;                     // first (global) super-chunk of this workgroup

    stamp<TS>(p, wave, lane, NWV, 0);
    const char* W1 = (const char*)p.w1f + lane16;
    const char* W2 = (const char*)p.w2f + lane16;
    v8 ring[DEPTH][8];
    // step (sc, u) into ring slot `slot`; u and slot are compile-time at every call site (the super-chunk loop is not unrolled, a
    // super-chunk is SPS steps and DEPTH divides SPS, so the slot of step SPS sc + u is u % DEPTH)
#ifdef FVIT_DIAG
    const bool abl_w = p.ablate & 1, abl_b = p.ablate & 2, abl_g = p.ablate & 4;
#else
    constexpr bool abl_w = false, abl_b = false, abl_g = false;   // the shipped library has no knob that can make a kernel skip work
#endif
    bool in_loop = false;   // the prologue's first DEPTH steps are always loaded
    auto issue = [&](int sc, int u, int slot) {
        if (sc < NSC && !(abl_w && in_loop)) {
            if (u < F1T) {    // fc1, term u / F1S: chunk 8 sc + wave, k steps 4uu .. 4uu + 3 (uu = u % F1S), slot (kk - 4uu) * 2 + hb
                const int uu = u % F1S;
                const char* b = W1 + (size_t)(u / F1S) * WBYTES + (size_t)((sc0 + sc) * NW + wave) * 2 * KK * 1024;
#pragma unroll
                for (int i = 0; i < 8; ++i) ring[slot][i] = *(const v8*)(b + ((i & 1) * KK + 4 * uu + (i >> 1)) * 1024);
            } else {          // fc2, term (u - F1T) / F2S: chunks 8 sc + CPS vv + c (vv = (u - F1T) % F2S), channel blocks CBW wave + q, slot c * CBW + q
                const int vv = (u - F1T) % F2S;
                const char* b = W2 + (size_t)((u - F1T) / F2S) * WBYTES + ((size_t)((sc0 + sc) * NW + CPS * vv) * CB + CBW * wave) * 1024;
#pragma unroll
                for (int i = 0; i < 8; ++i) ring[slot][i] = *(const v8*)(b + ((i / CBW) * CB + (i % CBW)) * 1024);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- phase A: LayerNorm; WPR waves share a row block: each reads the full rows and writes KK / WPR of the k steps ----
    // request order (r03): fc1 bias, first ring steps, the rows -- all in flight before the first wait.  The bias goes to LDS (an ordinary load
    // inside the chunk loops would queue behind the ring's prefetches and drain it); its LDS write used to sit BEFORE the row loads were even
    // requested: one memory round trip (3.8 us at C = 512, profiles/r03_winmlp_phase_timeline.log) in front of the prologue.
    constexpr int NT = 64 * NW;
    // (r06: NRB > NW -- four waves x 128 rows -- takes ceil(NRB / NW) passes: wave w normalises row blocks w, w + NW, ...)
    constexpr int NPASS = LN_EVEN ? 1 : (NRB + NW - 1) / NW;
    float c1[HID / NT];
#pragma unroll
    for (int i = 0; i < HID / NT; ++i) c1[i] = p.b1[tid + NT * i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < DEPTH; ++t) issue(0, t, t);
    in_loop = true;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
    const int ln_rb_raw = LN_EVEN ? wave / WPR : ps * NW + wave;
    const bool ln_wave = LN_EVEN || ln_rb_raw < NRB;
    const int ln_rb = ln_wave ? ln_rb_raw : 0, ln_part = LN_EVEN ? wave % WPR : 0;
    f4 v[2 * KK];
    {
        const int row = min(row0 + ln_rb * 16 + s, p.M - 1);
        const float* src = p.x + (size_t)row * C;
        if (ln_wave) {
#pragma unroll
            for (int i = 0; i < 2 * KK; ++i) v[i] = *(const f4*)(src + (i >> 2) * 64 + g * 16 + (i & 3) * 4);   // i = 2 * kk + h2
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ps == 0) {
#pragma unroll
        for (int i = 0; i < HID / NT; ++i) b1s[tid + NT * i] = c1[i];
        stamp<TS>(p, wave, lane, NWV, 1);
    }
    if (ln_wave) {
        constexpr int KP = KK / WPR;
        const int rb = ln_rb, part = ln_part;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 2 * KK; ++i) sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        sum = sum_xor32(sum_xor16(sum));
        if (ps == 0) stamp<TS>(p, wave, lane, NWV, 2);
        const float mean = sum / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 2 * KK; ++i) {
            const f4 d = v[i] - mean;
            sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        sq = sum_xor32(sum_xor16(sq));
        const float rstd = rsqrtf(sq / (float)C + p.eps);
#pragma unroll
        for (int k8 = 0; k8 < KP; ++k8) {
            v8 o;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                // wave-uniform choice of the part without dynamic register indexing
                f4 xs = v[2 * k8 + h2];
#pragma unroll
                for (int pp = 1; pp < WPR; ++pp) xs = part == pp ? v[2 * (k8 + pp * KP) + h2] : xs;
                const int kk = k8 + part * KP;
                const int co = (kk >> 1) * 64 + g * 16 + (kk & 1) * 8 + h2 * 4;
                const f4 w = *(const f4*)(p.ln_w + co);
                const f4 bb = *(const f4*)(p.ln_b + co);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[h2 * 4 + r] = sat16<T>((xs[r] - mean) * rstd * w[r] + bb[r]);
            }
            *(v8*)(smem + ((rb * KK + k8 + part * KP) * 1024) + lane16) = o;
        }
    }
    }
    __syncthreads();
    stamp<TS>(p, wave, lane, NWV, 3);

    const char* xn = smem + lane16;
    f4 acc2[CBW][NRB];
#pragma unroll
    for (int q = 0; q < CBW; ++q)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc2[q][rb] = (f4){0.f, 0.f, 0.f, 0.f};

    if constexpr (!PIPE) {
#pragma unroll 1
    for (int sc = 0; sc < NSC; ++sc) {
        // ---- B: H^T of chunk 8 sc + wave: [32 units][16 NRB rows] ----
        f4 acc1[2][NRB];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc1[hb][rb] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < F1T; ++u) {
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                v8 xb[NRB];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) xb[rb] = *(const v8*)(xn + (rb * KK + 4 * (u % F1S) + k4) * 1024);
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) acc1[hb][rb] = Op16<T>::mfma(ring[u % DEPTH][k4 * 2 + hb], xb[rb], acc1[hb][rb]);
            }
            __builtin_amdgcn_sched_barrier(0);
            issue(u + DEPTH < SPS ? sc : sc + 1, (u + DEPTH) % SPS, u % DEPTH);
        }
        const int j = (sc0 + sc) * NW + wave;
        const f4 bA = *(const f4*)(b1s + j * 32 + g * 4);
        const f4 bB = *(const f4*)(b1s + j * 32 + 16 + g * 4);
        const int hbuf = HBUF == 2 ? (sc & 1) : 0;
        char* hw = smem + OFF_H + (hbuf * NW + wave) * NRB * 1024 + lane16;
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            float hv[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                hv[r] = acc1[0][rb][r] + bA[r];
                hv[4 + r] = acc1[1][rb][r] + bB[r];
            }
            if (!abl_g) gelu_fast_n<8>(hv);   // eight independent Horner chains in lockstep (fvit_common.h), bitwise gelu_fast
            v8 pf;
#pragma unroll
            for (int r = 0; r < 8; ++r) pf[r] = sat16<T>(hv[r]);
            *(v8*)(hw + rb * 1024) = pf;
        }
        if (!abl_b) __syncthreads();   // H of this super-chunk visible (HBUF = 2: the other buffer's last readers are past their fc2 of super-chunk sc - 1)
        // ---- C: out^T[this wave's channels][rows] += W2[:, chunk] . H^T over the 8 chunks ----
        const char* hr = smem + OFF_H + hbuf * NW * NRB * 1024 + lane16;
#pragma unroll
        for (int u = F1T; u < SPS; ++u) {
#pragma unroll
            for (int c = 0; c < CPS; ++c) {
                v8 hb4[NRB];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) hb4[rb] = *(const v8*)(hr + ((CPS * ((u - F1T) % F2S) + c) * NRB + rb) * 1024);
#pragma unroll
                for (int q = 0; q < CBW; ++q)
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) acc2[q][rb] = Op16<T>::mfma(ring[u % DEPTH][c * CBW + q], hb4[rb], acc2[q][rb]);
            }
            __builtin_amdgcn_sched_barrier(0);
            issue(u + DEPTH < SPS ? sc : sc + 1, (u + DEPTH) % SPS, u % DEPTH);
        }
        if (HBUF == 1) __syncthreads();   // single H buffer: every wave is done reading it before the next super-chunk overwrites it
        stamp<TS>(p, wave, lane, NWV, sc + 4 < 13 ? sc + 4 : 13);
    }
    } else {
        // ---- the software-pipelined form (r06, fvit_tune "win_mlp_pipe"): the same steps in the order
        //          F1(0) GELU(0) | F1(1) B F2(0)+GELU(1) | F1(2) B F2(1)+GELU(2) | ... | B F2(NSC-1)          (B = the workgroup barrier)
        // fc1 of super-chunk sc + 1 runs BEFORE the barrier that publishes H(sc) (the barrier's skew and the H write's round trip hide behind 32 .. 128
        // MFMAs), and bias + GELU + narrowing + publishing of H(sc + 1) are issued INSIDE fc2 of super-chunk sc, one row block per fc2 step, so that the VALU
        // work sits in the shadow of the wave's own MFMAs instead of between two MFMA phases (r05: 496 VALU instructions behind every fc1, the MFMA pipe
        // idle for all waves of the workgroup at once since the barrier keeps them in phase).  Needs the double-buffered H: H(sc + 1) is written while
        // other waves still read H(sc); H(sc + 2) is written after barrier sc + 1, which every wave passes after its fc2(sc).
        // The weight stream is the same list of steps in the new order; a block of the loop is still SPS steps and DEPTH divides F1T and F2T, so ring
        // slots stay static.  Per value the operations and their order are those of the plain form: bitwise the same result.
        static_assert(HBUF == 2, "the pipelined form needs the double-buffered H");
        static_assert(F1T % DEPTH == 0 && F2T % DEPTH == 0, "ring slots must be static inside the pipelined loop");
        auto load_f1 = [&](int slot, int sc, int u) { issue(sc, u, slot); };                 // fc1 step u of super-chunk sc (nothing beyond the last)
        auto load_f2 = [&](int slot, int sc, int v) { issue(sc, F1T + v, slot); };           // fc2 step v of super-chunk sc
        f4 acc1[2][NRB];
        auto fc1 = [&](int sc, bool first) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc1[hb][rb] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < F1T; ++u) {
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    v8 xb[NRB];
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) xb[rb] = *(const v8*)(xn + (rb * KK + 4 * (u % F1S) + k4) * 1024);
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                        for (int rb = 0; rb < NRB; ++rb) acc1[hb][rb] = Op16<T>::mfma(ring[u % DEPTH][k4 * 2 + hb], xb[rb], acc1[hb][rb]);
                }
                __builtin_amdgcn_sched_barrier(0);
                // the stream after fc1(sc): fc1's own later steps, then fc2(sc - 1) -- or, behind the very first fc1, fc1(1) (fc2(0) when there is one super-chunk)
                const int nu = u + DEPTH;
                if (nu < F1T) load_f1(u % DEPTH, sc, nu);
                else if (first) { if (NSC > 1) load_f1(u % DEPTH, 1, nu - F1T); else load_f2(u % DEPTH, 0, nu - F1T); }
                else load_f2(u % DEPTH, sc - 1, nu - F1T);
            }
        };
        // bias + GELU + narrowing of row block rb of the accumulators -> H fragment of chunk j, published in buffer hbuf
        auto gelu_publish = [&](int j, int hbuf, int rb) {
            const f4 bA = *(const f4*)(b1s + j * 32 + g * 4);
            const f4 bB = *(const f4*)(b1s + j * 32 + 16 + g * 4);
            float hv[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                hv[r] = acc1[0][rb][r] + bA[r];
                hv[4 + r] = acc1[1][rb][r] + bB[r];
            }
            if constexpr (NW == 8) {   // 8 waves x 256 registers: two groups of four chains (eight at once spill 7 registers inside the loop)
                float h0[4] = {hv[0], hv[1], hv[2], hv[3]}, h1[4] = {hv[4], hv[5], hv[6], hv[7]};
                gelu_fast_n<4>(h0);
                gelu_fast_n<4>(h1);
#pragma unroll
                for (int r = 0; r < 4; ++r) { hv[r] = h0[r]; hv[4 + r] = h1[r]; }
            } else {
                gelu_fast_n<8>(hv);
            }
            v8 pf;
#pragma unroll
            for (int r = 0; r < 8; ++r) pf[r] = sat16<T>(hv[r]);
            *(v8*)(smem + OFF_H + ((hbuf * NW + wave) * NRB + rb) * 1024 + lane16) = pf;
        };
        fc1(0, true);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) gelu_publish(sc0 * NW + wave, 0, rb);
#pragma unroll 1
        for (int sc = 0; sc < NSC; ++sc) {
            const bool has_next = sc + 1 < NSC;
            if (has_next) fc1(sc + 1, false);
            if (!abl_b) __syncthreads();   // H(sc) of every wave visible; every wave is past fc2(sc - 1), i.e. done reading the other buffer
            const int hbuf = sc & 1;
            const char* hr = smem + OFF_H + hbuf * NW * NRB * 1024 + lane16;
            // this wave's chunk of the next super-chunk.  Behind the LAST super-chunk there is none: the same instructions then turn the stale fc1 accumulators into an
            // H fragment nobody reads (a branch around them would end the basic block and put them BEHIND the step's MFMAs instead of between them: r06 ISA check)
            const int jn = min((sc0 + sc + 1) * NW + wave, HID / 32 - 1);
#pragma unroll
            for (int v = 0; v < F2T; ++v) {
#pragma unroll
                for (int c = 0; c < CPS; ++c) {
                    v8 hb4[NRB];
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb) hb4[rb] = *(const v8*)(hr + ((CPS * (v % F2S) + c) * NRB + rb) * 1024);
#pragma unroll
                    for (int q = 0; q < CBW; ++q)
#pragma unroll
                        for (int rb = 0; rb < NRB; ++rb) acc2[q][rb] = Op16<T>::mfma(ring[v % DEPTH][c * CBW + q], hb4[rb], acc2[q][rb]);
                }
                // row blocks [v NRB / F2T, (v + 1) NRB / F2T) of H(sc + 1): VALU + one LDS write each, free to move between this step's MFMAs
#pragma unroll
                for (int rb = v * NRB / F2T; rb < (v + 1) * NRB / F2T; ++rb) gelu_publish(jn, hbuf ^ 1, rb);
                __builtin_amdgcn_sched_barrier(0);
                // the stream after fc2(sc): its own later steps, then fc1(sc + 2) -- or fc2(sc + 1) when sc + 1 is the last super-chunk (it has no fc1 block in front)
                const int nv = v + DEPTH;
                if (nv < F2T) load_f2(v % DEPTH, sc, nv);
                else if (sc + 2 < NSC) load_f1(v % DEPTH, sc + 2, nv - F2T);
                else load_f2(v % DEPTH, sc + 1, nv - F2T);
            }
            stamp<TS>(p, wave, lane, NWV, sc + 4 < 13 ? sc + 4 : 13);
        }
    }
    stamp<TS>(p, wave, lane, NWV, 14);

    if constexpr (NSPLIT > 1) {
        // ---- partial sums of the NSPLIT sibling workgroups meet in L2; the last arriver finishes the row group ----
        constexpr int FPW = CBW * NRB;   // accumulator fragments per wave
        f4* const slab_rg = (f4*)p.slab + (size_t)rg * NSPLIT * NW * FPW * 64;
        {
            f4* mine = slab_rg + ((size_t)sp * NW + wave) * FPW * 64 + lane;
#pragma unroll
            for (int q = 0; q < CBW; ++q)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) mine[(q * NRB + rb) * 64] = acc2[q][rb];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* const tick = (int*)(smem + OFF_B1);   // the fc1 bias copy is dead: every wave is past its last read (barriers of the last super-chunk)
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the post-write-back wait where the compiler cannot drop it
            *tick = __hip_atomic_fetch_add(p.counters + rg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (*tick != NSPLIT - 1) return;              // not the last of this row group: done
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            p.counters[rg] = 0;                        // every ticket of this launch is drawn; the next launch finds zero
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < CBW; ++q) {
#pragma unroll
            for (int rh = 0; rh < NRB; rh += 2) {     // two row blocks at a time: 2 NSPLIT 16-byte loads in flight per lane
                f4 part[NSPLIT][2];
#pragma unroll
                for (int s2 = 0; s2 < NSPLIT; ++s2)   // all NSPLIT partials are loaded the same way (own one included): no per-element select
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) part[s2][r2] = slab_rg[(((size_t)s2 * NW + wave) * FPW + q * NRB + rh + r2) * 64 + lane];
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {
                    f4 sum = part[0][r2];
#pragma unroll
                    for (int s2 = 1; s2 < NSPLIT; ++s2) sum += part[s2][r2];   // fixed order
                    acc2[q][rh + r2] = sum;
                }
            }
        }
    }

    // ---- epilogue: x[row][c] += gamma * (out + b2); fragment cb = CBW w + q, slot 4g + r <-> channel (cb>>2)*64 + 16g + (cb&3)*4 + r, row rb * 16 + s ----
    // Two passes (r03): ALL loads first, then the updates and stores.  Written as one load-update-store loop the compiler had to keep
    // every store ahead of the next iteration's loads (they may alias): 16 dependent L2 / HBM round trips per lane, 8-12 us of a 45-76 us
    // workgroup (phase timeline, profiles/r03_winmlp_phase_timeline.log).
    const bool has_g = p.gamma != nullptr;
    f4 bvq[CBW], glq[CBW], xv[NRB][CBW];
#pragma unroll
    for (int q = 0; q < CBW; ++q) {
        const int cb = CBW * wave + q;
        const int c0 = (cb >> 2) * 64 + g * 16 + (cb & 3) * 4;
        bvq[q] = *(const f4*)(p.b2 + c0);
        glq[q] = *(const f4*)((has_g ? p.gamma : p.b2) + c0);
    }
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
        const int row = min(row0 + rb * 16 + s, p.M - 1);   // clamped for the load; rows >= M are not stored
#pragma unroll
        for (int q = 0; q < CBW; ++q) {
            const int cb = CBW * wave + q;
            xv[rb][q] = *(const f4*)(p.x + (size_t)row * C + (cb >> 2) * 64 + g * 16 + (cb & 3) * 4);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
        const int row = row0 + rb * 16 + s;
        if (row < p.M) {
#pragma unroll
            for (int q = 0; q < CBW; ++q) {
                const int cb = CBW * wave + q;
                f4 o = xv[rb][q];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] += (has_g ? glq[q][r] : 1.f) * (acc2[q][rb][r] + bvq[q][r]);
                *(f4*)(p.x + (size_t)row * C + (cb >> 2) * 64 + g * 16 + (cb & 3) * 4) = o;
            }
        }
    }
    if constexpr (TS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp<TS>(p, wave, lane, NWV, 15);
    }
}

template <typename T, int CC, int HID, int NRB, int DEPTH, int NWV = 8, int SP = 1, int NSPLIT = 1, bool TS = false, bool PIPE = false>
__global__ __launch_bounds__(64 * NWV, winmlp_one_per_cu(NWV, NRB) ? 1 : 2) void winmlp_kernel(WinMlpParams p) {
    __shared__ __attribute__((aligned(16))) char smem[winmlp_lds_bytes<CC, HID, NRB, NWV>()];
    winmlp_body<T, CC, HID, NRB, DEPTH, NWV, SP, NSPLIT, TS, PIPE>(p, smem, blockIdx.x);
}

inline WinMlpParams make_winmlp_params(const MlpFusedCall& c) {
    WinMlpParams p;
    p.x = c.x; p.ln_w = c.ln_w; p.ln_b = c.ln_b; p.w1f = c.w1f; p.b1 = c.b1; p.w2f = c.w2f; p.b2 = c.b2; p.gamma = c.gamma; p.eps = c.eps; p.M = c.M;
    p.slab = c.slab; p.counters = c.counters; p.ts = (unsigned long long*)c.ts;
    p.ablate = diag_knob("wm_ablate");
    return p;
}

}  // namespace

#ifndef FVIT_BODIES_ONLY
size_t winmlp_split_slab_bytes(int64_t M, int C, int nsplit) { return (size_t)((M + 63) / 64) * (size_t)nsplit * 64 * C * 4; }

bool winmlp_supported(int C, int hidden) { return (C == 512 && hidden == 2048) || (C == 256 && hidden == 1024); }

int launch_winmlp(const MlpFusedCall& c, hipStream_t stream) {
    if (!winmlp_supported(c.C, c.hidden) || c.M <= 0 || !c.x || !c.w1f || !c.w2f) {
        set_error("win_mlp: unsupported arguments C=%d hidden=%d M=%d", c.C, c.hidden, c.M);
        return FVIT_EINVAL;
    }
    if (ablate_skip(c.C == 512 ? 2 : 1)) return FVIT_OK;
    const WinMlpParams p = make_winmlp_params(c);
    // (the 4-way split measured in profiles/r03_stage3_split_over_sibling_workgroups_ab.log -- 288 workgroups, two rounds, 90 us -- is no longer
    // instantiated: git history, commit "split-hidden form of the C = 512 MLP kernel")
    const int nsplit = (c.C == 512 && c.slab && c.counters && c.nsplit == 2) ? 2 : 1;
    const double flops = 4.0 * c.M * (double)c.C * c.hidden;
    const double bytes = 8.0 * c.M * (double)c.C + 4.0 * c.C * (double)c.hidden;
    ProfScope prof(FVIT_K_MLP_FUSED, flops, bytes, stream);
    // C = 256 (stage 2): 4 waves x 64 rows, 68 KiB of LDS, two workgroups per CU.  (r02 / r03 also measured 8 waves x 128 rows -- no gain -- and 8 waves x 80 / 96 rows:
    // the launch 24 % shorter, 50.4 -> 38.2 us, and the STEP 1.5 % slower, 124 KiB of LDS and 8 x 200 registers take the whole CU from the other stream shard's kernels;
    // fvit_tune "win_mlp256" = 1 / 3.  Not instantiated since r06: git history, profiles/HISTORY.md.)
    // r06, whole-batch launches with two steps in flight (inference.PipelinedInference): the 8-wave x 128-row form again behind fvit_tune "win_mlp256" = 1
    // (every weight fragment feeds 8 MFMAs, half the L2 -> CU weight stream per row; one 133-KiB workgroup per CU)
    // (r06 also measured FOUR waves x 128 rows -- one workgroup per CU, ONE wave per SIMD, up to 512 registers; weight ring 2 / 4 deep, plain / pipelined loop: the template
    // takes NRB = 8, NWV = 4 since then -- at 126-134 us per whole-batch launch against 97-99 us for the default and 106-108 us for the 8-wave form: halving the L2 -> CU
    // weight stream buys nothing when one wave per SIMD has to hide every latency by itself; profiles/r06_steps_in_flight_ab.log, call 17.  Not instantiated.)
    const int form256 = c.C == 256 && !c.ts ? tune_get("win_mlp256", 2) : 2;
    const bool wide256 = form256 == 1;
    const int rows_per_wg = wide256 ? 128 : 64;
    const int nrg = (c.M + rows_per_wg - 1) / rows_per_wg;
    const int grid = nsplit > 1 ? (nrg + 7) / 8 * 8 * nsplit : nrg;
    prof_note(c.C == 512 ? (nsplit == 2 ? "winmlp_kernel<512,split2>" : "winmlp_kernel<512>")
                         : (wide256 ? "winmlp_kernel<256,128rows>" : "winmlp_kernel<256>"), grid);
    if (c.dtype != FVIT_F16 && c.dtype != FVIT_BF16) { set_error("win_mlp: operand dtype %d not supported", c.dtype); return FVIT_EINVAL; }
    if (c.terms != 1 && c.terms != 2) { set_error("win_mlp: weight terms %d not supported", c.terms); return FVIT_EINVAL; }
#define FVIT_WINMLP(T, CC_, HID_, NRB_, NWV_, SP_) \
    hipLaunchKernelGGL((winmlp_kernel<T, CC_, HID_, NRB_, 2, NWV_, SP_>), dim3(grid), dim3(64 * NWV_), 0, stream, p)
#define FVIT_WINMLP_T(CC_, HID_, NRB_, NWV_)                                                     \
    do {                                                                                         \
        if (c.dtype == FVIT_F16) { if (c.terms == 2) FVIT_WINMLP(_Float16, CC_, HID_, NRB_, NWV_, 2); else FVIT_WINMLP(_Float16, CC_, HID_, NRB_, NWV_, 1); } \
        else { if (c.terms == 2) FVIT_WINMLP(__bf16, CC_, HID_, NRB_, NWV_, 2); else FVIT_WINMLP(__bf16, CC_, HID_, NRB_, NWV_, 1); } \
    } while (0)
    // the software-pipelined main loop (r06; fvit_tune "win_mlp_pipe", see winmlp_body): C = 512 only.  Measured in one box, three interleaved pairs
    // (profiles/r06_winmlp_pipelined_loop_ab.log): winmlp<512> 83.8 -> 80.9 us per launch (-3.4 %, bitwise the same result); the 4-wave C = 256 form gets SLOWER
    // with it (54.0 -> 56.9 us: 246 instead of 208 registers, and its fc2 has only 64 MFMAs per super-chunk to put 480 VALU instructions behind) and keeps the
    // plain loop.  What the loop's parts cost (diagnosis build, wm_ablate; profiles/r06_winmlp_ablation.log): weight loads 11-14 %, the barrier 7-10 %, GELU 8-13 % of
    // a launch; with all three removed 39.8 / 60.1 us remain (C = 256 / 512) = prologue + epilogue + MFMA issue + LDS reads.
#define FVIT_WINMLP_P(T, CC_, HID_, NRB_, NWV_, SP_) \
    hipLaunchKernelGGL((winmlp_kernel<T, CC_, HID_, NRB_, 2, NWV_, SP_, 1, false, true>), dim3(grid), dim3(64 * NWV_), 0, stream, p)
#define FVIT_WINMLP_PT(CC_, HID_, NRB_, NWV_)                                                    \
    do {                                                                                         \
        if (c.dtype == FVIT_F16) { if (c.terms == 2) FVIT_WINMLP_P(_Float16, CC_, HID_, NRB_, NWV_, 2); else FVIT_WINMLP_P(_Float16, CC_, HID_, NRB_, NWV_, 1); } \
        else { if (c.terms == 2) FVIT_WINMLP_P(__bf16, CC_, HID_, NRB_, NWV_, 2); else FVIT_WINMLP_P(__bf16, CC_, HID_, NRB_, NWV_, 1); } \
    } while (0)
    const int pipe = tune_get("win_mlp_pipe", 1);
#define FVIT_WINMLP_S(T, SP_, NS_) hipLaunchKernelGGL((winmlp_kernel<T, 512, 2048, 4, 2, 8, SP_, NS_>), dim3(grid), dim3(512), 0, stream, p)
#define FVIT_WINMLP_ST(NS_)                                                                      \
    do {                                                                                         \
        if (c.dtype == FVIT_F16) { if (c.terms == 2) FVIT_WINMLP_S(_Float16, 2, NS_); else FVIT_WINMLP_S(_Float16, 1, NS_); } \
        else { if (c.terms == 2) FVIT_WINMLP_S(__bf16, 2, NS_); else FVIT_WINMLP_S(__bf16, 1, NS_); } \
    } while (0)
#ifdef FVIT_DIAG
    if (c.ts && c.dtype == FVIT_F16 && c.terms == 1 && nsplit == 1) {   // timeline instances (fvit_debug_win_mlp_timeline: the diagnosis build only)
        if (c.C == 512) hipLaunchKernelGGL((winmlp_kernel<_Float16, 512, 2048, 4, 2, 8, 1, 1, true>), dim3(grid), dim3(512), 0, stream, p);
        else hipLaunchKernelGGL((winmlp_kernel<_Float16, 256, 1024, 4, 2, 4, 1, 1, true>), dim3(grid), dim3(256), 0, stream, p);
    } else
#endif
    if (c.C == 512 && nsplit == 2) FVIT_WINMLP_ST(2);
    // (a 4-deep ring for C = 512 needs 105 spilled registers at 8 waves x 256: not instantiated)
    else if (c.C == 512 && pipe) FVIT_WINMLP_PT(512, 2048, 4, 8);
    else if (c.C == 512) FVIT_WINMLP_T(512, 2048, 4, 8);
    else if (wide256) FVIT_WINMLP_T(256, 1024, 8, 8);
    else FVIT_WINMLP_T(256, 1024, 4, 4);
#undef FVIT_WINMLP_ST
#undef FVIT_WINMLP_S
#undef FVIT_WINMLP_PT
#undef FVIT_WINMLP_P
#undef FVIT_WINMLP_T
#undef FVIT_WINMLP
    return check_launch("winmlp_kernel");
}

#endif  // FVIT_BODIES_ONLY

}  // namespace fvit
