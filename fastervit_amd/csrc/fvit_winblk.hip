// fvit_winblk.hip -- attention sub-block of a HAT block for C = 512 / 16 heads of 32, one window of 49..64 tokens per workgroup (stage 3
// of FasterViT-0), with the waves splitting the N dimension (gfx950):
//
//   x_out = x_in + gamma * proj( softmax( q k^T * scale + bias ) v ),   [q|k|v] = qkv( LayerNorm(x_in) ),   x_in = row (+ pos-embed row)
//
// Same contract as fvit_attn_block_fused (AR:671-696 / FV:665-690).  The per-row-block form of that kernel (fvit_attnblk.hip, every wave
// 16 rows and ALL weights through LDS) loses at this shape: a workgroup streams 2 MiB of weights for 49 rows in lockstep.  Unfused it is
// gather-LayerNorm + qkv GEMM + attention + proj GEMM: four chip-wide launches of 264-1054 workgroups for 4 214 rows per stream shard.
//
// Here (the work split of fvit_ctblk.hip): a workgroup of 8 waves owns one window = 4 row blocks of 16 tokens.
//   A  LayerNorm of the 64 row slots into MFMA B-operand fragments in LDS (XN[rb][kk], 64 KiB), wave w: row block w >> 1, k steps of half w & 1.
//   B  wave w computes heads 2w, 2w + 1 for ALL four row blocks: its 96 KiB qkv weight slice per head comes straight from L2 into a
//      register ring (one step = the six q0 q1 k0 k1 v0 v1 fragments of one k step), XN fragments are read once per k step and feed six
//      MFMAs each; the wave's accumulators are its own Q / K / V operands (no exchange); the head's bias tiles ride in the same ring.
//   C  the normalised O^T fragments of all heads are exchanged through LDS, then wave w computes output channels
//      64w .. 64w + 63 for all rows (proj weights of its 4 channel blocks through the ring) and applies the residual.
// Every weight fragment is read from L2 exactly once per workgroup and feeds four MFMAs.
//
// Measured (FasterViT-0 stage 3, 86 windows, r02 calls r3v / r3w): 55.6 us per launch -- NOT faster than the four unfused launches it replaces
// (8.8 + 18.2 + 11.8 + ~15 us; stage-3 forward alone 504 -> 540 us): the weights of a block are cold and one CU pulls only ~65 GB/s from
// the memory side (scripts/probes/regstream_probe.hip), 2.3 MB per workgroup.  But it occupies 86 CUs instead of the whole chip four times,
// and with three stream shards sharing the GPU that is what counts: +0.7 % / +1.9 % images/s end to end on two boxes.  On by default
// (fvit_tune "win_fused").
// The C = 256 / 8-head instance for stage 2 (4 waves, two workgroups per CU; fvit_tune "win_fused256", off) is bitwise the same result as
// attnblk_kernel<256> (same summation orders) and loses to it: 767-772 vs 755-774 us per stage-2 forward, 77.4k-78.0k vs 79.2k images/s
// (call r4l): 344 window workgroups are a chip-wide launch either way, and attnblk's per-row-block waves need no O^T exchange.
#include "fvit_common.h"

namespace fvit {

namespace {

struct WinBlkParams {
    const float* srcA;
    const float* srcB;
    const int32_t* src_idx;
    const int32_t* add_idx;
    const float* add;
    const float* ln_w;
    const float* ln_b;
    float eps;
    int rowsA, rowsB, rows_per_image;
    const void* wqkv_f;   // op16 [heads][6][C/32][64][8]
    const float* bqkv;    // f32  [heads][96]
    const void* wproj_f;  // op16 [heads][C/16][64][8]
    const float* bproj;
    const float* gamma;
    const float* bias;    // f32 [heads][64][64]
    float* x_out;
    int nwin, S;
    float scale;
    float* slab;     // NSPLIT > 1: f32 partial outputs [window][NSPLIT][waves][16][64 lanes][4] (64 x C x 4 bytes per (window, split))
    int* counters;   // NSPLIT > 1: one arrival counter per window, zero before the launch, zero again after it
};

// CC = 512: 8 waves, one workgroup per CU.  CC = 256 (stage 2, 8 heads): 4 waves (two heads and four channel blocks each, like the 8-wave
// form), <= 256 registers and 67 KiB of LDS, so two workgroups share a CU and their phases interleave.
// NSPLIT = 2 (r03, C = 512): the 16 heads of a window are split over two sibling workgroups (8 heads each, one per wave): each streams
// half of the qkv / bias / proj weights (1.15 MiB instead of 2.3 MiB through one CU's L2 port), computes the proj partial sum over ITS
// heads for all C channels, and the two partials meet in L2 exactly as in winmlp_kernel (fp32 partial stored, agent-scope release,
// ticket; the last arriver acquires, adds the partials in split order and applies the residual; nobody waits).
// WT = 2 (r03): two-term weights (hi + lo 16-bit images, the lo image after the hi image in the fragment arrays): the qkv and proj k loops run
// twice over the same activation fragments, once per weight image -- same registers and LDS, twice the weight stream.
// LDS bytes of a workgroup: XN (4 row blocks x C / 32 k steps), O^T (4 x heads of this workgroup), the qkv bias copy
template <int CC, int NSPLIT>
constexpr int winblk_lds_bytes() { return 4 * (CC / 32) * 1024 + 4 * (CC / 32 / NSPLIT) * 1024 + (CC / 32) * 96 * 4; }

// The kernel body as a device function (blk = blockIdx.x of a stand-alone launch): fvit_stage3.hip runs it as one phase of a persistent workgroup.
template <typename T, int CC, int NWV, int NSPLIT = 1, int WT = 1>
__device__ __forceinline__ void winblk_body(const WinBlkParams& p, char* const smem, const int blk) {
    typedef typename Op16<T>::v8 v8;
    constexpr int C = CC, KK = C / 32, CB = C / 16, HEADS = C / 32, NW = NWV, NRB = 4, SP = 64;
    constexpr int HW = HEADS / NSPLIT;             // heads of this workgroup
    constexpr int NH = HW / NW;                    // heads per wave (2, or 1 when split)
    static_assert(HW == NH * NW && NH >= 1 && CB == 4 * NW, "heads split evenly over the waves; four output channel blocks per wave");
    constexpr int KQ = KK * WT, PW = HW * WT;      // qkv steps per head / proj steps, over the weight terms
    constexpr int SPH = KQ + 4;                    // steps per head: KQ qkv steps + 4 bias-tile steps
    constexpr size_t QKV_IMG = (size_t)3 * C * C * 2, PROJ_IMG = (size_t)C * C * 2;   // bytes of one weight image
    constexpr int DEPTH = 2;                       // ring slots of 6 fragments (6 KiB) per wave (3 slots spill at the 256-register budget of 8 waves)
    constexpr int OFF_O = NRB * KK * 1024;         // XN: 64 KiB, then O: 64 KiB (a separate region: each head's O^T fragments leave the
    constexpr int OFF_BQ = OFF_O + NRB * HW * 1024;      // registers at once -- held across the next head they spilled, and a scratch reload
                                                         // inside the loop queues behind the ring's prefetches and drains it)
    static_assert(OFF_BQ + HEADS * 96 * 4 == winblk_lds_bytes<CC, NSPLIT>(), "LDS layout");
    float* bqs = (float*)(smem + OFF_BQ);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, s = lane & 15;
    const int lane16 = lane * 16;
    int win = blk, sp = 0;
    if constexpr (NSPLIT > 1) {
        const int xcd = blk & 7, j = blk >> 3;   // siblings: block ids that differ by a multiple of 8 (same XCD, speed only)
        win = (j / NSPLIT) * 8 + xcd;
        sp = j % NSPLIT;
        if (win >= p.nwin) return;
    }
    const int h0 = sp * HW;                         // first head of this workgroup

    // ---- the wave's stream: head hh of {2w, 2w+1}: 16 k steps of 6 qkv fragments, then 16 bias tiles (4 query blocks x 4 key blocks);
    //      then 16 heads x 4 proj fragments (this wave's channel blocks).  Steps are addressed by a running index. ----
    const char* Wq = (const char*)p.wqkv_f + lane16;
    const char* Wp = (const char*)p.wproj_f + lane16;
    const char* Bz = (const char*)p.bias + (s * SP + g * 4) * 4;   // tile (qb, kb) of head h: + ((h * 64 + qb * 16) * 64 + kb * 16) * 4
    v8 ring[DEPTH][6];
    // step kinds: QKV(h, kk): 6 fragments; BIAS(h, qb): 4 tiles (kb = 0..3) in slots 0..3; PROJ(h): 4 fragments in slots 0..3
    auto load_qkv = [&](int slot, int h, int u) {   // u = term * KK + kk
        const char* base = Wq + (u / KK) * QKV_IMG;
        const int kk = u % KK;
#pragma unroll
        for (int ub = 0; ub < 6; ++ub) ring[slot][ub] = *(const v8*)(base + (((size_t)h * 6 + ub) * KK + kk) * 1024);
    };
    auto load_bias = [&](int slot, int h, int qb) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) ring[slot][kb] = *(const v8*)(Bz + ((size_t)(h * SP + qb * 16) * SP + kb * 16) * 4);
    };
    auto load_proj = [&](int slot, int hp) {        // hp = term * HW + local head
        const char* base = Wp + (hp / HW) * PROJ_IMG;
        const int h = h0 + hp % HW;
#pragma unroll
        for (int q = 0; q < 4; ++q) ring[slot][q] = *(const v8*)(base + ((size_t)h * CB + 4 * wave + q) * 1024);
    };
    // per head KK + 4 steps; two heads; then one proj step per head of the layer.  issue(t) requests step t into slot t % DEPTH.
    auto issue = [&](int t) {
        if (t < NH * SPH) {
            const int hh = t / SPH, u = t - hh * SPH, h = h0 + NH * wave + hh;
            if (u < KQ) load_qkv(t % DEPTH, h, u);
            else load_bias(t % DEPTH, h, u - KQ);
        } else if (t < NH * SPH + PW) {
            load_proj(t % DEPTH, t - NH * SPH);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    for (int i = tid; i < HEADS * 96; i += 64 * NW) bqs[i] = p.bqkv[i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < DEPTH; ++t) issue(t);

    // ---- phase A: gather (+ add) and LayerNorm; WPR waves share a row block: each reads the full rows and writes KK / WPR of the k steps ----
    {
        constexpr int WPR = NW / NRB, KP = KK / WPR;
        const int rb = wave / WPR, part = wave % WPR;
        const int tok = rb * 16 + s;
        const int64_t row = (int64_t)win * p.S + (tok < p.S ? tok : p.S - 1);   // clamped: always a real row
        const int b = (int)(row / p.rows_per_image), pr = (int)(row - (int64_t)b * p.rows_per_image);
        const float* src;
        if (p.src_idx) {
            const int si = p.src_idx[pr];
            src = si >= 0 ? p.srcA + ((size_t)b * p.rowsA + si) * C : p.srcB + ((size_t)b * p.rowsB + (-si - 1)) * C;
        } else {
            src = p.srcA + (size_t)row * C;
        }
        const int ai = p.add ? (p.add_idx ? p.add_idx[pr] : pr) : -1;
        const bool has_add = ai >= 0;
        const float* addp = has_add ? p.add + (size_t)ai * C : src;
        f4 v[2 * KK];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 2 * KK; ++i) {
            const int co = (i >> 2) * 64 + g * 16 + (i & 3) * 4;   // i = 2 * kk + h2
            f4 t = *(const f4*)(src + co);
            const f4 a = *(const f4*)(addp + co);
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] += has_add ? a[r] : 0.f;
            v[i] = t;
            sum += (t[0] + t[1]) + (t[2] + t[3]);
        }
        sum = sum_xor32(sum_xor16(sum));
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
    __syncthreads();

    // ---- phase B: heads h0 + NH w .. + NH - 1 ----
    const char* xn = smem + lane16;
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
        const int h = h0 + NH * wave + hh;
        f4 acc[6][NRB];
#pragma unroll
        for (int ub = 0; ub < 6; ++ub)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[ub][rb] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < KQ; ++u) {
            const int t = hh * SPH + u, kk = u % KK;
            v8 xb[NRB];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) xb[rb] = *(const v8*)(xn + (rb * KK + kk) * 1024);
#pragma unroll
            for (int ub = 0; ub < 6; ++ub)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
                    acc[ub][rb] = ub < 4 ? Op16<T>::mfma(ring[t % DEPTH][ub], xb[rb], acc[ub][rb])     // q0 q1 k0 k1: D[dim][token]
                                         : Op16<T>::mfma(xb[rb], ring[t % DEPTH][ub], acc[ub][rb]);    // v0 v1: D[token][dim]
            __builtin_amdgcn_sched_barrier(0);
            issue(t + DEPTH);
        }
        // accumulators -> operand fragments
        const float* bq = bqs + h * 96;
        v8 qf[NRB], kf[NRB], vf[2][2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const f4 bqv = *(const f4*)(bq + blk * 16 + g * 4);
            const f4 bkv = *(const f4*)(bq + 32 + blk * 16 + g * 4);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    qf[rb][blk * 4 + r] = sat16<T>(acc[blk][rb][r] + bqv[r]);
                    kf[rb][blk * 4 + r] = sat16<T>(acc[2 + blk][rb][r] + bkv[r]);
                }
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const float bv = bq[64 + db * 16 + s];
#pragma unroll
            for (int k32 = 0; k32 < 2; ++k32)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vf[db][k32][r] = sat16<T>(acc[4 + db][2 * k32][r] + bv);          // keys 32 k32 + 4g + r
                    vf[db][k32][4 + r] = sat16<T>(acc[4 + db][2 * k32 + 1][r] + bv);  // keys 32 k32 + 16 + 4g + r
                }
        }
        // scores^T, softmax over keys, O^T, per query row block; the bias tiles of (h, qb) are step hh * SPH + KK + qb
#pragma unroll
        for (int qb = 0; qb < NRB; ++qb) {
            const int t = hh * SPH + KQ + qb;
            f4 sc[NRB];
            float mx = -3.0e38f;
#pragma unroll
            for (int kb = 0; kb < NRB; ++kb) {
                f4 a = Op16<T>::mfma(kf[kb], qf[qb], (f4){0.f, 0.f, 0.f, 0.f});
                const f4 bz = __builtin_bit_cast(f4, ring[t % DEPTH][kb]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a[r] = a[r] * p.scale + bz[r];
                    mx = fmaxf(mx, a[r]);
                }
                sc[kb] = a;
            }
            __builtin_amdgcn_sched_barrier(0);
            issue(t + DEPTH);
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
            for (int k32 = 0; k32 < 2; ++k32) {
                v8 pf;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf[r] = (T)sc[2 * k32][r];
                    pf[4 + r] = (T)sc[2 * k32 + 1][r];
                }
#pragma unroll
                for (int db = 0; db < 2; ++db) o[db] = Op16<T>::mfma(vf[db][k32], pf, o[db]);
            }
            v8 of;   // normalised O^T fragment of (head h, query block qb) -> O[qb][h] for phase C
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                of[r] = sat16<T>(o[0][r] * inv);
                of[4 + r] = sat16<T>(o[1][r] * inv);
            }
            *(v8*)(smem + OFF_O + ((qb * HW + (h - h0)) * 1024) + lane16) = of;
        }
    }
    __syncthreads();   // O^T fragments of all heads visible

    // ---- phase C: output channels 64 * wave .. + 63 (channel blocks 4w .. 4w + 3) for all rows: out^T += Wproj[:, head] . O^T ----
    f4 oacc[4][NRB];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) oacc[q][rb] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int hp = 0; hp < PW; ++hp) {   // local head index (per weight term); the ring step holds the proj fragments of head h0 + h
        const int t = NH * SPH + hp, h = hp % HW;
        v8 ob[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) ob[rb] = *(const v8*)(xn + OFF_O + (rb * HW + h) * 1024);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) oacc[q][rb] = Op16<T>::mfma(ring[t % DEPTH][q], ob[rb], oacc[q][rb]);
        __builtin_amdgcn_sched_barrier(0);
        issue(t + DEPTH);
    }

    if constexpr (NSPLIT > 1) {
        // ---- the two head halves meet in L2; the last arriver finishes the window (see winmlp_kernel) ----
        f4* const slab_w = (f4*)p.slab + (size_t)win * NSPLIT * NW * 16 * 64;
        {
            f4* mine = slab_w + ((size_t)sp * NW + wave) * 16 * 64 + lane;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) mine[(q * NRB + rb) * 64] = oacc[q][rb];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* const tick = (int*)(smem + OFF_BQ);   // the qkv bias copy is dead after phase B
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            *tick = __hip_atomic_fetch_add(p.counters + win, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (*tick != NSPLIT - 1) return;
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            p.counters[win] = 0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f4 part[NSPLIT][NRB];
#pragma unroll
            for (int s2 = 0; s2 < NSPLIT; ++s2)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) part[s2][rb] = slab_w[(((size_t)s2 * NW + wave) * 16 + q * NRB + rb) * 64 + lane];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                f4 sum = part[0][rb];
#pragma unroll
                for (int s2 = 1; s2 < NSPLIT; ++s2) sum += part[s2][rb];   // fixed order
                oacc[q][rb] = sum;
            }
        }
    }

    // ---- epilogue: x_out[row][c] = x_in + gamma * (out + bproj); fragment (4w + q), slot 4g + r <-> channel 64w + 16g + 4q + r, row rb * 16 + s ----
    // (r03 tried the two-pass form of winmlp_kernel here -- all row / add loads first: no shorter launch, 57.4 vs 57.8 us, and at 256 registers it
    // tipped instances into scratch; the r02 loop stays)
    const bool has_g = p.gamma != nullptr;
    const int c0 = wave * 64 + g * 16;
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
        const int tok = rb * 16 + s;
        if (tok < p.S) {
            const int64_t row = (int64_t)win * p.S + tok;
            const int b = (int)(row / p.rows_per_image), pr = (int)(row - (int64_t)b * p.rows_per_image);
            const float* src;
            if (p.src_idx) {
                const int si = p.src_idx[pr];
                src = si >= 0 ? p.srcA + ((size_t)b * p.rowsA + si) * C : p.srcB + ((size_t)b * p.rowsB + (-si - 1)) * C;
            } else {
                src = p.srcA + (size_t)row * C;
            }
            const int ai = p.add ? (p.add_idx ? p.add_idx[pr] : pr) : -1;
            float* px = p.x_out + (size_t)row * C + c0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f4 xv = *(const f4*)(src + c0 + q * 4);
                if (ai >= 0) xv += *(const f4*)(p.add + (size_t)ai * C + c0 + q * 4);
                const f4 bv = *(const f4*)(p.bproj + c0 + q * 4);
                const f4 gl = *(const f4*)((has_g ? p.gamma : p.bproj) + c0 + q * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) xv[r] += (has_g ? gl[r] : 1.f) * (oacc[q][rb][r] + bv[r]);
                *(f4*)(px + q * 4) = xv;
            }
        }
    }
}

template <typename T, int CC, int NWV, int NSPLIT = 1, int WT = 1>
__global__ __launch_bounds__(64 * NWV, NWV == 8 ? 1 : 2) void winblk_kernel(WinBlkParams p) {
    __shared__ __attribute__((aligned(16))) char smem[winblk_lds_bytes<CC, NSPLIT>()];
    winblk_body<T, CC, NWV, NSPLIT, WT>(p, smem, blockIdx.x);
}

inline WinBlkParams make_winblk_params(const AttnBlkCall& c) {
    WinBlkParams p;
    p.srcA = c.srcA; p.srcB = c.srcB; p.src_idx = c.src_idx; p.add_idx = c.add_idx; p.add = c.add; p.ln_w = c.ln_w; p.ln_b = c.ln_b; p.eps = c.eps;
    p.rowsA = c.rowsA; p.rowsB = c.rowsB; p.rows_per_image = c.rows_per_image > 0 ? c.rows_per_image : 1;
    p.wqkv_f = c.wqkv_f; p.bqkv = c.bqkv; p.wproj_f = c.wproj_f; p.bproj = c.bproj; p.gamma = c.gamma; p.bias = c.bias; p.x_out = c.x_out;
    p.nwin = c.nwin; p.S = c.S; p.scale = c.scale;
    p.slab = c.slab; p.counters = c.counters;
    return p;
}

}  // namespace

#ifndef FVIT_BODIES_ONLY
bool winblk_supported(int C, int heads, int S) { return ((C == 512 && heads == 16) || (C == 256 && heads == 8)) && S > 48 && S <= 64; }

int launch_winblk(const AttnBlkCall& c, hipStream_t stream) {
    if (!winblk_supported(c.C, c.heads, c.S) || c.nwin <= 0 || !c.wqkv_f || !c.wproj_f || !c.x_out || !c.bias || !c.bqkv) {
        set_error("win_block: unsupported arguments C=%d heads=%d S=%d nwin=%d", c.C, c.heads, c.S, c.nwin);
        return FVIT_EINVAL;
    }
    if (ablate_skip(4)) return FVIT_OK;
    const WinBlkParams p = make_winblk_params(c);
    if (c.terms != 1 && !(c.terms == 2 && c.C == 512)) {
        set_error("win_block: weight terms %d with C = %d (two-term weights: C = 512 only)", c.terms, c.C);
        return FVIT_EINVAL;
    }
    const bool split = c.C == 512 && c.terms == 1 && c.nsplit == 2 && c.slab && c.counters;
    const double rows = (double)c.nwin * c.S;
    const double flops = rows * (2.0 * c.C * 3 * c.C + 4.0 * c.S * c.C + 2.0 * c.C * c.C);
    const double bytes = rows * c.C * 8.0 + c.terms * 2.0 * 4.0 * c.C * c.C;
    ProfScope prof(FVIT_K_ATTN_FUSED, flops, bytes, stream);
    const int grid = split ? (c.nwin + 7) / 8 * 8 * 2 : c.nwin;
    prof_note(c.C == 512 ? (split ? "winblk_kernel<512,S64,split2>" : "winblk_kernel<512,S64>") : "winblk_kernel<256,S64>", grid);
    if (c.dtype != FVIT_F16 && c.dtype != FVIT_BF16) { set_error("win_block: operand dtype %d not supported", c.dtype); return FVIT_EINVAL; }
    if (c.terms == 2) {
        if (c.dtype == FVIT_F16) hipLaunchKernelGGL((winblk_kernel<_Float16, 512, 8, 1, 2>), dim3(c.nwin), dim3(512), 0, stream, p);
        else hipLaunchKernelGGL((winblk_kernel<__bf16, 512, 8, 1, 2>), dim3(c.nwin), dim3(512), 0, stream, p);
    } else if (split) {
        if (c.dtype == FVIT_F16) hipLaunchKernelGGL((winblk_kernel<_Float16, 512, 8, 2>), dim3(grid), dim3(512), 0, stream, p);
        else hipLaunchKernelGGL((winblk_kernel<__bf16, 512, 8, 2>), dim3(grid), dim3(512), 0, stream, p);
    } else if (c.C == 512) {
        if (c.dtype == FVIT_F16) hipLaunchKernelGGL((winblk_kernel<_Float16, 512, 8>), dim3(c.nwin), dim3(512), 0, stream, p);
        else hipLaunchKernelGGL((winblk_kernel<__bf16, 512, 8>), dim3(c.nwin), dim3(512), 0, stream, p);
    } else {
        if (c.dtype == FVIT_F16) hipLaunchKernelGGL((winblk_kernel<_Float16, 256, 4>), dim3(c.nwin), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((winblk_kernel<__bf16, 256, 4>), dim3(c.nwin), dim3(256), 0, stream, p);
    }
    return check_launch("winblk_kernel");
}

#endif  // FVIT_BODIES_ONLY

}  // namespace fvit
