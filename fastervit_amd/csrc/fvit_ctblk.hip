// fvit_ctblk.hip -- the whole carrier-token branch of one HAT block in ONE kernel (C = 256, 8 heads of 32, hidden 1024, <= 16 carrier
// tokens per image: stage 2 of FasterViT-0), gfx950:
//
//   ct  = ct_dewindow(carrier rows of X) (+ hat_pos_embed)                         (AR:679-682)
//   ct += gamma1 * hat_attn(hat_norm1(ct));  ct += gamma2 * hat_mlp(hat_norm2(ct)) (AR:685-686)   -> R [B * G][C] fp32
//
// Unfused this is gather-LayerNorm, qkv GEMM, attention, proj GEMM, LayerNorm, fc1 GEMM, fc2 GEMM on 1360 rows per stream shard: seven
// launches of 42-176 workgroups that each run a K loop of dependent LDS-DMA round trips (9-19 us each, 61 us per block, 0.04-0.05 of
// their roof: r02 bench line).  The per-row-block fused kernels (fvit_attnblk / fvit_mlp) lose here as well: a workgroup streams ALL
// 1.5 MB of weights through LDS in lockstep for 64 rows.
//
// Measured (FasterViT-0 stage 2, 86 images): 33 us per launch instead of ~59 us in seven launches; 826 -> 776 us per stage forward,
// 73.5k -> 76.2k images/s end to end (r02 calls r3l / r3o).
//
// Here a workgroup of 4 waves owns ONE image (one 16-token row block) and the waves split the N dimension: wave w computes heads 2w,
// 2w + 1 of the attention and hidden units 256w .. 256w + 255 of the MLP, so each wave reads only ITS quarter of the weights (384 KiB),
// straight from L2 into registers in MFMA fragment order (the packed images of fvit_attnblk / fvit_mlp: a fragment = 64 lanes x 16 B,
// one global_load_dwordx4 per lane), through a 3-deep register ring of 8-fragment steps (24 KiB in flight per wave); every fragment
// feeds exactly one MFMA, nothing is staged in LDS.  Partial sums over heads (proj) and over hidden units (fc2) are exchanged once each
// through LDS (fp32, 4 x 16 KiB) and added in a fixed order.  All lane reductions are VALU swaps (fvit_common.h).
#include "fvit_common.h"

namespace fvit {

namespace {

struct CtBlkParams {
    const float* X;          // window tensor rows [B * rowsA][C]
    const int32_t* src_idx;  // [G] row of X (inside the image) of carrier token i in raster order
    const float* add;        // hat_pos_embed rows [G][C] or null
    float* R;                // out [B * G][C]
    int rowsA, B, G;
    const float* ln1_w; const float* ln1_b;
    const void* wqkv_f;      // op16 [heads][6][C/32][64][8]
    const float* bqkv;       // f32  [heads][96]
    const void* wproj_f;     // op16 [heads][C/16][64][8]
    const float* bproj; const float* gamma1;
    const float* bias;       // f32 [heads][16][16] (mask on padded keys)
    float scale;
    const float* ln2_w; const float* ln2_b;
    const void* w1f;         // op16 [hidden/32][2][C/32][64][8]
    const float* b1;
    const void* w2f;         // op16 [hidden/32][C/16][64][8]
    const float* b2; const float* gamma2;
    float eps;
    int touch;
    unsigned long long* ts;   // ctblk8_kernel TS instance (fvit_debug_ct_block_timeline): s_memtime stamps [image][wave][16]
};

// DEPTH: steps of the register ring in flight; MINB: workgroups per CU the register budget is sized for (1: one wave per SIMD, 512
// registers, the ring and every phase's operands fit without scratch; 2: 256 registers, other kernels' waves can share the SIMD)
// WT = 2 (r03): two-term weights (hi + lo 16-bit images, the lo image after the hi image in each fragment array): every 8-fragment step of
// the stream is followed by the same step of the lo image and both accumulate into the same registers before the narrowing.
template <typename T, int DEPTH, int MINB, int WT = 1>
__global__ __launch_bounds__(256, MINB) void ctblk_kernel(CtBlkParams p) {
    typedef typename Op16<T>::v8 v8;
    typedef typename Op16<T>::v4 v4;
    constexpr int C = 256, KK = 8, CB = 16, NW = 4, HID = 1024;
    constexpr int CPW = HID / 32 / NW;        // hidden chunks (of 32 units) per wave
    constexpr int SA = 16 * WT;               // steps of the attention phase
    constexpr int NSTEP = SA + 4 * WT * CPW;  // 8-fragment steps of a wave's weight stream: 2 heads x (6 qkv + 2 proj), CPW chunks x (2 fc1 + 2 fc2), x WT
    constexpr size_t QKV_IMG = (size_t)3 * C * C * 2, PROJ_IMG = (size_t)C * C * 2, FC_IMG = (size_t)C * HID * 2;   // bytes of one weight image
    // partial accumulators [wave][cb][lane] x 16 B, then the small constants of the inner loops (fc1 bias, qkv bias, attention bias tables):
    // an ordinary global load inside the loops would queue behind the ring's prefetches in the in-order vmcnt counter, and waiting for it
    // would drain the ring (first version: 56 us per launch instead of ~20)
    constexpr int OFF_B1 = NW * CB * 1024, OFF_BQ = OFF_B1 + HID * 4, OFF_BZ = OFF_BQ + 8 * 96 * 4;
    __shared__ __attribute__((aligned(16))) char smem[OFF_BZ + 8 * 256 * 4];
    float* b1s = (float*)(smem + OFF_B1);
    float* bqs = (float*)(smem + OFF_BQ);
    float* bzs = (float*)(smem + OFF_BZ);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, s = lane & 15;
    const int lane16 = lane * 16;
    const int img = blockIdx.x;
    const int tok = s < p.G ? s : p.G - 1;    // padded tokens recompute the last one; masked as keys by the bias table, never stored
    const bool row_ok = s < p.G;

    const char* Wq = (const char*)p.wqkv_f + lane16;
    const char* Wp = (const char*)p.wproj_f + lane16;
    const char* W1 = (const char*)p.w1f + lane16;
    const char* W2 = (const char*)p.w2f + lane16;
    auto step_ptr = [&](int t) -> const char* {
        if (t < SA) {
            const int h = 2 * wave + t / (8 * WT), r = t % (8 * WT), u = r / WT, term = r % WT;
            return u < 6 ? Wq + term * QKV_IMG + ((size_t)h * 48 + u * 8) * 1024 : Wp + term * PROJ_IMG + ((size_t)h * 16 + (u - 6) * 8) * 1024;
        }
        const int m = t - SA, j = CPW * wave + m / (4 * WT), r = m % (4 * WT), u = r / WT, term = r % WT;
        return (u < 2 ? W1 : W2) + term * FC_IMG + ((size_t)j * 16 + (u & 1) * 8) * 1024;
    };
    v8 ring[DEPTH][8];
    constexpr bool in_attention = true;   // shadowed in the MLP phase: steps >= 16 are requested only after the exchange
#define FVIT_CT_LOAD(t)                                                                                  \
    if ((t) < NSTEP && !(in_attention && (t) >= SA)) {                                                                                   \
        const char* sp_ = step_ptr(t);                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) ring[(t) % DEPTH][i_] = *(const v8*)(sp_ + i_ * 1024); \
    }                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);

    // Optional L2 touch (fvit_tune "ct_touch", off): one dword per 128-byte line of the wave's whole stream, all 48 loads in flight at once,
    // before the ring starts.  The idea: a block's weights are cold (60 MB of weights per model against 4 MiB of L2 per XCD) and the
    // workgroups of an XCD want the same lines at the same time.  Measured (r02 call r3o): 40.0 vs 33.4 us per launch -- the kernel runs at the
    // ~40 GB/s a CU pulls from the memory side whatever the ring depth (1.5 MB per workgroup; DEPTH 2 / 3 / 4: 794 / 776 / 786 us per stage),
    // and the touch pays that rate once more.
    if (p.touch) {
        // plain C++ loads folded into a value the compiler cannot discard (an asm load with a dead output would let the register
        // allocator reuse the destination while the load is still in flight)
        unsigned sink = 0;
        const unsigned* tq = (const unsigned*)((const char*)p.wqkv_f + (size_t)(2 * wave) * 48 * 1024 + lane * 128);
        const unsigned* tp = (const unsigned*)((const char*)p.wproj_f + (size_t)(2 * wave) * 16 * 1024 + lane * 128);
        const unsigned* t1 = (const unsigned*)((const char*)p.w1f + (size_t)(CPW * wave) * 16 * 1024 + lane * 128);
        const unsigned* t2 = (const unsigned*)((const char*)p.w2f + (size_t)(CPW * wave) * 16 * 1024 + lane * 128);
#pragma unroll
        for (int i = 0; i < 12; ++i) sink ^= tq[i * 2048];
#pragma unroll
        for (int i = 0; i < 4; ++i) sink ^= tp[i * 2048];
#pragma unroll
        for (int i = 0; i < 2 * CPW; ++i) sink ^= t1[i * 2048];
#pragma unroll
        for (int i = 0; i < 2 * CPW; ++i) sink ^= t2[i * 2048];
        if (sink == 0x9E3779B1u && p.touch == 12345) p.R[0] = 0.f;   // never true: keeps the loads
    }
    {   // constants first (oldest in the vmcnt queue), then the first DEPTH steps of the weight stream
        float c1[HID / 256], c2[3], c3[8];
#pragma unroll
        for (int i = 0; i < HID / 256; ++i) c1[i] = p.b1[tid + 256 * i];
#pragma unroll
        for (int i = 0; i < 3; ++i) c2[i] = p.bqkv[tid + 256 * i];
#pragma unroll
        for (int i = 0; i < 8; ++i) c3[i] = p.bias[tid + 256 * i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < DEPTH; ++t) { FVIT_CT_LOAD(t) }
#pragma unroll
        for (int i = 0; i < HID / 256; ++i) b1s[tid + 256 * i] = c1[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) bqs[tid + 256 * i] = c2[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) bzs[tid + 256 * i] = c3[i];
    }

    // ---- gather (+ position embedding) and LayerNorm: lane (g, s) holds token s, channels (cb>>2)*64 + 16g + (cb&3)*4 .. +3 in v[cb] ----
    const float* src = p.X + ((size_t)img * p.rowsA + p.src_idx[tok]) * C + g * 16;
    // optional inputs are read through a valid stand-in pointer and masked with a scalar select: a branch per load splits the phase into
    // basic blocks and the partial sums spill across them
    const bool has_add = p.add != nullptr, has_g1 = p.gamma1 != nullptr, has_g2 = p.gamma2 != nullptr;
    const float* addp = has_add ? p.add + (size_t)tok * C + g * 16 : src;
    auto gather = [&](f4 (&v)[CB]) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) v[cb] = *(const f4*)(src + (cb >> 2) * 64 + (cb & 3) * 4);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const f4 a = *(const f4*)(addp + (cb >> 2) * 64 + (cb & 3) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[cb][r] += has_add ? a[r] : 0.f;
        }
    };
    // v[cb] = channels of GEMM k step kk = cb >> 1, half cb & 1 (kch order): xf[kk] = normalised (v[2kk], v[2kk+1])
    auto layernorm = [&](const f4 (&v)[CB], const float* lw, const float* lb, v8 (&xf)[KK]) {
        float sum = 0.f;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) sum += (v[cb][0] + v[cb][1]) + (v[cb][2] + v[cb][3]);
        sum = sum_xor32(sum_xor16(sum));
        const float mean = sum / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const f4 d = v[cb] - mean;
            sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        sq = sum_xor32(sum_xor16(sq));
        const float rstd = rsqrtf(sq / (float)C + p.eps);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            v8 o;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int cb = 2 * kk + h2;
                const int co = (cb >> 2) * 64 + g * 16 + (cb & 3) * 4;
                const f4 w = *(const f4*)(lw + co);
                const f4 b = *(const f4*)(lb + co);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[h2 * 4 + r] = sat16<T>((v[cb][r] - mean) * rstd * w[r] + b[r]);
            }
            xf[kk] = o;
        }
    };

    v8 xf[KK];
    {
        f4 v[CB];
        gather(v);
        layernorm(v, p.ln1_w, p.ln1_b, xf);
    }

    __syncthreads();   // constants visible (plain loads in flight survive the barrier)
    // ---- attention: heads 2 * wave, 2 * wave + 1; out^T partial over these heads in oacc ----
    f4 oacc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) oacc[cb] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int h = 2 * wave + hh;
        const float* bq = bqs + h * 96;
        v8 qf, kf, vf[2];
#pragma unroll
        for (int ub = 0; ub < 6; ++ub) {
            // two independent accumulator chains (even / odd k steps): eight dependent MFMAs would wait 32 cycles each on the previous one
            f4 a = (f4){0.f, 0.f, 0.f, 0.f}, ao = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int term = 0; term < WT; ++term) {
                const int t = (hh * 8 + ub) * WT + term;
#pragma unroll
                for (int kk = 0; kk < KK; kk += 2) {
                    a = ub < 4 ? Op16<T>::mfma(ring[t % DEPTH][kk], xf[kk], a)     // q0 q1 k0 k1: weights are the A operand -> D[dim][token]
                               : Op16<T>::mfma(xf[kk], ring[t % DEPTH][kk], a);    // v0 v1: activations are A -> D[token][dim]
                    ao = ub < 4 ? Op16<T>::mfma(ring[t % DEPTH][kk + 1], xf[kk + 1], ao) : Op16<T>::mfma(xf[kk + 1], ring[t % DEPTH][kk + 1], ao);
                }
                __builtin_amdgcn_sched_barrier(0);
                FVIT_CT_LOAD(t + DEPTH)
            }
            a += ao;
            // the accumulator becomes an operand fragment at once (keeps 4 instead of 24 accumulator registers alive)
            if (ub < 4) {
                const f4 bb = *(const f4*)(bq + (ub >> 1) * 32 + (ub & 1) * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (ub < 2) qf[(ub & 1) * 4 + r] = sat16<T>(a[r] + bb[r]);
                    else kf[(ub & 1) * 4 + r] = sat16<T>(a[r] + bb[r]);
                }
            } else {
                const float bv = bq[64 + (ub - 4) * 16 + s];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vf[ub - 4][r] = sat16<T>(a[r] + bv);   // keys 4g + r of the only 16-key block; k slots 4..7 = keys 16.. do not exist
                    vf[ub - 4][4 + r] = (T)0.f;
                }
            }
        }
        // scores^T[key][query] = K . Q^T * scale + bias, softmax down the key axis (in-lane over r, lane swaps over g)
        f4 sc = Op16<T>::mfma(kf, qf, (f4){0.f, 0.f, 0.f, 0.f});
        const f4 bz = *(const f4*)(bzs + (h * 16 + tok) * 16 + g * 4);
        float mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[r] = sc[r] * p.scale + bz[r];
            mx = fmaxf(mx, sc[r]);
        }
        mx = max_xor32(max_xor16(mx));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[r] = __expf(sc[r] - mx);
            sum += sc[r];
        }
        sum = sum_xor32(sum_xor16(sum));
        const float inv = 1.0f / sum;
        v8 pf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pf[r] = (T)sc[r];
            pf[4 + r] = (T)0.f;
        }
        v8 of;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const f4 o = Op16<T>::mfma(vf[db], pf, (f4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int r = 0; r < 4; ++r) of[db * 4 + r] = sat16<T>(o[r] * inv);
        }
        // out^T += Wproj[:, head h] . O^T
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int term = 0; term < WT; ++term) {
                const int t = (hh * 8 + 6 + half) * WT + term;
#pragma unroll
                for (int i = 0; i < 8; ++i) oacc[half * 8 + i] = Op16<T>::mfma(ring[t % DEPTH][i], of, oacc[half * 8 + i]);
                __builtin_amdgcn_sched_barrier(0);
                FVIT_CT_LOAD(t + DEPTH)
            }
    }

    // ---- exchange 1: sum of the four head-pair partials in a fixed order; ct1 = ct0 + gamma1 * (sum + bproj) ----
    char* ex = smem + lane16;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) *(f4*)(ex + (wave * CB + cb) * 1024) = oacc[cb];
    __syncthreads();
    f4 r1q[4];   // this wave's channel quarter of ct1, kept for the final residual
    {
        f4 v[CB];
        gather(v);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            if ((cb & 3) == 0) __builtin_amdgcn_sched_barrier(0);   // four channel blocks at a time: left alone the compiler requests all 64 partial reads first
            f4 a = *(const f4*)(ex + (0 * CB + cb) * 1024);
#pragma unroll
            for (int w = 1; w < NW; ++w) a += *(const f4*)(ex + (w * CB + cb) * 1024);
            const int co = (cb >> 2) * 64 + g * 16 + (cb & 3) * 4;
            const f4 bv = *(const f4*)(p.bproj + co);
            const f4 gl = *(const f4*)((has_g1 ? p.gamma1 : p.bproj) + co);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[cb][r] += (has_g1 ? gl[r] : 1.f) * (a[r] + bv[r]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // cb = 4 * wave + q without dynamic register indexing
            f4 t = v[q];
#pragma unroll
            for (int w = 1; w < NW; ++w) t = wave == w ? v[4 * w + q] : t;
            r1q[q] = t;
        }
        layernorm(v, p.ln2_w, p.ln2_b, xf);
    }
    __syncthreads();   // every wave has read all partials: the buffer is free for the second exchange

    {
        constexpr bool in_attention = false;
#pragma unroll
        for (int t = SA; t < SA + DEPTH; ++t) { FVIT_CT_LOAD(t) }
    }
    // ---- MLP: hidden units 32 * (CPW * wave + c) .. + 31 per chunk; fc2 partial over this wave's units in acc2 ----
    f4 acc2[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc2[cb] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        constexpr bool in_attention = false;
        const int j = CPW * wave + c;
        f4 a1[2];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            f4 a = (f4){0.f, 0.f, 0.f, 0.f}, ao = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int term = 0; term < WT; ++term) {
                const int t = SA + (4 * c + hb) * WT + term;
#pragma unroll
                for (int kk = 0; kk < KK; kk += 2) {
                    a = Op16<T>::mfma(ring[t % DEPTH][kk], xf[kk], a);
                    ao = Op16<T>::mfma(ring[t % DEPTH][kk + 1], xf[kk + 1], ao);
                }
                __builtin_amdgcn_sched_barrier(0);
                FVIT_CT_LOAD(t + DEPTH)
            }
            a1[hb] = a + ao;
        }
        const f4 bA = *(const f4*)(b1s + j * 32 + g * 4);
        const f4 bB = *(const f4*)(b1s + j * 32 + 16 + g * 4);
        float hv[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hv[r] = a1[0][r] + bA[r];
            hv[4 + r] = a1[1][r] + bB[r];
        }
        gelu_fast_n<8>(hv);   // eight Horner chains in lockstep, bitwise gelu_fast (fvit_common.h)
        v8 pf;
#pragma unroll
        for (int r = 0; r < 8; ++r) pf[r] = sat16<T>(hv[r]);
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int term = 0; term < WT; ++term) {
                const int t = SA + (4 * c + 2 + half) * WT + term;
#pragma unroll
                for (int i = 0; i < 8; ++i) acc2[half * 8 + i] = Op16<T>::mfma(ring[t % DEPTH][i], pf, acc2[half * 8 + i]);
                __builtin_amdgcn_sched_barrier(0);
                FVIT_CT_LOAD(t + DEPTH)
            }
    }
#undef FVIT_CT_LOAD

    // ---- exchange 2: wave w finishes channels 64w .. 64w + 63: ct2 = ct1 + gamma2 * (sum + b2) ----
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) *(f4*)(ex + (wave * CB + cb) * 1024) = acc2[cb];
    __syncthreads();
    if (row_ok) {
        float* pr = p.R + ((size_t)img * p.G + s) * C + wave * 64 + g * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cbo = (4 * wave + q) * 1024;
            f4 a = *(const f4*)(ex + (0 * CB) * 1024 + cbo);
#pragma unroll
            for (int w = 1; w < NW; ++w) a += *(const f4*)(ex + (w * CB) * 1024 + cbo);
            const int co = wave * 64 + g * 16 + q * 4;
            const f4 bv = *(const f4*)(p.b2 + co);
            const f4 gl = *(const f4*)((has_g2 ? p.gamma2 : p.b2) + co);
            f4 o = r1q[q];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] += (has_g2 ? gl[r] : 1.f) * (a[r] + bv[r]);
            *(f4*)(pr + q * 4) = o;
        }
    }
}


// ------------------------------------------------------------------------------------------------------------
// The 8-wave form (r03, fvit_tune "ct_variant" = 3): the same branch, one image per workgroup, with the waves splitting OUTPUT channels
// instead of the reduction dimension:
//   attention  wave w = head w (q^T, k^T, v of its head: 48 weight fragments; scores, softmax, O^T in registers); O^T (1 KiB) goes to LDS
//   proj       wave w = channel fragments 2w, 2w + 1 over ALL heads (16 fragments of w_proj_frag), O^T fragments of the eight heads from LDS
//   fc1        wave w = hidden chunks 4w .. 4w + 3 (64 fragments); GELU(H^T) fragments (1 KiB per chunk) go to LDS
//   fc2        wave w = channel fragments 2w, 2w + 1 over all 32 chunks (64 fragments), H^T fragments from LDS
// No fp32 partial sums are exchanged (ctblk_kernel: 2 x 64 KiB through LDS, added in wave order): what crosses waves is the 16-bit operand
// the next GEMM reads anyway, plus the 16-KiB fp32 carrier rows for the second LayerNorm, which every wave recomputes for itself (as it
// does the first one: the gather is 16 KiB per wave out of L2).  Every wave streams 192 KiB of weights instead of 384 (the same arrays, other
// fragment addresses), eight rings of three 8-fragment steps are in flight per CU instead of four.
// ------------------------------------------------------------------------------------------------------------
// NIMG (r06): images per workgroup.  One image is ONE 16-row MFMA block: every weight fragment of the 1.5 MB stream feeds one MFMA, and a
// launch of B workgroups pulls B x 1.5 MB through L2 for 6 % of a block's FLOPs (r05: 0.04 of the MFMA peak, 17.4 MB moved for 5.8 MB,
// 128 CUs held for 32 us per shard launch).  With NIMG = 2 a fragment feeds two MFMAs (one per image) and the launch needs half the
// workgroups for about the same duration (the duration is the weight stream through one CU): half the CU x time, which is what the step
// pays while two stream shards share the chip.  Per-image arithmetic is unchanged (same operations in the same order): bitwise the NIMG = 1 result.
template <typename T, int WT, int DEPTH, bool TS = false, int NIMG = 1>
__global__ __launch_bounds__(512, 1) void ctblk8_kernel(CtBlkParams p) {
    typedef typename Op16<T>::v8 v8;
    static_assert(!TS || NIMG == 1, "timeline instance: one image per workgroup");
    // stamps: 0 entry, 1 rows + constants landed (constants in LDS), first ring steps requested, 3 barrier passed, 2 LayerNorm 1 done, 4 attention done, 5 barrier,
    // 6 proj + residual done, 7 barrier, 8 LayerNorm 2 done, 9 fc1 + GELU done, 10 barrier, 11 fc2 done, 12 end (stores drained)
#define FVIT_CT8_STAMP(k, dep) if constexpr (TS) { asm volatile("s_nop 0" ::"v"(dep) : "memory"); if ((threadIdx.x & 63) == 0) p.ts[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (k)] = __builtin_amdgcn_s_memtime(); }
    FVIT_CT8_STAMP(0, threadIdx.x)
    constexpr int C = 256, KK = 8, CB = 16, NW = 8, HID = 1024;
    constexpr int S_QKV = 6 * WT, S_PROJ = 2 * WT, S_FC1 = 8 * WT, S_FC2 = 8 * WT;
    constexpr int T_PROJ = S_QKV, T_FC1 = T_PROJ + S_PROJ, T_FC2 = T_FC1 + S_FC1, NSTEP = T_FC2 + S_FC2;
    constexpr size_t QKV_IMG = (size_t)3 * C * C * 2, PROJ_IMG = (size_t)C * C * 2, FC_IMG = (size_t)C * HID * 2;
    // per image: O^T (8 KiB), H^T (32 KiB), ct1 rows (16 KiB); then the constants shared by the images
    constexpr int OT_B = 8 * 1024, H_B = 32 * 1024, CT_B = CB * 1024;
    // the constants and the regions every phase touches first, H^T (written and read in ONE phase each) last: DS instructions carry a 16-bit
    // offset, and with the H^T regions in front every access beyond 64 KiB got its own address register (r06 ISA check: 135 spilled registers
    // at NIMG = 2).  O^T and H^T go through their own base pointers (smO, smH below).
    constexpr int OFF_B1 = 0, OFF_BQ = OFF_B1 + HID * 4, OFF_BZ = OFF_BQ + 8 * 96 * 4, OFF_VEC = OFF_BZ + 8 * 256 * 4;
    constexpr int OFF_CT = OFF_VEC + 8 * C * 4, OFF_OT = OFF_CT + NIMG * CT_B, OFF_H = OFF_OT + NIMG * OT_B, LDS_BYTES = OFF_H + NIMG * H_B;
    static_assert(OFF_OT <= 65536 && NIMG * OT_B <= 65536 && NIMG * H_B <= 65536, "three bases (constants + ct1 rows, O^T, H^T), 16-bit offsets from each");
    // the eight per-channel vectors of the branch (ln1 w / b, proj bias, gamma1, ln2 w / b, fc2 bias, gamma2; a missing gamma is stored as ones):
    // read from global inside the phases they would queue behind the ring's prefetches in the in-order vmcnt counter and drain it
    // (timeline of the first version: proj + residual 2.9 us for 16 MFMAs, LayerNorm 2 2.5 us)
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    float* b1s = (float*)(smem + OFF_B1);
    float* bqs = (float*)(smem + OFF_BQ);
    float* bzs = (float*)(smem + OFF_BZ);
    float* vecs = (float*)(smem + OFF_VEC);   // [8][256]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, s = lane & 15;
    const int lane16 = lane * 16;
    char* smH = smem + OFF_H + lane16;          // bases of the H^T / O^T regions (fragment j of image im: + im * H_B + j * 1024), opaque to the compiler so
    char* smO = smem + OFF_OT + lane16;         // that each is kept as ONE register + immediate offsets instead of smem + a 17-bit constant per access
    asm volatile("" : "+v"(smH), "+v"(smO));
    const int img0 = blockIdx.x * NIMG;
    const int tok = s < p.G ? s : p.G - 1;
    const bool row_ok = s < p.G;

    // ---- the rows first (the longest round trip: they come from the memory side).  r06: wave w gathers only ITS two channel fragments (2w, 2w + 1:
    //      lane (g, s) = token s, channels (cb>>2)*64 + 16g + (cb&3)*4 .. +3) of every image and parks them in the ct1 region of LDS, from where
    //      every wave reads the full rows for its LayerNorm after the constants barrier (the second LayerNorm reads the same region the same way).
    //      r03 had every wave gather all 16 fragments for itself: 64 registers per image next to the ring's first steps, 8 x the gather traffic.
    //      An image beyond the batch (odd batch, NIMG = 2) recomputes the last one and is not stored ----
    const bool has_add = p.add != nullptr;
    f4 rowq[NIMG][2];
#pragma unroll
    for (int im = 0; im < NIMG; ++im) {
        const int img = min(img0 + im, p.B - 1);
        const float* src = p.X + ((size_t)img * p.rowsA + p.src_idx[tok]) * C + g * 16;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int cb = 2 * wave + q;
            rowq[im][q] = *(const f4*)(src + (cb >> 2) * 64 + (cb & 3) * 4);
        }
    }
    if (has_add) {   // hat_pos_embed rows (L2): requested here too -- after the barrier they would queue behind the first ring steps
        const float* addp = p.add + (size_t)tok * C + g * 16;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int cb = 2 * wave + q;
            const f4 a = *(const f4*)(addp + (cb >> 2) * 64 + (cb & 3) * 4);
#pragma unroll
            for (int im = 0; im < NIMG; ++im) rowq[im][q] += a;
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    const char* Wq = (const char*)p.wqkv_f + lane16;
    const char* Wp = (const char*)p.wproj_f + lane16;
    const char* W1 = (const char*)p.w1f + lane16;
    const char* W2 = (const char*)p.w2f + lane16;
    v8 ring[DEPTH][8];
    // fragment i of stream step t (compile-time t): the term is the fastest index inside each GEMM's step list
    auto frag_ptr = [&](int t, int i) -> const char* {
        if (t < T_PROJ) {            // qkv: step = (ub, term): the 8 k fragments of (head = wave, ub)
            const int ub = t / WT, term = t % WT;
            return Wq + term * QKV_IMG + ((size_t)wave * 48 + ub * 8 + i) * 1024;
        }
        if (t < T_FC1) {             // proj: step = (half, term): heads 4 half .. 4 half + 3 x channel fragments 2w, 2w + 1
            const int u = t - T_PROJ, half = u / WT, term = u % WT;
            const int h = 4 * half + (i >> 1), cb = 2 * wave + (i & 1);
            return Wp + term * PROJ_IMG + ((size_t)h * CB + cb) * 1024;
        }
        if (t < T_FC2) {             // fc1: step = (chunk c, hb, term): the 8 k fragments of (j = 4w + c, hb)
            const int u = t - T_FC1, ch = u / WT, term = u % WT;
            const int j = 4 * wave + (ch >> 1), hb = ch & 1;
            return W1 + term * FC_IMG + ((size_t)j * 16 + hb * 8 + i) * 1024;
        }
        {                            // fc2: step = (quad q, term): chunks 4q .. 4q + 3 x channel fragments 2w, 2w + 1
            const int u = t - T_FC2, q = u / WT, term = u % WT;
            const int j = 4 * q + (i >> 1), cb = 2 * wave + (i & 1);
            return W2 + term * FC_IMG + ((size_t)j * CB + cb) * 1024;
        }
    };
#define FVIT_CT8_LOAD(t)                                                                                              \
    if ((t) < NSTEP) {                                                                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) ring[(t) % DEPTH][i_] = *(const v8*)frag_ptr((t), i_);      \
    }                                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);

    {   // constants (behind the rows in the vmcnt queue), then the first DEPTH steps of the weight stream
        float c1[2], c3[4], c4[4];
        float c2 = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) c1[i] = p.b1[tid + 512 * i];
        if (tid < 8 * 96 - 512) c2 = p.bqkv[tid + 512];
        const float c2a = p.bqkv[tid];
#pragma unroll
        for (int i = 0; i < 4; ++i) c3[i] = p.bias[tid + 512 * i];
        {
            const float* vp[8] = {p.ln1_w, p.ln1_b, p.bproj, p.gamma1, p.ln2_w, p.ln2_b, p.b2, p.gamma2};
            const int hi = tid >> 8, ch = tid & 255;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* q = hi ? vp[2 * i + 1] : vp[2 * i];
                c4[i] = q ? q[ch] : 1.0f;   // only the gammas can be null
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < DEPTH; ++t) { FVIT_CT8_LOAD(t) }
#pragma unroll
        for (int i = 0; i < 2; ++i) b1s[tid + 512 * i] = c1[i];
        bqs[tid] = c2a;
        if (tid < 8 * 96 - 512) bqs[tid + 512] = c2;
#pragma unroll
        for (int i = 0; i < 4; ++i) bzs[tid + 512 * i] = c3[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) vecs[(2 * i + (tid >> 8)) * C + (tid & 255)] = c4[i];
#pragma unroll
        for (int im = 0; im < NIMG; ++im)
#pragma unroll
            for (int q = 0; q < 2; ++q) *(f4*)(smem + OFF_CT + im * CT_B + (2 * wave + q) * 1024 + lane16) = rowq[im][q];
        FVIT_CT8_STAMP(1, c3[3])
    }
    __syncthreads();   // constants and gathered rows visible (plain loads in flight survive the barrier)
    FVIT_CT8_STAMP(3, rowq[0][0])

    // LayerNorm of the 16 rows parked in LDS (fragment cb at rows + cb * 1024, lane-linear) in three streaming passes -- sum, squared deviations,
    // normalise -- four fragments at a time: the r03 form held all 64 values of a lane in registers next to the ring's steps in flight and the
    // other image's fragments (250 registers at NIMG = 2).  48 ds_read_b128 per image instead of 16; same operations in the same order.
    auto layernorm = [&](const char* rows, const float* lw, const float* lb, v8 (&xf)[KK]) {
        float sum = 0.f;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            if ((cb & 3) == 0) __builtin_amdgcn_sched_barrier(0);
            const f4 t = *(const f4*)(rows + cb * 1024);
            sum += (t[0] + t[1]) + (t[2] + t[3]);
        }
        sum = sum_xor32(sum_xor16(sum));
        const float mean = sum / (float)C;
        asm volatile("" ::: "memory");   // the next pass RE-READS the rows: without the clobber the three passes' loads are merged and the 64 values stay in registers
        float sq = 0.f;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            if ((cb & 3) == 0) __builtin_amdgcn_sched_barrier(0);
            const f4 d = *(const f4*)(rows + cb * 1024) - mean;
            sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        sq = sum_xor32(sum_xor16(sq));
        const float rstd = rsqrtf(sq / (float)C + p.eps);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            if ((kk & 1) == 0) __builtin_amdgcn_sched_barrier(0);
            v8 o;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int cb = 2 * kk + h2;
                const int co = (cb >> 2) * 64 + g * 16 + (cb & 3) * 4;
                const f4 t = *(const f4*)(rows + cb * 1024);
                const f4 w = *(const f4*)(lw + co);
                const f4 b = *(const f4*)(lb + co);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[h2 * 4 + r] = sat16<T>((t[r] - mean) * rstd * w[r] + b[r]);
            }
            xf[kk] = o;
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    v8 xf[NIMG][KK];
#pragma unroll
    for (int im = 0; im < NIMG; ++im) layernorm(smem + OFF_CT + im * CT_B + lane16, vecs + 0 * C, vecs + 1 * C, xf[im]);
    FVIT_CT8_STAMP(2, xf[NIMG - 1][KK - 1])

    // ---- attention of head = wave ----
    {
        const int h = wave;
        const float* bq = bqs + h * 96;
        v8 qf[NIMG], kf[NIMG], vf[NIMG][2];
#pragma unroll
        for (int ub = 0; ub < 6; ++ub) {
            f4 a[NIMG], ao[NIMG];
#pragma unroll
            for (int im = 0; im < NIMG; ++im) { a[im] = (f4){0.f, 0.f, 0.f, 0.f}; ao[im] = (f4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int term = 0; term < WT; ++term) {
                const int t = ub * WT + term;
#pragma unroll
                for (int kk = 0; kk < KK; kk += 2) {
#pragma unroll
                    for (int im = 0; im < NIMG; ++im) {
                        a[im] = ub < 4 ? Op16<T>::mfma(ring[t % DEPTH][kk], xf[im][kk], a[im]) : Op16<T>::mfma(xf[im][kk], ring[t % DEPTH][kk], a[im]);
                        ao[im] = ub < 4 ? Op16<T>::mfma(ring[t % DEPTH][kk + 1], xf[im][kk + 1], ao[im]) : Op16<T>::mfma(xf[im][kk + 1], ring[t % DEPTH][kk + 1], ao[im]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                FVIT_CT8_LOAD(t + DEPTH)
            }
#pragma unroll
            for (int im = 0; im < NIMG; ++im) {
                const f4 aa = a[im] + ao[im];
                if (ub < 4) {
                    const f4 bb = *(const f4*)(bq + (ub >> 1) * 32 + (ub & 1) * 16 + g * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (ub < 2) qf[im][(ub & 1) * 4 + r] = sat16<T>(aa[r] + bb[r]);
                        else kf[im][(ub & 1) * 4 + r] = sat16<T>(aa[r] + bb[r]);
                    }
                } else {
                    const float bv = bq[64 + (ub - 4) * 16 + s];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        vf[im][ub - 4][r] = sat16<T>(aa[r] + bv);
                        vf[im][ub - 4][4 + r] = (T)0.f;
                    }
                }
            }
        }
        const f4 bz = *(const f4*)(bzs + (h * 16 + tok) * 16 + g * 4);
#pragma unroll
        for (int im = 0; im < NIMG; ++im) {
            f4 sc = Op16<T>::mfma(kf[im], qf[im], (f4){0.f, 0.f, 0.f, 0.f});
            float mx = -3.0e38f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc[r] = sc[r] * p.scale + bz[r];
                mx = fmaxf(mx, sc[r]);
            }
            mx = max_xor32(max_xor16(mx));
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc[r] = __expf(sc[r] - mx);
                sum += sc[r];
            }
            sum = sum_xor32(sum_xor16(sum));
            const float inv = 1.0f / sum;
            v8 pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = (T)sc[r];
                pf[4 + r] = (T)0.f;
            }
            v8 of;
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const f4 o = Op16<T>::mfma(vf[im][db], pf, (f4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int r = 0; r < 4; ++r) of[db * 4 + r] = sat16<T>(o[r] * inv);
            }
            *(v8*)(smO + im * OT_B + h * 1024) = of;
            if (im == NIMG - 1) { FVIT_CT8_STAMP(4, of) }
        }
    }
    __syncthreads();   // O^T of all heads visible
    FVIT_CT8_STAMP(5, xf[0][0])

    // ---- proj over all heads for channel fragments 2w, 2w + 1; ct1 = ct0 + gamma1 * (out + bproj) -> LDS (fp32, lane-linear per fragment) ----
    {
        f4 oacc[NIMG][2];
#pragma unroll
        for (int im = 0; im < NIMG; ++im) { oacc[im][0] = (f4){0.f, 0.f, 0.f, 0.f}; oacc[im][1] = (f4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            v8 ob[NIMG][4];
#pragma unroll
            for (int im = 0; im < NIMG; ++im)
#pragma unroll
                for (int i = 0; i < 4; ++i) ob[im][i] = *(const v8*)(smO + im * OT_B + (4 * half + i) * 1024);
#pragma unroll
            for (int term = 0; term < WT; ++term) {
                const int t = T_PROJ + half * WT + term;
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int im = 0; im < NIMG; ++im) oacc[im][i & 1] = Op16<T>::mfma(ring[t % DEPTH][i], ob[im][i >> 1], oacc[im][i & 1]);
                __builtin_amdgcn_sched_barrier(0);
                FVIT_CT8_LOAD(t + DEPTH)
            }
        }
#pragma unroll
        for (int im = 0; im < NIMG; ++im)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int cb = 2 * wave + q;
                const int co = (cb >> 2) * 64 + g * 16 + (cb & 3) * 4;
                const f4 bv = *(const f4*)(vecs + 2 * C + co);
                const f4 gl = *(const f4*)(vecs + 3 * C + co);
                // the gathered rows ct0 of this fragment are still where this wave parked them: every wave finished reading them for its LayerNorm
                // before the O^T barrier, and only this wave writes fragment cb
                f4 o = *(const f4*)(smem + OFF_CT + im * CT_B + cb * 1024 + lane16);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] += gl[r] * (oacc[im][q][r] + bv[r]);
                *(f4*)(smem + OFF_CT + im * CT_B + cb * 1024 + lane16) = o;
            }
        FVIT_CT8_STAMP(6, oacc[NIMG - 1][1])
    }
    __syncthreads();   // ct1 of all channel fragments visible
    FVIT_CT8_STAMP(7, xf[0][0])

    // ---- second LayerNorm (every wave for itself), fc1 + GELU of hidden chunks 4w .. 4w + 3 -> LDS ----
#pragma unroll
    for (int im = 0; im < NIMG; ++im) layernorm(smem + OFF_CT + im * CT_B + lane16, vecs + 4 * C, vecs + 5 * C, xf[im]);
    FVIT_CT8_STAMP(8, xf[NIMG - 1][KK - 1])
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int j = 4 * wave + c;
        f4 a1[NIMG][2];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            f4 a[NIMG], ao[NIMG];
#pragma unroll
            for (int im = 0; im < NIMG; ++im) { a[im] = (f4){0.f, 0.f, 0.f, 0.f}; ao[im] = (f4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int term = 0; term < WT; ++term) {
                const int t = T_FC1 + (2 * c + hb) * WT + term;
#pragma unroll
                for (int kk = 0; kk < KK; kk += 2) {
#pragma unroll
                    for (int im = 0; im < NIMG; ++im) {
                        a[im] = Op16<T>::mfma(ring[t % DEPTH][kk], xf[im][kk], a[im]);
                        ao[im] = Op16<T>::mfma(ring[t % DEPTH][kk + 1], xf[im][kk + 1], ao[im]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                FVIT_CT8_LOAD(t + DEPTH)
            }
#pragma unroll
            for (int im = 0; im < NIMG; ++im) a1[im][hb] = a[im] + ao[im];
        }
        const f4 bA = *(const f4*)(b1s + j * 32 + g * 4);
        const f4 bB = *(const f4*)(b1s + j * 32 + 16 + g * 4);
#pragma unroll
        for (int im = 0; im < NIMG; ++im) {
            float hv[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                hv[r] = a1[im][0][r] + bA[r];
                hv[4 + r] = a1[im][1][r] + bB[r];
            }
            gelu_fast_n<8>(hv);   // eight Horner chains in lockstep, bitwise gelu_fast (fvit_common.h)
            v8 pf;
#pragma unroll
            for (int r = 0; r < 8; ++r) pf[r] = sat16<T>(hv[r]);
            *(v8*)(smH + im * H_B + j * 1024) = pf;
            if (c == 3 && im == NIMG - 1) { FVIT_CT8_STAMP(9, pf) }
        }
    }
    __syncthreads();   // H^T of all 32 chunks visible
    FVIT_CT8_STAMP(10, xf[0][0])

    // ---- fc2 over all chunks for channel fragments 2w, 2w + 1; ct2 = ct1 + gamma2 * (out + b2) -> R ----
    {
        f4 acc2[NIMG][2];
#pragma unroll
        for (int im = 0; im < NIMG; ++im) { acc2[im][0] = (f4){0.f, 0.f, 0.f, 0.f}; acc2[im][1] = (f4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            v8 hb4[NIMG][4];
#pragma unroll
            for (int im = 0; im < NIMG; ++im)
#pragma unroll
                for (int i = 0; i < 4; ++i) hb4[im][i] = *(const v8*)(smH + im * H_B + (4 * q + i) * 1024);
#pragma unroll
            for (int term = 0; term < WT; ++term) {
                const int t = T_FC2 + q * WT + term;
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int im = 0; im < NIMG; ++im) acc2[im][i & 1] = Op16<T>::mfma(ring[t % DEPTH][i], hb4[im][i >> 1], acc2[im][i & 1]);
                __builtin_amdgcn_sched_barrier(0);
                FVIT_CT8_LOAD(t + DEPTH)
            }
        }
        FVIT_CT8_STAMP(11, acc2[NIMG - 1][1])
#pragma unroll
        for (int im = 0; im < NIMG; ++im) {
            if (row_ok && img0 + im < p.B) {
                float* pr = p.R + ((size_t)(img0 + im) * p.G + s) * C;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int cb = 2 * wave + q;
                    const int co = (cb >> 2) * 64 + g * 16 + (cb & 3) * 4;
                    const f4 bv = *(const f4*)(vecs + 6 * C + co);
                    const f4 gl = *(const f4*)(vecs + 7 * C + co);
                    f4 o = *(const f4*)(smem + OFF_CT + im * CT_B + cb * 1024 + lane16);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] += gl[r] * (acc2[im][q][r] + bv[r]);
                    *(f4*)(pr + co) = o;
                }
            }
        }
    }
    if constexpr (TS) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    FVIT_CT8_STAMP(12, threadIdx.x)
#undef FVIT_CT8_STAMP
#undef FVIT_CT8_LOAD
}

}  // namespace

bool ctblk_supported(int C, int heads, int G, int hidden) { return C == 256 && heads == 8 && hidden == 1024 && G >= 1 && G <= 16; }

int launch_ctblk(const CtBlkCall& c, hipStream_t stream) {
    if (ablate_skip(16)) return FVIT_OK;
    if (!ctblk_supported(c.C, c.heads, c.G, c.hidden) || c.batch <= 0 || !c.X || !c.src_idx || !c.R || !c.wqkv_f || !c.wproj_f || !c.w1f || !c.w2f ||
        !c.bias || !c.bqkv) {
        set_error("ct_block: unsupported arguments C=%d heads=%d G=%d hidden=%d batch=%d", c.C, c.heads, c.G, c.hidden, c.batch);
        return FVIT_EINVAL;
    }
    CtBlkParams p;
    p.X = c.X; p.src_idx = c.src_idx; p.add = c.add; p.R = c.R; p.rowsA = c.rowsA; p.B = c.batch; p.G = c.G;
    p.ln1_w = c.ln1_w; p.ln1_b = c.ln1_b; p.wqkv_f = c.wqkv_f; p.bqkv = c.bqkv; p.wproj_f = c.wproj_f; p.bproj = c.bproj; p.gamma1 = c.gamma1;
    p.bias = c.bias; p.scale = c.scale;
    p.ln2_w = c.ln2_w; p.ln2_b = c.ln2_b; p.w1f = c.w1f; p.b1 = c.b1; p.w2f = c.w2f; p.b2 = c.b2; p.gamma2 = c.gamma2; p.eps = c.eps;
    const double rows = (double)c.batch * c.G;
    const double flops = rows * (2.0 * c.C * 3 * c.C + 4.0 * c.G * c.C + 2.0 * c.C * c.C + 4.0 * c.C * c.hidden);
    const double bytes = rows * c.C * 8.0 + c.terms * 2.0 * (4.0 * c.C * c.C + 2.0 * c.C * c.hidden);
    ProfScope prof(FVIT_K_ATTN_FUSED, flops, bytes, stream);
    prof_note(tune_get("ct_variant", 3) == 3 ? "ctblk8_kernel<256,G16>" : "ctblk_kernel<256,G16>", c.batch);
    p.touch = tune_get("ct_touch", 0);
    p.ts = (unsigned long long*)c.ts;
    if (c.ts) {   // timeline instance: the 8-wave fp16 form, single-term weights
        if (c.dtype != FVIT_F16 || c.terms != 1) { set_error("ct_block timeline: fp16, one weight term only"); return FVIT_EINVAL; }
        hipLaunchKernelGGL((ctblk8_kernel<_Float16, 1, 2, true>), dim3(c.batch), dim3(512), 0, stream, p);
        return check_launch("ctblk8_kernel");
    }
    // 3 (default since r03): the 8-wave form; 0 / 1 / 2: the 4-wave form with ring depth 3 / 2 (256 registers) / 4
    const int variant = tune_get("ct_variant", 3);
    if (c.terms != 1 && c.terms != 2) { set_error("ct_block: weight terms %d (1 or 2)", c.terms); return FVIT_EINVAL; }
    if (variant == 3) {   // the 8-wave form: waves split output channels, no fp32 partial exchange
        // (r06 measured two images per workgroup -- ctblk8_kernel<.., NIMG = 2>: every weight fragment feeds two MFMAs, half the workgroups -- at -4.5 % images/s,
        // 80.0k vs 83.7k in three interleaved pairs, profiles/r06_ct_two_images_per_workgroup_ab.log: the workgroup's life is its chain of small dependent phases,
        // not its weight stream, and the chain doubles; not instantiated.  What stayed from that work: the rows parked in LDS and the streaming LayerNorm.)
        const int depth = tune_get("ct8_depth", 3);   // ring steps of 8 fragments in flight per wave (2 / 3 / 4)
#define FVIT_CT8(T_, WT_) do { \
            if (depth == 2) hipLaunchKernelGGL((ctblk8_kernel<T_, WT_, 2>), dim3(c.batch), dim3(512), 0, stream, p); \
            else if (depth == 4) hipLaunchKernelGGL((ctblk8_kernel<T_, WT_, 4>), dim3(c.batch), dim3(512), 0, stream, p); \
            else hipLaunchKernelGGL((ctblk8_kernel<T_, WT_, 3>), dim3(c.batch), dim3(512), 0, stream, p); } while (0)
        if (c.dtype == FVIT_F16) {
            if (c.terms == 2) FVIT_CT8(_Float16, 2); else FVIT_CT8(_Float16, 1);
        } else if (c.dtype == FVIT_BF16) {
            if (c.terms == 2) FVIT_CT8(__bf16, 2); else FVIT_CT8(__bf16, 1);
#undef FVIT_CT8
        } else { set_error("ct_block: operand dtype %d not supported", c.dtype); return FVIT_EINVAL; }
        return check_launch("ctblk8_kernel");
    }
    if (c.dtype == FVIT_F16 && c.terms == 2) {
        hipLaunchKernelGGL((ctblk_kernel<_Float16, 3, 1, 2>), dim3(c.batch), dim3(256), 0, stream, p);
    } else if (c.dtype == FVIT_F16) {
        if (variant == 1) hipLaunchKernelGGL((ctblk_kernel<_Float16, 2, 2>), dim3(c.batch), dim3(256), 0, stream, p);
        else if (variant == 2) hipLaunchKernelGGL((ctblk_kernel<_Float16, 4, 1>), dim3(c.batch), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((ctblk_kernel<_Float16, 3, 1>), dim3(c.batch), dim3(256), 0, stream, p);
    } else if (c.dtype == FVIT_BF16) {
        if (c.terms == 2) hipLaunchKernelGGL((ctblk_kernel<__bf16, 3, 1, 2>), dim3(c.batch), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((ctblk_kernel<__bf16, 3, 1>), dim3(c.batch), dim3(256), 0, stream, p);
    } else { set_error("ct_block: operand dtype %d not supported", c.dtype); return FVIT_EINVAL; }
    return check_launch("ctblk_kernel");
}

}  // namespace fvit
