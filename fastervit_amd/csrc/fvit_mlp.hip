// fvit_mlp.hip -- fused MLP sub-block of HAT (gfx950):
//
//     x += gamma * ( fc2( GELU( fc1( LayerNorm(x) ) ) ) )            (AR:697 with AR:399-408; FV:691)
//
// in ONE kernel.  The unfused path moves the 4C-wide hidden activation through HBM twice and the
// fp32 residual stream three times (LayerNorm, fc1, fc2 launches); here a workgroup reads its X
// rows once, keeps LayerNorm(x) and both accumulators in registers, streams the weights from L2
// through LDS, and writes X once: HBM traffic per row drops from ~18 C to 8 C bytes (+weights
// once per XCD L2), which moves the sub-block from the HBM roof to the MFMA roof.
//
// How the hidden activation stays in registers ("transposed chaining"):
//   GEMM1 is issued as H^T = W1 . Xn^T with v_mfma_f32_16x16x32 (A = W1 rows = hidden units,
//   B = normalised activation rows).  Its accumulator layout -- lane (g = lane>>4, s = lane&15)
//   holds H^T[unit = 4g + r][row = s] -- is exactly the B-operand layout of GEMM2
//   OUT^T = W2 . H^T (k index = hidden unit, column = row) up to a permutation of the k slots,
//   and an MFMA is invariant under any permutation of k applied to both operands.  So GELU(acc1)
//   is converted to 16 bit and fed straight back as the B operand; W2 is pre-packed with its k
//   slots in the matching order (fragment-major, see fvit_hip.h: w_fc1_frag / w_fc2_frag).
//
// Work split: 4 wave64 per workgroup, each wave owns RB*16 rows for the whole kernel (RB = 2 for
// C = 256: 32 rows/wave, 128 rows/workgroup).  The hidden dimension is walked in chunks of 32
// units: per chunk a wave issues RB*2*C/32 MFMAs for GEMM1 and RB*C/16 for GEMM2 against
// 2*C/32 + C/16 ds_read_b128 of the shared weight chunk (0.5 KiB LDS per MFMA).  The chunk's
// W1 (32 x C) and W2 (C x 32) slices arrive by 16-byte global_load_lds into a double-buffered
// LDS image (2 x 2 x 16 KiB at C = 256; two workgroups per CU).  Both are pre-packed in MFMA
// fragment order, so the copy is linear and every read is a conflict-free lane-linear
// ds_read_b128 at an immediate offset; one barrier per chunk.
#include "fvit_common.h"

namespace fvit {

namespace {

struct MlpParams {
    float* x;            // [M][C] fp32 residual stream, updated in place
    const float* ln_w;
    const float* ln_b;
    const void* w1f;     // op16 fc1 weight, fragment-major: [hidden/32][2][C/32][64 lanes][8]
    const float* b1;     // [hidden]
    const void* w2f;     // op16 fc2 weight, fragment-major: [hidden/32][C/16][64 lanes][8]
    const float* b2;     // [C]
    const float* gamma;  // [C] or null
    float eps;
    int M, hidden;
    float* dbgx;     // diagnosis only: the kernel stores the input rows exactly as its loads returned them ([M][C])
    unsigned* dbg;   // diagnosis only (fvit_debug_mlp_trace_begin): per-lane hashes of the kernel's intermediate state; null in production
    int stagger;  // 1: workgroup b walks the hidden chunks starting at chunk b % nchunk; 2: offsets spread evenly over the workgroups of an XCD
    int ablate;   // timing experiments only (results are wrong): bit 1 = skip weight staging (bit 0, GELU -> identity, was removed in r02:
                  // a runtime branch around every GELU serialised the chunk body)
};

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// Weights are static, so they are pre-packed in the exact MFMA fragment order (hat_runtime.frag_pack_*):
// one fragment = 64 lanes x 16 bytes = 1 KiB = one global_load_lds instruction = one conflict-free
// lane-linear ds_read_b128.  The HBM image of a 32-unit chunk IS its LDS image: staging is a linear copy and
// every fragment address is (buffer base + lane*16 + compile-time immediate) -- one address register.
// KEEPX: keep the fp32 input rows in registers for the residual epilogue instead of re-reading X (a third of the kernel's HBM
// traffic).  It works because the k-slot order of GEMM1 is free: k slot (kk, g, e) is mapped to channel
//     kch(kk, g, e) = (kk>>1)*64 + g*16 + (kk&1)*8 + e
// (w_fc1_frag is packed in that order), so the 64 input values a lane loads per row are exactly the 64 output channels it owns
// in GEMM2's accumulator layout (fragment cb, slot 4g + r <-> channel (cb>>2)*64 + 16g + (cb&3)*4 + r).
// NB: depth of the weight-chunk ring in LDS.  2 = double buffer, drained barrier per chunk, two workgroups per CU (C = 256).  4 = three
// chunks in flight behind a COUNTED s_waitcnt vmcnt + raw s_barrier, one workgroup per CU: for launches of <= ~1 workgroup per CU,
// where the chunk loop is otherwise one LDS-DMA round trip per chunk (r02 probe, 32 KiB steps, small grids: 0.88 us per step with a
// 2-deep ring and every workgroup in lockstep on a cold weight stream vs 0.30 us with a 4-deep ring and staggered chunk order).
template <typename T, int C, int RB, int NW, int MINW, bool KEEPX, int NB = 2, bool TRACE = false>
__global__ __launch_bounds__(64 * NW, MINW) void mlp_fused_kernel(MlpParams p) {
    constexpr int NBUF = NB;
    typedef typename Op16<T>::v8 v8;
    constexpr int KK = C / 32;             // GEMM1 k-steps
    constexpr int CB = C / 16;             // output channel blocks
    constexpr int W1_FRAGS = 2 * KK;       // fragments (KiB) of one W1 chunk: 32 hidden units x C
    constexpr int W2_FRAGS = CB;           // fragments of one W2 chunk: C channels x 32 hidden units
    constexpr int W1_BYTES = W1_FRAGS * 1024, W2_BYTES = W2_FRAGS * 1024;
    constexpr int BUF_BYTES = W1_BYTES + W2_BYTES;
    constexpr int ROWS_PER_WAVE = RB * 16;
    constexpr int MAX_HIDDEN = 4 * C;      // fc1 bias staged in LDS once: no ordinary global loads inside the chunk loop
                                           // (they queue behind the LDS-DMA prefetch in the in-order vmcnt counter and drain it)
    // + fc2 bias and gamma (2 x C floats): the epilogue then has no global loads, so its 16-byte stores issue back to back (with
    // b2 / gamma fetched from L2 the compiler put an s_waitcnt vmcnt(0) -- which also waits for the PREVIOUS store -- in front of
    // every one of the C/16 store groups: 16 serial memory round trips per workgroup)
    __shared__ __attribute__((aligned(16))) char smem[NBUF * BUF_BYTES + MAX_HIDDEN * 4 + 2 * C * 4];
    float* b1s = (float*)(smem + NBUF * BUF_BYTES);
    float* b2s = b1s + MAX_HIDDEN;
    float* gms = b2s + C;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, s = lane & 15;
    const int row0 = blockIdx.x * (NW * ROWS_PER_WAVE) + wave * ROWS_PER_WAVE;
    const int nchunk = p.hidden / 32;
    int jofs = 0;
    if (p.stagger == 1) jofs = (int)(blockIdx.x % (unsigned)nchunk);
    else if (p.stagger == 2) {   // even spread over the workgroups of one XCD (blockIdx % 8 = XCD, observed)
        const int per_xcd = (gridDim.x + 7) >> 3, idx = blockIdx.x >> 3;
        jofs = per_xcd < nchunk ? (idx * nchunk) / per_xcd : idx % nchunk;
    }
    auto chunk_of = [&](int it) { int j = it + jofs; return j >= nchunk ? j - nchunk : j; };

    const char* __restrict__ W1 = (const char*)p.w1f;
    const char* __restrict__ W2 = (const char*)p.w2f;
    const int lane16 = lane * 16;
    // diagnosis trace: slot q of this wave = 64 lanes x 1 word; q = 0: LN fragments, 1 + 5 * it + {0: W1 fragments as read, 1: pre-GELU
    // accumulators, 2: GELU output fragment, 3: W2 fragments as read, 4: output accumulators after the chunk}
    unsigned* dq = TRACE && p.dbg ? p.dbg + ((size_t)(blockIdx.x * NW + wave) * (size_t)(4 + 5 * nchunk)) * 64 + lane : nullptr;
    auto fold8 = [](unsigned h, const v8& v) {
        const uint4 u = __builtin_bit_cast(uint4, v);
        h = h * 31u + u.x; h = h * 31u + u.y; h = h * 31u + u.z; h = h * 31u + u.w;
        return h;
    };
    auto fold4 = [](unsigned h, const f4& v) {
        const uint4 u = __builtin_bit_cast(uint4, v);
        h = h * 31u + u.x; h = h * 31u + u.y; h = h * 31u + u.z; h = h * 31u + u.w;
        return h;
    };

    // wave w copies fragments w, w+NW, w+2NW, ... of the chunk (W1 fragments first, then W2)
    auto stage = [&](int j, char* buf) {
        const char* s1 = W1 + (size_t)j * W1_BYTES + lane16;
        const char* s2 = W2 + (size_t)j * W2_BYTES + lane16;
#pragma unroll
        for (int i = 0; i < W1_FRAGS / NW; ++i) glds16(s1 + (wave + NW * i) * 1024, buf + (wave + NW * i) * 1024);
#pragma unroll
        for (int i = 0; i < W2_FRAGS / NW; ++i) glds16(s2 + (wave + NW * i) * 1024, buf + W1_BYTES + (wave + NW * i) * 1024);
    };

    constexpr int GLDS_PER_STAGE = (W1_FRAGS + W2_FRAGS) / NW;   // LDS-DMA instructions one wave issues per chunk
    static_assert(NBUF == 2 || (NBUF - 2) * GLDS_PER_STAGE <= 48, "vmcnt is a 6-bit counter");
    for (int i = tid; i < p.hidden; i += 64 * NW) b1s[i] = p.b1[i];
    for (int i = tid; i < C; i += 64 * NW) {
        b2s[i] = p.b2[i];
        gms[i] = p.gamma ? p.gamma[i] : 1.0f;
    }
#pragma unroll
    for (int st = 0; st < NBUF - 1; ++st)
        if (st < nchunk) stage(chunk_of(st), smem + st * BUF_BYTES);

    // ---- LayerNorm of this wave's rows straight into B-operand fragments ----
    // lane (g, s) holds channels kch(kk, g, 0..7) = (kk>>1)*64 + g*16 + (kk&1)*8 .. +8 (kk = 0..KK-1) of row rb*16 + s
    v8 xf[RB][KK];
    f4 xk[KEEPX ? RB : 1][KK][2];   // KEEPX: the fp32 rows, alive until the epilogue
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int row = min(row0 + rb * 16 + s, p.M - 1);  // tail rows recompute the last row; never stored
        const float* xr = p.x + (size_t)row * C + g * 16;
        f4 (&v)[KK][2] = xk[KEEPX ? rb : 0];
        float sum = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            v[kk][0] = *(const f4*)(xr + (kk >> 1) * 64 + (kk & 1) * 8);
            v[kk][1] = *(const f4*)(xr + (kk >> 1) * 64 + (kk & 1) * 8 + 4);
            sum += (v[kk][0][0] + v[kk][0][1]) + (v[kk][0][2] + v[kk][0][3]) + (v[kk][1][0] + v[kk][1][1]) + (v[kk][1][2] + v[kk][1][3]);
        }
        sum = sum_xor32(sum_xor16(sum));   // VALU lane swaps, not ds_bpermute: see fvit_common.h
        const float mean = sum / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f4 d = v[kk][h] - mean;
                sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
        sq = sum_xor32(sum_xor16(sq));
        const float rstd = rsqrtf(sq / (float)C + p.eps);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const float* lw = p.ln_w + (kk >> 1) * 64 + g * 16 + (kk & 1) * 8;
            const float* lb = p.ln_b + (kk >> 1) * 64 + g * 16 + (kk & 1) * 8;
            v8 o;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f4 w = *(const f4*)(lw + h * 4);
                const f4 b = *(const f4*)(lb + h * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[h * 4 + r] = sat16<T>((v[kk][h][r] - mean) * rstd * w[r] + b[r]);
            }
            xf[rb][kk] = o;
        }
    }

    if (TRACE && dq) {
        unsigned h = 0;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) h = fold8(h, xf[0][kk]);
        dq[0] = h;
    }
    if (TRACE && p.dbgx) {   // the normalised fragments themselves: [workgroup * NW + wave][kk][lane] x 16 bytes
        uint4* dx = (uint4*)p.dbgx + ((size_t)(blockIdx.x * NW + wave) * KK) * 64 + lane;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) dx[kk * 64] = __builtin_bit_cast(uint4, xf[0][kk]);
    }
    f4 acc2[CB][RB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc2[cb][rb] = (f4){0.f, 0.f, 0.f, 0.f};

    for (int it = 0; it < nchunk; ++it) {
        const int j = chunk_of(it);
        const char* buf = smem + (it % NBUF) * BUF_BYTES + lane16;
        if (NBUF == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
            // chunk `it` has landed once at most the groups of the chunks after it are outstanding; raw barrier: __syncthreads()
            // would drain the whole DMA queue (vmcnt(0))
            const int ahead = min(NBUF - 2, nchunk - 1 - it);
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * GLDS_PER_STAGE) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(GLDS_PER_STAGE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        // refill the slot chunk it - 1 was read from (every wave is past it: it arrived at this barrier)
        if (TRACE && (p.ablate & 64)) { asm volatile("s_sleep 4\n\ts_barrier" ::: "memory"); }   // debug: settle time + second barrier between the DMA wait and the first read
        if (TRACE && (p.ablate & 128)) {   // debug: stage through registers (plain loads + ds_write, no LDS-DMA), synchronously
            if (it == 0) {      // chunk 0 was staged by DMA in the prologue: restage it too
                const char* s1 = W1 + (size_t)chunk_of(0) * W1_BYTES + lane16;
                const char* s2 = W2 + (size_t)chunk_of(0) * W2_BYTES + lane16;
                for (int i = 0; i < W1_FRAGS / NW; ++i) *(uint4*)(smem + (wave + NW * i) * 1024 + lane16) = *(const uint4*)(s1 + (wave + NW * i) * 1024);
                for (int i = 0; i < W2_FRAGS / NW; ++i) *(uint4*)(smem + W1_BYTES + (wave + NW * i) * 1024 + lane16) = *(const uint4*)(s2 + (wave + NW * i) * 1024);
            }
            if (it + 1 < nchunk) {
                char* dst = smem + ((it + 1) % NBUF) * BUF_BYTES;
                const char* s1 = W1 + (size_t)chunk_of(it + 1) * W1_BYTES + lane16;
                const char* s2 = W2 + (size_t)chunk_of(it + 1) * W2_BYTES + lane16;
                for (int i = 0; i < W1_FRAGS / NW; ++i) *(uint4*)(dst + (wave + NW * i) * 1024 + lane16) = *(const uint4*)(s1 + (wave + NW * i) * 1024);
                for (int i = 0; i < W2_FRAGS / NW; ++i) *(uint4*)(dst + W1_BYTES + (wave + NW * i) * 1024 + lane16) = *(const uint4*)(s2 + (wave + NW * i) * 1024);
            }
            __syncthreads();
        } else if (it + NBUF - 1 < nchunk && !(p.ablate & 2)) stage(chunk_of(it + NBUF - 1), smem + ((it + NBUF - 1) % NBUF) * BUF_BYTES);
        if (p.ablate & 32) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }   // debug: synchronous weight DMA
        if constexpr (RB == 1 || NW == 2) {   // NW == 2: one wave per SIMD (two 2-wave workgroups per CU), 512-register budget
            // The chunk body is written for a wave that is ALONE on its SIMD (small grids: the carrier-token branch, stage 3, the
            // shard-sized launches of the stream-sharded plan): left to the compiler the loop was "2 ds_read, s_waitcnt lgkmcnt(0),
            // 2 MFMA" sixteen times over plus eight serial GELU chains behind runtime branches -- ~3100 cycles per chunk for 512
            // cycles of MFMA issue (r02 ISA audit).  Here every fragment of a GEMM is requested before its first MFMA (one exposed
            // LDS round trip per GEMM instead of eight), GEMM1 runs KSPLIT independent accumulator chains per (unit block, row
            // block), and GEMM2's fragments are in flight while the VALU does bias + GELU.
            constexpr int FB = RB == 1 ? 16 : 8;         // fragments requested per batch (64 / 32 VGPRs)
            constexpr int KSPLIT = RB == 1 ? 2 : 1;      // independent accumulator chains per output block in GEMM1 (RB = 2: the two row blocks are the two chains)
            constexpr int KH = KK / KSPLIT;
            // ---- GEMM1: H^T[unit][row], 2 unit blocks x RB row blocks; unit = hb*16 + 4g + r ----
            f4 acc1[2][RB][KSPLIT];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int ks = 0; ks < KSPLIT; ++ks) acc1[hb][rb][ks] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int f0 = 0; f0 < W1_FRAGS; f0 += FB) {
                v8 wf[FB];
#pragma unroll
                for (int i = 0; i < FB; ++i) wf[i] = *(const v8*)(buf + (f0 + i) * 1024);   // fragment index = hb * KK + kk
                __builtin_amdgcn_sched_barrier(0);
                if (TRACE && dq) {
                    unsigned h = 0;
#pragma unroll
                    for (int i = 0; i < FB; ++i) h = fold8(h, wf[i]);
                    dq[(size_t)(4 + 5 * it + 0) * 64] = h;
                }
                // issue order kk-major inside the batch so that consecutive MFMAs hit different accumulators
#pragma unroll
                for (int i = 0; i < FB; ++i) {
                    const int f = f0 + i, hb = f / KK, kk = f - hb * KK;
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) acc1[hb][rb][kk / KH] = Op16<T>::mfma(wf[i], xf[rb][kk], acc1[hb][rb][kk / KH]);
                }
            }
            // ---- GEMM2's first fragment batch goes in flight now and lands while the VALU runs bias + GELU (pinned: left alone the
            // compiler sinks these reads below the GELU block again to save registers) ----
            constexpr int HB2 = RB == 1 ? 8 : 4;     // fragments per GEMM2 batch (32 / 16 VGPRs)
            v8 w2a[HB2], w2b[HB2];
#pragma unroll
            for (int i = 0; i < HB2; ++i) w2a[i] = *(const v8*)(buf + W1_BYTES + i * 1024);
            __builtin_amdgcn_sched_barrier(0);
            // ---- bias + GELU, straight into GEMM2's B operand (k slot 8g + i <-> unit (i>>2)*16 + 4g + (i&3)) ----
            const f4 bA = *(const f4*)(b1s + j * 32 + g * 4);
            const f4 bB = *(const f4*)(b1s + j * 32 + 16 + g * 4);
            v8 pf[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                f4 h0 = acc1[0][rb][0], h1 = acc1[1][rb][0];
#pragma unroll
                for (int ks = 1; ks < KSPLIT; ++ks) { h0 += acc1[0][rb][ks]; h1 += acc1[1][rb][ks]; }
                if (TRACE && dq) dq[(size_t)(4 + 5 * it + 1) * 64] = fold4(fold4(0u, h0), h1);
                float hv[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    hv[r] = h0[r] + bA[r];
                    hv[4 + r] = h1[r] + bB[r];
                }
                gelu_fast_n<8>(hv);   // eight Horner chains in lockstep, bitwise gelu_fast (fvit_common.h)
#pragma unroll
                for (int r = 0; r < 8; ++r) pf[rb][r] = sat16<T>(hv[r]);
            }
            if (TRACE && dq) dq[(size_t)(4 + 5 * it + 2) * 64] = fold8(0u, pf[0]);
            unsigned hw2 = 0;
            // ---- GEMM2: OUT^T[channel][row] += W2[channel][chunk units] . H^T; batch c+1 is requested before the MFMAs of batch c ----
#pragma unroll
            for (int c0 = 0; c0 < CB; c0 += 2 * HB2) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < HB2; ++i) w2b[i] = *(const v8*)(buf + W1_BYTES + (c0 + HB2 + i) * 1024);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < HB2; ++i)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) acc2[c0 + i][rb] = Op16<T>::mfma(w2a[i], pf[rb], acc2[c0 + i][rb]);
                if (TRACE && dq) {
#pragma unroll
                    for (int i = 0; i < HB2; ++i) hw2 = fold8(hw2, w2a[i]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (c0 + 2 * HB2 < CB) {
#pragma unroll
                    for (int i = 0; i < HB2; ++i) w2a[i] = *(const v8*)(buf + W1_BYTES + (c0 + 2 * HB2 + i) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < HB2; ++i)
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) acc2[c0 + HB2 + i][rb] = Op16<T>::mfma(w2b[i], pf[rb], acc2[c0 + HB2 + i][rb]);
                if (TRACE && dq) {
#pragma unroll
                    for (int i = 0; i < HB2; ++i) hw2 = fold8(hw2, w2b[i]);
                }
            }
            if (TRACE && dq) {
                dq[(size_t)(4 + 5 * it + 3) * 64] = hw2;
                unsigned h = 0;
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) h = fold4(h, acc2[cb][0]);
                dq[(size_t)(4 + 5 * it + 4) * 64] = h;
            }
        } else {
            // 32 rows per wave (whole-batch launches, two workgroups per CU): 128 accumulator registers leave no room for fragment
            // batches (they spill); the partner wave on the SIMD hides the LDS round trips instead
            // ---- GEMM1: H^T[unit][row], 2 unit blocks x RB row blocks; unit = hb*16 + 4g + r ----
            f4 acc1[2][RB];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc1[hb][rb] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    const v8 wf = *(const v8*)(buf + (hb * KK + kk) * 1024);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) acc1[hb][rb] = Op16<T>::mfma(wf, xf[rb][kk], acc1[hb][rb]);
                }
            }
            // ---- bias + GELU, straight into GEMM2's B operand (k slot 8g + i <-> unit (i>>2)*16 + 4g + (i&3)) ----
            const f4 bA = *(const f4*)(b1s + j * 32 + g * 4);
            const f4 bB = *(const f4*)(b1s + j * 32 + 16 + g * 4);
            v8 pf[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                float hv[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    hv[r] = acc1[0][rb][r] + bA[r];
                    hv[4 + r] = acc1[1][rb][r] + bB[r];
                }
                gelu_fast_n<8>(hv);
#pragma unroll
                for (int r = 0; r < 8; ++r) pf[rb][r] = sat16<T>(hv[r]);
            }
            // ---- GEMM2: OUT^T[channel][row] += W2[channel][chunk units] . H^T ----
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const v8 wf = *(const v8*)(buf + W1_BYTES + cb * 1024);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc2[cb][rb] = Op16<T>::mfma(wf, pf[rb], acc2[cb][rb]);
            }
        }
        if (p.ablate & 16) __syncthreads();   // debug: extra barrier at the end of the chunk body
    }

    // ---- epilogue: x[row][c] += gamma[c] * (acc2 + b2[c]) ----
    // fragment cb, A-row slot 4g + r <-> channel (cb>>2)*64 + 16g + (cb&3)*4 + r: a lane owns 16 consecutive channels per 4 blocks
#pragma unroll
    for (int cg = 0; cg < CB / 4; ++cg) {
        const int c0 = cg * 64 + g * 16;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int row = row0 + rb * 16 + s;
            if (row < p.M) {
                float* px = p.x + (size_t)row * C + c0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f4 bv = *(const f4*)(b2s + c0 + q * 4);
                    const f4 gv = *(const f4*)(gms + c0 + q * 4);
                    f4 xv;
                    if (KEEPX) xv = xk[KEEPX ? rb : 0][2 * cg + (q >> 1)][q & 1];
                    else xv = *(f4*)(px + q * 4);
                    const f4 a = acc2[cg * 4 + q][rb];
#pragma unroll
                    for (int r = 0; r < 4; ++r) xv[r] += gv[r] * (a[r] + bv[r]);
                    *(f4*)(px + q * 4) = xv;
                }
            }
        }
    }
}

struct MlpDbg {
    unsigned* buf = nullptr; long long cap = 0, used = 0; int nlaunch = 0; long long off[64]; int rows[64];
    float* xbuf = nullptr; long long xcap = 0, xused = 0; int nx = 0; long long xoff[64]; int xrows[64];
};
MlpDbg g_mdbg;

template <typename T>
int launch_t(const MlpFusedCall& c, hipStream_t stream) {
    MlpParams p;
    p.x = c.x; p.ln_w = c.ln_w; p.ln_b = c.ln_b; p.w1f = c.w1f; p.b1 = c.b1; p.w2f = c.w2f; p.b2 = c.b2; p.gamma = c.gamma;
    p.eps = c.eps; p.M = c.M; p.hidden = c.hidden;
    // chunk-order stagger is off by default: it buys no throughput (the chunk loop is not DMA-bound: profiles/r02_ring_depth_ab.log).
    // (r02 first blamed it for run-to-run differences under concurrent stream shards; the cause was the ds_bpermute lane exchange of the
    // LayerNorm prologue, see fvit_common.h / profiles/r02_repeatability_hunt.log -- the stagger only moved the timing.)
    p.stagger = tune_get("mlp_stagger", 0);
    p.ablate = diag_knob("mlp_ablate");
    p.dbg = nullptr;
    p.dbgx = nullptr;
    std::lock_guard<std::recursive_mutex> dlock(diag_mutex());   // trace bookkeeping + the timer record of this launch
    if (g_mdbg.xbuf && c.C == 256 && g_mdbg.nx < 64 && g_mdbg.xused + (long long)c.M * c.C <= g_mdbg.xcap) {
        p.dbgx = g_mdbg.xbuf + g_mdbg.xused;
        g_mdbg.xoff[g_mdbg.nx] = g_mdbg.xused;
        g_mdbg.xrows[g_mdbg.nx++] = c.M;
        g_mdbg.xused += (long long)c.M * c.C;
    }
    if (g_mdbg.buf && c.C == 256 && g_mdbg.nlaunch < 64) {
        const long long need = (long long)((c.M + 63) / 64) * 4 * (4 + 5 * (c.hidden / 32)) * 64;
        if (g_mdbg.used + need <= g_mdbg.cap) {
            p.dbg = g_mdbg.buf + g_mdbg.used;
            g_mdbg.off[g_mdbg.nlaunch] = g_mdbg.used;
            g_mdbg.rows[g_mdbg.nlaunch++] = c.M;
            g_mdbg.used += need;
        }
    }
    const double flops = 4.0 * c.M * (double)c.C * c.hidden;
    const double bytes = 8.0 * c.M * (double)c.C + 4.0 * c.C * (double)c.hidden;
    ProfScope prof(FVIT_K_MLP_FUSED, flops, bytes, stream);
    prof_note(c.C == 256 ? "mlp_fused_kernel<256>" : "mlp_fused_kernel<512>", (c.M + 63) / 64);
    if (c.C == 256) {
        // variants for within-process A/B (fvit_tune "mlp_variant"); 0 (16 rows per wave, input rows kept in registers) is the default;
        // 3 = the r01 v1-v10 default (32 rows per wave, X re-read in the epilogue)
        int variant = tune_get("mlp_variant", -1);
        // auto: 32 rows per wave (half the weight traffic per row) once 128-row workgroups fill the chip with two per CU, else 16 rows per
        // wave with the input rows kept in registers (M = 54272: 96-107 vs 117-134 us; M = 18020: 72-74 vs 58-59 us, r01 sweep r27)
        if (variant < 0) variant = (c.M + 127) / 128 >= 400 ? 3 : 0;
        // (r06: the opt-in variants that lost their A/Bs -- the 4-deep ring, 137 KiB of LDS ("mlp_ring4_max_grid", r02: 77.9 vs 52.7 us at 282 workgroups); 8 waves x 128
        // rows; 64-row workgroups with X re-read; 32 rows per wave with the rows in registers (spills); two waves of 32 rows (437 VGPRs: 57.3 vs 49.0 us) -- are no longer
        // instantiated: git history, profiles/HISTORY.md)
        switch (variant) {
            case 3: hipLaunchKernelGGL((mlp_fused_kernel<T, 256, 2, 4, 2, false>), dim3((c.M + 127) / 128), dim3(256), 0, stream, p); break;
            default:
                if (p.dbg || p.dbgx || (p.ablate & (64 | 128)))   // diagnosis build of the same kernel (fvit_debug_mlp_*)
                    hipLaunchKernelGGL((mlp_fused_kernel<T, 256, 1, 4, 2, true, 2, true>), dim3((c.M + 63) / 64), dim3(256), 0, stream, p);
                else
                    hipLaunchKernelGGL((mlp_fused_kernel<T, 256, 1, 4, 2, true>), dim3((c.M + 63) / 64), dim3(256), 0, stream, p);
                break;
        }
    } else if (c.C == 512) {
        // stage 3 of FasterViT-0: 16 rows per wave, one workgroup (136 KiB of LDS: two 64-KiB weight chunks) per CU; the 128
        // accumulator VGPRs leave no room to keep the fp32 rows, so the epilogue re-reads X
        hipLaunchKernelGGL((mlp_fused_kernel<T, 512, 1, 4, 1, false>), dim3((c.M + 63) / 64), dim3(256), 0, stream, p);
    } else {
        set_error("mlp_fused: C=%d has no fused instance", c.C);
        return FVIT_EINVAL;
    }
    return check_launch("mlp_fused_kernel");
}

}  // namespace

bool mlp_fused_supported(int C, int hidden) { return (C == 256 || C == 512) && hidden % 32 == 0 && hidden > 0 && hidden <= 4 * C; }

int launch_mlp_fused(const MlpFusedCall& c, hipStream_t stream) {
    if (!mlp_fused_supported(c.C, c.hidden) || c.M <= 0 || !c.x || !c.w1f || !c.w2f) {
        set_error("mlp_fused: unsupported arguments C=%d hidden=%d M=%d", c.C, c.hidden, c.M);
        return FVIT_EINVAL;
    }
    if (c.dtype == FVIT_F16) return launch_t<_Float16>(c, stream);
    if (c.dtype == FVIT_BF16) return launch_t<__bf16>(c, stream);
    set_error("mlp_fused: operand dtype %d not supported", c.dtype);
    return FVIT_EINVAL;
}

}  // namespace fvit

#ifdef FVIT_DIAG
extern "C" {
/* diagnosis: while active, every default-variant C = 256 fused-MLP launch writes per-lane hashes of its intermediate state (see dq in the
 * kernel) to consecutive slices of buf; _end returns the number of traced launches with their slice offsets (words) and row counts */
int fvit_debug_mlp_trace_begin(void* buf, int64_t capacity_words) {
    std::lock_guard<std::recursive_mutex> lock(fvit::diag_mutex());
    fvit::g_mdbg.buf = (unsigned*)buf; fvit::g_mdbg.cap = capacity_words; fvit::g_mdbg.used = 0; fvit::g_mdbg.nlaunch = 0;
    return FVIT_OK;
}
int fvit_debug_mlp_inputs_begin(void* buf, int64_t capacity_floats) {
    std::lock_guard<std::recursive_mutex> lock(fvit::diag_mutex());
    fvit::g_mdbg.xbuf = (float*)buf; fvit::g_mdbg.xcap = capacity_floats; fvit::g_mdbg.xused = 0; fvit::g_mdbg.nx = 0;
    return FVIT_OK;
}
int fvit_debug_mlp_inputs_end(int64_t* offsets, int32_t* rows, int32_t max_launches) {
    std::lock_guard<std::recursive_mutex> lock(fvit::diag_mutex());
    const int n = fvit::g_mdbg.nx < max_launches ? fvit::g_mdbg.nx : max_launches;
    for (int i = 0; i < n; ++i) { if (offsets) offsets[i] = fvit::g_mdbg.xoff[i]; if (rows) rows[i] = fvit::g_mdbg.xrows[i]; }
    fvit::g_mdbg.xbuf = nullptr;
    return n;
}
int fvit_debug_mlp_trace_end(int64_t* offsets, int32_t* rows, int32_t max_launches) {
    std::lock_guard<std::recursive_mutex> lock(fvit::diag_mutex());
    const int n = fvit::g_mdbg.nlaunch < max_launches ? fvit::g_mdbg.nlaunch : max_launches;
    for (int i = 0; i < n; ++i) { if (offsets) offsets[i] = fvit::g_mdbg.off[i]; if (rows) rows[i] = fvit::g_mdbg.rows[i]; }
    fvit::g_mdbg.buf = nullptr;
    return n;
}
}
#endif  // FVIT_DIAG
