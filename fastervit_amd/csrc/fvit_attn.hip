// fvit_attn.hip -- windowed multi-head attention core on packed q/k/v (gfx950).
//
//   out[w, q, head, :] = softmax_k( q.k * scale + bias[head, q, k] ) @ v     per (window w, head)
//
// Replaces the middle of WindowAttention.forward (AR:562-566 / FV:561-565) including the additive
// relative-position bias of PosEmbMLPSwinv2D (AR:310), which is folded to a constant table at load.
// The same kernel serves the window attention (S = ws^2 + cw^2 <= 208 tokens) and the carrier-token
// "global" attention (S = G tokens of one image).
//
// Design (CDNA4): one wave64 per (window, head), everything after the loads stays in registers.
//   * Scores are computed TRANSPOSED, S^T[key][query] = K . Q^T, with v_mfma_f32_16x16x32: K rows
//     are the A operand, Q rows the B operand, both read straight from HBM/L2 as 16-byte pieces
//     of the packed qkv rows (head_dim is padded to 32 or 64 at weight-pack time, so every
//     fragment is 16-byte aligned, also for head_dim 49).
//   * In that orientation a lane owns one query column: the softmax max/sum are 4-register
//     in-lane reductions plus two cross-lane steps (xor 16, 32) -- no LDS, no serial lanes.
//   * The probabilities are then ALREADY in the B-operand layout of the second MFMA
//     O^T[dim][query] = V^T . P^T (the k index of an MFMA may be permuted freely as long as both
//     operands agree), so P never leaves the registers.  Only V is staged through LDS, written
//     once per wave as V^T in exactly that k-slot order, and read back as one ds_read_b128 per
//     fragment.  V^T rows are assigned to A-row slots so that a lane ends up with 8 (dpad 32) or
//     16 (dpad 64) consecutive output channels: 16/32-byte stores.
//   * Padded key columns carry FVIT_MASK_BIAS in the folded bias table, so masking costs nothing.
#include "fvit_common.h"

namespace fvit {

namespace {

struct AttnParams {
    const void* qkv;
    void* out;
    const float* bias;
    int ldq, ldo;
    int nwin, S, heads;
    float scale;
    int q_lo_off, o_lo_off;   // TT instances: column offset (elements) of the lo image inside a qkv row / an output row
    const void* drop;         // attn_drop mask op16 [item][S][SP] (0 or 1 / keep) applied to the probabilities AFTER the softmax sum, or null
};

// SB: number of 16-token blocks (Spad / 16); DP: padded head dim (32, 64 or 96)
// TT (r04, weight_terms 3): every activation as TWO 16-bit terms -- q, k, v read as hi + lo images of the qkv rows, scores =
// qh.kh + qh.kl + ql.kh, the probabilities split in registers (ph + pl), O = ph.vh + ph.vl + pl.vh, the output stored as hi + lo.
// Two waves per workgroup (the V^T image is held twice); the K lo fragments are re-read per query block instead of living in registers.
template <typename T, int SB, int DP, bool TT = false>
__global__ __launch_bounds__(TT ? 128 : 256) void attn_kernel(AttnParams p) {
    typedef typename Op16<T>::v8 v8;
    constexpr int SP = SB * 16;
    constexpr int KD = DP / 32;            // k-steps over head_dim for the score MFMA
    constexpr int DB = DP / 16;            // output-channel blocks
    constexpr int KB = (SB + 1) / 2;       // 32-key blocks for the PV MFMA
    constexpr int VROW = KB * 32 + 8;      // V^T row length in elements (+8: keeps rows 16-B aligned, breaks bank stride)
    constexpr int NWV = TT ? 2 : 4, NT = TT ? 2 : 1;
    __shared__ __attribute__((aligned(16))) T vt_all[NWV][NT * DP * VROW];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * NWV + wave;  // (window, head)
    if (item >= p.nwin * p.heads) return;    // no block-wide barriers below
    const int win = item / p.heads, head = item - win * p.heads;
    const int HD = p.heads * DP;             // columns per q/k/v section
    const int S = p.S;
    const int g = lane >> 4, s = lane & 15;

    const T* __restrict__ qkv = (const T*)p.qkv + (size_t)win * S * p.ldq + head * DP;
    T* vt = vt_all[wave];

    // ---- stage V^T into LDS in PV k-slot order ----
    // key -> position: jb = key>>4, gg = (key>>2)&3, r = key&3; kb = jb>>1; i = (jb&1)*4 + r; pos = kb*32 + gg*8 + i
    {
        constexpr int CH = DP / 8;  // 16-byte chunks per V row
        for (int e = lane; e < KB * 32 * CH; e += 64) {
            const int key = e / CH, ch = e - key * CH;
            v8 val;
#pragma unroll
            for (int j = 0; j < 8; ++j) val[j] = (T)0.f;
            if (key < S) val = *(const v8*)(qkv + (size_t)key * p.ldq + 2 * HD + ch * 8);
            const int jb = key >> 4, gg = (key >> 2) & 3, r = key & 3;
            const int pos = (jb >> 1) * 32 + gg * 8 + (jb & 1) * 4 + r;
#pragma unroll
            for (int j = 0; j < 8; ++j) vt[(ch * 8 + j) * VROW + pos] = val[j];
            if constexpr (TT) {
#pragma unroll
                for (int j = 0; j < 8; ++j) val[j] = (T)0.f;
                if (key < S) val = *(const v8*)(qkv + p.q_lo_off + (size_t)key * p.ldq + 2 * HD + ch * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) vt[DP * VROW + (ch * 8 + j) * VROW + pos] = val[j];
            }
        }
    }

    // ---- K fragments (A operand of the score MFMA), kept in registers ----
    v8 kf[SB][KD];
#pragma unroll
    for (int jb = 0; jb < SB; ++jb) {
        const int key = jb * 16 + s;
#pragma unroll
        for (int kd = 0; kd < KD; ++kd) {
            v8 val;
#pragma unroll
            for (int j = 0; j < 8; ++j) val[j] = (T)0.f;
            if (key < S) val = *(const v8*)(qkv + (size_t)key * p.ldq + HD + kd * 32 + g * 8);
            kf[jb][kd] = val;
        }
    }

    // TT: the K lo fragments too, when they fit (r05: SB * KD <= 16 fragments = 64 registers; 7 x 7 windows + 4 carrier tokens at head_dim <= 64 are 4 x 2): they used
    // to be re-read from L2 inside the query loop for every (query block, key block, k step) -- the kernel ran at wait 0.75, MFMA busy 0.035
    // (profiles/r05_sq_counters_by_kernel_faster_vit_4_224_precise.json).  Same arithmetic, same order: bitwise the same result.
    constexpr bool KEEP_KL = TT && SB * KD <= 16;
    v8 klf[KEEP_KL ? SB : 1][KEEP_KL ? KD : 1];
    if constexpr (KEEP_KL) {
#pragma unroll
        for (int jb = 0; jb < SB; ++jb) {
            const int key = jb * 16 + s;
#pragma unroll
            for (int kd = 0; kd < KD; ++kd) {
                v8 val;
#pragma unroll
                for (int j = 0; j < 8; ++j) val[j] = (T)0.f;
                if (key < S) val = *(const v8*)(qkv + p.q_lo_off + (size_t)key * p.ldq + HD + kd * 32 + g * 8);
                klf[jb][kd] = val;
            }
        }
    }

    // V^T rows for A-row slot s of output block db: dim = (s>>2)*(DP/4) + db*4 + (s&3)
    const T* vrow[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) vrow[db] = vt + ((s >> 2) * (DP / 4) + db * 4 + (s & 3)) * VROW + g * 8;

    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    const float* __restrict__ bias = p.bias + (size_t)head * SP * SP;
    T* __restrict__ out = (T*)p.out + (size_t)win * S * p.ldo + head * DP;
    const int nqb = (S + 15) >> 4;

    for (int qb = 0; qb < nqb; ++qb) {
        const int qi = qb * 16 + s;  // this lane's query
        v8 qf[KD];
#pragma unroll
        for (int kd = 0; kd < KD; ++kd) {
            v8 val;
#pragma unroll
            for (int j = 0; j < 8; ++j) val[j] = (T)0.f;
            if (qi < S) val = *(const v8*)(qkv + (size_t)qi * p.ldq + kd * 32 + g * 8);
            qf[kd] = val;
        }
        v8 ql[TT ? KD : 1];
        if constexpr (TT) {
#pragma unroll
            for (int kd = 0; kd < KD; ++kd) {
                v8 val;
#pragma unroll
                for (int j = 0; j < 8; ++j) val[j] = (T)0.f;
                if (qi < S) val = *(const v8*)(qkv + p.q_lo_off + (size_t)qi * p.ldq + kd * 32 + g * 8);
                ql[kd] = val;
            }
        }
        // scores^T: lane holds keys jb*16 + g*4 + r (r = 0..3) of query qi
        f4 sc[SB];
        float mx = -3.0e38f;
#pragma unroll
        for (int jb = 0; jb < SB; ++jb) {
            f4 a = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kd = 0; kd < KD; ++kd) {
                if constexpr (TT) {   // the small products first: qh.kl + ql.kh, then qh.kh
                    v8 kl;
                    if constexpr (KEEP_KL) {
                        kl = klf[jb][kd];
                    } else {
                        const int key = jb * 16 + s;
#pragma unroll
                        for (int j = 0; j < 8; ++j) kl[j] = (T)0.f;
                        if (key < S) kl = *(const v8*)(qkv + p.q_lo_off + (size_t)key * p.ldq + HD + kd * 32 + g * 8);
                    }
                    a = Op16<T>::mfma(kl, qf[kd], a);
                    a = Op16<T>::mfma(kf[jb][kd], ql[kd], a);
                }
                a = Op16<T>::mfma(kf[jb][kd], qf[kd], a);
            }
            const f4 bz = *(const f4*)(bias + (size_t)qi * SP + jb * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a[r] = a[r] * p.scale + bz[r];
                mx = fmaxf(mx, a[r]);
            }
            sc[jb] = a;
        }
        mx = max_xor32(max_xor16(mx));
        float sum = 0.f;
#pragma unroll
        for (int jb = 0; jb < SB; ++jb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(sc[jb][r] - mx);
                sc[jb][r] = e;
                sum += e;
            }
        }
        sum = sum_xor32(sum_xor16(sum));
        const float inv = 1.0f / sum;
        if (p.drop) {   // attn_drop (train mode): P <- P . mask after the normalisation constant is fixed (FV:563-564: softmax, then Dropout)
            typedef typename Op16<T>::v4 v4;
            const T* dm = (const T*)p.drop + ((size_t)item * S + min(qi, S - 1)) * SP + g * 4;
#pragma unroll
            for (int jb = 0; jb < SB; ++jb) {
                const v4 m = *(const v4*)(dm + jb * 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[jb][r] *= (float)m[r];
            }
        }

        // O^T[dim][q] = V^T . P^T
        f4 o[DB];
#pragma unroll
        for (int db = 0; db < DB; ++db) o[db] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            v8 pf, pl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e0 = sc[2 * kb][r], e1 = (2 * kb + 1 < SB) ? sc[(2 * kb + 1 < SB) ? 2 * kb + 1 : 0][r] : 0.f;
                pf[r] = (T)e0;
                pf[4 + r] = (T)e1;
                if constexpr (TT) {
                    pl[r] = (T)(e0 - (float)pf[r]);
                    pl[4 + r] = (T)(e1 - (float)pf[4 + r]);
                }
            }
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const v8 vf = *(const v8*)(vrow[db] + kb * 32);
                if constexpr (TT) {
                    const v8 vl = *(const v8*)(vrow[db] + DP * VROW + kb * 32);
                    o[db] = Op16<T>::mfma(vl, pf, o[db]);
                    o[db] = Op16<T>::mfma(vf, pl, o[db]);
                }
                o[db] = Op16<T>::mfma(vf, pf, o[db]);
            }
        }
        // lane holds channels g*(DP/4) + db*4 + r of query qi: DP/4 consecutive channels
        if (qi < S) {
            T* po = out + (size_t)qi * p.ldo + g * (DP / 4);
            if (DP == 32) {
                v8 ov, ol;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ov[db * 4 + r] = sat16<T>(o[db][r] * inv);
                        ol[db * 4 + r] = sat16<T>(o[db][r] * inv - (float)ov[db * 4 + r]);
                    }
                *(v8*)po = ov;
                if constexpr (TT) *(v8*)(po + p.o_lo_off) = ol;
            } else {
#pragma unroll
                for (int hseg = 0; hseg < DB / 2; ++hseg) {
                    v8 ov, ol;
#pragma unroll
                    for (int d2 = 0; d2 < 2; ++d2)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            ov[d2 * 4 + r] = sat16<T>(o[hseg * 2 + d2][r] * inv);
                            ol[d2 * 4 + r] = sat16<T>(o[hseg * 2 + d2][r] * inv - (float)ov[d2 * 4 + r]);
                        }
                    *(v8*)(po + hseg * 8) = ov;
                    if constexpr (TT) *(v8*)(po + p.o_lo_off + hseg * 8) = ol;
                }
            }
        }
    }
}

template <typename T, int SB, int DP>
void launch_inst(const AttnParams& p, hipStream_t stream) {
    const int items = p.nwin * p.heads;
    if (p.q_lo_off > 0) hipLaunchKernelGGL((attn_kernel<T, SB, DP, true>), dim3((items + 1) / 2), dim3(128), 0, stream, p);
    else hipLaunchKernelGGL((attn_kernel<T, SB, DP>), dim3((items + 3) / 4), dim3(256), 0, stream, p);
}

template <typename T, int DP>
int launch_sb(const AttnParams& p, int sb, hipStream_t stream) {
    switch (sb) {
        case 1: launch_inst<T, 1, DP>(p, stream); break;
        case 2: launch_inst<T, 2, DP>(p, stream); break;
        case 3: launch_inst<T, 3, DP>(p, stream); break;
        case 4: launch_inst<T, 4, DP>(p, stream); break;
        case 5: launch_inst<T, 5, DP>(p, stream); break;
        case 6: launch_inst<T, 6, DP>(p, stream); break;
        case 7: launch_inst<T, 7, DP>(p, stream); break;
        case 8: launch_inst<T, 8, DP>(p, stream); break;
        case 10: if constexpr (DP <= 64) { launch_inst<T, 10, DP>(p, stream); break; }
        case 13: if constexpr (DP <= 64) { launch_inst<T, 13, DP>(p, stream); break; }
        default:
            set_error("attention: padded sequence of %d tokens has no kernel instance (supported: 16..128, 160, 208)",
                      sb * 16);
            return FVIT_EINVAL;
    }
    return check_launch("attn_kernel");
}

}  // namespace

bool attention_dense(int S, int dpad) {
    // the in-register kernel keeps all K fragments (SB x DP/32 x 4 VGPRs) and a V^T image (4 waves x DP x ~S x 2 B of LDS):
    // with the 96-wide head padding (head_dim 65..96: FasterViT-5/6) that fits up to 128 tokens
    return S >= 1 && S <= FVIT_MAX_DENSE_SEQ && (dpad <= 64 || S <= 128);
}

int launch_attention(const AttnCall& c, hipStream_t stream) {
    const bool tt = c.q_lo_off > 0 || c.o_lo_off > 0;
    if (!attention_dense(c.S, c.dpad)) return launch_attention_long(c, stream);   // (r06: the long kernel has its own two-term instances and argument checks)
    if (tt && (c.q_lo_off < 3 * c.heads * c.dpad || c.ldq < c.q_lo_off + 3 * c.heads * c.dpad || (c.q_lo_off % 8) ||
               c.o_lo_off < c.heads * c.dpad || c.ldo < c.o_lo_off + c.heads * c.dpad || (c.o_lo_off % 8))) {
        set_error("attention: two-term activations need rows holding [hi | lo] images (ldq=%d q_lo_off=%d ldo=%d o_lo_off=%d)", c.ldq, c.q_lo_off, c.ldo, c.o_lo_off);
        return FVIT_EINVAL;
    }
    if (c.S <= 0 || c.nwin <= 0 || (c.dpad != 32 && c.dpad != 64 && c.dpad != 96) || (c.ldq % 8) || (c.ldo % 8) || !c.bias) {
        set_error("attention: unsupported geometry S=%d nwin=%d dpad=%d ldq=%d ldo=%d bias=%p", c.S, c.nwin, c.dpad, c.ldq, c.ldo, (const void*)c.bias);
        return FVIT_EINVAL;
    }
    int sb = (c.S + 15) / 16;
    if (sb == 9) sb = 10;
    if (sb == 11 || sb == 12) sb = 13;
    AttnParams p;
    p.qkv = c.qkv; p.out = c.out; p.bias = c.bias; p.ldq = c.ldq; p.ldo = c.ldo;
    p.nwin = c.nwin; p.S = c.S; p.heads = c.heads; p.scale = c.scale;
    p.q_lo_off = c.q_lo_off; p.o_lo_off = c.o_lo_off;
    p.drop = c.drop_mask;
    if (c.drop_mask && tt) { set_error("attention: an attn_drop mask with two-term activations is not supported"); return FVIT_EINVAL; }
    // algorithmic FLOPs count the real head_dim (49 of FasterViT-4 runs on dpad = 64; padding work is not credited)
    const double flops = 4.0 * c.nwin * (double)c.heads * c.S * (double)c.S * (c.d > 0 && c.d <= c.dpad ? c.d : c.dpad);
    const double bytes = 2.0 * c.nwin * (double)c.S * c.heads * c.dpad * 4.0;  // q,k,v read + o write (16-bit)
    ProfScope prof(FVIT_K_ATTENTION, flops, bytes, stream);
    prof_note(tt ? "attn_kernel<two-term>" : "attn_kernel", tt ? (c.nwin * c.heads + 1) / 2 : (c.nwin * c.heads + 3) / 4);
#define FVIT_ATTN_DP(T) (c.dpad == 32 ? launch_sb<T, 32>(p, sb, stream) : c.dpad == 64 ? launch_sb<T, 64>(p, sb, stream) : launch_sb<T, 96>(p, sb, stream))
    if (c.dtype == FVIT_F16) return FVIT_ATTN_DP(_Float16);
    if (c.dtype == FVIT_BF16) return FVIT_ATTN_DP(__bf16);
#undef FVIT_ATTN_DP
    set_error("attention: operand dtype %d not supported", c.dtype);
    return FVIT_EINVAL;
}

}  // namespace fvit
