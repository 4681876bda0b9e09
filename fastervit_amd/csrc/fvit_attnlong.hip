// fvit_attnlong.hip -- windowed multi-head attention for LONG windows (S > 208 tokens), gfx950.
//
//   out[w, q, head, :] = softmax_k( q.k * scale + bias(head, q, k) ) @ v        per (window w, head)
//
// Same contract as fvit_attn.hip (WindowAttention.forward FV:561-565 + PosEmbMLPSwinv2D FV:266-310) for the window sizes
// that kernel cannot hold in registers: the ImageNet-21k fine-tunes run stage 2 / 3 with ONE window of 24^2, 32^2 or 48^2
// = 576 .. 2304 tokens (faster_vit_4_21k_{384,512,768}: window_size [7,7,24,12] / [7,7,32,16] / [7,7,48,24]), and any-res
// inputs can produce more than 208 carrier tokens per image for hat_attn.
//
// Two things change with the length:
//   * the scores no longer fit a wave's registers -> key tiles of 32 with an ONLINE softmax (running max / sum per query,
//     the output accumulator rescaled by exp(m_old - m_new) per tile).  The transposed orientation of fvit_attn.hip is kept:
//     S^T[key][query] = K . Q^T, so a lane owns one query column, the running statistics are per-lane scalars, the tile's
//     probabilities are already the B operand of O^T[dim][query] += V^T . P^T, and the rescale is one multiply per register;
//   * the folded bias table [heads][S][S] would be 0.3 GB per block at S = 2304 -> the bias is looked up from the COMPACT
//     table t[head][(2w-1)^2] = 16 * sigmoid(cpb_mlp(relative_coords_table)) (FV:276-280 before the index gather), staged in
//     LDS per workgroup; relative_position_index (FV:243-258) is evaluated arithmetically:
//         index(q, k) = (yq - yk + w - 1) * (2w - 1) + (xq - xk + w - 1) = qbase(q) - kpos(k),
//     with kpos(k) = yk * (2w - 1) + xk precomputed per key in LDS.  The first n_g = S - w^2 tokens (carrier tokens in front
//     of the window, or the zero-padded part of a non-square carrier grid, FV:282-299) get bias 0 on their rows and columns.
//
// Workgroup = 4 waves = 64 queries of one (window, head); K fragments come straight from global/L2 (prefetched one tile
// ahead), V tiles are transposed through LDS once per workgroup (double-buffered, one barrier per tile).
//
// TT (r06; weight_terms 3 = the x3 operand modes, as fvit_attn.hip's TT instances): q, k, v are read as hi + lo images of the qkv rows (lo at column
// q_lo_off), scores = kh.qh + kl.qh + kh.ql, the tile's probabilities are split in registers (ph = round(e), pl = round(e - ph); the running sum takes
// the fp32 e), O += vh.ph + vl.ph + vh.pl with both V^T images in LDS, and the output leaves as hi + lo (lo at column o_lo_off).  This is what the
// 21k 384 / 512 / 768 fine-tunes (one window of 24^2 .. 48^2 tokens) and large carrier grids needed to run the ABSOLUTE-tolerance plan (missing 6).
#include "fvit_common.h"

namespace fvit {

namespace {

struct AttnLongParams {
    const void* qkv;
    void* out;
    const float* rel_table;   // f32 [heads][(2w-1)^2] or null (no bias)
    int ldq, ldo;
    int nwin, S, heads;
    int w, ng;                // window side of the bias table, leading tokens without bias
    int nqt;                  // 64-query tiles per (window, head)
    int tab_in_lds;           // the head's table fits the dynamic LDS allocation
    float scale;
    int q_lo_off, o_lo_off;   // TT: column offset (elements) of the lo image inside a qkv row / an output row
};

constexpr int LONG_VROW = 40;   // V^T row: 32 key slots + 8 pad elements (rows stay 16-byte aligned, bank stride broken)

template <typename T, int DP, bool TT = false>
__global__ __launch_bounds__(256) void attn_long_kernel(AttnLongParams p) {
    typedef typename Op16<T>::v8 v8;
    constexpr int KD = DP / 32;   // k-steps over head_dim for the score MFMA
    constexpr int DB = DP / 16;   // output-channel blocks
    constexpr int CH = DP / 8;    // 16-byte chunks per V row
    constexpr int NT = TT ? 2 : 1;             // operand terms
    constexpr int VT_TILE = DP * LONG_VROW;    // elements of one V^T image of one tile
    extern __shared__ __attribute__((aligned(16))) char smem_long[];
    // layout: [V^T tiles 2 buffers x NT terms x DP x LONG_VROW op16][kpos int32 x Spad32][table f32 x (2w-1)^2]
    T* vt_base = (T*)smem_long;
    const int Spad32 = (p.S + 31) & ~31;
    int* kpos = (int*)(smem_long + 2 * NT * VT_TILE * 2);
    float* tab = (float*)(kpos + Spad32);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, s = lane & 15;
    const int item = blockIdx.x / p.nqt, qt = blockIdx.x - item * p.nqt;
    const int win = item / p.heads, head = item - win * p.heads;
    const int HD = p.heads * DP;
    const int S = p.S;
    const int tw = 2 * p.w - 1;

    const T* __restrict__ qkv = (const T*)p.qkv + (size_t)win * S * p.ldq + head * DP;
    const float* __restrict__ gtab = p.rel_table ? p.rel_table + (size_t)head * tw * tw : nullptr;

    // ---- per-workgroup tables ----
    for (int k = tid; k < Spad32; k += 256) {
        int v = -1;                                   // no bias: carrier token or padding
        if (k >= p.ng && k < S) {
            const int l = k - p.ng, y = l / p.w;
            v = y * tw + (l - y * p.w);
        }
        kpos[k] = v;
    }
    if (gtab && p.tab_in_lds)
        for (int i = tid; i < tw * tw; i += 256) tab[i] = gtab[i];
    const float* __restrict__ btab = p.tab_in_lds ? tab : gtab;

    // ---- this lane's query ----
    const int qi = qt * 64 + wave * 16 + s;
    v8 qf[NT][KD];
#pragma unroll
    for (int tm = 0; tm < NT; ++tm)
#pragma unroll
        for (int kd = 0; kd < KD; ++kd) {
            v8 val;
#pragma unroll
            for (int j = 0; j < 8; ++j) val[j] = (T)0.f;
            if (qi < S) val = *(const v8*)(qkv + tm * p.q_lo_off + (size_t)qi * p.ldq + kd * 32 + g * 8);
            qf[tm][kd] = val;
        }
    int qbase = -1;                                   // < 0: this query row carries no bias
    if (gtab && qi >= p.ng && qi < S) {
        const int l = qi - p.ng, y = l / p.w;
        qbase = (y + p.w - 1) * tw + (l - y * p.w) + p.w - 1;
    }

    // V^T rows for A-row slot s of output block db: dim = (s>>2)*(DP/4) + db*4 + (s&3)  (as in fvit_attn.hip)
    int vrow[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) vrow[db] = ((s >> 2) * (DP / 4) + db * 4 + (s & 3)) * LONG_VROW + g * 8;

    // staging role of this thread: 16-byte chunk e = i*256 + tid of the tile's 32 x CH chunks -> key e / CH, chunk e % CH
    constexpr int NST = (32 * CH + 255) / 256;   // 1 (DP 32, 64) or 2 (DP 96)
    auto load_v = [&](int t, v8 (&vr)[NT][NST]) {
#pragma unroll
        for (int tm = 0; tm < NT; ++tm)
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int e = i * 256 + tid, key_l = e / CH, ch = e - key_l * CH;
                v8 val;
#pragma unroll
                for (int j = 0; j < 8; ++j) val[j] = (T)0.f;
                const int key = t * 32 + key_l;
                if (key_l < 32 && key < S) val = *(const v8*)(qkv + tm * p.q_lo_off + (size_t)key * p.ldq + 2 * HD + ch * 8);
                vr[tm][i] = val;
            }
    };
    auto load_k = [&](int t, v8 (&kf)[NT][2][KD]) {
#pragma unroll
        for (int tm = 0; tm < NT; ++tm)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const int key = t * 32 + jb * 16 + s;
#pragma unroll
                for (int kd = 0; kd < KD; ++kd) {
                    v8 val;
#pragma unroll
                    for (int j = 0; j < 8; ++j) val[j] = (T)0.f;
                    if (key < S) val = *(const v8*)(qkv + tm * p.q_lo_off + (size_t)key * p.ldq + HD + kd * 32 + g * 8);
                    kf[tm][jb][kd] = val;
                }
            }
    };

    const int ntiles = (S + 31) >> 5;
    float m = -3.0e38f, lsum = 0.f;
    f4 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) o[db] = (f4){0.f, 0.f, 0.f, 0.f};

    v8 vreg[NT][NST];
    load_v(0, vreg);
    v8 kf[NT][2][KD];
    load_k(0, kf);

    for (int t = 0; t < ntiles; ++t) {
        T* vt = vt_base + (t & 1) * NT * VT_TILE;   // hi image, then (TT) the lo image
#pragma unroll
        for (int tm = 0; tm < NT; ++tm)
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int e = i * 256 + tid, key_l = e / CH, ch = e - key_l * CH;
                if (key_l < 32) {
                    const int pos = ((key_l >> 2) & 3) * 8 + ((key_l >> 4) & 1) * 4 + (key_l & 3);
#pragma unroll
                    for (int j = 0; j < 8; ++j) vt[tm * VT_TILE + (ch * 8 + j) * LONG_VROW + pos] = vreg[tm][i][j];
                }
            }
        __syncthreads();   // tile t staged (and, for t = 0, the tables); buffer (t+1)&1 is free: its readers passed this barrier
        v8 kn[NT][2][KD];
        if (t + 1 < ntiles) {
            load_v(t + 1, vreg);
            load_k(t + 1, kn);
        }

        // scores^T: lane holds keys t*32 + jb*16 + g*4 + r of query qi
        f4 sc[2];
        float mx = -3.0e38f;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            f4 a = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kd = 0; kd < KD; ++kd) {
                a = Op16<T>::mfma(kf[0][jb][kd], qf[0][kd], a);
                if constexpr (TT) {
                    a = Op16<T>::mfma(kf[1][jb][kd], qf[0][kd], a);
                    a = Op16<T>::mfma(kf[0][jb][kd], qf[1][kd], a);
                }
            }
            const int k0 = t * 32 + jb * 16 + g * 4;
            const int4 kp = *(const int4*)(kpos + k0);
            const int kpv[4] = {kp.x, kp.y, kp.z, kp.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float b = 0.f;
                if (qbase >= 0 && kpv[r] >= 0) b = btab[qbase - kpv[r]];
                float v = a[r] * p.scale + b;
                if (k0 + r >= S) v = -3.0e38f;
                a[r] = v;
                mx = fmaxf(mx, v);
            }
            sc[jb] = a;
        }
        mx = max_xor32(max_xor16(mx));
        const float mn = fmaxf(m, mx);
        const float alpha = __expf(m - mn);
        m = mn;
        float rs = 0.f;
        v8 pf, pl;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(sc[jb][r] - mn);
                rs += e;
                const T eh = (T)e;
                pf[jb * 4 + r] = eh;
                if constexpr (TT) pl[jb * 4 + r] = (T)(e - (float)eh);
            }
        lsum = lsum * alpha + rs;   // per-lane partial sum (alpha is identical in the 4 lanes of a query); reduced at the end
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            f4 a = o[db];
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] *= alpha;
            const v8 vf = *(const v8*)(vt + vrow[db]);
            a = Op16<T>::mfma(vf, pf, a);
            if constexpr (TT) {
                const v8 vl = *(const v8*)(vt + VT_TILE + vrow[db]);
                a = Op16<T>::mfma(vl, pf, a);
                a = Op16<T>::mfma(vf, pl, a);
            }
            o[db] = a;
        }
        if (t + 1 < ntiles) {
#pragma unroll
            for (int tm = 0; tm < NT; ++tm)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int kd = 0; kd < KD; ++kd) kf[tm][jb][kd] = kn[tm][jb][kd];
        }
    }
    lsum = sum_xor32(sum_xor16(lsum));
    const float inv = 1.0f / lsum;

    // lane holds channels g*(DP/4) + db*4 + r of query qi: DP/4 consecutive channels
    if (qi < S) {
        T* po = (T*)p.out + ((size_t)win * S + qi) * p.ldo + head * DP + g * (DP / 4);
#pragma unroll
        for (int hseg = 0; hseg < DB / 2; ++hseg) {
            v8 ov, ol;
#pragma unroll
            for (int d2 = 0; d2 < 2; ++d2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y = o[hseg * 2 + d2][r] * inv;
                    const T yh = sat16<T>(y);
                    ov[d2 * 4 + r] = yh;
                    if constexpr (TT) ol[d2 * 4 + r] = sat16<T>(y - (float)yh);
                }
            *(v8*)(po + hseg * 8) = ov;
            if constexpr (TT) *(v8*)(po + p.o_lo_off + hseg * 8) = ol;
        }
    }
}

template <typename T, int DP, bool TT>
int launch_long_t(AttnLongParams& p, hipStream_t stream) {
    const int tw = 2 * p.w - 1;
    const size_t fixed = (TT ? 2 : 1) * 2 * DP * LONG_VROW * 2 + (size_t)((p.S + 31) & ~31) * 4;
    const size_t tabb = p.rel_table ? (size_t)tw * tw * 4 : 0;
    p.tab_in_lds = tabb > 0 && fixed + tabb <= 150 * 1024;
    const size_t lds = fixed + (p.tab_in_lds ? tabb : 0);
    if (fixed > 150 * 1024) {
        set_error("attention(long): %d tokens per window exceed the key-position table in LDS", p.S);
        return FVIT_EINVAL;
    }
    static DeviceOnce once;   // opt in to > 64 KiB of dynamic LDS, once per device and kernel instance
    if (once.first_on_current_device())
        (void)hipFuncSetAttribute((const void*)attn_long_kernel<T, DP, TT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    const int64_t grid = (int64_t)p.nwin * p.heads * p.nqt;
    if (grid > 0x7fffffff) {
        set_error("attention(long): grid too large");
        return FVIT_EINVAL;
    }
    prof_note(TT ? "attn_long_kernel<two-term>" : "attn_long_kernel", (int)grid);
    hipLaunchKernelGGL((attn_long_kernel<T, DP, TT>), dim3((unsigned)grid), dim3(256), lds, stream, p);
    return check_launch("attn_long_kernel");
}

}  // namespace

int launch_attention_long(const AttnCall& c, hipStream_t stream) {
    if (c.S <= 0 || c.nwin <= 0 || (c.dpad != 32 && c.dpad != 64 && c.dpad != 96) || (c.ldq % 8) || (c.ldo % 8)) {
        set_error("attention(long): unsupported geometry S=%d nwin=%d dpad=%d ldq=%d ldo=%d", c.S, c.nwin, c.dpad, c.ldq, c.ldo);
        return FVIT_EINVAL;
    }
    if (c.rel_table && (c.rel_w <= 0 || c.rel_ng < 0 || c.rel_ng + c.rel_w * c.rel_w != c.S)) {
        set_error("attention(long): bias table geometry w=%d n_g=%d does not cover S=%d tokens (need n_g + w^2 == S)", c.rel_w, c.rel_ng, c.S);
        return FVIT_EINVAL;
    }
    const bool tt = c.q_lo_off > 0 || c.o_lo_off > 0;
    if (tt && (c.q_lo_off < 3 * c.heads * c.dpad || c.ldq < c.q_lo_off + 3 * c.heads * c.dpad || (c.q_lo_off % 8) || c.o_lo_off < c.heads * c.dpad ||
               c.ldo < c.o_lo_off + c.heads * c.dpad || (c.o_lo_off % 8))) {
        set_error("attention(long): two-term activations need rows holding [hi | lo] images (ldq=%d q_lo_off=%d ldo=%d o_lo_off=%d)", c.ldq, c.q_lo_off, c.ldo, c.o_lo_off);
        return FVIT_EINVAL;
    }
    if (c.drop_mask) { set_error("attention(long): an attn_drop mask is not supported for windows beyond the dense kernel"); return FVIT_EINVAL; }
    AttnLongParams p;
    p.q_lo_off = c.q_lo_off; p.o_lo_off = c.o_lo_off;
    p.qkv = c.qkv; p.out = c.out; p.rel_table = c.rel_table; p.ldq = c.ldq; p.ldo = c.ldo;
    p.nwin = c.nwin; p.S = c.S; p.heads = c.heads; p.scale = c.scale;
    p.w = c.rel_table ? c.rel_w : 1; p.ng = c.rel_table ? c.rel_ng : c.S;
    p.nqt = (c.S + 63) / 64;
    p.tab_in_lds = 0;
    const double flops = 4.0 * c.nwin * (double)c.heads * c.S * (double)c.S * (c.d > 0 && c.d <= c.dpad ? c.d : c.dpad);
    const double bytes = (tt ? 2.0 : 1.0) * 2.0 * c.nwin * (double)c.S * c.heads * c.dpad * 4.0;
    ProfScope prof(FVIT_K_ATTENTION, flops, bytes, stream);
#define FVIT_LONG_DP(T, TT_) (c.dpad == 32 ? launch_long_t<T, 32, TT_>(p, stream) : c.dpad == 64 ? launch_long_t<T, 64, TT_>(p, stream) : launch_long_t<T, 96, TT_>(p, stream))
    if (c.dtype == FVIT_F16) return tt ? FVIT_LONG_DP(_Float16, true) : FVIT_LONG_DP(_Float16, false);
    if (c.dtype == FVIT_BF16) return tt ? FVIT_LONG_DP(__bf16, true) : FVIT_LONG_DP(__bf16, false);
#undef FVIT_LONG_DP
    set_error("attention(long): operand dtype %d not supported", c.dtype);
    return FVIT_EINVAL;
}

}  // namespace fvit
