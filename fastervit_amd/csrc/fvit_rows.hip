// fvit_rows.hip -- HBM-bound row kernels of the HAT path (gfx950):
//   * gather + add + LayerNorm  (nn.LayerNorm AR:616,636,648-649; PosEmbMLPSwinv1D add AR:671,682;
//                                ct_dewindow / ct_window / torch.cat as row gathers AR:679,689-693)
//   * window_partition          (AR:84-88)  feature map -> f32 token rows (+ carrier tokens in front)
//   * window_reverse            (AR:91-94)  token rows -> feature map, with the any-res crop
//                                (AR:866-867) and the carrier-token propagation (AR:703-706) fused
// All of them move each byte once; they are priced against the HBM roof, not MFMA.
#include "fvit_common.h"

namespace fvit {

namespace {

// ------------------------------------------------------------------------------------------
// gather + add + LayerNorm: one wave64 per row, the row lives in registers (float4 per lane,
// up to MAXV of them => C <= 256 * MAXV), two-pass mean / variance like the fp32 reference.
// ------------------------------------------------------------------------------------------
struct LnParams {
    const float* srcA;
    const float* srcB;
    const int32_t* src_idx;
    const int32_t* add_idx;
    const float* add;
    float* x_out;
    void* n_out;
    const float* ln_w;
    const float* ln_b;
    float eps;
    int rowsA, rowsB, ldn;
    int rows, rows_per_image, C;
    int lw;       // columns of one term in an n_out row (pad columns C .. lw - 1 are zero-filled); = ldn unless lo_off > 0
    int lo_off;   // > 0: second term lo = round(y - hi) at column lo_off + c (two-term activations, weight_terms 3)
};

template <typename T, int MAXV>
__global__ __launch_bounds__(256) void ln_kernel(LnParams p) {
    typedef typename Op16<T>::v4 v4;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int C4 = p.C >> 2;
    const float* src;
    const float* addp = nullptr;
    const int b = row / p.rows_per_image, pr = row - b * p.rows_per_image;
    if (p.src_idx) {
        const int si = p.src_idx[pr];
        src = si >= 0 ? p.srcA + ((size_t)b * p.rowsA + si) * p.C : p.srcB + ((size_t)b * p.rowsB + (-si - 1)) * p.C;
    } else {
        src = p.srcA + (size_t)row * p.C;
    }
    if (p.add) {  // add row: add_idx[pr] (negative: none) or pr itself when no table is given
        const int ai = p.add_idx ? p.add_idx[pr] : pr;
        if (ai >= 0) addp = p.add + (size_t)ai * p.C;
    }
    f4 v[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c4 = lane + i * 64;
        if (c4 < C4) {
            f4 t = *(const f4*)(src + c4 * 4);
            if (addp) {
                const f4 a = *(const f4*)(addp + c4 * 4);
                t += a;
            }
            v[i] = t;
            sum += (t[0] + t[1]) + (t[2] + t[3]);
        } else {
            v[i] = (f4){0.f, 0.f, 0.f, 0.f};
        }
    }
    sum = group_sum<64>(sum);
    const float mean = sum / (float)p.C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c4 = lane + i * 64;
        if (c4 < C4) {
            const f4 d = v[i] - mean;
            sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    sq = group_sum<64>(sq);
    const float rstd = rsqrtf(sq / (float)p.C + p.eps);
    float* xo = p.x_out ? p.x_out + (size_t)row * p.C : nullptr;
    T* no = (T*)p.n_out + (size_t)row * p.ldn;
    const int L4 = p.lw >> 2;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c4 = lane + i * 64;
        if (c4 < C4) {
            if (xo) *(f4*)(xo + c4 * 4) = v[i];
            const f4 w = *(const f4*)(p.ln_w + c4 * 4);
            const f4 bb = *(const f4*)(p.ln_b + c4 * 4);
            v4 o, l;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = (v[i][r] - mean) * rstd * w[r] + bb[r];
                o[r] = sat16<T>(y);
                l[r] = sat16<T>(y - (float)o[r]);
            }
            *(v4*)(no + c4 * 4) = o;
            if (p.lo_off > 0) *(v4*)(no + p.lo_off + c4 * 4) = l;
        } else if (c4 < L4) {
            v4 z;
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = (T)0.f;
            *(v4*)(no + c4 * 4) = z;
            if (p.lo_off > 0) *(v4*)(no + p.lo_off + c4 * 4) = z;
        }
    }
}

template <typename T>
int launch_ln_t(const LnParams& p, hipStream_t stream) {
    const int grid = (p.rows + 3) / 4;
    const int c4 = p.C / 4;
    if (c4 <= 64 * 2 && p.lw / 4 <= 64 * 2) hipLaunchKernelGGL((ln_kernel<T, 2>), dim3(grid), dim3(256), 0, stream, p);
    else if (c4 <= 64 * 4 && p.lw / 4 <= 64 * 4) hipLaunchKernelGGL((ln_kernel<T, 4>), dim3(grid), dim3(256), 0, stream, p);
    else if (c4 <= 64 * 8 && p.lw / 4 <= 64 * 8) hipLaunchKernelGGL((ln_kernel<T, 8>), dim3(grid), dim3(256), 0, stream, p);
    else if (c4 <= 64 * 12 && p.lw / 4 <= 64 * 12) hipLaunchKernelGGL((ln_kernel<T, 12>), dim3(grid), dim3(256), 0, stream, p);
    else {
        set_error("layernorm: C=%d too wide (max 3072)", p.C);
        return FVIT_EINVAL;
    }
    return check_launch("ln_kernel");
}

// ------------------------------------------------------------------------------------------
// feature map <-> token rows.  Two thread mappings chosen on the map's channel stride:
//   channels-last maps (stride_c == 1): a wave walks the channels of one pixel -> both sides
//     coalesced, no LDS.
//   NCHW maps: a 64-pixel x 32-channel tile is transposed through LDS so that the map side is
//     read/written along W*H (contiguous) and the token side along C (contiguous).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float load_elem(const void* base, int64_t off, int dtype) {
    if (dtype == FVIT_F32) return ((const float*)base)[off];
    if (dtype == FVIT_F16) return (float)((const _Float16*)base)[off];
    return (float)((const __bf16*)base)[off];
}
__device__ __forceinline__ void store_elem(void* base, int64_t off, int dtype, float v) {
    if (dtype == FVIT_F32) ((float*)base)[off] = v;
    else if (dtype == FVIT_F16) ((_Float16*)base)[off] = (_Float16)v;
    else ((__bf16*)base)[off] = (__bf16)v;
}

struct MapParams {
    FvitMapView map;
    float* x;               // token rows (partition: dst, reverse: src)
    const float* ct;        // partition: optional carrier tokens (B, nW*ncw, C)
    const float* gamma;     // reverse: propagation scale (or null => 1)
    const int32_t* up_idx;  // reverse: propagation carrier slot per window token (or null)
    int batch, C, Hp, Wp, H, W, ws;
    int rows_per_win, row_off, ncw;
    int nwx;                // windows per row of windows (Wp / ws)
    int nw;                 // windows per image
};

// token row (relative to the whole x tensor) of padded pixel (b, y, x)
__device__ __forceinline__ int64_t token_row(const MapParams& p, int b, int y, int xx) {
    const int wy = y / p.ws, iy = y - wy * p.ws, wx = xx / p.ws, ix = xx - wx * p.ws;
    const int64_t win = (int64_t)b * p.nw + wy * p.nwx + wx;
    return win * p.rows_per_win + p.row_off + iy * p.ws + ix;
}

// --- channels-last: one wave per pixel ---
// MT: element type of the map (compile-time since r03: load_elem / store_elem's runtime dtype switch cost a branch per element and, on the
// 16-bit paths, a drained load per element).  VEC: C % 8 == 0 and a 16-byte aligned, channel-contiguous map: a lane moves 8 channels at a time
// (one 16-byte map access, two 16-byte row accesses) instead of one.
template <bool REVERSE, typename MT, bool VEC>
__global__ __launch_bounds__(256) void map_rows_cl_kernel(MapParams p) {
    const int lane = threadIdx.x & 63;
    const int hh = REVERSE ? p.H : p.Hp, wwid = REVERSE ? p.W : p.Wp;
    const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= (int64_t)p.batch * hh * wwid) return;
    const int b = (int)(pix / (hh * wwid));
    const int rem = (int)(pix - (int64_t)b * hh * wwid);
    const int y = rem / wwid, xx = rem - y * wwid;
    const int64_t trow = token_row(p, b, y, xx);
    float* xr = p.x + trow * p.C;
    MT* mp = (MT*)p.map.data + b * p.map.stride_b + y * p.map.stride_h + xx * p.map.stride_w;
    const float* cr = nullptr;
    if (REVERSE && p.up_idx) {
        const int tok = (int)((trow % p.rows_per_win) - p.row_off);
        cr = p.x + (trow - (trow % p.rows_per_win) + p.up_idx[tok]) * p.C;
    }
    if constexpr (VEC) {
        typedef MT m8 __attribute__((ext_vector_type(8)));
        for (int c = lane * 8; c < p.C; c += 512) {
            if (!REVERSE) {
                const m8 v = *(const m8*)(mp + c);
                f4 a, bq;
#pragma unroll
                for (int r = 0; r < 4; ++r) { a[r] = (float)v[r]; bq[r] = (float)v[4 + r]; }
                *(f4*)(xr + c) = a;
                *(f4*)(xr + c + 4) = bq;
            } else {
                f4 a = *(const f4*)(xr + c), bq = *(const f4*)(xr + c + 4);
                if (cr) {
                    const f4 ca = *(const f4*)(cr + c), cb = *(const f4*)(cr + c + 4);
                    const f4 ga = p.gamma ? *(const f4*)(p.gamma + c) : (f4){1.f, 1.f, 1.f, 1.f};
                    const f4 gb = p.gamma ? *(const f4*)(p.gamma + c + 4) : (f4){1.f, 1.f, 1.f, 1.f};
                    a += ga * ca;
                    bq += gb * cb;
                }
                m8 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = (MT)a[r]; v[4 + r] = (MT)bq[r]; }
                *(m8*)(mp + c) = v;
            }
        }
    } else {
        if (!REVERSE) {
            for (int c = lane; c < p.C; c += 64) xr[c] = (float)mp[c * p.map.stride_c];
        } else {
            for (int c = lane; c < p.C; c += 64) {
                float v = xr[c];
                if (cr) v += (p.gamma ? p.gamma[c] : 1.f) * cr[c];
                mp[c * p.map.stride_c] = (MT)v;
            }
        }
    }
}

// --- generic strides (NCHW): 64 pixels x 32 channels tile through LDS ---
template <bool REVERSE>
__global__ __launch_bounds__(256) void map_rows_tiled_kernel(MapParams p) {
    __shared__ float tile[32][65];
    const int hh = REVERSE ? p.H : p.Hp, wwid = REVERSE ? p.W : p.Wp;
    const int npix = hh * wwid;
    const int tiles_p = (npix + 63) / 64;
    const int tiles_c = (p.C + 31) / 32;
    int t = blockIdx.x;
    const int tc = t % tiles_c; t /= tiles_c;
    const int tp = t % tiles_p; t /= tiles_p;
    const int b = t;
    const int tid = threadIdx.x;
    if (!REVERSE) {
        // read along pixels (map side)
        const int pl = tid & 63, cl0 = tid >> 6;
        const int pix = tp * 64 + pl;
        if (pix < npix) {
            const int y = pix / wwid, xx = pix - y * wwid;
            const int64_t moff = b * p.map.stride_b + y * p.map.stride_h + xx * p.map.stride_w;
            for (int cl = cl0; cl < 32; cl += 4) {
                const int c = tc * 32 + cl;
                if (c < p.C) tile[cl][pl] = load_elem(p.map.data, moff + c * p.map.stride_c, p.map.dtype);
            }
        }
        __syncthreads();
        // write along channels (token side)
        const int cl = tid & 31, pl0 = tid >> 5;
        const int c = tc * 32 + cl;
        for (int pp = pl0; pp < 64; pp += 8) {
            const int pix2 = tp * 64 + pp;
            if (pix2 < npix && c < p.C) {
                const int y = pix2 / wwid, xx = pix2 - y * wwid;
                p.x[token_row(p, b, y, xx) * p.C + c] = tile[cl][pp];
            }
        }
    } else {
        const int cl = tid & 31, pl0 = tid >> 5;
        const int c = tc * 32 + cl;
        for (int pp = pl0; pp < 64; pp += 8) {
            const int pix2 = tp * 64 + pp;
            if (pix2 < npix && c < p.C) {
                const int y = pix2 / wwid, xx = pix2 - y * wwid;
                const int64_t trow = token_row(p, b, y, xx);
                float v = p.x[trow * p.C + c];
                if (p.up_idx) {
                    const int tok = (int)((trow % p.rows_per_win) - p.row_off);
                    const int64_t crow = trow - (trow % p.rows_per_win) + p.up_idx[tok];
                    v += (p.gamma ? p.gamma[c] : 1.f) * p.x[crow * p.C + c];
                }
                tile[cl][pp] = v;
            }
        }
        __syncthreads();
        const int pl = tid & 63, cl0 = tid >> 6;
        const int pix = tp * 64 + pl;
        if (pix < npix) {
            const int y = pix / wwid, xx = pix - y * wwid;
            const int64_t moff = b * p.map.stride_b + y * p.map.stride_h + xx * p.map.stride_w;
            for (int cl2 = cl0; cl2 < 32; cl2 += 4) {
                const int c2 = tc * 32 + cl2;
                if (c2 < p.C) store_elem(p.map.data, moff + c2 * p.map.stride_c, p.map.dtype, tile[cl2][pl]);
            }
        }
    }
}

// carrier tokens (B, nW*ncw, C) windowed order <-> rows [0, ncw) of every window in x
__global__ __launch_bounds__(256) void ct_rows_kernel(float* x, float* ct, int rows_per_win, int row_off, int ncw,
                                                      int64_t nrows, int C4, int to_x) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * C4) return;
    const int64_t r = i / C4;
    const int c4 = (int)(i - r * C4);
    const int64_t win = r / ncw;
    const int j = (int)(r - win * ncw);
    f4* px = (f4*)(x + (win * rows_per_win + row_off + j) * (int64_t)C4 * 4) + c4;
    f4* pc = (f4*)(ct + r * (int64_t)C4 * 4) + c4;
    if (to_x) *px = *pc; else *pc = *px;
}

// block-level API only: x[win][ncw + t] += gamma * x[win][up_idx[t]]  (AR:703-706)
__global__ __launch_bounds__(256) void propagate_kernel(float* x, const float* gamma, const int32_t* up_idx, int rows_per_win,
                                                        int ncw, int nloc, int64_t nwin, int C4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwin * nloc * C4) return;
    const int c4 = (int)(i % C4);
    const int64_t r = i / C4;
    const int64_t win = r / nloc;
    const int t = (int)(r - win * nloc);
    f4* px = (f4*)(x + (win * rows_per_win + ncw + t) * (int64_t)C4 * 4) + c4;
    const f4 cv = *((const f4*)(x + (win * rows_per_win + up_idx[t]) * (int64_t)C4 * 4) + c4);
    const f4 gv = gamma ? *((const f4*)gamma + c4) : (f4){1.f, 1.f, 1.f, 1.f};
    *px = *px + gv * cv;
}

// ------------------------------------------------------------------------------------------
// TokenInitializer (AR:715-750 / FV:709-738) in one pass: depthwise 3x3 conv (pad 1) + bias, AvgPool2d(k, s), and the
// view/permute into per-window carrier order, written as the f32 (B, G, C) tensor the stage kernel takes as ct_init.
// One thread per (image, carrier token, channel), channels fastest (coalesced on channels-last maps).
// ------------------------------------------------------------------------------------------
struct TokParams {
    FvitMapView in;      // (B, C, Hp, Wp) padded map
    const float* w;      // [C][3][3]
    const float* bias;   // [C]
    float* out;          // [B][G][C]
    int B, C, Hp, Wp, kh, kw, sh, sw, Ho, Wo, cw;
};

// IN: element type of the map (compile-time since r03: the runtime dtype switch of load_elem put a branch and -- on the 16-bit paths -- an
// s_waitcnt vmcnt(0) around every one of the 49 patch elements: 49 dependent L2 round trips per thread, 33 us for a 0.4-MFLOP op).
// SMALL: the (kh + 2) x (kw + 2) input patch fits 8 x 8 (pooling windows up to 6 x 6: every 224 / any-res entrypoint): the loops are fully
// unrolled, every load is issued (clamped address, masked weight) before the first use.  Larger windows (21k 384 / 512 / 768) take the loop form.
template <typename IN, bool SMALL>
__global__ __launch_bounds__(256) void token_init_kernel(TokParams p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int G = p.Ho * p.Wo;
    if (i >= (int64_t)p.B * G * p.C) return;
    const int c = (int)(i % p.C);
    const int64_t t = i / p.C;
    const int tok = (int)(t % G), b = (int)(t / G);
    // windowed order: tok = ((a * (Wo/cw) + bb) * cw + ii) * cw + kk  <->  pooled pixel (a*cw + ii, bb*cw + kk)
    const int kk = tok % p.cw, r1 = tok / p.cw, ii = r1 % p.cw, r2 = r1 / p.cw, nbw = p.Wo / p.cw, bb = r2 % nbw, a = r2 / nbw;
    const int oy = a * p.cw + ii, ox = bb * p.cw + kk;
    float wv[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) wv[j] = p.w[c * 9 + j];
    const IN* __restrict__ src = (const IN*)p.in.data + b * p.in.stride_b + c * p.in.stride_c;
    // sum over the pooling window of the 3x3 depthwise responses == weighted sum over the (kh+2) x (kw+2) input patch:
    // input (dy, dx) feeds conv outputs (dy - ky, dx - kx) relative to the window, valid for ky in [max(0, dy-kh+1), min(2, dy)]
    // and the same for kx -- the effective weight is separable into a valid-ky column sum and a valid-kx range sum
    float acc = 0.f;
    const int y0 = oy * p.sh - 1, x0 = ox * p.sw - 1;
    if constexpr (SMALL) {
        float val[8][8];
#pragma unroll
        for (int dy = 0; dy < 8; ++dy) {
            const int y = min(max(y0 + dy, 0), p.Hp - 1);
#pragma unroll
            for (int dx = 0; dx < 8; ++dx) {
                const int x = min(max(x0 + dx, 0), p.Wp - 1);
                val[dy][dx] = (float)src[y * p.in.stride_h + x * p.in.stride_w];   // clamped: always a legal address; masked by the weight below
            }
        }
#pragma unroll
        for (int dy = 0; dy < 8; ++dy) {
            const int y = y0 + dy;
            const bool yok = dy < p.kh + 2 && y >= 0 && y < p.Hp;
            const int ky_lo = max(0, dy - p.kh + 1), ky_hi = min(2, dy);
            const float m0 = (yok && ky_lo <= 0 && ky_hi >= 0) ? 1.f : 0.f, m1 = (yok && ky_lo <= 1 && ky_hi >= 1) ? 1.f : 0.f, m2 = (yok && ky_hi >= 2) ? 1.f : 0.f;
            const float cw0 = m0 * wv[0] + m1 * wv[3] + m2 * wv[6];
            const float cw1 = m0 * wv[1] + m1 * wv[4] + m2 * wv[7];
            const float cw2 = m0 * wv[2] + m1 * wv[5] + m2 * wv[8];
#pragma unroll
            for (int dx = 0; dx < 8; ++dx) {
                const int x = x0 + dx;
                const bool xok = dx < p.kw + 2 && x >= 0 && x < p.Wp;
                const int kx_lo = max(0, dx - p.kw + 1), kx_hi = min(2, dx);
                const float wsum = (kx_lo <= 0 && kx_hi >= 0 ? cw0 : 0.f) + (kx_lo <= 1 && kx_hi >= 1 ? cw1 : 0.f) + (kx_hi >= 2 ? cw2 : 0.f);
                acc += (xok ? wsum : 0.f) * val[dy][dx];
            }
        }
    } else {
        for (int dy = 0; dy < p.kh + 2; ++dy) {
            const int y = y0 + dy;
            if (y < 0 || y >= p.Hp) continue;
            const int ky_lo = max(0, dy - p.kh + 1), ky_hi = min(2, dy);
            // static register indexing only (a runtime-indexed register array would go to scratch)
            const float m0 = (ky_lo <= 0 && ky_hi >= 0) ? 1.f : 0.f, m1 = (ky_lo <= 1 && ky_hi >= 1) ? 1.f : 0.f, m2 = ky_hi >= 2 ? 1.f : 0.f;
            const float cw0 = m0 * wv[0] + m1 * wv[3] + m2 * wv[6];
            const float cw1 = m0 * wv[1] + m1 * wv[4] + m2 * wv[7];
            const float cw2 = m0 * wv[2] + m1 * wv[5] + m2 * wv[8];
            const IN* rowp = src + y * p.in.stride_h;
            for (int dx = 0; dx < p.kw + 2; ++dx) {
                const int x = x0 + dx;
                if (x < 0 || x >= p.Wp) continue;
                const int kx_lo = max(0, dx - p.kw + 1), kx_hi = min(2, dx);
                const float wsum = (kx_lo <= 0 && kx_hi >= 0 ? cw0 : 0.f) + (kx_lo <= 1 && kx_hi >= 1 ? cw1 : 0.f) + (kx_hi >= 2 ? cw2 : 0.f);
                acc += wsum * (float)rowp[x * p.in.stride_w];
            }
        }
    }
    p.out[i] = acc / (float)(p.kh * p.kw) + p.bias[c];
}

int launch_map(const MapParams& p, bool reverse, hipStream_t stream) {
    const int hh = reverse ? p.H : p.Hp, wwid = reverse ? p.W : p.Wp;
    if (p.map.stride_c == 1) {
        const int64_t pix = (int64_t)p.batch * hh * wwid;
        const int grid = (int)((pix + 3) / 4);
        const int esz = p.map.dtype == FVIT_F32 ? 4 : 2;
        const bool vec = esz == 2 && p.C % 8 == 0 && ((uintptr_t)p.map.data % 16) == 0 && (p.map.stride_b * esz) % 16 == 0 && (p.map.stride_h * esz) % 16 == 0 &&
                         (p.map.stride_w * esz) % 16 == 0;
#define FVIT_MAP_CL(MT_, VEC_) do { if (reverse) hipLaunchKernelGGL((map_rows_cl_kernel<true, MT_, VEC_>), dim3(grid), dim3(256), 0, stream, p); \
                                    else hipLaunchKernelGGL((map_rows_cl_kernel<false, MT_, VEC_>), dim3(grid), dim3(256), 0, stream, p); } while (0)
        if (p.map.dtype == FVIT_F32) FVIT_MAP_CL(float, false);
        else if (p.map.dtype == FVIT_F16) { if (vec) FVIT_MAP_CL(_Float16, true); else FVIT_MAP_CL(_Float16, false); }
        else { if (vec) FVIT_MAP_CL(__bf16, true); else FVIT_MAP_CL(__bf16, false); }
#undef FVIT_MAP_CL
    } else {
        const int tiles_p = (hh * wwid + 63) / 64, tiles_c = (p.C + 31) / 32;
        const int grid = p.batch * tiles_p * tiles_c;
        if (reverse) hipLaunchKernelGGL((map_rows_tiled_kernel<true>), dim3(grid), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((map_rows_tiled_kernel<false>), dim3(grid), dim3(256), 0, stream, p);
    }
    return check_launch(reverse ? "window_reverse" : "window_partition");
}

int elem_size(int dtype) { return dtype == FVIT_F32 ? 4 : 2; }

}  // namespace

int launch_gather_layernorm(const LnCall& c, hipStream_t stream) {
    if (c.rows <= 0 || (c.C % 4) || (c.ldn % 4) || c.ldn < c.C || c.lo_off < 0 || (c.lo_off % 4) || (c.lo_off > 0 && (c.lo_off < c.C || c.ldn < 2 * c.lo_off))) {
        set_error("layernorm: unsupported shape rows=%d C=%d ldn=%d", c.rows, c.C, c.ldn);
        return FVIT_EINVAL;
    }
    LnParams p;
    p.srcA = c.srcA; p.srcB = c.srcB; p.src_idx = c.src_idx; p.add_idx = c.add_idx; p.add = c.add;
    p.x_out = c.x_out; p.n_out = c.n_out; p.ln_w = c.ln_w; p.ln_b = c.ln_b; p.eps = c.eps;
    p.rowsA = c.rowsA; p.rowsB = c.rowsB; p.ldn = c.ldn; p.rows = c.rows;
    p.rows_per_image = c.rows_per_image > 0 ? c.rows_per_image : 1; p.C = c.C;
    p.lo_off = c.lo_off; p.lw = c.lo_off > 0 ? c.lo_off : c.ldn;
    const double bytes = (double)c.rows * c.C * (4.0 + (c.x_out ? 4.0 : 0.0) + 2.0);
    ProfScope prof(FVIT_K_LAYERNORM, 8.0 * c.rows * (double)c.C, bytes, stream);
    prof_note("ln_kernel", (c.rows + 3) / 4);
    if (c.dtype == FVIT_F16) return launch_ln_t<_Float16>(p, stream);
    if (c.dtype == FVIT_BF16) return launch_ln_t<__bf16>(p, stream);
    set_error("layernorm: operand dtype %d not supported", c.dtype);
    return FVIT_EINVAL;
}

int launch_partition(const PartitionCall& c, hipStream_t stream) {
    if (c.ws <= 0 || (c.Hp % c.ws) || (c.Wp % c.ws) || (c.C % 4)) {
        set_error("window_partition: map %dx%d is not a multiple of the window %d (or C %% 4 != 0)", c.Hp, c.Wp, c.ws);
        return FVIT_EINVAL;
    }
    MapParams p;
    p.map = c.in; p.x = c.x; p.ct = c.ct; p.gamma = nullptr; p.up_idx = nullptr;
    p.batch = c.batch; p.C = c.C; p.Hp = c.Hp; p.Wp = c.Wp; p.H = c.Hp; p.W = c.Wp; p.ws = c.ws;
    p.rows_per_win = c.rows_per_win; p.row_off = c.row_off; p.ncw = c.ncw;
    p.nwx = c.Wp / c.ws; p.nw = (c.Hp / c.ws) * p.nwx;
    const double bytes = (double)c.batch * c.C * c.Hp * c.Wp * (elem_size(c.in.dtype) + 4.0);
    int rc;
    {
        ProfScope prof(FVIT_K_PARTITION, 0.0, bytes, stream);
        rc = launch_map(p, false, stream);
    }
    if (rc) return rc;
    if (c.ct && c.ncw > 0)
        rc = launch_ct_copy(c.x, c.rows_per_win, 0, c.ncw, const_cast<float*>(c.ct), c.batch * p.nw, c.C, 1, stream);
    return rc;
}

int launch_reverse(const ReverseCall& c, hipStream_t stream) {
    if (c.ws <= 0 || (c.Hp % c.ws) || (c.Wp % c.ws) || c.H > c.Hp || c.W > c.Wp) {
        set_error("window_reverse: bad geometry Hp=%d Wp=%d H=%d W=%d ws=%d", c.Hp, c.Wp, c.H, c.W, c.ws);
        return FVIT_EINVAL;
    }
    MapParams p;
    p.map = c.out; p.x = const_cast<float*>(c.x); p.ct = nullptr; p.gamma = c.gamma; p.up_idx = c.up_idx;
    p.batch = c.batch; p.C = c.C; p.Hp = c.Hp; p.Wp = c.Wp; p.H = c.H; p.W = c.W; p.ws = c.ws;
    p.rows_per_win = c.rows_per_win; p.row_off = c.row_off; p.ncw = c.row_off;
    p.nwx = c.Wp / c.ws; p.nw = (c.Hp / c.ws) * p.nwx;
    const double bytes = (double)c.batch * c.C * c.H * c.W * (elem_size(c.out.dtype) + 4.0);
    ProfScope prof(FVIT_K_REVERSE, 0.0, bytes, stream);
    return launch_map(p, true, stream);
}

int launch_ct_copy(float* x, int rows_per_win, int row_off, int ncw, float* ct, int nwin_total, int C, int to_x,
                   hipStream_t stream) {
    const int64_t nrows = (int64_t)nwin_total * ncw;
    const int C4 = C / 4;
    const int64_t n = nrows * C4;
    ProfScope prof(FVIT_K_OTHER, 0.0, 8.0 * nrows * C, stream);
    hipLaunchKernelGGL(ct_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, ct, rows_per_win, row_off, ncw,
                       nrows, C4, to_x);
    return check_launch("ct_rows_kernel");
}

int launch_token_init(const FvitMapView& in, const float* w, const float* bias, float* out, int B, int C, int Hp, int Wp, int kh, int kw,
                      int sh, int sw, int cw, hipStream_t stream) {
    TokParams p;
    p.in = in; p.w = w; p.bias = bias; p.out = out; p.B = B; p.C = C; p.Hp = Hp; p.Wp = Wp;
    p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw; p.cw = cw;
    p.Ho = (Hp - kh) / sh + 1;
    p.Wo = (Wp - kw) / sw + 1;
    if (kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || cw <= 0 || p.Ho <= 0 || p.Wo <= 0 || (p.Ho % cw) || (p.Wo % cw)) {
        set_error("token_init: bad pooling geometry k=%dx%d s=%dx%d on %dx%d (cw=%d)", kh, kw, sh, sw, Hp, Wp, cw);
        return FVIT_EINVAL;
    }
    const int64_t n = (int64_t)B * p.Ho * p.Wo * C;
    ProfScope prof(FVIT_K_OTHER, 0.0, (double)B * C * Hp * Wp * 2.0 + 4.0 * n, stream);
    const dim3 grid((unsigned)((n + 255) / 256));
    const bool small_patch = kh + 2 <= 8 && kw + 2 <= 8;
#define FVIT_TOK(IN_) do { if (small_patch) hipLaunchKernelGGL((token_init_kernel<IN_, true>), grid, dim3(256), 0, stream, p); \
                           else hipLaunchKernelGGL((token_init_kernel<IN_, false>), grid, dim3(256), 0, stream, p); } while (0)
    if (in.dtype == FVIT_F32) FVIT_TOK(float);
    else if (in.dtype == FVIT_F16) FVIT_TOK(_Float16);
    else if (in.dtype == FVIT_BF16) FVIT_TOK(__bf16);
    else { set_error("token_init: map dtype %d not supported", in.dtype); return FVIT_EINVAL; }
#undef FVIT_TOK
    return check_launch("token_init_kernel");
}

int launch_propagate(float* x, const float* gamma, const int32_t* up_idx, int rows_per_win, int ncw, int nloc, int nwin_total,
                     int C, hipStream_t stream) {
    const int C4 = C / 4;
    const int64_t n = (int64_t)nwin_total * nloc * C4;
    ProfScope prof(FVIT_K_OTHER, 0.0, 12.0 * nwin_total * nloc * (double)C, stream);
    hipLaunchKernelGGL(propagate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, gamma, up_idx, rows_per_win, ncw,
                       nloc, (int64_t)nwin_total, C4);
    return check_launch("propagate_kernel");
}

}  // namespace fvit
