// fvit_api.hip -- C ABI of libfvit_hip.so (see include/fvit_hip.h) and the host-side sequencing of
// one HAT stage: workspace layout, per-block launch order, error plumbing, kernel timer.
//
// The launch order of one block follows HAT.forward (AR:668-707 / FV:662-701):
//   carrier branch (hier only)  : gather(ct_dewindow)+pe+LN -> qkv -> attention -> proj+res
//                                 -> LN -> fc1+GELU -> fc2+res
//   window branch               : gather(cat(ct_window(ct), x+pe))+LN -> qkv -> attention -> proj+res
//                                 -> LN -> fc1+GELU -> fc2+res
// The fp32 residual stream X holds the carrier tokens of every window in front of its ws^2 local
// tokens for the whole stage, so torch.cat / split (AR:693,701) cost nothing.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "fvit_common.h"

namespace fvit {

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return FVIT_ELAUNCH;
    }
    return FVIT_OK;
}

// ------------------------------------------------------------------------------------------
// tuning knobs
// ------------------------------------------------------------------------------------------
static std::map<std::string, int>& tune_map() {
    static std::map<std::string, int> m;
    return m;
}
static std::mutex g_tune_mu;   // launches may come from several host threads (one per device under nn.DataParallel)

int tune_get(const char* key, int dflt) {
    std::lock_guard<std::mutex> lock(g_tune_mu);
    auto& m = tune_map();
    auto it = m.find(key);
    if (it != m.end()) return it->second;
    // environment override FVIT_TUNE_<key>=<int>, read once per key
    std::string env = std::string("FVIT_TUNE_") + key;
    const char* v = getenv(env.c_str());
    const int val = v ? atoi(v) : dflt;
    m[key] = val;
    return val;
}

// ------------------------------------------------------------------------------------------
// kernel timer: an event pair around every launch, on the launch stream
// ------------------------------------------------------------------------------------------
struct ProfRec {
    int kind;
    double flops, bytes;
    int grid;
    char name[40];
};
std::recursive_mutex& diag_mutex() {
    static std::recursive_mutex m;
    return m;
}
typedef std::lock_guard<std::recursive_mutex> DiagLock;
static bool g_prof_on = false;
static std::vector<hipEvent_t> g_events;  // 2 per record
static std::vector<ProfRec> g_recs;

ProfScope::ProfScope(int kind, double flops, double bytes, hipStream_t stream) : slot_(-1), stream_(stream) {
    DiagLock lock(diag_mutex());
    dbg_poison_before_launch(stream);   // no-op unless fvit_debug_poison_launches(sink != null) is active
    if (!g_prof_on) return;
    slot_ = (int)g_recs.size();
    while ((int)g_events.size() < 2 * (slot_ + 1)) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { slot_ = -1; return; }
        g_events.push_back(e);
    }
    g_recs.push_back({kind, flops, bytes, 0, {0}});
    hipEventRecord(g_events[2 * slot_], stream_);
}

ProfScope::~ProfScope() {
    DiagLock lock(diag_mutex());
    if (slot_ >= 0 && 2 * slot_ + 1 < (int)g_events.size()) hipEventRecord(g_events[2 * slot_ + 1], stream_);
}

void prof_note(const char* kernel, int grid) {
    DiagLock lock(diag_mutex());
    if (!g_prof_on || g_recs.empty()) return;
    ProfRec& r = g_recs.back();
    r.grid = grid;
    strncpy(r.name, kernel ? kernel : "", sizeof(r.name) - 1);
    r.name[sizeof(r.name) - 1] = 0;
}

// ------------------------------------------------------------------------------------------
// stage workspace layout
// ------------------------------------------------------------------------------------------
extern "C" int fvit_attention_spad(int32_t S);

struct StageLayout {
    int nW, nloc, ncw, S, G;
    int64_t Mx, Mc;          // window-tensor rows, carrier rows
    int ldn, ldqkv, ldao, ldh;
    size_t off_X, off_Xn, off_QKV, off_AO, off_H, off_R, off_Rn, off_RQKV, off_RAO, off_RH, off_SLAB, off_CNT, off_SPLITK, splitk_bytes, total;
};

static bool make_layout(const FvitStageDesc& d, StageLayout& L) {
    if (d.batch <= 0 || d.C <= 0 || d.heads <= 0 || d.C % d.heads || d.ws <= 0 || d.Hp % d.ws || d.Wp % d.ws ||
        (d.dpad != 32 && d.dpad != 64 && d.dpad != 96) || d.dpad < d.C / d.heads || d.hidden <= 0 || (d.C % 16) || (d.hidden % 16) ||
        (d.operand_dtype != FVIT_F16 && d.operand_dtype != FVIT_BF16) || (d.hier && d.cw <= 0) || d.weight_terms < 1 || d.weight_terms > 3) {
        set_error("stage descriptor rejected: batch=%d C=%d heads=%d dpad=%d ws=%d Hp=%d Wp=%d hidden=%d hier=%d cw=%d dtype=%d weight_terms=%d",
                  d.batch, d.C, d.heads, d.dpad, d.ws, d.Hp, d.Wp, d.hidden, d.hier, d.cw, d.operand_dtype, d.weight_terms);
        return false;
    }
    L.nW = (d.Hp / d.ws) * (d.Wp / d.ws);
    L.nloc = d.ws * d.ws;
    L.ncw = d.hier ? d.cw * d.cw : 0;
    L.S = L.nloc + L.ncw;
    L.G = L.ncw * L.nW;
    L.Mx = (int64_t)d.batch * L.nW * L.S;
    L.Mc = (int64_t)d.batch * L.G;
    L.ldn = round_up(d.C, FVIT_TILE_K);
    L.ldqkv = 3 * d.heads * d.dpad;
    L.ldao = round_up(d.heads * d.dpad, FVIT_TILE_K);
    L.ldh = round_up(d.hidden, FVIT_TILE_K);
    const int64_t Mxp = round_up64(L.Mx, 128), Mcp = round_up64(L.Mc, 128);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    L.off_X = take((size_t)L.Mx * d.C * 4);
    // weight_terms 3 ("x3"): every 16-bit activation row holds TWO terms [hi | lo] (row stride 2 x ld*, lo image at column ld*)
    const size_t at = d.weight_terms == 3 ? 2 : 1;
    L.off_Xn = take((size_t)Mxp * L.ldn * 2 * at);
    L.off_QKV = take((size_t)Mxp * L.ldqkv * 2 * at);
    L.off_AO = take((size_t)Mxp * L.ldao * 2 * at);
    L.off_H = take((size_t)Mxp * L.ldh * 2 * at);
    if (d.hier) {
        L.off_R = take((size_t)L.Mc * d.C * 4);
        L.off_Rn = take((size_t)Mcp * L.ldn * 2 * at);
        L.off_RQKV = take((size_t)Mcp * L.ldqkv * 2 * at);
        L.off_RAO = take((size_t)Mcp * L.ldao * 2 * at);
        L.off_RH = take((size_t)Mcp * L.ldh * 2 * at);
    } else {
        L.off_R = L.off_Rn = L.off_RQKV = L.off_RAO = L.off_RH = 0;
    }
    // split-hidden form of the C = 512 MLP kernel (fvit_winmlp.hip): fp32 partial outputs of up to 4 sibling workgroups per 64-row
    // group + one arrival counter per group (zero from the workspace's one-time zero fill; the kernel leaves them zero)
    L.off_SLAB = L.off_CNT = 0;
    if (d.C == 512 && winmlp_supported(d.C, d.hidden)) {
        L.off_SLAB = take(winmlp_split_slab_bytes(L.Mx, d.C, 4));
        L.off_CNT = take((size_t)(L.Mx / 32 + 8) * 4);   // >= windows (winblk split) and >= 64-row groups (winmlp split)
    }
    // deterministic split-K of the small-grid residual GEMMs (fvit_gemm.hip): fp32 partials [splits <= 8][rows][C] for the row counts whose 128 x 128
    // tile grid is small (the carrier-token branch; a narrow window branch such as stage 3 of FasterViT-4 at batch 43)
    // Reserved only when the opt-in knob is set AT LAYOUT TIME (fvit_tune "gemm_splitk" = 1 before the workspace is sized; ADVICE r04: the slab was
    // up to ~30 MB per stage, slot and geometry for a default-off feature).  Sized for the worst case launch_t can pick: 64-row tiles double the
    // tile count, so the split count computed here from 128-row tiles is an upper bound; launch_t re-checks splitk_bytes and falls back to no split.
    L.off_SPLITK = 0;
    L.splitk_bytes = 0;
    if (tune_get("gemm_splitk", 0)) {
        const int slots = tune_get("gemm_splitk_slots", 460);
        auto need = [&](int64_t rows) -> size_t {
            if (rows <= 0) return 0;
            const int64_t tiles = ((rows + 127) / 128) * ((d.C + 127) / 128);
            const int64_t sp = std::min<int64_t>(8, slots / std::max<int64_t>(tiles, 1));
            return sp >= 2 ? (size_t)sp * rows * d.C * 4 : 0;
        };
        const size_t nb = std::max(need(L.Mx), d.hier ? need(L.Mc) : (size_t)0);
        if (nb > 0) { L.off_SPLITK = take(nb); L.splitk_bytes = nb; }
    }
    L.total = off;
    const int want_s = fvit_attention_spad(L.S), want_g = d.hier ? fvit_attention_spad(L.G) : d.gpad;
    if (d.spad != want_s || d.gpad != want_g) {
        set_error("stage descriptor: spad/gpad %d/%d do not match fvit_attention_spad (%d/%d)", d.spad, d.gpad, want_s, want_g);
        return false;
    }
    return true;
}

#define FVIT_TRY(expr)            \
    do {                          \
        int rc__ = (expr);        \
        if (rc__ != FVIT_OK) return rc__; \
    } while (0)

// LayerNorm folded into the following Linear's A staging (fvit_lngemm.hip): C = 256 / 512 and launches small enough that the separate
// LayerNorm kernel is a dispatch-floor launch of its own (carrier-token branch, stage 3, shard-sized launches).  OPT-IN (fvit_tune
// "ln_gemm" = 1): measured r02 it removes 17-22 launches per stream shard but LOSES end to end (60.9k vs 74.3k images/s): its
// resident operand panel needs 96-128 KiB of LDS per workgroup, so no other stream shard's kernel can share the CU with it -- and
// that co-residency is what the stream shards' gain comes from (profiles/r02_ln_gemm_ab.log)
static bool use_ln_gemm(const FvitStageDesc& d, int N, int ldw, int ldo, int64_t rows) {
    return d.weight_terms == 1 && ln_gemm_supported(d.C, N, ldw, ldo) && tune_get("ln_gemm", 0) && rows <= tune_get("ln_gemm_max_rows", 16384);
}

// LN -> qkv -> attention -> proj + gamma-residual, on `rows` rows of the f32 stream `x`
static int run_attn(const FvitStageDesc& d, const StageLayout& L, const FvitAttnWeights& w, float* x, int64_t rows, void* xn,
                    void* qkv, void* ao, int nwin, int S, bool ln_done, hipStream_t st, bool qkv_done = false, char* wsb = nullptr) {
    const int dt = d.operand_dtype;
    const int T = d.weight_terms;   // K-concatenated weight terms: the GEMMs run K = T x ld against the activation columns (GemmCall.ka)
    const int AT = T == 3 ? 2 : 1;  // activation terms: rows of xn / qkv / ao hold [hi | lo] images (stride AT x ld, lo at column ld)
    if (!ln_done && !qkv_done) {
        LnCall ln = {dt, x, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, xn, AT * L.ldn, w.ln_w, w.ln_b, 1e-5f, (int)rows, 1, d.C};
        if (AT == 2) ln.lo_off = L.ldn;
        FVIT_TRY(launch_gather_layernorm(ln, st));
    }
    dbg_rowhash("attn.xn", xn, rows, AT * L.ldn * 2, st);
    if (!qkv_done) {
        GemmCall g1 = {dt, xn, AT * L.ldn, w.w_qkv, T * L.ldn, w.b_qkv, nullptr, qkv, AT * L.ldqkv, (int)rows, L.ldqkv, T * L.ldn, 0};
        g1.ka = L.ldn;
        if (AT == 2) g1.out_lo_off = L.ldqkv;
        FVIT_TRY(launch_gemm(g1, st));
    }
    dbg_rowhash("attn.qkv", qkv, rows, AT * L.ldqkv * 2, st);
    const float scale = (d.qk_scale > 0.f ? d.qk_scale : 1.0f / sqrtf((float)(d.C / d.heads)));
    AttnCall at = {dt, qkv, AT * L.ldqkv, ao, AT * L.ldao, w.bias, nwin, S, d.heads, d.dpad, scale, w.rel_table, w.rel_w, w.rel_ng, d.C / d.heads};
    if (AT == 2) { at.q_lo_off = L.ldqkv; at.o_lo_off = L.ldao; }
    FVIT_TRY(launch_attention(at, st));
    dbg_rowhash("attn.ao", ao, rows, AT * L.ldao * 2, st);
    GemmCall g2 = {dt, ao, AT * L.ldao, w.w_proj, T * L.ldao, w.b_proj, w.gamma, x, d.C, (int)rows, d.C, T * L.ldao, 2};
    g2.ka = L.ldao;
    if (wsb && L.splitk_bytes) { g2.splitk_slab = (float*)(wsb + L.off_SPLITK); g2.splitk_bytes = L.splitk_bytes; }
    FVIT_TRY(launch_gemm(g2, st));
    dbg_rowhash("attn.out", x, rows, d.C * 4, st);
    return FVIT_OK;
}

// the fused attention block wins once the launch fills the chip (92 vs 117 us at 54k rows); on the carrier branch (4k rows,
// 32 workgroups) it is latency-bound and loses (38 vs 30 us)
// C = 512 / 16 heads (stage 3): the fused instance is correct but loses end to end (66.0k vs 71.9k images/s, r01 sweep r41): one
// workgroup per 49-token window streams 2 MiB of weights for 49 rows => opt-in (attn_fused512_min_rows)
static bool fused_attn_ok(const FvitStageDesc& d, const FvitAttnWeights& w, int S, int64_t rows) {
    // two-term weights (r04): the C = 256 window instance takes them ([hi image | lo image] fragment arrays); fvit_tune "attn_fused_x2" = 0 restores the r03 chain
    const bool terms_ok = d.weight_terms == 1 || (d.weight_terms == 2 && d.C == 256 && S > 48 && tune_get("attn_fused_x2", 1));
    return terms_ok && w.w_qkv_frag && w.b_qkv_heads && w.w_proj_frag && d.dpad == 32 && d.C / d.heads == 32 && attnblk_supported(d.C, d.heads, S) &&
           rows >= (d.C == 256 ? tune_get("attn_fused_min_rows", 16384) : tune_get("attn_fused512_min_rows", 1 << 30)) && tune_get("attn_fused", 1);
}

// LN -> fc1 + GELU -> fc2 + gamma-residual
struct NextPe {   // position-embedding rows of the NEXT block, applied in this block's fc2 epilogue (window branch, local-only stages)
    const float* add = nullptr;
    const int32_t* add_idx = nullptr;
    int rows_per_image = 1;
};

static bool win_mlp_ok(const FvitStageDesc& d, const FvitMlpWeights& w, int64_t rows) {
    if (d.weight_terms > 2 || !winmlp_supported(d.C, d.hidden) || !w.w_fc1_frag || !w.w_fc2_frag) return false;
    if (d.C == 512) return tune_get("win_mlp", 1) != 0;
    return rows >= tune_get("mlp_fused_min_rows", 16384) && tune_get("win_mlp256", 2) != 0;   // C = 256: 4-wave 64-row workgroups, two per CU (0: fvit_mlp_fused's kernel)
}

static bool mlp_takes_fused_kernel(const FvitStageDesc& d, const FvitMlpWeights& w, int64_t rows) {
    if (d.weight_terms != 1) return false;
    const int64_t fused_min = d.C == 256 ? tune_get("mlp_fused_min_rows", 16384) : tune_get("mlp_fused512_min_rows", 1 << 30);
    return w.w_fc1_frag && w.w_fc2_frag && mlp_fused_supported(d.C, d.hidden) && rows >= fused_min && tune_get("mlp_fused", 1);
}

static int run_mlp(const FvitStageDesc& d, const StageLayout& L, const FvitMlpWeights& w, float* x, int64_t rows, void* xn, void* h,
                   hipStream_t st, const NextPe* next_pe = nullptr, char* slab = nullptr) {
    const int dt = d.operand_dtype;
    const int T = d.weight_terms;
    // the fused kernel streams all MLP weights per 128-row workgroup: it wins once the launch fills the chip
    // (>= ~16k rows; 104 vs 137 us at 54k rows) and loses on the latency-bound carrier branch (4k rows: 83 vs 31 us)
    // C = 512 (stage 3): the fused instance is correct but slower than LN + 2 GEMMs at these row counts (65-196 workgroups, each
    // streaming 4 MiB of weights: 65.8k vs 71.0k images/s end to end, r01 sweep r31) => opt-in
    if (win_mlp_ok(d, w, rows)) {
        // C = 512 (stage 3 of FasterViT-0): 64-row workgroups whose waves split hidden units / output channels (fvit_winmlp.hip)
        if (next_pe && next_pe->add) { set_error("internal: position-embedding pre-add requested on the fused MLP path"); return FVIT_EINVAL; }
        MlpFusedCall mc = {dt, x, (int)rows, d.C, d.hidden, w.ln_w, w.ln_b, 1e-5f, w.w_fc1_frag, w.b_fc1, w.w_fc2_frag, w.b_fc2, w.gamma, T};
        if (slab && L.off_SLAB && rows <= L.Mx) {   // window branch of a C = 512 stage: hidden units split over sibling workgroups
            mc.slab = (float*)(slab + L.off_SLAB);
            mc.counters = (int*)(slab + L.off_CNT);
            mc.nsplit = tune_get("win_mlp_split", 1);
        }
        FVIT_TRY(launch_winmlp(mc, st));
        dbg_rowhash("winmlp.out", x, rows, d.C * 4, st);
        return FVIT_OK;
    }
    if (mlp_takes_fused_kernel(d, w, rows)) {
        if (next_pe && next_pe->add) { set_error("internal: position-embedding pre-add requested on the fused MLP path"); return FVIT_EINVAL; }
        MlpFusedCall mc = {dt, x, (int)rows, d.C, d.hidden, w.ln_w, w.ln_b, 1e-5f, w.w_fc1_frag, w.b_fc1, w.w_fc2_frag, w.b_fc2, w.gamma};
        dbg_rowhash("mlpf.in", x, rows, d.C * 4, st);
        FVIT_TRY(launch_mlp_fused(mc, st));
        dbg_rowhash("mlpf.out", x, rows, d.C * 4, st);
        return FVIT_OK;
    }
    const int AT = T == 3 ? 2 : 1;   // activation terms (see run_attn)
    LnCall ln = {dt, x, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, xn, AT * L.ldn, w.ln_w, w.ln_b, 1e-5f, (int)rows, 1, d.C};
    if (AT == 2) ln.lo_off = L.ldn;
    if (use_ln_gemm(d, d.hidden, L.ldn, L.ldh, rows)) {
        // norm2 -> fc1 -> GELU in one kernel (AR:697 with AR:401-403): the normalised rows never reach HBM
        LnGemmCall lg = {ln, w.w_fc1, L.ldn, w.b_fc1, h, L.ldh, d.hidden, 1};
        FVIT_TRY(launch_ln_gemm(lg, st));
    } else {
        FVIT_TRY(launch_gather_layernorm(ln, st));
        GemmCall g1 = {dt, xn, AT * L.ldn, w.w_fc1, T * L.ldn, w.b_fc1, nullptr, h, AT * L.ldh, (int)rows, d.hidden, T * L.ldn, 1};
        g1.ka = L.ldn;
        if (AT == 2) g1.out_lo_off = L.ldh;
        FVIT_TRY(launch_gemm(g1, st));
    }
    dbg_rowhash("mlp.h", h, rows, AT * L.ldh * 2, st);
    GemmCall g2 = {dt, h, AT * L.ldh, w.w_fc2, T * L.ldh, w.b_fc2, w.gamma, x, d.C, (int)rows, d.C, T * L.ldh, 2};
    g2.ka = L.ldh;
    if (next_pe && next_pe->add) { g2.add = next_pe->add; g2.add_idx = next_pe->add_idx; g2.rows_per_image = next_pe->rows_per_image; }
    if (slab && L.splitk_bytes) { g2.splitk_slab = (float*)(slab + L.off_SPLITK); g2.splitk_bytes = L.splitk_bytes; }
    FVIT_TRY(launch_gemm(g2, st));
    dbg_rowhash("mlp.out", x, rows, d.C * 4, st);
    return FVIT_OK;
}

// Local-only stages (no carrier tokens: stage 3 of FasterViT-0) with the LayerNorm-in-GEMM kernels: the in-place `x = x + pos_embed`
// of block i + 1 (AR:671) is applied by block i's fc2 epilogue, so block i + 1's norm1 is a plain LayerNorm of X and folds into its
// qkv GEMM.  Same two fp32 additions in the same order as the separate kernel (bitwise the same residual stream).
// C = 512 (stage 3 of FasterViT-0): the attention sub-block with the waves of a window splitting heads / output channels (fvit_winblk.hip)
static bool win_fused_ok(const FvitStageDesc& d, const FvitAttnWeights& w, int S) {
    if (d.weight_terms > 2 || (d.weight_terms != 1 && d.C != 512) || !winblk_supported(d.C, d.heads, S) || d.dpad != 32 || !w.w_qkv_frag || !w.b_qkv_heads || !w.w_proj_frag || !w.bias) return false;
    return d.C == 512 ? tune_get("win_fused", 1) != 0 : tune_get("win_fused256", 0) != 0;   // C = 256: the 4-wave form, two workgroups per CU
}

static bool pe_preadd_chain(const FvitStageDesc& d, const StageLayout& L, const FvitBlockWeights& w) {
    return !d.hier && !win_fused_ok(d, w.attn, L.S) && !win_mlp_ok(d, w.mlp, L.Mx) && use_ln_gemm(d, L.ldqkv, L.ldn, L.ldqkv, L.Mx) && !fused_attn_ok(d, w.attn, L.S, L.Mx) && !mlp_takes_fused_kernel(d, w.mlp, L.Mx) &&
           tune_get("pe_preadd", 1);
}

static int run_block(const FvitStageDesc& d, const StageLayout& L, const FvitBlockWeights& w, const FvitStageTables& t, char* ws,
                     hipStream_t st, bool pe_preadded = false, const NextPe* next_pe = nullptr) {
    const int dt = d.operand_dtype;
    float* X = (float*)(ws + L.off_X);
    void* Xn = ws + L.off_Xn;
    void* QKV = ws + L.off_QKV;
    void* AO = ws + L.off_AO;
    void* Hb = ws + L.off_H;
    float* R = (float*)(ws + L.off_R);
    const int rpi = L.nW * L.S;  // window-tensor rows per image
    if (d.hier) {
        void* Rn = ws + L.off_Rn;
        void* RQKV = ws + L.off_RQKV;
        void* RAO = ws + L.off_RAO;
        void* RH = ws + L.off_RH;
        const float scale = (d.qk_scale > 0.f ? d.qk_scale : 1.0f / sqrtf((float)(d.C / d.heads)));
        bool ct_done = false;
        if (d.weight_terms <= 2 && ctblk_supported(d.C, d.heads, L.G, d.hidden) && d.dpad == 32 && w.hat_attn.w_qkv_frag && w.hat_attn.b_qkv_heads && w.hat_attn.w_proj_frag &&
            w.hat_attn.bias && w.hat_mlp.w_fc1_frag && w.hat_mlp.w_fc2_frag && tune_get("ct_fused", 1)) {
            // the whole carrier-token branch (AR:679-686) in one kernel, one workgroup per image
            CtBlkCall cb = {dt, X, rpi, t.ct_src, (d.square ? w.pe_ct : nullptr), R, d.batch, L.G, d.heads, d.C, d.hidden,
                            w.hat_attn.ln_w, w.hat_attn.ln_b, w.hat_attn.w_qkv_frag, w.hat_attn.b_qkv_heads, w.hat_attn.w_proj_frag, w.hat_attn.b_proj,
                            w.hat_attn.gamma, w.hat_attn.bias, scale, w.hat_mlp.ln_w, w.hat_mlp.ln_b, w.hat_mlp.w_fc1_frag, w.hat_mlp.b_fc1,
                            w.hat_mlp.w_fc2_frag, w.hat_mlp.b_fc2, w.hat_mlp.gamma, 1e-5f, d.weight_terms};
            FVIT_TRY(launch_ctblk(cb, st));
            ct_done = true;
            dbg_rowhash("ct.block", R, L.Mc, d.C * 4, st);
        } else if (fused_attn_ok(d, w.hat_attn, L.G, L.Mc)) {
            // ct_dewindow gather (+ hat_pos_embed), LN, qkv, attention over the G carrier tokens, proj, gamma1-residual -> R
            AttnBlkCall ab = {dt, X, rpi, nullptr, 0, t.ct_src, nullptr, (d.square ? w.pe_ct : nullptr), w.hat_attn.ln_w, w.hat_attn.ln_b,
                              1e-5f, L.G, w.hat_attn.w_qkv_frag, w.hat_attn.b_qkv_heads, w.hat_attn.w_proj_frag, w.hat_attn.b_proj,
                              w.hat_attn.gamma, w.hat_attn.bias, R, d.batch, L.G, d.heads, d.C, scale};
            FVIT_TRY(launch_attnblk(ab, st));
            dbg_rowhash("ct.attnblk", R, L.Mc, d.C * 4, st);
        } else {
            // ct_dewindow gather (+ hat_pos_embed) -> R, LN(hat_norm1) -> Rn
            const int AT = d.weight_terms == 3 ? 2 : 1;
            LnCall ln = {dt, X, rpi, nullptr, 0, t.ct_src, nullptr, (d.square ? w.pe_ct : nullptr), R, Rn, AT * L.ldn,
                         w.hat_attn.ln_w, w.hat_attn.ln_b, 1e-5f, (int)L.Mc, L.G, d.C};
            if (AT == 2) ln.lo_off = L.ldn;
            if (use_ln_gemm(d, L.ldqkv, L.ldn, L.ldqkv, L.Mc)) {
                // ct_dewindow gather + hat_pos_embed + hat_norm1 + hat_attn.qkv in one kernel (AR:679-686); R (the fp32 carrier stream)
                // is written by the kernel's first column group
                LnGemmCall lg = {ln, w.hat_attn.w_qkv, L.ldn, w.hat_attn.b_qkv, RQKV, L.ldqkv, L.ldqkv, 0};
                FVIT_TRY(launch_ln_gemm(lg, st));
                FVIT_TRY(run_attn(d, L, w.hat_attn, R, L.Mc, Rn, RQKV, RAO, d.batch, L.G, true, st, true, ws));
            } else {
                FVIT_TRY(launch_gather_layernorm(ln, st));
                dbg_rowhash("ct.gather", R, L.Mc, d.C * 4, st);
                FVIT_TRY(run_attn(d, L, w.hat_attn, R, L.Mc, Rn, RQKV, RAO, d.batch, L.G, true, st, false, ws));
            }
        }
        if (!ct_done) FVIT_TRY(run_mlp(d, L, w.hat_mlp, R, L.Mc, Rn, RH, st, nullptr, ws));
    }
    if (win_fused_ok(d, w.attn, L.S)) {
        // C = 512 (stage 3 of FasterViT-0): the same sub-block with the waves of a window splitting heads / output channels
        const float scale = (d.qk_scale > 0.f ? d.qk_scale : 1.0f / sqrtf((float)(d.C / d.heads)));
        AttnBlkCall ab = {dt, X, rpi, R, L.G, (d.hier ? t.ln1_src : nullptr), t.ln1_add, w.pe_x, w.attn.ln_w, w.attn.ln_b, 1e-5f, rpi,
                          w.attn.w_qkv_frag, w.attn.b_qkv_heads, w.attn.w_proj_frag, w.attn.b_proj, w.attn.gamma, w.attn.bias, X,
                          d.batch * L.nW, L.S, d.heads, d.C, scale};
        if (L.off_SLAB && d.C == 512) {   // heads of a window split over two sibling workgroups (fvit_tune "win_blk_split" = 2)
            ab.slab = (float*)(ws + L.off_SLAB);
            ab.counters = (int*)(ws + L.off_CNT);
            ab.nsplit = tune_get("win_blk_split", 1);
        }
        ab.terms = d.weight_terms;
        FVIT_TRY(launch_winblk(ab, st));
        dbg_rowhash("win.winblk", X, L.Mx, d.C * 4, st);
    } else if (fused_attn_ok(d, w.attn, L.S, L.Mx)) {
        // cat(ct_window(ct), x + pos_embed) gather, LN(norm1), qkv, window attention, proj, gamma3-residual -> X, one kernel
        const float scale = (d.qk_scale > 0.f ? d.qk_scale : 1.0f / sqrtf((float)(d.C / d.heads)));
        AttnBlkCall ab = {dt, X, rpi, R, L.G, (d.hier ? t.ln1_src : nullptr), t.ln1_add, w.pe_x, w.attn.ln_w, w.attn.ln_b, 1e-5f, rpi,
                          w.attn.w_qkv_frag, w.attn.b_qkv_heads, w.attn.w_proj_frag, w.attn.b_proj, w.attn.gamma, w.attn.bias, X,
                          d.batch * L.nW, L.S, d.heads, d.C, scale};
        ab.terms = d.weight_terms;
        FVIT_TRY(launch_attnblk(ab, st));
        dbg_rowhash("win.attnblk", X, L.Mx, d.C * 4, st);
    } else if (pe_preadded) {
        // X already holds x + pos_embed (added by the previous block's fc2 epilogue): norm1 + qkv in one kernel
        LnCall ln1 = {dt, X, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, w.attn.ln_w, w.attn.ln_b, 1e-5f, (int)L.Mx, 1, d.C};
        LnGemmCall lg = {ln1, w.attn.w_qkv, L.ldn, w.attn.b_qkv, QKV, L.ldqkv, L.ldqkv, 0};
        FVIT_TRY(launch_ln_gemm(lg, st));
        FVIT_TRY(run_attn(d, L, w.attn, X, L.Mx, Xn, QKV, AO, d.batch * L.nW, L.S, true, st, true, ws));
    } else {
        // cat(ct_window(ct), x + pos_embed) gather -> X, LN(norm1) -> Xn
        const int AT = d.weight_terms == 3 ? 2 : 1;
        LnCall ln1 = {dt, X, rpi, R, L.G, (d.hier ? t.ln1_src : nullptr), t.ln1_add, w.pe_x, X, Xn, AT * L.ldn,
                      w.attn.ln_w, w.attn.ln_b, 1e-5f, (int)L.Mx, rpi, d.C};
        if (AT == 2) ln1.lo_off = L.ldn;
        FVIT_TRY(launch_gather_layernorm(ln1, st));
        dbg_rowhash("win.gather", X, L.Mx, d.C * 4, st);
        FVIT_TRY(run_attn(d, L, w.attn, X, L.Mx, Xn, QKV, AO, d.batch * L.nW, L.S, true, st, false, ws));
    }
    FVIT_TRY(run_mlp(d, L, w.mlp, X, L.Mx, Xn, Hb, st, next_pe, ws));
    return FVIT_OK;
}

static bool check_tables(const FvitStageDesc& d, const FvitStageTables* t) {
    if (!t || !t->ln1_add || (d.hier && (!t->ln1_src || !t->ct_src)) || (d.hier && d.do_propagation && !t->up_idx)) {
        set_error("stage tables missing (ln1_add%s)", d.hier ? ", ln1_src, ct_src, up_idx" : "");
        return false;
    }
    return true;
}

}  // namespace fvit

using namespace fvit;

extern "C" {

int fvit_abi_version(void) { return FVIT_ABI_VERSION; }
const char* fvit_last_error(void) { return g_err; }

int fvit_attention_dense(int32_t S, int32_t dpad) { return attention_dense(S, dpad) ? 1 : 0; }

int fvit_attention_spad(int32_t S) {
    int sb = (S + 15) / 16;
    if (sb == 9) sb = 10;
    if (sb == 11 || sb == 12) sb = 13;
    return sb * 16;
}

size_t fvit_stage_workspace_bytes(const FvitStageDesc* desc) {
    StageLayout L;
    if (!desc || !make_layout(*desc, L)) return 0;
    return L.total;
}

int fvit_workspace_init(const FvitStageDesc* desc, void* workspace, size_t bytes, fvit_stream_t stream) {
    StageLayout L;
    if (!desc || !make_layout(*desc, L)) return FVIT_EINVAL;
    if (!workspace || bytes < L.total) {
        set_error("workspace too small: %zu < %zu", bytes, L.total);
        return FVIT_EWORKSPACE;
    }
    if (hipMemsetAsync(workspace, 0, L.total, (hipStream_t)stream) != hipSuccess) return check_launch("workspace memset");
    return FVIT_OK;
}

int fvit_hat_stage_forward(const FvitStageDesc* desc, const FvitBlockWeights* blocks, const FvitStageTables* tables,
                           const FvitMapView* in, const float* ct_init, const FvitMapView* out, void* workspace,
                           size_t workspace_bytes, fvit_stream_t stream) {
    StageLayout L;
    if (!desc || !blocks || !in || !out || !make_layout(*desc, L)) {
        if (!desc || !blocks || !in || !out) set_error("null argument");
        return FVIT_EINVAL;
    }
    const FvitStageDesc& d = *desc;
    if (!check_tables(d, tables)) return FVIT_EINVAL;
    if (!workspace || workspace_bytes < L.total) {
        set_error("workspace too small: %zu < %zu", workspace_bytes, L.total);
        return FVIT_EWORKSPACE;
    }
    if (d.hier && !ct_init) {
        set_error("hierarchical stage needs ct_init");
        return FVIT_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* X = (float*)(ws + L.off_X);
    PartitionCall pc = {*in, d.batch, d.C, d.Hp, d.Wp, d.ws, X, L.S, L.ncw, d.hier ? ct_init : nullptr, L.ncw};
    FVIT_TRY(launch_partition(pc, st));
    dbg_rowhash("partition", X, L.Mx, d.C * 4, st);
    // (r03-r05 carried an opt-in ONE-launch form of a non-hierarchical C = 512 stage -- persistent per-window workgroups running the two kernel bodies alternately,
    // fvit_stage3.hip: measured -3 % in the joined launch structure, removed in r06: git history, profiles/HISTORY.md)
    bool pre = false;   // does X already include block i's position embedding?
    for (int i = 0; i < d.depth; ++i) {
        NextPe np;
        const bool chain = i + 1 < d.depth && pe_preadd_chain(d, L, blocks[i]) && pe_preadd_chain(d, L, blocks[i + 1]) && blocks[i + 1].pe_x;
        if (chain) { np.add = blocks[i + 1].pe_x; np.add_idx = tables->ln1_add; np.rows_per_image = L.nW * L.S; }
        FVIT_TRY(run_block(d, L, blocks[i], *tables, ws, st, pre, chain ? &np : nullptr));
        pre = chain;
    }
    const bool prop = d.hier && d.do_propagation && d.depth > 0 && blocks[d.depth - 1].last;
    ReverseCall rc = {X, L.S, L.ncw, d.batch, d.C, d.Hp, d.Wp, d.H, d.W, d.ws, *out,
                      prop ? blocks[d.depth - 1].hat_attn.gamma : nullptr, prop ? tables->up_idx : nullptr};
    FVIT_TRY(launch_reverse(rc, st));
    return FVIT_OK;
}

int fvit_hat_block_forward(const FvitStageDesc* desc, const FvitBlockWeights* block, const FvitStageTables* tables, float* x,
                           float* ct, void* workspace, size_t workspace_bytes, fvit_stream_t stream) {
    StageLayout L;
    if (!desc || !block || !x || !make_layout(*desc, L)) {
        if (!desc || !block || !x) set_error("null argument");
        return FVIT_EINVAL;
    }
    const FvitStageDesc& d = *desc;
    if (!check_tables(d, tables)) return FVIT_EINVAL;
    if (!workspace || workspace_bytes < L.total) {
        set_error("workspace too small: %zu < %zu", workspace_bytes, L.total);
        return FVIT_EWORKSPACE;
    }
    if (d.hier && !ct) {
        set_error("hierarchical block needs carrier tokens");
        return FVIT_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* X = (float*)(ws + L.off_X);
    const int nwin = d.batch * L.nW;
    FVIT_TRY(launch_ct_copy(X, L.S, L.ncw, L.nloc, x, nwin, d.C, 1, st));
    if (d.hier) FVIT_TRY(launch_ct_copy(X, L.S, 0, L.ncw, ct, nwin, d.C, 1, st));
    FVIT_TRY(run_block(d, L, *block, *tables, ws, st));
    if (d.hier && block->last && d.do_propagation)
        FVIT_TRY(launch_propagate(X, block->hat_attn.gamma, tables->up_idx, L.S, L.ncw, L.nloc, nwin, d.C, st));
    FVIT_TRY(launch_ct_copy(X, L.S, L.ncw, L.nloc, x, nwin, d.C, 0, st));
    if (d.hier) FVIT_TRY(launch_ct_copy(X, L.S, 0, L.ncw, ct, nwin, d.C, 0, st));
    return FVIT_OK;
}

int fvit_token_init(const FvitMapView* in, const float* weight, const float* bias, float* ct_out, int32_t batch, int32_t C, int32_t Hp,
                    int32_t Wp, int32_t pool_kh, int32_t pool_kw, int32_t pool_sh, int32_t pool_sw, int32_t cw, fvit_stream_t stream) {
    if (!in || !weight || !bias || !ct_out) { set_error("null argument"); return FVIT_EINVAL; }
    return launch_token_init(*in, weight, bias, ct_out, batch, C, Hp, Wp, pool_kh, pool_kw, pool_sh, pool_sw, cw, (hipStream_t)stream);
}

int fvit_window_partition(const FvitMapView* in, int32_t batch, int32_t C, int32_t Hp, int32_t Wp, int32_t ws, float* windows,
                          fvit_stream_t stream) {
    if (!in || !windows) { set_error("null argument"); return FVIT_EINVAL; }
    PartitionCall pc = {*in, batch, C, Hp, Wp, ws, windows, ws * ws, 0, nullptr, 0};
    return launch_partition(pc, (hipStream_t)stream);
}

int fvit_window_reverse(const float* windows, int32_t batch, int32_t C, int32_t Hp, int32_t Wp, int32_t H, int32_t W, int32_t ws,
                        const FvitMapView* out, fvit_stream_t stream) {
    if (!out || !windows) { set_error("null argument"); return FVIT_EINVAL; }
    ReverseCall rc = {windows, ws * ws, 0, batch, C, Hp, Wp, H, W, ws, *out, nullptr, nullptr};
    return launch_reverse(rc, (hipStream_t)stream);
}

int fvit_gemm_bias_act(int32_t operand_dtype, const void* A, int32_t lda, const void* Wt, int32_t ldw, const float* bias, void* out,
                       int32_t ldo, int32_t M, int32_t N, int32_t K, int32_t act, fvit_stream_t stream) {
    GemmCall g = {operand_dtype, A, lda, Wt, ldw, bias, nullptr, out, ldo, M, N, K, act ? 1 : 0};
    return launch_gemm(g, (hipStream_t)stream);
}

int fvit_gemm_residual(int32_t operand_dtype, const void* A, int32_t lda, const void* Wt, int32_t ldw, const float* bias,
                       const float* gamma, float* x, int32_t ldx, int32_t M, int32_t N, int32_t K, fvit_stream_t stream) {
    GemmCall g = {operand_dtype, A, lda, Wt, ldw, bias, gamma, x, ldx, M, N, K, 2};
    return launch_gemm(g, (hipStream_t)stream);
}

int fvit_gemm_residual_splitk(int32_t operand_dtype, const void* A, int32_t lda, const void* Wt, int32_t ldw, const float* bias, const float* gamma,
                              float* x, int32_t ldx, int32_t M, int32_t N, int32_t K, float* slab, size_t slab_bytes, fvit_stream_t stream) {
    GemmCall g = {operand_dtype, A, lda, Wt, ldw, bias, gamma, x, ldx, M, N, K, 2};
    g.splitk_slab = slab;
    g.splitk_bytes = slab_bytes;
    return launch_gemm(g, (hipStream_t)stream);
}

int fvit_gemm_terms(int32_t operand_dtype, const void* A, int32_t lda, const void* Wt, int32_t ldw, const float* bias, const float* gamma,
                    void* out, int32_t ldo, int32_t M, int32_t N, int32_t K, int32_t ka, int32_t epilogue, fvit_stream_t stream) {
    if (epilogue < 0 || epilogue > 2) { set_error("gemm_terms: epilogue %d", epilogue); return FVIT_EINVAL; }
    GemmCall g = {operand_dtype, A, lda, Wt, ldw, bias, epilogue == 2 ? gamma : nullptr, out, ldo, M, N, K, epilogue};
    g.ka = ka;
    return launch_gemm(g, (hipStream_t)stream);
}

int fvit_gemm_terms_lo(int32_t operand_dtype, const void* A, int32_t lda, const void* Wt, int32_t ldw, const float* bias, const float* gamma,
                       void* out, int32_t ldo, int32_t out_lo_off, int32_t M, int32_t N, int32_t K, int32_t ka, int32_t epilogue,
                       fvit_stream_t stream) {
    if (epilogue < 0 || epilogue > 2) { set_error("gemm_terms_lo: epilogue %d", epilogue); return FVIT_EINVAL; }
    GemmCall g = {operand_dtype, A, lda, Wt, ldw, bias, epilogue == 2 ? gamma : nullptr, out, ldo, M, N, K, epilogue};
    g.ka = ka;
    g.out_lo_off = out_lo_off;
    return launch_gemm(g, (hipStream_t)stream);
}

int fvit_window_attention_terms(int32_t operand_dtype, const void* qkv, int32_t ldq, int32_t q_lo_off, void* out, int32_t ldo,
                                int32_t o_lo_off, const float* bias, int32_t nwin, int32_t S, int32_t heads, int32_t dpad, float scale,
                                fvit_stream_t stream) {
    AttnCall a = {operand_dtype, qkv, ldq, out, ldo, bias, nwin, S, heads, dpad, scale, nullptr, 0, 0, 0};
    a.q_lo_off = q_lo_off;
    a.o_lo_off = o_lo_off;
    if (q_lo_off <= 0 || o_lo_off <= 0) { set_error("window_attention_terms: q_lo_off / o_lo_off must be > 0"); return FVIT_EINVAL; }
    return launch_attention(a, (hipStream_t)stream);
}

int fvit_gather_layernorm_terms(int32_t operand_dtype, const float* srcA, int32_t rowsA, const float* srcB, int32_t rowsB,
                                const int32_t* src_idx, const int32_t* add_idx, const float* add, float* x_out, void* n_out, int32_t ldn,
                                int32_t lo_off, const float* ln_w, const float* ln_b, float eps, int32_t rows, int32_t rows_per_image,
                                int32_t C, fvit_stream_t stream) {
    LnCall c = {operand_dtype, srcA, rowsA, srcB, rowsB, src_idx, add_idx, add, x_out, n_out, ldn, ln_w, ln_b, eps, rows,
                rows_per_image, C};
    c.lo_off = lo_off;
    return launch_gather_layernorm(c, (hipStream_t)stream);
}

int fvit_window_attention(int32_t operand_dtype, const void* qkv, int32_t ldq, void* out, int32_t ldo, const float* bias,
                          int32_t nwin, int32_t S, int32_t heads, int32_t dpad, float scale, fvit_stream_t stream) {
    if (!attention_dense(S, dpad)) {
        set_error("window_attention: S=%d at dpad=%d is outside the dense-bias kernel (fvit_attention_dense); use fvit_window_attention_long", S, dpad);
        return FVIT_EINVAL;
    }
    AttnCall a = {operand_dtype, qkv, ldq, out, ldo, bias, nwin, S, heads, dpad, scale, nullptr, 0, 0, 0};
    return launch_attention(a, (hipStream_t)stream);
}

int fvit_window_attention_drop(int32_t operand_dtype, const void* qkv, int32_t ldq, void* out, int32_t ldo, const float* bias, int32_t nwin, int32_t S,
                               int32_t heads, int32_t dpad, float scale, const void* drop_mask, fvit_stream_t stream) {
    if (!attention_dense(S, dpad)) {
        set_error("window_attention_drop: S=%d at dpad=%d is outside the dense-bias kernel", S, dpad);
        return FVIT_EINVAL;
    }
    AttnCall a = {operand_dtype, qkv, ldq, out, ldo, bias, nwin, S, heads, dpad, scale, nullptr, 0, 0, 0};
    a.drop_mask = drop_mask;
    return launch_attention(a, (hipStream_t)stream);
}

int fvit_window_attention_long(int32_t operand_dtype, const void* qkv, int32_t ldq, void* out, int32_t ldo, const float* rel_table,
                               int32_t rel_w, int32_t rel_ng, int32_t nwin, int32_t S, int32_t heads, int32_t dpad, float scale,
                               fvit_stream_t stream) {
    AttnCall a = {operand_dtype, qkv, ldq, out, ldo, nullptr, nwin, S, heads, dpad, scale, rel_table, rel_w, rel_ng, 0};
    return launch_attention_long(a, (hipStream_t)stream);
}

int fvit_window_attention_long_terms(int32_t operand_dtype, const void* qkv, int32_t ldq, int32_t q_lo_off, void* out, int32_t ldo, int32_t o_lo_off,
                                     const float* rel_table, int32_t rel_w, int32_t rel_ng, int32_t nwin, int32_t S, int32_t heads, int32_t dpad, float scale,
                                     fvit_stream_t stream) {
    AttnCall a = {operand_dtype, qkv, ldq, out, ldo, nullptr, nwin, S, heads, dpad, scale, rel_table, rel_w, rel_ng, 0};
    a.q_lo_off = q_lo_off;
    a.o_lo_off = o_lo_off;
    if (q_lo_off <= 0 || o_lo_off <= 0) { set_error("window_attention_long_terms: q_lo_off / o_lo_off must be > 0"); return FVIT_EINVAL; }
    return launch_attention_long(a, (hipStream_t)stream);
}

int fvit_gather_layernorm(int32_t operand_dtype, const float* srcA, int32_t rowsA, const float* srcB, int32_t rowsB,
                          const int32_t* src_idx, const int32_t* add_idx, const float* add, float* x_out, void* n_out, int32_t ldn,
                          const float* ln_w, const float* ln_b, float eps, int32_t rows, int32_t rows_per_image, int32_t C,
                          fvit_stream_t stream) {
    LnCall c = {operand_dtype, srcA, rowsA, srcB, rowsB, src_idx, add_idx, add, x_out, n_out, ldn, ln_w, ln_b, eps, rows,
                rows_per_image, C};
    return launch_gather_layernorm(c, (hipStream_t)stream);
}

int fvit_ln_gemm_supported(int32_t C, int32_t N, int32_t ldw, int32_t ldo) { return ln_gemm_supported(C, N, ldw, ldo) ? 1 : 0; }

int fvit_ln_gemm(int32_t operand_dtype, const float* srcA, int32_t rowsA, const float* srcB, int32_t rowsB, const int32_t* src_idx,
                 const int32_t* add_idx, const float* add, float* x_out, const float* ln_w, const float* ln_b, float eps, int32_t rows,
                 int32_t rows_per_image, int32_t C, const void* Wt, int32_t ldw, const float* bias, void* out, int32_t ldo, int32_t N,
                 int32_t act, fvit_stream_t stream) {
    LnCall ln = {operand_dtype, srcA, rowsA, srcB, rowsB, src_idx, add_idx, add, x_out, nullptr, 0, ln_w, ln_b, eps, rows, rows_per_image, C};
    LnGemmCall c = {ln, Wt, ldw, bias, out, ldo, N, act ? 1 : 0};
    return launch_ln_gemm(c, (hipStream_t)stream);
}

int fvit_attn_block_supported(int32_t C, int32_t heads, int32_t S) { return attnblk_supported(C, heads, S) ? 1 : 0; }

int fvit_attn_block_fused(int32_t operand_dtype, const float* srcA, int32_t rowsA, const float* srcB, int32_t rowsB, const int32_t* src_idx,
                          const int32_t* add_idx, const float* add, const float* ln_w, const float* ln_b, float eps,
                          int32_t rows_per_image, const void* w_qkv_frag, const float* b_qkv_heads, const void* w_proj_frag,
                          const float* b_proj, const float* gamma, const float* bias, float* x_out, int32_t nwin, int32_t S,
                          int32_t heads, int32_t C, float scale, fvit_stream_t stream) {
    AttnBlkCall ab = {operand_dtype, srcA, rowsA, srcB, rowsB, src_idx, add_idx, add, ln_w, ln_b, eps, rows_per_image, w_qkv_frag,
                      b_qkv_heads, w_proj_frag, b_proj, gamma, bias, x_out, nwin, S, heads, C, scale};
    return launch_attnblk(ab, (hipStream_t)stream);
}

#ifdef FVIT_DIAG
int fvit_debug_attn_block_timeline(const float* srcA, int32_t rowsA, const float* srcB, int32_t rowsB, const int32_t* src_idx,
                                   const int32_t* add_idx, const float* add, const float* ln_w, const float* ln_b, float eps,
                                   int32_t rows_per_image, const void* w_qkv_frag, const float* b_qkv_heads, const void* w_proj_frag,
                                   const float* b_proj, const float* gamma, const float* bias, float* x_out, int32_t nwin, int32_t S,
                                   int32_t heads, int32_t C, float scale, void* stamps, fvit_stream_t stream) {
    if (!stamps) { set_error("debug_attn_block_timeline: null stamp buffer"); return FVIT_EINVAL; }
    AttnBlkCall ab = {FVIT_F16, srcA, rowsA, srcB, rowsB, src_idx, add_idx, add, ln_w, ln_b, eps, rows_per_image, w_qkv_frag,
                      b_qkv_heads, w_proj_frag, b_proj, gamma, bias, x_out, nwin, S, heads, C, scale};
    ab.ts = stamps;
    return launch_attnblk(ab, (hipStream_t)stream);
}
#endif  // FVIT_DIAG

int fvit_win_mlp_supported(int32_t C, int32_t hidden) { return winmlp_supported(C, hidden) ? 1 : 0; }

int fvit_win_mlp_fused(int32_t operand_dtype, float* x, int32_t M, int32_t C, int32_t hidden, const float* ln_w, const float* ln_b,
                       float eps, const void* w_fc1_frag, const float* b_fc1, const void* w_fc2_frag, const float* b_fc2,
                       const float* gamma, fvit_stream_t stream) {
    MlpFusedCall mc = {operand_dtype, x, M, C, hidden, ln_w, ln_b, eps, w_fc1_frag, b_fc1, w_fc2_frag, b_fc2, gamma};
    return launch_winmlp(mc, (hipStream_t)stream);
}

int fvit_win_mlp_fused_terms(int32_t operand_dtype, float* x, int32_t M, int32_t C, int32_t hidden, const float* ln_w, const float* ln_b,
                             float eps, const void* w_fc1_frag, const float* b_fc1, const void* w_fc2_frag, const float* b_fc2,
                             const float* gamma, int32_t terms, fvit_stream_t stream) {
    MlpFusedCall mc = {operand_dtype, x, M, C, hidden, ln_w, ln_b, eps, w_fc1_frag, b_fc1, w_fc2_frag, b_fc2, gamma, terms};
    return launch_winmlp(mc, (hipStream_t)stream);
}

size_t fvit_win_mlp_split_bytes(int32_t M, int32_t C, int32_t nsplit) { return winmlp_split_slab_bytes(M, C, nsplit); }

int fvit_win_mlp_fused_split(int32_t operand_dtype, float* x, int32_t M, int32_t C, int32_t hidden, const float* ln_w, const float* ln_b,
                             float eps, const void* w_fc1_frag, const float* b_fc1, const void* w_fc2_frag, const float* b_fc2,
                             const float* gamma, int32_t terms, float* slab, int32_t* counters, int32_t nsplit, fvit_stream_t stream) {
    if (C != 512 || (nsplit != 1 && nsplit != 2) || (nsplit > 1 && (!slab || !counters))) {
        set_error("win_mlp_fused_split: C = %d nsplit = %d (C must be 512, nsplit 1 / 2 with scratch)", C, nsplit);
        return FVIT_EINVAL;
    }
    MlpFusedCall mc = {operand_dtype, x, M, C, hidden, ln_w, ln_b, eps, w_fc1_frag, b_fc1, w_fc2_frag, b_fc2, gamma, terms};
    mc.slab = slab; mc.counters = counters; mc.nsplit = nsplit;
    return launch_winmlp(mc, (hipStream_t)stream);
}

int fvit_win_block_supported(int32_t C, int32_t heads, int32_t S) { return winblk_supported(C, heads, S) ? 1 : 0; }

int fvit_win_block_fused(int32_t operand_dtype, const float* srcA, int32_t rowsA, const float* srcB, int32_t rowsB, const int32_t* src_idx,
                         const int32_t* add_idx, const float* add, const float* ln_w, const float* ln_b, float eps,
                         int32_t rows_per_image, const void* w_qkv_frag, const float* b_qkv_heads, const void* w_proj_frag,
                         const float* b_proj, const float* gamma, const float* bias, float* x_out, int32_t nwin, int32_t S,
                         int32_t heads, int32_t C, float scale, fvit_stream_t stream) {
    AttnBlkCall ab = {operand_dtype, srcA, rowsA, srcB, rowsB, src_idx, add_idx, add, ln_w, ln_b, eps, rows_per_image, w_qkv_frag,
                      b_qkv_heads, w_proj_frag, b_proj, gamma, bias, x_out, nwin, S, heads, C, scale};
    return launch_winblk(ab, (hipStream_t)stream);
}

#ifdef FVIT_DIAG
int fvit_debug_win_mlp_timeline(float* x, int32_t M, int32_t C, int32_t hidden, const float* ln_w, const float* ln_b, float eps,
                                const void* w_fc1_frag, const float* b_fc1, const void* w_fc2_frag, const float* b_fc2, const float* gamma,
                                void* stamps, fvit_stream_t stream) {
    if (!stamps) { set_error("debug_win_mlp_timeline: null stamp buffer"); return FVIT_EINVAL; }
    MlpFusedCall mc = {FVIT_F16, x, M, C, hidden, ln_w, ln_b, eps, w_fc1_frag, b_fc1, w_fc2_frag, b_fc2, gamma, 1};
    mc.ts = stamps;
    return launch_winmlp(mc, (hipStream_t)stream);
}
#endif  // FVIT_DIAG

int fvit_win_block_fused_split(int32_t operand_dtype, const float* srcA, int32_t rowsA, const float* srcB, int32_t rowsB, const int32_t* src_idx,
                               const int32_t* add_idx, const float* add, const float* ln_w, const float* ln_b, float eps,
                               int32_t rows_per_image, const void* w_qkv_frag, const float* b_qkv_heads, const void* w_proj_frag,
                               const float* b_proj, const float* gamma, const float* bias, float* x_out, int32_t nwin, int32_t S,
                               int32_t heads, int32_t C, float scale, float* slab, int32_t* counters, int32_t nsplit, fvit_stream_t stream) {
    if (C != 512 || (nsplit != 1 && nsplit != 2) || (nsplit > 1 && (!slab || !counters))) {
        set_error("win_block_fused_split: C = %d nsplit = %d (C must be 512, nsplit 1 / 2 with scratch)", C, nsplit);
        return FVIT_EINVAL;
    }
    AttnBlkCall ab = {operand_dtype, srcA, rowsA, srcB, rowsB, src_idx, add_idx, add, ln_w, ln_b, eps, rows_per_image, w_qkv_frag,
                      b_qkv_heads, w_proj_frag, b_proj, gamma, bias, x_out, nwin, S, heads, C, scale};
    ab.slab = slab; ab.counters = counters; ab.nsplit = nsplit;
    return launch_winblk(ab, (hipStream_t)stream);
}

int fvit_win_block_fused_terms(int32_t operand_dtype, const float* srcA, int32_t rowsA, const float* srcB, int32_t rowsB, const int32_t* src_idx,
                               const int32_t* add_idx, const float* add, const float* ln_w, const float* ln_b, float eps,
                               int32_t rows_per_image, const void* w_qkv_frag, const float* b_qkv_heads, const void* w_proj_frag,
                               const float* b_proj, const float* gamma, const float* bias, float* x_out, int32_t nwin, int32_t S,
                               int32_t heads, int32_t C, float scale, int32_t terms, fvit_stream_t stream) {
    AttnBlkCall ab = {operand_dtype, srcA, rowsA, srcB, rowsB, src_idx, add_idx, add, ln_w, ln_b, eps, rows_per_image, w_qkv_frag,
                      b_qkv_heads, w_proj_frag, b_proj, gamma, bias, x_out, nwin, S, heads, C, scale};
    ab.terms = terms;
    return launch_winblk(ab, (hipStream_t)stream);
}

int fvit_attn_block_fused_terms(int32_t operand_dtype, const float* srcA, int32_t rowsA, const float* srcB, int32_t rowsB, const int32_t* src_idx,
                                const int32_t* add_idx, const float* add, const float* ln_w, const float* ln_b, float eps,
                                int32_t rows_per_image, const void* w_qkv_frag, const float* b_qkv_heads, const void* w_proj_frag,
                                const float* b_proj, const float* gamma, const float* bias, float* x_out, int32_t nwin, int32_t S,
                                int32_t heads, int32_t C, float scale, int32_t terms, fvit_stream_t stream) {
    AttnBlkCall ab = {operand_dtype, srcA, rowsA, srcB, rowsB, src_idx, add_idx, add, ln_w, ln_b, eps, rows_per_image, w_qkv_frag,
                      b_qkv_heads, w_proj_frag, b_proj, gamma, bias, x_out, nwin, S, heads, C, scale};
    ab.terms = terms;
    return launch_attnblk(ab, (hipStream_t)stream);
}

int fvit_ct_block_supported(int32_t C, int32_t heads, int32_t G, int32_t hidden) { return ctblk_supported(C, heads, G, hidden) ? 1 : 0; }

int fvit_ct_block_fused(int32_t operand_dtype, const float* X, int32_t rowsA, const int32_t* src_idx, const float* add, float* R,
                        int32_t batch, int32_t G, int32_t heads, int32_t C, int32_t hidden, const float* ln1_w, const float* ln1_b,
                        const void* w_qkv_frag, const float* b_qkv_heads, const void* w_proj_frag, const float* b_proj, const float* gamma1,
                        const float* bias, float scale, const float* ln2_w, const float* ln2_b, const void* w_fc1_frag, const float* b_fc1,
                        const void* w_fc2_frag, const float* b_fc2, const float* gamma2, float eps, fvit_stream_t stream) {
    CtBlkCall cb = {operand_dtype, X, rowsA, src_idx, add, R, batch, G, heads, C, hidden, ln1_w, ln1_b, w_qkv_frag, b_qkv_heads, w_proj_frag,
                    b_proj, gamma1, bias, scale, ln2_w, ln2_b, w_fc1_frag, b_fc1, w_fc2_frag, b_fc2, gamma2, eps};
    return launch_ctblk(cb, (hipStream_t)stream);
}

int fvit_ct_block_fused_terms(int32_t operand_dtype, const float* X, int32_t rowsA, const int32_t* src_idx, const float* add, float* R,
                              int32_t batch, int32_t G, int32_t heads, int32_t C, int32_t hidden, const float* ln1_w, const float* ln1_b,
                              const void* w_qkv_frag, const float* b_qkv_heads, const void* w_proj_frag, const float* b_proj, const float* gamma1,
                              const float* bias, float scale, const float* ln2_w, const float* ln2_b, const void* w_fc1_frag, const float* b_fc1,
                              const void* w_fc2_frag, const float* b_fc2, const float* gamma2, float eps, int32_t terms, fvit_stream_t stream) {
    CtBlkCall cb = {operand_dtype, X, rowsA, src_idx, add, R, batch, G, heads, C, hidden, ln1_w, ln1_b, w_qkv_frag, b_qkv_heads, w_proj_frag,
                    b_proj, gamma1, bias, scale, ln2_w, ln2_b, w_fc1_frag, b_fc1, w_fc2_frag, b_fc2, gamma2, eps, terms};
    return launch_ctblk(cb, (hipStream_t)stream);
}

#ifdef FVIT_DIAG
int fvit_debug_ct_block_timeline(const float* X, int32_t rowsA, const int32_t* src_idx, const float* add, float* R,
                                 int32_t batch, int32_t G, int32_t heads, int32_t C, int32_t hidden, const float* ln1_w, const float* ln1_b,
                                 const void* w_qkv_frag, const float* b_qkv_heads, const void* w_proj_frag, const float* b_proj, const float* gamma1,
                                 const float* bias, float scale, const float* ln2_w, const float* ln2_b, const void* w_fc1_frag, const float* b_fc1,
                                 const void* w_fc2_frag, const float* b_fc2, const float* gamma2, float eps, void* stamps, fvit_stream_t stream) {
    if (!stamps) { set_error("debug_ct_block_timeline: null stamp buffer"); return FVIT_EINVAL; }
    CtBlkCall cb = {FVIT_F16, X, rowsA, src_idx, add, R, batch, G, heads, C, hidden, ln1_w, ln1_b, w_qkv_frag, b_qkv_heads, w_proj_frag,
                    b_proj, gamma1, bias, scale, ln2_w, ln2_b, w_fc1_frag, b_fc1, w_fc2_frag, b_fc2, gamma2, eps, 1};
    cb.ts = stamps;
    return launch_ctblk(cb, (hipStream_t)stream);
}
#endif  // FVIT_DIAG

int fvit_mlp_fused_supported(int32_t C, int32_t hidden) { return mlp_fused_supported(C, hidden) ? 1 : 0; }

int fvit_mlp_fused(int32_t operand_dtype, float* x, int32_t M, int32_t C, int32_t hidden, const float* ln_w, const float* ln_b,
                   float eps, const void* w_fc1_frag, const float* b_fc1, const void* w_fc2_frag, const float* b_fc2,
                   const float* gamma, fvit_stream_t stream) {
    MlpFusedCall mc = {operand_dtype, x, M, C, hidden, ln_w, ln_b, eps, w_fc1_frag, b_fc1, w_fc2_frag, b_fc2, gamma};
    return launch_mlp_fused(mc, (hipStream_t)stream);
}

int fvit_tune(const char* key, int32_t value) {
    if (!key) return FVIT_EINVAL;
    std::lock_guard<std::mutex> lock(g_tune_mu);
    tune_map()[key] = value;
    return FVIT_OK;
}

int fvit_prof_enable(int on) {
    DiagLock lock(diag_mutex());
    g_prof_on = on != 0;
    if (g_prof_on) g_recs.clear();
    return FVIT_OK;
}

int fvit_prof_collect(FvitProfEntry* out) {
    if (!out) return FVIT_EINVAL;
    DiagLock lock(diag_mutex());
    memset(out, 0, sizeof(FvitProfEntry) * FVIT_PROF_KINDS);
    for (size_t i = 0; i < g_recs.size(); ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(g_events[2 * i + 1]) != hipSuccess ||
            hipEventElapsedTime(&ms, g_events[2 * i], g_events[2 * i + 1]) != hipSuccess) {
            set_error("profiler: event read failed (was the region captured into a graph?)");
            (void)hipGetLastError();
            return FVIT_ELAUNCH;
        }
        const ProfRec& r = g_recs[i];
        FvitProfEntry& e = out[r.kind < 0 || r.kind >= FVIT_PROF_KINDS ? FVIT_K_OTHER : r.kind];
        e.launches += 1;
        e.ms += ms;
        e.flops += r.flops;
        e.bytes += r.bytes;
    }
    return FVIT_OK;
}

int fvit_prof_records(FvitProfRecord* out, int32_t max_records) {
    if (!out || max_records < 0) return FVIT_EINVAL;
    DiagLock lock(diag_mutex());
    int n = 0;
    for (size_t i = 0; i < g_recs.size() && n < max_records; ++i, ++n) {
        float ms = 0.f;
        if (hipEventSynchronize(g_events[2 * i + 1]) != hipSuccess ||
            hipEventElapsedTime(&ms, g_events[2 * i], g_events[2 * i + 1]) != hipSuccess) {
            set_error("profiler: event read failed (was the region captured into a graph?)");
            (void)hipGetLastError();
            return FVIT_ELAUNCH;
        }
        const ProfRec& r = g_recs[i];
        FvitProfRecord& o = out[n];
        o.kind = r.kind;
        o.grid = r.grid;
        o.ms = ms;
        o.flops = r.flops;
        o.bytes = r.bytes;
        memcpy(o.name, r.name, sizeof(o.name));
    }
    return n;
}

const char* fvit_prof_kind_name(int kind) {
    static const char* names[FVIT_PROF_KINDS] = {"window_partition", "gather_layernorm", "gemm_bias", "gemm_gelu",
                                                 "gemm_residual",    "window_attention", "window_reverse", "other",
                                                 "mlp_fused",        "conv3x3",          "attn_block_fused"};
    return (kind >= 0 && kind < FVIT_PROF_KINDS) ? names[kind] : "?";
}

}  // extern "C"
