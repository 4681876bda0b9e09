// fvit_stage3.hip -- a whole non-hierarchical C = 512 stage (stage 3 of FasterViT-0: depth x [window attention sub-block, MLP sub-block], FV:690-691
// with ct = None) as ONE launch: one persistent workgroup per window runs winblk_body and winmlp_body (the bodies of fvit_winblk.hip / fvit_winmlp.hip,
// unchanged) alternately for every block of the stage.  Without carrier tokens a window's tokens never meet another window's inside the stage, so no
// workgroup ever waits for another one: the hand-off between sub-blocks is the workgroup's own rows through global memory behind a __syncthreads().
//
// Why ("where this architecture stops"): per block the two stand-alone kernels spend 5-14 us before their first MFMA (launch, index
// tables, rows from the memory side because XCD L2s carry nothing across kernels, LayerNorm) and the stage is 2 x depth launches per stream shard.  Here
// the rows a phase reads were written by the same CU one barrier earlier, and the stage is one graph node.  Cost: the MLP phase runs per window (49 real
// rows in a 64-row tile, 86 workgroups) instead of over packed 64-row groups (66 workgroups).
// fvit_tune "win_stage3" (default: see fvit_api.hip).  Same arithmetic per row as the two kernels: results are bitwise those of the unfused launches.
#define FVIT_BODIES_ONLY
#include "fvit_winblk.hip"
#include "fvit_winmlp.hip"
#undef FVIT_BODIES_ONLY

namespace fvit {

namespace {

constexpr int S3_MAX_DEPTH = 8;

struct Stage3Params {
    int depth, S;
    WinBlkParams attn[S3_MAX_DEPTH];
    WinMlpParams mlp[S3_MAX_DEPTH];
};

template <typename T>
__global__ __launch_bounds__(512, 1) void win_stage3_kernel(Stage3Params p) {
    constexpr int LDS_A = winblk_lds_bytes<512, 1>(), LDS_M = winmlp_lds_bytes<512, 2048, 4, 8>();
    __shared__ __attribute__((aligned(16))) char smem[LDS_A > LDS_M ? LDS_A : LDS_M];
    const int win = blockIdx.x;
#pragma unroll 1
    for (int b = 0; b < p.depth; ++b) {
        winblk_body<T, 512, 8, 1, 1>(p.attn[b], smem, win);
        // hand-off inside the workgroup: stores drained + barrier, then THIS CU's vector L1 is invalidated -- the phase that follows re-reads rows this CU
        // loaded before it rewrote them, and a write-through store does not refresh the L1 line (first version: wrong rows at 86 windows, right at 4 / 256)
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        WinMlpParams pm = p.mlp[b];
        pm.x += (size_t)win * p.S * 512;   // this window's rows as a one-tile problem: rows >= S of the 64-row tile are clamped / masked by the body
        pm.M = p.S;
        winmlp_body<T, 512, 2048, 4, 2, 8, 1, 1, false>(pm, smem, 0);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

}  // namespace

bool win_stage3_supported(int C, int heads, int hidden, int S, int depth) {
    return C == 512 && heads == 16 && hidden == 2048 && S > 48 && S <= 64 && depth >= 1 && depth <= S3_MAX_DEPTH;
}

int launch_win_stage3(const AttnBlkCall* attn, const MlpFusedCall* mlp, int depth, hipStream_t stream) {
    if (!attn || !mlp || depth < 1 || depth > S3_MAX_DEPTH) { set_error("win_stage3: bad depth %d", depth); return FVIT_EINVAL; }
    const AttnBlkCall& a0 = attn[0];
    if (!win_stage3_supported(a0.C, a0.heads, mlp[0].hidden, a0.S, depth) || a0.nwin <= 0) {
        set_error("win_stage3: unsupported geometry C=%d heads=%d hidden=%d S=%d", a0.C, a0.heads, mlp[0].hidden, a0.S);
        return FVIT_EINVAL;
    }
    Stage3Params p;
    p.depth = depth; p.S = a0.S;
    double flops = 0, bytes = 0;
    for (int b = 0; b < depth; ++b) {
        const AttnBlkCall& c = attn[b];
        const MlpFusedCall& m = mlp[b];
        if (c.C != a0.C || c.S != a0.S || c.nwin != a0.nwin || c.dtype != a0.dtype || m.dtype != a0.dtype || m.C != a0.C || m.terms != 1 || c.terms != 1 ||
            !c.wqkv_f || !c.wproj_f || !c.bias || !c.bqkv || !c.x_out || !m.x || !m.w1f || !m.w2f || m.M != a0.nwin * a0.S || (float*)m.x != c.x_out) {
            set_error("win_stage3: block %d does not fit the one-launch form", b);
            return FVIT_EINVAL;
        }
        p.attn[b] = make_winblk_params(c);
        p.mlp[b] = make_winmlp_params(m);
        const double rows = (double)c.nwin * c.S;
        flops += rows * (2.0 * c.C * 3 * c.C + 4.0 * c.S * c.C + 2.0 * c.C * c.C) + 4.0 * rows * c.C * m.hidden;
        bytes += 2.0 * (4.0 * c.C * c.C + 2.0 * c.C * m.hidden);
    }
    bytes += 8.0 * a0.nwin * a0.S * a0.C;   // the stage's rows in and out once; everything between stays on chip (L2)
    ProfScope prof(FVIT_K_ATTN_FUSED, flops, bytes, stream);
    prof_note("win_stage3_kernel<512>", a0.nwin);
    if (a0.dtype == FVIT_F16) hipLaunchKernelGGL((win_stage3_kernel<_Float16>), dim3(a0.nwin), dim3(512), 0, stream, p);
    else if (a0.dtype == FVIT_BF16) hipLaunchKernelGGL((win_stage3_kernel<__bf16>), dim3(a0.nwin), dim3(512), 0, stream, p);
    else { set_error("win_stage3: operand dtype %d not supported", a0.dtype); return FVIT_EINVAL; }
    return check_launch("win_stage3_kernel");
}

}  // namespace fvit
