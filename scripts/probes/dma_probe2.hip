// LDS-DMA probe 2: what a SMALL grid of weight-streaming workgroups gets out of the memory system (the regime of the carrier-token
// branch and of stage 3: 20-260 workgroups that all stream the same 0.5-4 MB of weights).
//
// Every workgroup (4 wave64) streams `region` bytes in 32-KiB steps through an LDS ring (global_load_lds, 16 B per lane) and touches
// each tile once.  Modes:
//   private   : workgroup b streams its own region (no sharing: every byte is a first touch)
//   lockstep  : all workgroups stream the SAME region in the same order (today's fused kernels / GEMM weight panels)
//   stagger   : same region, workgroup b starts at step (b / 8) * nsteps / (grid / 8) (b % 8 = XCD, observed): a chunk is first
//               touched by one workgroup per XCD and found in L2 by the others
//   warm      : lockstep, launched right after an identical launch with no flush in between (does L2 survive a kernel boundary?)
// Between measured launches a 768-MiB buffer is streamed to push the region out of L2 and the Infinity Cache (except `warm`).
// build: hipcc -O3 --offload-arch=gfx950 scripts/probes/dma_probe2.hip -o /tmp/dma_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

constexpr int STEP = 32 * 1024;

template <int NSTAGE>
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ src, int nsteps, size_t wg_stride, int stagger, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)blockIdx.x * wg_stride;
    int pos = 0;
    if (stagger) {
        const int per_xcd = (gridDim.x + 7) / 8;
        pos = (int)(((long)(blockIdx.x >> 3) * nsteps) / per_xcd) % nsteps;
    }
    auto stage = [&](int slot) {
        const char* s = base + (size_t)pos * STEP + wave * 8192 + lane * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) glds16(s + i * 1024, smem + slot * STEP + (wave * 8 + i) * 1024);
        pos = pos + 1 == nsteps ? 0 : pos + 1;
    };
#pragma unroll
    for (int st = 0; st < NSTAGE - 1; ++st) stage(st);
    float acc = 0.f;
    int cur = 0;
    for (int it = 0; it < nsteps; ++it) {
        if (NSTAGE == 2) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");   // NSTAGE 4: two steps stay in flight
        int slot = cur + NSTAGE - 1;
        if (slot >= NSTAGE) slot -= NSTAGE;
        if (it + NSTAGE - 1 < nsteps) stage(slot);
        else if (NSTAGE > 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += *(const float*)(smem + cur * STEP + threadIdx.x * 16);
        cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 123.456f) sink[0] = acc;
}

__global__ void flush_kernel(const float4* __restrict__ p, size_t n, float* sink) {
    float a = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i].x;
    if (a == 123.456f) sink[0] = a;
}

template <int NSTAGE>
float run_once(const char* src, int grid, int nsteps, size_t stride, int stagger, float* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<NSTAGE>), dim3(grid), dim3(256), NSTAGE * STEP, 0, src, nsteps, stride, stagger, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e3f;
}

int main() {
    const size_t flush_bytes = 768u << 20, pool_bytes = 512u << 20;
    char *pool, *fl;
    float* sink;
    hipMalloc(&pool, pool_bytes); hipMalloc(&fl, flush_bytes); hipMalloc(&sink, 4);
    hipMemset(pool, 1, pool_bytes); hipMemset(fl, 1, flush_bytes);
    hipFuncSetAttribute((const void*)stream_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)stream_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    auto flush = [&]() { hipLaunchKernelGGL(flush_kernel, dim3(2048), dim3(256), 0, 0, (const float4*)fl, flush_bytes / 16, sink); hipDeviceSynchronize(); };
    const int regions_kb[] = {512, 1536, 4096};
    const int grids[] = {16, 32, 64, 128, 256};
    printf("%-9s %5s %6s %2s | %8s %9s %9s\n", "mode", "grid", "regKB", "ns", "us", "GB/s/WG", "us/step");
    for (int region_kb : regions_kb)
        for (int grid : grids) {
            const int nsteps = region_kb * 1024 / STEP;
            const size_t region = (size_t)region_kb * 1024;
            for (int ns = 2; ns <= 4; ns += 2) {
                for (int mode = 0; mode < 4; ++mode) {
                    const char* name = mode == 0 ? "private" : mode == 1 ? "lockstep" : mode == 2 ? "stagger" : "warm";
                    if (mode == 0 && (size_t)grid * region > pool_bytes) continue;
                    float best = 1e30f, sum = 0.f;
                    const int reps = 3;
                    for (int r = 0; r < reps; ++r) {
                        if (mode != 3) flush();
                        else { if (ns == 2) run_once<2>(pool, grid, nsteps, 0, 0, sink); else run_once<4>(pool, grid, nsteps, 0, 0, sink); }
                        const size_t stride = mode == 0 ? region : 0;
                        const int stg = mode == 2;
                        const float us = ns == 2 ? run_once<2>(pool, grid, nsteps, stride, stg, sink) : run_once<4>(pool, grid, nsteps, stride, stg, sink);
                        best = us < best ? us : best;
                        sum += us;
                    }
                    printf("%-9s %5d %6d %2d | %8.1f %9.1f %9.3f   (mean %.1f us)\n", name, grid, region_kb, ns, best, region / best / 1e3, best / nsteps, sum / reps);
                }
            }
        }
    return 0;
}
