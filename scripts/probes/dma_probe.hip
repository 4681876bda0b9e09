// LDS-DMA (global_load_lds, 16 B/lane) fill-rate probe: how many GB/s per CU can workgroups pull from an L2-resident or an
// HBM-sized buffer into LDS, as a function of bytes per step, ring depth (prefetch distance) and workgroups per CU?
// build: hipcc -O3 --offload-arch=gfx950 scripts/probes/dma_probe.hip -o /tmp/dma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// PIECES: 1-KiB pieces per wave and step (4 waves): bytes per step = 4 * PIECES KiB.  NSTAGE ring slots.
template <int PIECES, int NSTAGE>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ src, size_t src_bytes, int steps, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int STEP_BYTES = 4 * PIECES * 1024;
    const size_t nsteps_src = src_bytes / STEP_BYTES;
    size_t pos = (size_t)blockIdx.x * 7919 % nsteps_src;   // different workgroups start at different places
    auto stage = [&](int slot) {
        const char* s = src + pos * STEP_BYTES + (size_t)wave * PIECES * 1024 + lane * 16;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) glds16(s + i * 1024, smem + slot * STEP_BYTES + (wave * PIECES + i) * 1024);
        pos = pos + 1 == nsteps_src ? 0 : pos + 1;
    };
#pragma unroll
    for (int st = 0; st < NSTAGE - 1; ++st) stage(st);
    float acc = 0.f;
    int cur = 0;
    for (int it = 0; it < steps; ++it) {
        if (NSTAGE == 2) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        else if (NSTAGE == 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * PIECES) : "memory");
        int slot = cur + NSTAGE - 1;
        if (slot >= NSTAGE) slot -= NSTAGE;
        stage(slot);
        acc += *(const float*)(smem + cur * STEP_BYTES + threadIdx.x * 16);   // touch the tile (one ds_read per thread)
        cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 123.456f) sink[0] = acc;
}

template <int PIECES, int NSTAGE>
void run(const char* src, size_t src_bytes, int wgs_per_cu, float* sink, const char* label) {
    const int steps = 400;
    const size_t lds = (size_t)NSTAGE * 4 * PIECES * 1024;
    hipFuncSetAttribute((const void*)probe<PIECES, NSTAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)probe<PIECES, NSTAGE>, 256, lds);
    if (occ < wgs_per_cu) { printf("%-8s %3d KiB/step x%d stages, %d WG/CU: does not fit (occupancy %d)\n", label, 4 * PIECES, NSTAGE, wgs_per_cu, occ); return; }
    const int grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<PIECES, NSTAGE>), dim3(grid), dim3(256), lds, 0, src, src_bytes, 50, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<PIECES, NSTAGE>), dim3(grid), dim3(256), lds, 0, src, src_bytes, steps, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * steps * 4 * PIECES * 1024;
    printf("%-8s %3d KiB/step x%d stages, %d WG/CU: %7.1f us  %6.2f TB/s chip  %6.1f GB/s per CU  %5.2f us/step\n", label, 4 * PIECES, NSTAGE,
           wgs_per_cu, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256, ms * 1e3 / steps);
}

int main() {
    const size_t small = 2u << 20, big = 512u << 20;
    char *a, *b;
    float* sink;
    hipMalloc(&a, small); hipMalloc(&b, big); hipMalloc(&sink, 4);
    hipMemset(a, 1, small); hipMemset(b, 1, big);
    for (int pass = 0; pass < 2; ++pass) {
        const char* src = pass ? b : a;
        const size_t n = pass ? big : small;
        const char* label = pass ? "HBM512M" : "L2-2M";
        for (int w = 1; w <= 4; w *= 2) {
            run<6, 2>(src, n, w, sink, label);    // 24 KiB: the 64x128 GEMM tile
            run<6, 3>(src, n, w, sink, label);
            run<6, 4>(src, n, w, sink, label);
            run<8, 2>(src, n, w, sink, label);    // 32 KiB: 128x128 GEMM tile / MLP chunk
            run<8, 3>(src, n, w, sink, label);
            run<8, 4>(src, n, w, sink, label);
            run<16, 2>(src, n, w, sink, label);   // 64 KiB
        }
    }
    return 0;
}
