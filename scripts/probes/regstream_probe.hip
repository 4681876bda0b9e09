// Register-ring streaming probe: what does ONE workgroup per CU get from L2 when its waves stream disjoint slices of a weight image
// straight into registers (global_load_dwordx4, 1 KiB per wave instruction), as the carrier-branch kernel (fvit_ctblk.hip) does?
//   rate per workgroup vs waves per workgroup (4 / 8 / 16) x ring depth (steps of 8 KiB in flight per wave) x grid (22 / 86 / 256),
//   warm (same launch repeated: slices L2-resident) and cold (a 512-MiB buffer streamed in between).
// Every workgroup streams the SAME `region` bytes (like the kernels: all images use the same weights), wave w its own 1/NW of it.
// build: hipcc -O3 --offload-arch=gfx950 scripts/probes/regstream_probe.hip -o /tmp/regstream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int NW, int DEPTH>
__global__ __launch_bounds__(64 * NW, 1) void stream_kernel(const char* __restrict__ src, int steps_per_wave, unsigned* sink) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)wave * steps_per_wave * 8192 + lane * 16;
    u4 ring[DEPTH][8];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < 8; ++i) ring[d][i] = *(const u4*)(base + (size_t)d * 8192 + i * 1024);
    u4 acc = {0, 0, 0, 0};
    for (int t0 = 0; t0 < steps_per_wave; t0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int t = t0 + d;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc ^= ring[d][i];
            __builtin_amdgcn_sched_barrier(0);
            if (t + DEPTH < steps_per_wave) {
#pragma unroll
                for (int i = 0; i < 8; ++i) ring[d][i] = *(const u4*)(base + (size_t)(t + DEPTH) * 8192 + i * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = acc.x;
}

__global__ void flush_kernel(const u4* p, size_t n, unsigned* sink) {
    u4 a = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a ^= p[i];
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345u) sink[0] = a.x;
}

template <int NW, int DEPTH>
void run(const char* src, size_t region, int grid, const u4* fl, size_t fln, unsigned* sink) {
    const int spw = (int)(region / 8192 / NW);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int cold = 0; cold < 2; ++cold) {
        std::vector<float> ts;
        for (int r = 0; r < 9; ++r) {
            if (cold) hipLaunchKernelGGL(flush_kernel, dim3(2048), dim3(256), 0, 0, fl, fln, sink);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL((stream_kernel<NW, DEPTH>), dim3(grid), dim3(64 * NW), 0, 0, src, spw, sink);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r >= 2) ts.push_back(ms * 1000.f);
        }
        std::sort(ts.begin(), ts.end());
        const float us = ts[ts.size() / 2];
        printf("region %4zu KiB grid %3d waves %2d depth %d %s: %6.1f us  %6.1f GB/s per workgroup  (%5.1f KiB in flight per workgroup)\n", region >> 10, grid, NW, DEPTH,
               cold ? "cold" : "warm", us, region / us * 1e-3, NW * DEPTH * 8.0);
    }
}

int main() {
    const size_t region = 1536 << 10, fln = (512u << 20) / 16;
    char* src; u4* fl; unsigned* sink;
    hipMalloc(&src, region); hipMalloc(&fl, fln * 16); hipMalloc(&sink, 64);
    hipMemset(src, 1, region); hipMemset(fl, 2, fln * 16);
    for (int grid : {22, 86, 256}) {
        run<4, 2>(src, region, grid, fl, fln, sink);
        run<4, 3>(src, region, grid, fl, fln, sink);
        run<4, 6>(src, region, grid, fl, fln, sink);
        run<8, 2>(src, region, grid, fl, fln, sink);
        run<8, 3>(src, region, grid, fl, fln, sink);
        run<16, 2>(src, region, grid, fl, fln, sink);
        run<16, 3>(src, region, grid, fl, fln, sink);
    }
    return 0;
}
