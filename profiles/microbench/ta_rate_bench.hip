// ta_rate_bench.hip -- what a divergent global gather costs on MI355X as a function of the number of ACTIVE lanes: the bottom-up BFS probe
// (bfs.hip, vgl_k_bu_probe) executes its later probe rounds with a handful of lanes left.  Every wavefront issues `iters` x 8 independent
// 8-byte loads from random words of a table (2 MiB = the frontier bitmap of 16.8 M vertices: misses L1, hits L2; 16 KiB: hits L1); lanes
// >= ACTIVE are masked off.  Prints clocks per wavefront instruction per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o ta_rate_bench ta_rate_bench.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int WIDE>
__global__ __launch_bounds__(256) void k(const uint64_t *tab, uint32_t mask, int active, int iters, uint64_t *out)
{
    const int lane = threadIdx.x & 63;
    uint32_t x = (threadIdx.x + blockIdx.x * 256u) * 2654435761u + 12345u;
    uint64_t sink = 0;
    if (lane < active) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x = x * 1664525u + 1013904223u;
                const uint32_t idx = (x >> 9) & mask;
                if (WIDE == 2) sink += tab[idx];
                else if (WIDE == 1) sink += reinterpret_cast<const uint32_t *>(tab)[idx];
                else { const uint4 v = reinterpret_cast<const uint4 *>(tab)[idx >> 1]; sink += v.x + v.w; }
            }
        }
    }
    if (sink == 0x1234567ULL) out[0] = sink;
}

template <int WIDE> void run(const uint64_t *tab, uint64_t *out, uint32_t words, int active, const char *what)
{
    const int blocks = 2048, iters = 512;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<WIDE>, dim3(blocks), dim3(256), 0, 0, tab, words - 1, active, 8, out);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<WIDE>, dim3(blocks), dim3(256), 0, 0, tab, words - 1, active, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double instr_per_cu = (double)blocks * 4 * iters * 8 / 256;
    printf("%-28s %2d active lanes  %8.3f ms  %7.2f clk per wavefront instruction per CU (2.4 GHz)  %7.1f G lane-loads/s\n", what, active, ms,
           ms * 1e-3 * 2.4e9 / instr_per_cu, (double)blocks * 4 * active * iters * 8 / ms / 1e6);
}

int main()
{
    uint64_t *tab, *out;
    const uint32_t big = 1u << 18;                           // 2 MiB of 8-byte words
    hipMalloc(&tab, (size_t)big * 8);
    hipMemset(tab, 0, (size_t)big * 8);
    hipMalloc(&out, 64);
    for (int active : {64, 32, 16, 8, 4, 1}) run<2>(tab, out, big, active, "8 B loads, 2 MiB table");
    for (int active : {64, 16, 1}) run<2>(tab, out, 2048, active, "8 B loads, 16 KiB table");
    for (int active : {64, 16, 1}) run<1>(tab, out, big, active, "4 B loads, 2 MiB (words)");
    for (int active : {64, 16, 1}) run<4>(tab, out, big, active, "16 B loads, 2 MiB table");
    hipDeviceSynchronize();
    return 0;
}
