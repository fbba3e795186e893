// adv_order_kernel.hip -- microbenchmark kernel of profiles/microbench/adv_order_bench.py: the reference's BFS edge operator (bfs.hpp:28-36)
// over an explicit edge list (src, dst per edge), once in CSR order and once in (destination block, source block) order.
#include <hip/hip_runtime.h>
#include <cstdint>

template <int EPT>
__global__ __launch_bounds__(256) void k_edge_bfs(long long E, const int *src, const int *dst, const int *flags, int *levels, int cur)
{
    const long long base = (long long)blockIdx.x * (256 * EPT) + threadIdx.x;
    int s[EPT], d[EPT], f[EPT];
#pragma unroll
    for (int j = 0; j < EPT; j++) { const long long e = base + (long long)j * 256; s[j] = src[e < E ? e : E - 1]; d[j] = dst[e < E ? e : E - 1]; }
#pragma unroll
    for (int j = 0; j < EPT; j++) f[j] = flags[s[j]];
#pragma unroll
    for (int j = 0; j < EPT; j++) {
        const long long e = base + (long long)j * 256;
        if (e < E && f[j] > 0) {
            const int sl = levels[s[j]], dl = levels[d[j]];            // the operator as written by the user
            if (sl == cur && dl == -1) levels[d[j]] = cur + 1;
        }
    }
}

extern "C" int run_edge_bfs(long long E, const int *src, const int *dst, const int *flags, int *levels, int cur, int ept, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (ept == 4) hipLaunchKernelGGL(k_edge_bfs<4>, dim3((unsigned)((E + 1023) / 1024)), dim3(256), 0, st, E, src, dst, flags, levels, cur);
    else hipLaunchKernelGGL(k_edge_bfs<8>, dim3((unsigned)((E + 2047) / 2048)), dim3(256), 0, st, E, src, dst, flags, levels, cur);
    return (int)hipGetLastError();
}
