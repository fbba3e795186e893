// lds_atomic_bench.hip -- rate of random LDS atomics on MI355X (one 1024-thread workgroup per CU, 32 K-word window), the inner
// operation of the blocked advance's accumulate kernel (vgl_blocked.h).
// build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_bench lds_atomic_bench.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int WORDS = 32768;
enum { OP_F32_ADD, OP_U32_ADD, OP_I32_MIN, OP_U64_ADD, OP_U32_ADD_RTN, OP_READ, OP_F32_ADD_SORTED, N_OPS };
static const char *names[] = {"ds_add_f32", "ds_add_u32", "ds_min_i32", "ds_add_u64", "ds_add_rtn_u32", "ds_read_b32", "ds_add_f32 (lanes on distinct banks)"};

template <int OP>
__global__ __launch_bounds__(1024) void k(int iters, uint32_t *out)
{
    __shared__ __attribute__((aligned(8))) uint32_t s[WORDS];
    for (int i = threadIdx.x; i < WORDS; i += 1024) s[i] = 0;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    uint32_t sink = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x = x * 1664525u + 1013904223u;
            uint32_t idx = (x >> 12) & (WORDS - 1);
            if (OP == OP_F32_ADD_SORTED) idx = (idx & ~63u) | (threadIdx.x & 63);
            if (OP == OP_F32_ADD || OP == OP_F32_ADD_SORTED) __hip_atomic_fetch_add(reinterpret_cast<float *>(&s[idx]), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == OP_U32_ADD) __hip_atomic_fetch_add(&s[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == OP_I32_MIN) __hip_atomic_fetch_min(reinterpret_cast<int *>(&s[idx]), (int)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == OP_U64_ADD) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(__builtin_assume_aligned(&s[idx & ~1u], 8)), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == OP_U32_ADD_RTN) sink += __hip_atomic_fetch_add(&s[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else sink += s[idx];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[0] + sink;
}

template <int OP> void run(uint32_t *d_out, int blocks)
{
    const int iters = 2048;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, 16, d_out);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, iters, d_out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double ops = (double)blocks * 1024 * iters * 8;
    printf("%-40s %8.3f ms  %8.1f G lane-ops/s  %6.2f clk per wave-instruction per CU (2.4 GHz)\n", names[OP], ms, ops / ms / 1e6,
           ms * 1e-3 * 2.4e9 / (ops / blocks / 64));
}

int main()
{
    uint32_t *d_out;
    hipMalloc(&d_out, 4096);
    const int blocks = 256;
    run<OP_F32_ADD>(d_out, blocks);
    run<OP_U32_ADD>(d_out, blocks);
    run<OP_I32_MIN>(d_out, blocks);
    run<OP_U64_ADD>(d_out, blocks);
    run<OP_U32_ADD_RTN>(d_out, blocks);
    run<OP_READ>(d_out, blocks);
    run<OP_F32_ADD_SORTED>(d_out, blocks);
    return 0;
}
