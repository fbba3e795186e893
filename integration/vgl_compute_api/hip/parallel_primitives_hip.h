// parallel_primitives_hip.h -- ParallelPrimitives::copy_if_indexes for conditions that are DEVICE lambdas (included by
// vgl_runtime/helpers/parallel_primitives/primitives.h under __USE_HIP__, ahead of copy_if/copy_if.hpp).
//
// Under a GPU architecture __VGL_COPY_IF_INDEXES_ARGS__ is `__device__ (int idx)` (architecture_independent_api.h:30) and the only user,
// TransitiveClosure::vgl_purdoms (algorithms/tc/tc.hpp:86-97), captures EdgesArrays whose device accessor indexes the HBM buffer: the condition
// can only be evaluated by a kernel.  The reference's own GPU flavour hands such a lambda to the host-side OpenMP loop
// (copy_if/copy_if.hpp:285-298), which is why its tc does not build there; this is the missing device side.
//
//   count  : one workgroup per tile of 2048 consecutive indexes, eight per thread (ascending inside the thread) -> tile_counts[tile]
//   scan   : ONE workgroup turns the counts into exclusive offsets (a few hundred thousand tiles at 10^9 indexes) and leaves the total
//   write  : the tiles evaluate the condition again (it is a pure function of idx), rank their hits with a workgroup scan and store
//            idx + offset in ascending order -- the order of omp_copy_if_indexes / vector_copy_if_indexes (copy_if.hpp:128-191)
// _out_data is what the caller allocated with MemoryAPI::allocate_array (managed, host-resident, visible to kernels): the write pass stores there.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#define HIP_COPY_IF_TILE 2048
#define HIP_COPY_IF_THREADS 256
#define HIP_COPY_IF_PER_THREAD (HIP_COPY_IF_TILE / HIP_COPY_IF_THREADS)

template <typename Cond>
__device__ __forceinline__ unsigned hip_copy_if_bits(const Cond &cond, size_t size, size_t first)
{
    unsigned bits = 0;
#pragma unroll
    for (int j = 0; j < HIP_COPY_IF_PER_THREAD; j++)
        if (first + j < size && cond((int)(first + j)) > 0) bits |= 1u << j;
    return bits;
}
// exclusive prefix of `mine` over the workgroup (HIP_COPY_IF_THREADS threads), *total = the sum
__device__ __forceinline__ unsigned hip_copy_if_block_scan(unsigned mine, unsigned *total)
{
    __shared__ unsigned s_wave[HIP_COPY_IF_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = mine;
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    __syncthreads();                                     // (a second call may not overwrite s_wave while the first is still read)
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    unsigned base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < HIP_COPY_IF_THREADS / 64; w++) { if (w < wave) base += s_wave[w]; sum += s_wave[w]; }
    *total = sum;
    return base + inc - mine;
}
template <typename Cond>
__global__ __launch_bounds__(HIP_COPY_IF_THREADS) void hip_copy_if_count_kernel(Cond cond, size_t size, size_t tiles, unsigned *tile_counts)
{
    for (size_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const unsigned bits = hip_copy_if_bits(cond, size, tile * HIP_COPY_IF_TILE + (size_t)threadIdx.x * HIP_COPY_IF_PER_THREAD);
        unsigned total;
        hip_copy_if_block_scan(__popc(bits), &total);
        if (threadIdx.x == 0) tile_counts[tile] = total;
    }
}
// one workgroup: tile_offsets[t] = sum of tile_counts[0..t), *total = everything (64-bit: more than 2^32 hits cannot be returned as an int anyway,
// the caller checks)
__global__ __launch_bounds__(HIP_COPY_IF_THREADS) void hip_copy_if_scan_kernel(size_t tiles, const unsigned *tile_counts, unsigned long long *tile_offsets,
                                                                               unsigned long long *total)
{
    __shared__ unsigned long long s_run;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (size_t t0 = 0; t0 < tiles; t0 += HIP_COPY_IF_THREADS) {
        const size_t t = t0 + threadIdx.x;
        const unsigned mine = t < tiles ? tile_counts[t] : 0u;
        unsigned sum;
        const unsigned before = hip_copy_if_block_scan(mine, &sum);
        const unsigned long long run = s_run;
        if (t < tiles) tile_offsets[t] = run + before;
        __syncthreads();                                 // everyone has read s_run
        if (threadIdx.x == 0) s_run = run + sum;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_run;
}
template <typename Cond>
__global__ __launch_bounds__(HIP_COPY_IF_THREADS) void hip_copy_if_write_kernel(Cond cond, size_t size, size_t tiles, const unsigned long long *tile_offsets,
                                                                                int index_offset, int *out)
{
    for (size_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const size_t first = tile * HIP_COPY_IF_TILE + (size_t)threadIdx.x * HIP_COPY_IF_PER_THREAD;
        unsigned bits = hip_copy_if_bits(cond, size, first);
        unsigned total;
        unsigned long long pos = tile_offsets[tile] + hip_copy_if_block_scan(__popc(bits), &total);
        while (bits) {
            const int j = __ffs(bits) - 1;
            bits &= bits - 1;
            out[pos++] = (int)(first + j) + index_offset;
        }
    }
}

#define HIP_COPY_IF_RT(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) throw hipGetErrorString(_e); } while (0)

template <typename CopyCondition>
inline int hip_copy_if_indexes(CopyCondition &&_cond, int *_out_data, size_t _size, const int _index_offset)
{
    using C = typename std::decay<CopyCondition>::type;
    if (_size == 0) return 0;
    if (_size > (size_t)0x7fffffff) throw "Error in ParallelPrimitives::copy_if_indexes : the condition takes an int index (more than 2^31 - 1 elements)";
    hip_shadows_to_device(0);                            // the condition reads user arrays: whatever the host wrote since the last primitive goes up
    const size_t tiles = (_size + HIP_COPY_IF_TILE - 1) / HIP_COPY_IF_TILE;
    const unsigned grid = (unsigned)(tiles < 65536 ? tiles : 65536);
    unsigned *tile_counts = nullptr;
    unsigned long long *tile_offsets = nullptr;          // [tiles] offsets + [1] total
    HIP_COPY_IF_RT(hipMalloc((void **)&tile_counts, sizeof(unsigned) * tiles));
    HIP_COPY_IF_RT(hipMalloc((void **)&tile_offsets, sizeof(unsigned long long) * (tiles + 1)));
    hipLaunchKernelGGL((hip_copy_if_count_kernel<C>), dim3(grid), dim3(HIP_COPY_IF_THREADS), 0, 0, _cond, _size, tiles, tile_counts);
    hipLaunchKernelGGL(hip_copy_if_scan_kernel, dim3(1), dim3(HIP_COPY_IF_THREADS), 0, 0, tiles, (const unsigned *)tile_counts, tile_offsets, tile_offsets + tiles);
    hipLaunchKernelGGL((hip_copy_if_write_kernel<C>), dim3(grid), dim3(HIP_COPY_IF_THREADS), 0, 0, _cond, _size, tiles, (const unsigned long long *)tile_offsets,
                       _index_offset, _out_data);
    unsigned long long total = 0;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(&total, tile_offsets + tiles, sizeof(total), hipMemcpyDeviceToHost);      // (synchronises: _out_data is complete)
    (void)hipFree(tile_counts);
    (void)hipFree(tile_offsets);
    if (e != hipSuccess) throw hipGetErrorString(e);
    return (int)total;
}
