// exchange.hip -- device side of the "recently changed" vertex-array exchange of the edge-cut multi-GPU path
// (EXCHANGE_RECENTLY_CHANGED, vgl_compute_api/common/mpi_exchange.hpp:110-150: every process sends the (index, value) pairs of the
// entries it changed in the super-step and merges what it receives with the algorithm's operator).  Here: compaction of the
// entries that differ from their pre-step value into a pair list, and the merge of P such lists with min / max on the 4-byte
// patterns (distances, widths and labels are non-negative, so the integer order is the value order).  The collective between the
// two (RCCL all-gather of the lists) is issued by the host side, vectorgraphlibrary_amd/distributed.py.
#include "vgl_hip_internal.h"

// list layout: [0] = number of changed entries (may exceed cap), then pairs (index, value bits) for the first min(count, cap) of them,
// in no particular order (one cursor reservation per workgroup)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_diff_to_pairs(int32_t n, const uint32_t *before, const uint32_t *after, int32_t cap, int32_t *out)
{
    __shared__ int s_w[VGL_WAVES];
    __shared__ int s_base;
    const int32_t i = blockIdx.x * VGL_BLOCK + threadIdx.x;
    const bool changed = i < n && before[i] != after[i];
    const unsigned long long ballot = __ballot(changed);
    if (vgl_lane() == 0) s_w[vgl_wave()] = __popcll(ballot);
    __syncthreads();
    int mine = 0, total = 0;
#pragma unroll
    for (int w = 0; w < VGL_WAVES; w++) { if (w < vgl_wave()) mine += s_w[w]; total += s_w[w]; }
    if (threadIdx.x == 0) s_base = total ? atomicAdd(out, total) : 0;
    __syncthreads();
    if (changed) {
        const int pos = s_base + mine + __popcll(ballot & ((1ULL << vgl_lane()) - 1ULL));
        if (pos < cap) { out[1 + 2 * pos] = i; out[2 + 2 * pos] = (int32_t)after[i]; }
    }
}

template <bool TAKE_MIN>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_apply_pairs(int parts, int64_t stride, int skip_part, const int32_t *lists, int32_t n, int32_t *values, int64_t *counters)
{
    for (int p = blockIdx.y; p < parts; p += gridDim.y) {
        if (p == skip_part) continue;                                   // this rank's own changes are already in `values`
        const int32_t *list = lists + (size_t)p * (size_t)stride;
        const int32_t count = list[0];
        int any = 0;
        for (int32_t k = blockIdx.x * VGL_BLOCK + threadIdx.x; k < count; k += gridDim.x * VGL_BLOCK) {
            const int32_t idx = list[1 + 2 * k], v = list[2 + 2 * k];
            if (idx < 0 || idx >= n) continue;                          // (a corrupt list must not write outside the array)
            const int32_t before = TAKE_MIN ? atomicMin(values + idx, v) : atomicMax(values + idx, v);
            any |= TAKE_MIN ? (before > v) : (before < v);
        }
        if (__syncthreads_or(any) && threadIdx.x == 0) counters[C_CHANGED] = 1;
    }
}

extern "C" {

int vgl_hip_diff_to_pairs_u32(vgl_hip_ctx *c, int32_t n, const void *d_before, const void *d_after, int32_t cap, int32_t *d_out)
{
    if (!c || !d_before || !d_after || !d_out) VGL_FAIL("diff_to_pairs: null argument");
    if (n < 0 || cap < 0) VGL_FAIL("diff_to_pairs: bad size");
    VGL_HIP_TRY(hipMemsetAsync(d_out, 0, sizeof(int32_t), c->stream));
    if (n > 0)
        hipLaunchKernelGGL(vgl_k_diff_to_pairs, dim3((unsigned)vgl_ceil_div(n, VGL_BLOCK)), dim3(VGL_BLOCK), 0, c->stream, n, (const uint32_t *)d_before,
                           (const uint32_t *)d_after, cap, d_out);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_apply_pairs_u32(vgl_hip_ctx *c, int parts, int64_t stride, int skip_part, const int32_t *d_lists, int take_min, int32_t n, void *d_values, int *changed)
{
    if (!c || !d_lists || !d_values) VGL_FAIL("apply_pairs: null argument");
    if (parts < 1 || stride < 1 || n < 0) VGL_FAIL("apply_pairs: bad size");
    VGL_TRY(vgl_zero_counters(c, C_CHANGED, 1));
    const dim3 grid(256, (unsigned)std::min(parts, 64));
    if (take_min)
        hipLaunchKernelGGL(vgl_k_apply_pairs<true>, grid, dim3(VGL_BLOCK), 0, c->stream, parts, stride, skip_part, d_lists, n, (int32_t *)d_values, c->d_counters);
    else
        hipLaunchKernelGGL(vgl_k_apply_pairs<false>, grid, dim3(VGL_BLOCK), 0, c->stream, parts, stride, skip_part, d_lists, n, (int32_t *)d_values, c->d_counters);
    VGL_HIP_TRY(hipGetLastError());
    if (changed) {
        VGL_TRY(vgl_read_counters(c, false));
        *changed = (int)c->h_counters[C_CHANGED];
    }
    return 0;
}

}  // extern "C"
