// exchange.hip -- device side of the "recently changed" vertex-array exchange of the edge-cut multi-GPU path
// (EXCHANGE_RECENTLY_CHANGED, vgl_compute_api/common/mpi_exchange.hpp:110-150: every process sends the (index, value) pairs of the
// entries it changed in the super-step and merges what it receives with the algorithm's operator).  Here: compaction of the
// entries that differ from their pre-step value into a pair list, and the merge of P such lists with min / max on the 4-byte
// patterns (distances, widths and labels are non-negative, so the integer order is the value order).  The collective between the
// two is vgl_hip_exchange_changed_u32 below (RCCL all-gather of the lists on the context's stream).
#include "vgl_comm.h"

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
        // the producer's count may exceed what its list holds (vgl_k_diff_to_pairs keeps counting past cap): never read past the slot
        const int32_t count = (int32_t)min((int64_t)list[0], (stride - 1) / 2);
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

// P list heads -> P int64 counts (what the host needs to pick the exchange form; every rank reads the same numbers)
__global__ void vgl_k_list_counts(const int32_t *lists, int64_t stride, int parts, int64_t *out)
{
    if ((int)threadIdx.x < parts) out[threadIdx.x] = lists[(int64_t)threadIdx.x * stride];
}

constexpr int32_t VGL_CHANGED_SMALL = 2048;        // pairs per rank that ride in the first all-gather (16 KiB)

int vgl_hip_exchange_changed_u32(vgl_hip_comm *m, int32_t n, const void *d_before, void *d_values, int take_min, int *changed_anywhere)
{
    if (!m || !d_before || !d_values) VGL_FAIL("exchange_changed: null argument");
    if (n < 0) VGL_FAIL("exchange_changed: negative size");
    vgl_hip_ctx *c = m->ctx;
    const int P = m->world;
    if (P > 64) VGL_FAIL("exchange_changed: at most 64 ranks");
    const int32_t cap_big = std::max<int32_t>(VGL_CHANGED_SMALL, n / (2 * P));
    const int64_t small_stride = 1 + 2 * (int64_t)VGL_CHANGED_SMALL;
    int32_t *mine = nullptr, *all_small = nullptr;
    VGL_TRY(vgl_comm_scratch(m, 0, sizeof(int32_t) * (size_t)(1 + 2 * (int64_t)cap_big), (void **)&mine));
    VGL_TRY(vgl_comm_scratch(m, 1, sizeof(int32_t) * (size_t)(small_stride * P), (void **)&all_small));
    VGL_TRY(vgl_hip_diff_to_pairs_u32(c, n, d_before, d_values, cap_big, mine));
    // stage 1: the heads of the lists -- count + the first 2048 pairs -- from everybody
    VGL_TRY(vgl_comm_allgather(m, mine, all_small, small_stride * 4));
    hipLaunchKernelGGL(vgl_k_list_counts, dim3(1), dim3(64), 0, c->stream, all_small, small_stride, P, m->d_small);
    int64_t counts[64];
    VGL_TRY(vgl_comm_read_small(m, m->d_small, P, counts));
    int64_t most = 0;
    for (int p = 0; p < P; p++) most = std::max(most, counts[p]);
    if (changed_anywhere) *changed_anywhere = most > 0;
    if (most == 0 || !vgl_comm_active(m)) return 0;
    if (most <= VGL_CHANGED_SMALL) {
        m->stats.list_steps++;
        return vgl_hip_apply_pairs_u32(c, P, small_stride, m->rank, all_small, take_min, n, d_values, nullptr);
    }
    if (most <= cap_big) {                          // stage 2: whole lists, padded to the next power of two of the longest
        int64_t len = 1;
        while (len < most) len <<= 1;
        len = std::min<int64_t>(len, cap_big);
        const int64_t stride = 1 + 2 * len;
        int32_t *all = nullptr;
        VGL_TRY(vgl_comm_scratch(m, 2, sizeof(int32_t) * (size_t)(stride * P), (void **)&all));
        VGL_TRY(vgl_comm_allgather(m, mine, all, stride * 4));
        m->stats.list_steps++;
        return vgl_hip_apply_pairs_u32(c, P, stride, m->rank, all, take_min, n, d_values, nullptr);
    }
    // some rank changed more than n / (2 P) entries: the whole array moves fewer bytes than the lists would (EXCHANGE_ALL with the
    // operator, shortest_paths.hpp:136-141); values are non-negative, so their 4-byte patterns order like the values
    m->stats.dense_steps++;
    return vgl_comm_allreduce(m, d_values, n, VGL_DT_I32, take_min ? VGL_OP_MIN : VGL_OP_MAX);
}

}  // extern "C"
