// gen.hip -- device-side synthetic inputs (RMAT / uniform / weights), COO->CSR build, row partitioning.
// Graph construction is offline in the reference (import is not part of any timed region,
// vgl_runtime.hpp:27-60), so the sort/scan/select here use rocPRIM library primitives; the hot-path
// kernels (advance / GNF / reduce / fused algorithm steps) are hand-written in the other files.
#include "vgl_hip_internal.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

// ---- counter-based RNG: identical integer arithmetic on host and device (spec in DESIGN.md) ----
__host__ __device__ static inline uint64_t vgl_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

struct vgl_relabel_keys { uint32_t m[3], k[3], mask; int sh; };

static vgl_relabel_keys vgl_make_relabel(int scale, uint64_t seed)
{
    vgl_relabel_keys r;
    r.mask = (scale >= 32) ? 0xFFFFFFFFu : ((1u << scale) - 1u);
    r.sh = (scale + 1) / 2;
    for (int i = 0; i < 3; i++) {
        r.m[i] = (uint32_t)vgl_splitmix64(seed + 0x100 + (uint64_t)i) | 1u;
        r.k[i] = (uint32_t)(vgl_splitmix64(seed + 0x200 + (uint64_t)i) >> 32);
    }
    return r;
}
__device__ static inline uint32_t vgl_relabel(uint32_t v, const vgl_relabel_keys &r)
{
    uint32_t x = v & r.mask;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        x = (x * r.m[i]) & r.mask;
        x ^= x >> r.sh;
        x = (x + r.k[i]) & r.mask;
    }
    return x;
}

// R-MAT recursion of graph_generation.hpp:131-169 driven by the counter-based stream
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_gen_rmat(int scale, int64_t first_edge, int64_t count, uint64_t seed,
                                                            int a, int ab, int abc, int relabel, vgl_relabel_keys rk,
                                                            int32_t *src, int32_t *dst)
{
    for (int64_t t = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; t < count; t += (int64_t)gridDim.x * VGL_BLOCK) {
        const uint64_t idx = (uint64_t)(first_edge + t);
        const uint64_t key = vgl_splitmix64(seed ^ vgl_splitmix64(idx));
        uint32_t x = 1u << (scale - 1), y = 1u << (scale - 1);
        uint64_t word = 0;
        for (int i = 1; i < scale; i++) {
            const int q = i - 1;
            if ((q & 1) == 0) word = vgl_splitmix64(key + (uint64_t)(q >> 1));
            const uint32_t r32 = (q & 1) ? (uint32_t)(word >> 32) : (uint32_t)word;
            const uint32_t p = r32 % 100u;
            const uint32_t step = 1u << (scale - (i + 1));
            if (p < (uint32_t)a)        { x -= step; y -= step; }
            else if (p < (uint32_t)ab)  { x -= step; y += step; }
            else if (p < (uint32_t)abc) { x += step; y -= step; }
            else                        { x += step; y += step; }
        }
        const uint64_t flips = vgl_splitmix64(key + 64);
        if ((flips & 1) == 0) x--;
        if ((flips & 2) == 0) y--;
        if (relabel) { x = vgl_relabel(x, rk); y = vgl_relabel(y, rk); }
        src[t] = (int32_t)x;
        dst[t] = (int32_t)y;
    }
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_gen_uniform(uint32_t mask, int64_t first_edge, int64_t count, uint64_t seed,
                                                               int32_t *src, int32_t *dst)
{
    for (int64_t t = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; t < count; t += (int64_t)gridDim.x * VGL_BLOCK) {
        const uint64_t h = vgl_splitmix64((seed + 0x5151ULL) ^ vgl_splitmix64((uint64_t)(first_edge + t)));
        src[t] = (int32_t)((uint32_t)(h >> 32) & mask);
        dst[t] = (int32_t)((uint32_t)h & mask);
    }
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_gen_weights(int64_t first_edge, int64_t count, uint64_t seed, float *w)
{
    for (int64_t t = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; t < count; t += (int64_t)gridDim.x * VGL_BLOCK) {
        const uint64_t h = vgl_splitmix64((seed + 0x7777ULL) ^ vgl_splitmix64((uint64_t)(first_edge + t)));
        w[t] = __fmul_rn((float)(uint32_t)(h >> 40), 100.0f / 16777216.0f);
    }
}

template <class T>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_gather(int64_t n, const int64_t *perm, const T *in, T *out)
{
    for (int64_t i = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * VGL_BLOCK)
        out[i] = in[perm[i]];
}

// keys for the stable sort: local row of the i-th kept edge
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_keys(int64_t n, const int64_t *kept_idx, const int32_t *src, int32_t row_begin,
                                                        int32_t *keys, unsigned long long *rowcnt /* nrows+1, pre-zeroed */)
{
    for (int64_t i = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * VGL_BLOCK) {
        const int32_t r = src[kept_idx[i]] - row_begin;
        keys[i] = r;
        atomicAdd(&rowcnt[r], 1ULL);
    }
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_degree_hist(int64_t n, const int32_t *src, const int32_t *dst, int kind, uint32_t *deg)
{
    for (int64_t i = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * VGL_BLOCK) {
        if (kind != 1) atomicAdd(&deg[src[i]], 1u);
        if (kind != 0) atomicAdd(&deg[dst[i]], 1u);
    }
}
// key = ~degree (ascending radix sort of the complement == descending degree; stable => original id ascending)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_order_keys(int32_t V, const uint32_t *deg, uint32_t *keys, int32_t *ids)
{
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK) { keys[v] = ~deg[v]; ids[v] = v; }
}
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_invert(int32_t V, const int32_t *bwd, int32_t *fwd)
{
    for (int32_t s = blockIdx.x * VGL_BLOCK + threadIdx.x; s < V; s += gridDim.x * VGL_BLOCK) fwd[bwd[s]] = s;
}
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_relabel(int64_t n, const int32_t *map, const int32_t *in, int32_t *out)
{
    for (int64_t i = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * VGL_BLOCK) out[i] = map[in[i]];
}
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_permute(int64_t n, const int32_t *idx, const uint32_t *in, uint32_t *out)
{
    for (int64_t i = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * VGL_BLOCK) out[i] = in[idx[i]];
}
// scratch[label] = min original id over the component (label = sorted id of the component's root)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_cc_min_orig(int32_t V, const int32_t *comp, const int32_t *bwd, int32_t *scratch)
{
    for (int32_t s = blockIdx.x * VGL_BLOCK + threadIdx.x; s < V; s += gridDim.x * VGL_BLOCK) atomicMin(&scratch[comp[s]], bwd[s]);
}
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_cc_emit_orig(int32_t V, const int32_t *comp, const int32_t *fwd, const int32_t *scratch, int32_t *out)
{
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK) out[v] = scratch[comp[fwd[v]]];
}

struct vgl_in_range {
    const int32_t *src; int32_t lo, hi;
    __device__ bool operator()(const int64_t &i) const { const int32_t s = src[i]; return s >= lo && s < hi; }
};

static inline int vgl_grid_for(int64_t n) { int64_t b = vgl_ceil_div(n, VGL_BLOCK); return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); }

extern "C" {

int vgl_hip_gen_rmat(vgl_hip_ctx *c, int scale, int64_t first_edge, int64_t count, uint64_t seed,
                     int a, int b, int cc, int d, int relabel, int32_t *d_src, int32_t *d_dst)
{
    if (!c) VGL_FAIL("null context");
    if (scale < 1 || scale > 30) VGL_FAIL("gen_rmat: scale must be in [1,30]");
    if (a + b + cc + d != 100) VGL_FAIL("gen_rmat: probabilities must sum to 100");
    if (count <= 0) return 0;
    vgl_relabel_keys rk = vgl_make_relabel(scale, seed);
    hipLaunchKernelGGL(vgl_k_gen_rmat, dim3(vgl_grid_for(count)), dim3(VGL_BLOCK), 0, c->stream,
                       scale, first_edge, count, seed, a, a + b, a + b + cc, relabel, rk, d_src, d_dst);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_gen_uniform(vgl_hip_ctx *c, int scale, int64_t first_edge, int64_t count, uint64_t seed,
                        int32_t *d_src, int32_t *d_dst)
{
    if (!c) VGL_FAIL("null context");
    if (scale < 1 || scale > 30) VGL_FAIL("gen_uniform: scale must be in [1,30]");
    if (count <= 0) return 0;
    const uint32_t mask = (1u << scale) - 1u;
    hipLaunchKernelGGL(vgl_k_gen_uniform, dim3(vgl_grid_for(count)), dim3(VGL_BLOCK), 0, c->stream,
                       mask, first_edge, count, seed, d_src, d_dst);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_gen_weights(vgl_hip_ctx *c, int64_t first_edge, int64_t count, uint64_t seed, float *d_w)
{
    if (!c) VGL_FAIL("null context");
    if (count <= 0) return 0;
    hipLaunchKernelGGL(vgl_k_gen_weights, dim3(vgl_grid_for(count)), dim3(VGL_BLOCK), 0, c->stream,
                       first_edge, count, seed, d_w);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_gather_u32(vgl_hip_ctx *c, int64_t n, const int64_t *d_perm, const void *d_in, void *d_out)
{
    if (!c) VGL_FAIL("null context");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(vgl_k_gather<uint32_t>, dim3(vgl_grid_for(n)), dim3(VGL_BLOCK), 0, c->stream,
                       n, d_perm, (const uint32_t *)d_in, (uint32_t *)d_out);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_coo_to_csr(vgl_hip_ctx *c, int32_t V, int64_t count, const int32_t *d_src, const int32_t *d_dst,
                       int32_t row_begin, int32_t row_end, int64_t *d_rowptr, int32_t *d_adj, int64_t *d_perm,
                       int64_t *kept_out)
{
    if (!c) VGL_FAIL("null context");
    if (row_begin < 0 || row_end > V || row_begin > row_end) VGL_FAIL("coo_to_csr: bad row range");
    if (count < 0 || count > 0x7FFFFFF0LL) VGL_FAIL("coo_to_csr: at most 2^31-16 edges per call");
    const int32_t nrows = row_end - row_begin;
    hipStream_t st = c->stream;
    VGL_HIP_TRY(hipMemsetAsync(d_rowptr, 0, sizeof(int64_t) * ((size_t)nrows + 1), st));
    if (count == 0) { if (kept_out) *kept_out = 0; return vgl_hip_ctx_sync(c); }

    // 1. stable selection of the input indices whose source row is owned
    // whatever is still allocated when the function leaves -- normally or through VGL_HIP_TRY on a failed allocation -- is freed (an
    // out-of-memory in the middle of a scale-27 shard build must not leak the multi-GB temporaries of the piece before it).
    // Round 5, late: the multi-GB temporaries (about 26 bytes per edge) are plain hipMalloc blocks again -- vgl_pool_alloc sends blocks of 64 MiB and more
    // there since kernels touching a freshly grown pool block of that size faulted in 2 - 12 % of the processes (vgl_hip_internal.h).  The idea it replaces:
    // the temporaries (about 26 bytes per edge) come from the library's stream-ordered pool, like those of the plan builders: what the graph
    // build has touched once is what the blocked-plan build of the same graph gets next -- the PageRank plan of uniform-25 paid 1.2 s for FRESH memory
    // (first touch, ~14 ms per GB, and the allocator's stalls) in front of 60 ms of kernels while the build before it took and returned device memory directly
    vgl_scratch b_kept, b_sorted, b_nkept, b_keys, b_keys_sorted, b_temp;
    struct cleanup {
        hipStream_t st;
        vgl_scratch *all[6];
        ~cleanup() { for (vgl_scratch *b : all) vgl_scratch_free(st, b); }
    } guard{st, {&b_kept, &b_sorted, &b_nkept, &b_keys, &b_keys_sorted, &b_temp}};
    size_t temp_bytes = 0, need = 0;
    VGL_HIP_TRY(vgl_scratch_alloc(st, &b_kept, sizeof(int64_t) * (size_t)count));
    VGL_HIP_TRY(vgl_scratch_alloc(st, &b_nkept, sizeof(int64_t)));
    int64_t *kept_idx = (int64_t *)b_kept.p, *d_nkept = (int64_t *)b_nkept.p;
    rocprim::counting_iterator<int64_t> iota(0);
    vgl_in_range pred{d_src, row_begin, row_end};
    VGL_HIP_TRY(rocprim::select(nullptr, need, iota, kept_idx, (size_t *)d_nkept, (size_t)count, pred, st));
    temp_bytes = need;
    VGL_HIP_TRY(vgl_scratch_alloc(st, &b_temp, temp_bytes ? temp_bytes : 16));
    VGL_HIP_TRY(rocprim::select(b_temp.p, temp_bytes, iota, kept_idx, (size_t *)d_nkept, (size_t)count, pred, st));
    int64_t nkept = 0;
    VGL_HIP_TRY(hipMemcpyAsync(&nkept, d_nkept, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    VGL_HIP_TRY(hipStreamSynchronize(st));
    if (kept_out) *kept_out = nkept;
    if (nkept > 0) {
        // 2. keys = local rows, row histogram
        VGL_HIP_TRY(vgl_scratch_alloc(st, &b_keys, sizeof(int32_t) * (size_t)nkept));
        VGL_HIP_TRY(vgl_scratch_alloc(st, &b_keys_sorted, sizeof(int32_t) * (size_t)nkept));
        VGL_HIP_TRY(vgl_scratch_alloc(st, &b_sorted, sizeof(int64_t) * (size_t)nkept));
        int32_t *keys = (int32_t *)b_keys.p, *keys_sorted = (int32_t *)b_keys_sorted.p;
        int64_t *sorted_idx = (int64_t *)b_sorted.p;
        hipLaunchKernelGGL(vgl_k_keys, dim3(vgl_grid_for(nkept)), dim3(VGL_BLOCK), 0, st, nkept, kept_idx, d_src, row_begin,
                           keys, (unsigned long long *)(d_rowptr + 1));
        VGL_HIP_TRY(hipGetLastError());
        // 3. stable radix sort by row (LSD radix sort is stable => adjacency keeps input order)
        int bits = 1;
        while (bits < 31 && (1LL << bits) < (int64_t)nrows) bits++;
        need = 0;
        VGL_HIP_TRY(rocprim::radix_sort_pairs(nullptr, need, keys, keys_sorted, kept_idx, sorted_idx, (size_t)nkept, 0, bits, st));
        if (need > temp_bytes) { vgl_scratch_free(st, &b_temp); temp_bytes = need; VGL_HIP_TRY(vgl_scratch_alloc(st, &b_temp, temp_bytes)); }
        VGL_HIP_TRY(rocprim::radix_sort_pairs(b_temp.p, need, keys, keys_sorted, kept_idx, sorted_idx, (size_t)nkept, 0, bits, st));
        // 4. adjacency + optional permutation
        hipLaunchKernelGGL(vgl_k_gather<int32_t>, dim3(vgl_grid_for(nkept)), dim3(VGL_BLOCK), 0, st, nkept, sorted_idx, d_dst, d_adj);
        VGL_HIP_TRY(hipGetLastError());
        if (d_perm) VGL_HIP_TRY(hipMemcpyAsync(d_perm, sorted_idx, sizeof(int64_t) * (size_t)nkept, hipMemcpyDeviceToDevice, st));
        // 5. row offsets: inclusive scan of the histogram stored at rowptr[1..nrows]
        need = 0;
        VGL_HIP_TRY(rocprim::inclusive_scan(nullptr, need, d_rowptr + 1, d_rowptr + 1, (size_t)nrows, rocprim::plus<int64_t>(), st));
        if (need > temp_bytes) { vgl_scratch_free(st, &b_temp); temp_bytes = need; VGL_HIP_TRY(vgl_scratch_alloc(st, &b_temp, temp_bytes)); }
        VGL_HIP_TRY(rocprim::inclusive_scan(b_temp.p, need, d_rowptr + 1, d_rowptr + 1, (size_t)nrows, rocprim::plus<int64_t>(), st));
    }
    VGL_HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int vgl_hip_degree_hist_add(vgl_hip_ctx *c, int64_t count, const int32_t *d_src, const int32_t *d_dst, int degree_kind, uint32_t *d_degree)
{
    if (!c || !d_degree || (count > 0 && (!d_src || !d_dst))) VGL_FAIL("degree_hist_add: null argument");      // an empty edge list may come without arrays
    if (degree_kind < 0 || degree_kind > 2) VGL_FAIL("degree_hist_add: degree_kind must be 0 (out), 1 (in) or 2 (in+out)");
    if (count > 0) hipLaunchKernelGGL(vgl_k_degree_hist, dim3(vgl_grid_for(count)), dim3(VGL_BLOCK), 0, c->stream, count, d_src, d_dst, degree_kind, d_degree);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_degree_order(vgl_hip_ctx *c, int32_t V, int64_t count, const int32_t *d_src, const int32_t *d_dst, int degree_kind,
                         int32_t *d_fwd, int32_t *d_bwd)
{
    if (!c || !d_fwd || !d_bwd || (count > 0 && (!d_src || !d_dst))) VGL_FAIL("degree_order: null argument");
    uint32_t *deg = nullptr;
    VGL_HIP_TRY(hipMalloc((void **)&deg, sizeof(uint32_t) * (size_t)V));
    VGL_HIP_TRY(hipMemsetAsync(deg, 0, sizeof(uint32_t) * (size_t)V, c->stream));
    int rc = vgl_hip_degree_hist_add(c, count, d_src, d_dst, degree_kind, deg);
    if (rc == 0) rc = vgl_hip_degree_order_from_degrees(c, V, deg, d_fwd, d_bwd);
    hipStreamSynchronize(c->stream);
    hipFree(deg);
    return rc;
}

int vgl_hip_degree_order_from_degrees(vgl_hip_ctx *c, int32_t V, const uint32_t *deg, int32_t *d_fwd, int32_t *d_bwd)
{
    if (!c || !deg || !d_fwd || !d_bwd) VGL_FAIL("degree_order_from_degrees: null argument");
    hipStream_t st = c->stream;
    uint32_t *keys = nullptr, *keys_out = nullptr;
    int32_t *ids = nullptr;
    void *temp = nullptr;
    size_t need = 0;
    struct cleanup {
        hipStream_t st;
        uint32_t *&a, *&b; int32_t *&c; void *&d;
        ~cleanup() { for (void *p : {(void *)a, (void *)b, (void *)c, d}) vgl_pool_free(st, p); }
    } guard{st, keys, keys_out, ids, temp};
    VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&keys, sizeof(uint32_t) * (size_t)V));
    VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&keys_out, sizeof(uint32_t) * (size_t)V));
    VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&ids, sizeof(int32_t) * (size_t)V));
    hipLaunchKernelGGL(vgl_k_order_keys, dim3(vgl_grid_for(V)), dim3(VGL_BLOCK), 0, st, V, deg, keys, ids);
    VGL_HIP_TRY(hipGetLastError());
    VGL_HIP_TRY(rocprim::radix_sort_pairs(nullptr, need, keys, keys_out, ids, d_bwd, (size_t)V, 0, 32, st));
    VGL_HIP_TRY(vgl_pool_alloc(st, &temp, need ? need : 16));
    VGL_HIP_TRY(rocprim::radix_sort_pairs(temp, need, keys, keys_out, ids, d_bwd, (size_t)V, 0, 32, st));
    hipLaunchKernelGGL(vgl_k_invert, dim3(vgl_grid_for(V)), dim3(VGL_BLOCK), 0, st, V, d_bwd, d_fwd);
    VGL_HIP_TRY(hipGetLastError());
    VGL_HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int vgl_hip_relabel_i32(vgl_hip_ctx *c, int64_t n, const int32_t *d_map, const int32_t *d_in, int32_t *d_out)
{
    if (!c || !d_map || (n > 0 && (!d_in || !d_out))) VGL_FAIL("relabel_i32: null argument");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(vgl_k_relabel, dim3(vgl_grid_for(n)), dim3(VGL_BLOCK), 0, c->stream, n, d_map, d_in, d_out);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}
int vgl_hip_permute_u32(vgl_hip_ctx *c, int64_t n, const int32_t *d_idx, const void *d_in, void *d_out)
{
    if (!c || !d_idx || !d_in || !d_out) VGL_FAIL("permute_u32: null argument");
    if (d_in == d_out) VGL_FAIL("permute_u32: in-place permutation is not supported");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(vgl_k_permute, dim3(vgl_grid_for(n)), dim3(VGL_BLOCK), 0, c->stream, n, d_idx, (const uint32_t *)d_in, (uint32_t *)d_out);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}
int vgl_hip_cc_labels_to_original(vgl_hip_ctx *c, int32_t V, const int32_t *d_comp, const int32_t *d_fwd, const int32_t *d_bwd,
                                  int32_t *d_scratch, int32_t *d_out)
{
    if (!c || !d_comp || !d_fwd || !d_bwd || !d_scratch || !d_out) VGL_FAIL("cc_labels_to_original: null argument");
    VGL_HIP_TRY(hipMemsetAsync(d_scratch, 0x7f, sizeof(int32_t) * (size_t)V, c->stream));      // 0x7f7f7f7f > any id
    hipLaunchKernelGGL(vgl_k_cc_min_orig, dim3(vgl_grid_for(V)), dim3(VGL_BLOCK), 0, c->stream, V, d_comp, d_bwd, d_scratch);
    hipLaunchKernelGGL(vgl_k_cc_emit_orig, dim3(vgl_grid_for(V)), dim3(VGL_BLOCK), 0, c->stream, V, d_comp, d_fwd, d_scratch, d_out);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_partition_rows(vgl_hip_ctx *c, int32_t V, const int64_t *d_rowptr, int parts, int32_t *bounds_host)
{
    if (!c || !bounds_host) VGL_FAIL("null argument");
    if (parts < 1) VGL_FAIL("partition_rows: parts must be >= 1");
    // the row-offset array is small (8*(V+1) bytes); bring it to the host and cut it there.  Bounds are rounded
    // to multiples of 64 so every shard owns whole bitmap words.
    std::vector<int64_t> rp((size_t)V + 1);
    VGL_TRY(vgl_hip_memcpy_d2h(c, rp.data(), d_rowptr, sizeof(int64_t) * ((size_t)V + 1)));
    const int64_t E = rp[V];
    bounds_host[0] = 0;
    for (int p = 1; p < parts; p++) {
        const int64_t target = E * p / parts;
        int64_t lo = 0, hi = V;                       // first row whose start offset >= target
        while (lo < hi) { int64_t mid = (lo + hi) / 2; if (rp[mid] < target) lo = mid + 1; else hi = mid; }
        int64_t b = (lo + 32) / 64 * 64;
        if (b > V) b = V;
        if (b < bounds_host[p - 1]) b = bounds_host[p - 1];
        bounds_host[p] = (int32_t)b;
    }
    bounds_host[parts] = V;
    return 0;
}

}  // extern "C"
