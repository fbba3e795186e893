// pr.hip -- PageRank (PR::vgl_page_rank, algorithms/pr/pr.hpp:7-149; f32, d = 0.85, fixed iteration count).
//
// Per iteration (pr.hpp:83-136):
//   prepare : old = rank; contrib[v] = old[v] * rdeg[v]  (the per-edge product dst_rank * reversed_dst_links_num of
//             pr.hpp:111-116, hoisted: the f32 product is the same value for every edge that reads it);
//             dangling = sum over v with (indeg-loops)==0 of old[v] / V  -- accumulated in f64 in a fixed tree and rounded
//             to f32 once (the reference's f32 OpenMP reduction is thread-order dependent)
//   pull    : rank[src] = k + d * (sum_{src->dst, dst != src} contrib[dst] + dangling), the sum taken IN ADJACENCY ORDER
//             in f32 exactly like the reference's `+=` chain, so the result is bit-identical to seq_page_rank's
//             evaluation order (seq_pr.hpp:81-96): vgl_k_pull_sum<float, skip self loops> of vgl_pull.h (256 rows per workgroup,
//             edges staged through LDS; rows with >= 512 edges go through the hub schedule of the same launch).
// Algorithmic bytes per iteration: 8*E + 28*V.
#include "vgl_pull.h"
#include "vgl_blocked.h"
#include "vgl_comm.h"
#include <queue>
#include <cstdio>
#include <cstdlib>

static inline unsigned vgl_grid3(int64_t n, int64_t cap) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cap, vgl_ceil_div(n, VGL_BLOCK))); }

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_pr_sum_partial(int32_t V, const float *ranks, double *partials)
{
    __shared__ double s[VGL_WAVES];
    double acc = 0.0;
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK) acc += (double)ranks[v];
    acc = vgl_block_reduce_add(acc, s);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// setup (pr.hpp:37-73): ranks = float(1.0/V); rdeg = float(1.0/indeg) or 0
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_pr_setup(int32_t V, const int32_t *indeg, float *ranks, float *rdeg)
{
    const float init = (float)(1.0 / (double)V);
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK) {
        ranks[v] = init;
        const int32_t dg = indeg[v];
        rdeg[v] = (dg == 0) ? 0.0f : (float)(1.0 / (double)dg);
    }
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_pr_prepare(int32_t V, const int32_t *indeg, const float *rdeg, const float *ranks,
                                                              float *contrib, double *partials)
{
    __shared__ double s[VGL_WAVES];
    const float fV = (float)V;
    double acc = 0.0;
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK) {
        const float old = ranks[v];
        contrib[v] = __fmul_rn(old, rdeg[v]);
        if (indeg[v] == 0) acc += (double)__fdiv_rn(old, fV);     // old_page_ranks[src_id] / vertices_count (pr.hpp:98)
    }
    acc = vgl_block_reduce_add(acc, s);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_pr_dangling(int nparts, const double *partials, float *dangling_out)
{
    __shared__ double s[VGL_WAVES];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += VGL_BLOCK) acc += partials[i];
    acc = vgl_block_reduce_add(acc, s);
    if (threadIdx.x == 0) *dangling_out = (float)acc;
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_pull_find_hubs(int32_t nrows, const int64_t *rowptr, int32_t *hub_rows, int32_t *hub_deg,
                                                                  int32_t *hub_count)
{
    for (int32_t r = blockIdx.x * VGL_BLOCK + threadIdx.x; r < nrows; r += gridDim.x * VGL_BLOCK) {
        const int64_t dg = rowptr[r + 1] - rowptr[r];
        if (dg >= VGL_PULL_HUB_DEGREE) {                     // few thousand rows, once per graph
            const int slot = atomicAdd(hub_count, 1);
            hub_rows[slot] = r;
            hub_deg[slot] = (int32_t)min(dg, (int64_t)INT32_MAX);
        }
    }
}

struct vgl_pr_epilogue {                                     // k + d * (rank + dangling)  (pr.hpp:121)
    const float *dangling;
    float k, d;
    float *ranks_out;
    __device__ __forceinline__ void operator()(int32_t v, float acc) const
    {
        ranks_out[v] = __fadd_rn(k, __fmul_rn(d, __fadd_rn(acc, *dangling)));
    }
};

// indegree without self loops from an out-CSR shard (pr.hpp:31-65 computes it from the incoming graph; same numbers)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_indeg_noloops(const int64_t *rowptr, const int32_t *adj, const int32_t *tile_row,
                                                                 int64_t E, int32_t row_base, int32_t *indeg)
{
    __shared__ int s_map[VGL_TILE];
    __shared__ int s_w[VGL_WAVES];
    const int64_t e0 = (int64_t)blockIdx.x * VGL_TILE;
    const int n = (int)min((int64_t)VGL_TILE, E - e0);
    const int r_first = tile_row[blockIdx.x];
    const int r_last = tile_row[blockIdx.x + 1];
    vgl_tile_row_map(s_map, s_w, rowptr, e0, r_first, r_last);
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) {
        const int i = threadIdx.x + j * VGL_BLOCK;
        if (i < n) {
            const int32_t dst = adj[e0 + i];
            if (dst != row_base + r_first + s_map[i]) atomicAdd(indeg + dst, 1);
        }
    }
}

// the same numbers from the incoming CSR when the graph has one (what pr.hpp:31-65 does): in-degree from the row offsets, then one
// streaming pass over the outgoing edges takes the self loops off again (rare: a handful of atomics instead of one per edge --
// the per-edge atomics above serialise on the hubs' counters, 53 ms on a degree-sorted RMAT-24)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_indeg_from_in_rows(int32_t V, const int64_t *in_rowptr, int32_t *indeg)
{
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK) indeg[v] = (int32_t)(in_rowptr[v + 1] - in_rowptr[v]);
}
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_indeg_sub_loops(const int64_t *rowptr, const int32_t *adj, const int32_t *tile_row,
                                                                   int64_t E, int32_t row_base, int32_t *indeg)
{
    __shared__ int s_map[VGL_TILE];
    __shared__ int s_w[VGL_WAVES];
    const int64_t e0 = (int64_t)blockIdx.x * VGL_TILE;
    const int n = (int)min((int64_t)VGL_TILE, E - e0);
    const int r_first = tile_row[blockIdx.x];
    const int r_last = tile_row[blockIdx.x + 1];
    vgl_tile_row_map(s_map, s_w, rowptr, e0, r_first, r_last);
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) {
        const int i = threadIdx.x + j * VGL_BLOCK;
        if (i < n) {
            const int32_t dst = adj[e0 + i];
            if (dst == row_base + r_first + s_map[i]) atomicSub(indeg + dst, 1);
        }
    }
}


int vgl_pull_find_hubs(vgl_hip_ctx *c, vgl_hip_graph *g, vgl_dir_csr &dir)
{
    if (dir.hub_rows) return 0;
    int32_t *d_rows = nullptr, *d_deg = nullptr, *d_count = nullptr;
    const size_t cap = (size_t)std::max<int32_t>(g->nrows, 1);
    VGL_HIP_TRY(hipMalloc((void **)&d_rows, sizeof(int32_t) * cap));
    VGL_HIP_TRY(hipMalloc((void **)&d_deg, sizeof(int32_t) * cap));
    VGL_HIP_TRY(hipMalloc((void **)&d_count, sizeof(int32_t)));
    VGL_HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(int32_t), c->stream));
    hipLaunchKernelGGL(vgl_k_pull_find_hubs, dim3(vgl_grid3(g->nrows, 4096)), dim3(VGL_BLOCK), 0, c->stream, g->nrows, dir.rowptr, d_rows,
                       d_deg, d_count);
    VGL_HIP_TRY(hipGetLastError());
    VGL_TRY(vgl_hip_memcpy_d2h(c, &dir.nhubs, d_count, sizeof(int32_t)));
    const size_t n = (size_t)dir.nhubs;
    const int hub_blocks_cap = vgl_env(c, "VGL_PULL_HUB_BLOCKS") ? std::max(1, atoi(vgl_env(c, "VGL_PULL_HUB_BLOCKS"))) : VGL_PULL_HUB_BLOCKS;
    dir.hub_blocks = n ? (int)std::min<int64_t>(hub_blocks_cap, vgl_ceil_div((int64_t)n, VGL_WAVES)) : 0;
    const int W = dir.hub_blocks * VGL_WAVES;
    // device layout: [n hub rows grouped by wavefront][W+1 offsets]
    VGL_HIP_TRY(hipMalloc((void **)&dir.hub_rows, sizeof(int32_t) * (n + (size_t)W + 1)));
    if (n > 0) {
        std::vector<int32_t> rows(n), deg(n), order(n);
        VGL_TRY(vgl_hip_memcpy_d2h(c, rows.data(), d_rows, sizeof(int32_t) * n));
        VGL_TRY(vgl_hip_memcpy_d2h(c, deg.data(), d_deg, sizeof(int32_t) * n));
        for (size_t i = 0; i < n; i++) order[i] = (int32_t)i;
        std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return deg[x] != deg[y] ? deg[x] > deg[y] : rows[x] < rows[y]; });
        // giants (vgl_pull.h): the longest rows go to whole workgroups, longest-processing-time over at most VGL_PULL_GIANT_BLOCKS of them
        size_t ng = 0;
        const int giant_degree = vgl_env(c, "VGL_PULL_GIANT_DEGREE") ? atoi(vgl_env(c, "VGL_PULL_GIANT_DEGREE")) : VGL_PULL_GIANT_DEGREE;
        while (ng < n && deg[order[ng]] >= giant_degree) ng++;
        if (vgl_env(c, "VGL_PULL_NO_GIANTS")) ng = 0;
        if (vgl_env(c, "VGL_PULL_TRACE")) {
            int64_t ge = 0, he = 0;
            for (size_t i = 0; i < n; i++) (i < ng ? ge : he) += deg[order[i]];
            fprintf(stderr, "[vgl pull] %zu hubs (%lld entries), %zu giants (%lld entries, largest %d)\n", n, (long long)(ge + he), ng, (long long)ge, n ? deg[order[0]] : 0);
        }
        dir.ngiants = (int)ng;
        dir.giant_blocks = (int)std::min<size_t>(ng, VGL_PULL_GIANT_BLOCKS);
        if (ng > 0) {
            typedef std::pair<int64_t, int> gslot;
            std::priority_queue<gslot, std::vector<gslot>, std::greater<gslot>> gheap;
            for (int b = 0; b < dir.giant_blocks; b++) gheap.push(gslot(0, b));
            std::vector<std::vector<int32_t>> glists((size_t)dir.giant_blocks);
            for (size_t i = 0; i < ng; i++) {
                gslot sl = gheap.top();
                gheap.pop();
                glists[(size_t)sl.second].push_back(rows[order[i]]);
                gheap.push(gslot(sl.first + deg[order[i]], sl.second));
            }
            std::vector<int32_t> gp(ng + (size_t)dir.giant_blocks + 1);
            size_t gpos = 0;
            for (int b = 0; b < dir.giant_blocks; b++) {
                gp[ng + (size_t)b] = (int32_t)gpos;
                for (int32_t r : glists[(size_t)b]) gp[gpos++] = r;
            }
            gp[ng + (size_t)dir.giant_blocks] = (int32_t)gpos;
            VGL_HIP_TRY(hipMalloc((void **)&dir.giant_rows, sizeof(int32_t) * gp.size()));
            VGL_TRY(vgl_hip_memcpy_h2d(c, dir.giant_rows, gp.data(), sizeof(int32_t) * gp.size()));
        }
        // longest-processing-time list scheduling: the next-largest hub goes to the least loaded wavefront, so the critical path is
        // max(largest hub, total / W) edges; a fixed per-hub cost stands for the un-overlapped first batch
        typedef std::pair<int64_t, int> slot;
        std::priority_queue<slot, std::vector<slot>, std::greater<slot>> heap;
        for (int wv = 0; wv < W; wv++) heap.push(slot(0, wv));
        std::vector<std::vector<int32_t>> lists((size_t)W);
        for (size_t i = ng; i < n; i++) {                                  // (the giants have their own workgroups)
            slot s = heap.top();
            heap.pop();
            lists[(size_t)s.second].push_back(rows[order[i]]);
            heap.push(slot(s.first + deg[order[i]] + 2048, s.second));
        }
        std::vector<int32_t> packed(n + (size_t)W + 1);
        size_t pos = 0;
        for (int wv = 0; wv < W; wv++) {
            packed[n + (size_t)wv] = (int32_t)pos;
            for (int32_t r : lists[(size_t)wv]) packed[pos++] = r;
        }
        packed[n + (size_t)W] = (int32_t)pos;
        VGL_TRY(vgl_hip_memcpy_h2d(c, dir.hub_rows, packed.data(), sizeof(int32_t) * packed.size()));
        // chunk list of the unordered variant: hubs in row order, every hub cut into VGL_PULL_CHUNK-entry chunks
        std::vector<int32_t> by_row(n);
        for (size_t i = 0; i < n; i++) by_row[i] = (int32_t)i;
        std::sort(by_row.begin(), by_row.end(), [&](int32_t x, int32_t y) { return rows[x] < rows[y]; });
        std::vector<int32_t> chunks, hubs;
        for (size_t i = 0; i < n; i++) {
            const int32_t r = rows[by_row[i]], dg = deg[by_row[i]];
            const int32_t nc = (int32_t)vgl_ceil_div((int64_t)dg, VGL_PULL_CHUNK);
            hubs.push_back(r); hubs.push_back((int32_t)(chunks.size() / 2)); hubs.push_back(nc);
            for (int32_t k = 0; k < nc; k++) { chunks.push_back(r); chunks.push_back(k); }
        }
        dir.n_hub_chunks = (int)(chunks.size() / 2); dir.n_hub_list = (int)n;
        chunks.insert(chunks.end(), hubs.begin(), hubs.end());              // [2 * n_chunks | 3 * n_hubs]
        VGL_HIP_TRY(hipMalloc((void **)&dir.hub_chunks, sizeof(int32_t) * chunks.size()));
        VGL_TRY(vgl_hip_memcpy_h2d(c, dir.hub_chunks, chunks.data(), sizeof(int32_t) * chunks.size()));
        VGL_HIP_TRY(hipMalloc((void **)&dir.hub_chunk_sums, sizeof(double) * (size_t)std::max(1, dir.n_hub_chunks)));
    }
    VGL_HIP_TRY(hipFree(d_rows));
    VGL_HIP_TRY(hipFree(d_deg));
    VGL_HIP_TRY(hipFree(d_count));
    // row blocks of the ordinary workgroups: every aligned group of 256 rows is cut into 1, 2, 4, ... 32 equal parts until a part
    // holds about VGL_PULL_BLOCK_EDGES edges (hub rows count too: they only make their neighbourhood's parts smaller)
    {
        const int32_t nrows = g->nrows;
        const int64_t ngroups = vgl_ceil_div(nrows, VGL_BLOCK);
        std::vector<int64_t> gstart((size_t)ngroups + 1);
        {   // row offsets at the group boundaries (one strided device -> host copy)
            std::vector<int64_t> tmp((size_t)ngroups + 1);
            VGL_HIP_TRY(hipMemcpy2DAsync(tmp.data(), sizeof(int64_t), dir.rowptr, sizeof(int64_t) * VGL_BLOCK, sizeof(int64_t), (size_t)ngroups,
                                         hipMemcpyDeviceToHost, c->stream));
            VGL_HIP_TRY(hipMemcpyAsync(tmp.data() + ngroups, dir.rowptr + nrows, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
            VGL_HIP_TRY(hipStreamSynchronize(c->stream));
            gstart.swap(tmp);
        }
        std::vector<int32_t> blk_row;
        blk_row.reserve((size_t)ngroups + 64);
        for (int64_t gi = 0; gi < ngroups; gi++) {
            const int32_t r0 = (int32_t)(gi * VGL_BLOCK), r1 = (int32_t)std::min<int64_t>(nrows, (gi + 1) * VGL_BLOCK);
            const int64_t e = gstart[(size_t)gi + 1] - gstart[(size_t)gi];
            int parts = 1;
            while (parts < 32 && e > VGL_PULL_BLOCK_EDGES * parts && (r1 - r0) >= 2 * parts) parts *= 2;
            const int32_t step = (int32_t)vgl_ceil_div(r1 - r0, parts);
            for (int32_t r = r0; r < r1; r += step) blk_row.push_back(r);
        }
        blk_row.push_back(nrows);
        dir.pull_nblk = (int)blk_row.size() - 1;
        VGL_HIP_TRY(hipMalloc((void **)&dir.pull_blk_row, sizeof(int32_t) * blk_row.size()));
        VGL_TRY(vgl_hip_memcpy_h2d(c, dir.pull_blk_row, blk_row.data(), sizeof(int32_t) * blk_row.size()));
    }
    return 0;
}

// Blocked pull (vgl_blocked.h): contrib[dst] is read from a 128 KiB LDS window instead of one L2 line per edge and the per-row sums
// are kept in 64-bit FIXED POINT (unit 2^-62; LDS float atomics are ~30x slower than integer ones on this chip): every contribution
// of at least 2^-39 converts exactly, smaller ones lose less than 2^-62 each, integer addition is associative -- so the sum is the
// exact sum of the f32 products (rounded to f32 once at the end), the same bits whatever the order of arrival, from run to run and
// for any cut into units.  It is NOT the reference's f32 `+=` chain in adjacency order: the two differ by the chain's own rounding
// error (~sqrt(n) * 3e-8 for a row of n entries: 2-6e-7 on uniform-25, 1e-4 on the largest RMAT-24 hub), which is why AUTO keeps the
// ordered kernel whenever a row is long enough for that to approach the 1e-6 bar of the north star.
struct vgl_pr_blk_op {
    typedef unsigned long long acc_t;
    static constexpr bool MARK = false;
    const float *contrib;
    const float *dangling;
    float k, d;
    float *ranks_out;
    int32_t a_base;
    __device__ __forceinline__ uint32_t load(int32_t i) const { return __float_as_uint(contrib[i]); }
    __device__ __forceinline__ uint32_t edge(uint32_t x, float) const { return x; }
    __device__ __forceinline__ acc_t identity() const { return 0ull; }
    // f32 (non-negative, < 2) -> fixed point with 62 fraction bits, by the bits: mantissa << (exponent - 150 + 62)
    static __device__ __forceinline__ acc_t to_fixed(uint32_t bits)
    {
        const int ex = (int)(bits >> 23) & 0xFF;
        const unsigned long long m = (unsigned long long)((bits & 0x7FFFFFu) | (ex ? 0x800000u : 0u));
        const int sh = (ex ? ex : 1) - 150 + 62;
        return sh >= 0 ? m << sh : (sh > -24 ? m >> -sh : 0ull);
    }
    __device__ __forceinline__ void accumulate(acc_t *p, uint32_t v) const
    {
        __hip_atomic_fetch_add(p, to_fixed(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ acc_t combine(acc_t a, acc_t b) const { return a + b; }
    __device__ __forceinline__ bool partial(int32_t, acc_t) const { return false; }
    __device__ __forceinline__ void finish(int32_t i, acc_t acc) const
    {
        const float sum = (float)((double)acc * 0x1p-62);           // 53 of the 64 bits, then f32: the double rounding is below 2^-53 relative
        ranks_out[a_base + i] = __fadd_rn(k, __fmul_rn(d, __fadd_rn(sum, *dangling)));
    }
};

// The declared operator VGL_SUM_OVER_EDGES (hip/vgl_hip.hpp): sums[src] = sum of values[dst] over the edges src -> dst, dst != src -- the pull of
// pr.hpp:109-123 with the caller's own arrays.  Same fixed-point accumulation as vgl_pr_blk_op, the unit scaled by the caller's bound on a
// per-vertex sum (rounded up to a power of two, so the scaling is exact); the epilogue stores the sum, the caller's post operator does the rest.
struct vgl_sum_blk_op {
    typedef unsigned long long acc_t;
    static constexpr bool MARK = false;
    const float *values;
    float *sums;
    int32_t a_base;
    int bound_exp;                                          // per-vertex sums stay below 2^bound_exp
    __device__ __forceinline__ uint32_t load(int32_t i) const { return __float_as_uint(values[i]); }
    __device__ __forceinline__ uint32_t edge(uint32_t x, float) const { return x; }
    __device__ __forceinline__ acc_t identity() const { return 0ull; }
    __device__ __forceinline__ void accumulate(acc_t *p, uint32_t bits) const
    {
        const int ex = (int)(bits >> 23) & 0xFF;
        const unsigned long long m = (unsigned long long)((bits & 0x7FFFFFu) | (ex ? 0x800000u : 0u));
        const int sh = (ex ? ex : 1) - 150 + 62 - bound_exp;
        __hip_atomic_fetch_add(p, sh >= 0 ? m << sh : (sh > -24 ? m >> -sh : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ acc_t combine(acc_t a, acc_t b) const { return a + b; }
    __device__ __forceinline__ bool partial(int32_t, acc_t) const { return false; }
    __device__ __forceinline__ void finish(int32_t i, acc_t acc) const { sums[a_base + i] = (float)ldexp((double)acc, bound_exp - 62); }
};

// VGL_PR_MODE=0|1 overrides AUTO (anything else is refused)
int vgl_pr_env_mode(vgl_hip_ctx *c, int mode, int *out)
{
    *out = mode;
    if (mode != VGL_HIP_PR_AUTO) return 0;
    const char *s = vgl_env(c, "VGL_PR_MODE");
    if (!s || !*s) return 0;
    if ((s[0] != '0' && s[0] != '1') || s[1] != 0) VGL_FAIL("VGL_PR_MODE must be 0 (ordered chain) or 1 (blocked exact sums)");
    *out = s[0] - '0';
    return 0;
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_max_row(int32_t nrows, const int64_t *rowptr, unsigned long long *out)
{
    __shared__ unsigned long long s[VGL_WAVES];
    unsigned long long m = 0;
    for (int32_t r = blockIdx.x * VGL_BLOCK + threadIdx.x; r < nrows; r += gridDim.x * VGL_BLOCK) m = max(m, (unsigned long long)(rowptr[r + 1] - rowptr[r]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned long long)__shfl_xor(m, o));
    if (vgl_lane() == 0) s[vgl_wave()] = m;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = s[0]; for (int w = 1; w < VGL_WAVES; w++) t = max(t, s[w]); atomicMax(out, t); }
}

// AUTO: blocked for large graphs whose rows are all short (the exact sum then stays well inside 1e-6 of the ordered chain)
int vgl_pr_longest_row(vgl_hip_ctx *c, vgl_hip_graph *g, int64_t *out)
{
    if (g->out.max_row < 0) {
        VGL_TRY(vgl_zero_counters(c, C_TMP0, 1));
        hipLaunchKernelGGL(vgl_k_max_row, dim3(vgl_grid3(g->nrows, 1024)), dim3(VGL_BLOCK), 0, c->stream, g->nrows, g->out.rowptr,
                           (unsigned long long *)(c->d_counters + C_TMP0));
        VGL_TRY(vgl_read_counters(c, false));
        g->out.max_row = c->h_counters[C_TMP0];
    }
    *out = g->out.max_row;
    return 0;
}

static int vgl_pr_mode_auto(vgl_hip_ctx *c, vgl_hip_graph *g, int *mode)
{
    VGL_TRY(vgl_pr_env_mode(c, *mode, mode));
    if (*mode != VGL_HIP_PR_AUTO) return 0;
    *mode = VGL_HIP_PR_EXACT_ORDER;
    if (g->out.edges < (1LL << 25)) return 0;
    int64_t longest = 0;
    VGL_TRY(vgl_pr_longest_row(c, g, &longest));
    if (longest <= 256) *mode = VGL_HIP_PR_BLOCKED;
    return 0;
}

int vgl_pr_iteration(vgl_hip_ctx *c, vgl_hip_graph *g, const int32_t *indeg, const float *rdeg, float *ranks, float *contrib,
                     float *ranks_out, int mode)
{
    VGL_TRY(vgl_pr_mode_auto(c, g, &mode));
    if (mode == VGL_HIP_PR_BLOCKED) {
        const int32_t V = g->V;
        const float d = 0.85f;
        const float k = (float)((1.0 - (double)d) / (double)((float)V));
        const int npart = (int)vgl_grid3(V, 1024);
        if (!g->blk_pr) VGL_TRY(vgl_blocked_plan_build(c, g->out, g->nrows, g->row_begin, V, 0, 1, nullptr, VGL_BLK_BITS - 1, &g->blk_pr));
        VGL_TRY(vgl_ensure_partials(c, (size_t)npart + 8));
        float *dangling = reinterpret_cast<float *>(c->d_partials + npart);
        hipLaunchKernelGGL(vgl_k_pr_prepare, dim3(npart), dim3(VGL_BLOCK), 0, c->stream, V, indeg, rdeg, ranks, contrib, c->d_partials);
        hipLaunchKernelGGL(vgl_k_pr_dangling, dim3(1), dim3(VGL_BLOCK), 0, c->stream, npart, c->d_partials, dangling);
        const vgl_pr_blk_op op{contrib, dangling, k, d, ranks_out, g->row_begin};
        return vgl_blocked_pass<vgl_pr_blk_op, false, true>(c, g->blk_pr, op, "pr_blk_gather", "pr_blk_accumulate");
    }
    const int32_t V = g->V;
    const float d = 0.85f;
    const float k = (float)((1.0 - (double)d) / (double)((float)V));       // pr.hpp:37-38
    const int npart = (int)vgl_grid3(V, 1024);
    VGL_TRY(vgl_pull_find_hubs(c, g, g->out));
    const unsigned nblk = (unsigned)g->out.pull_nblk;
    VGL_TRY(vgl_ensure_partials(c, (size_t)npart + 8));
    float *dangling = reinterpret_cast<float *>(c->d_partials + npart);    // one slot after the prepare partials
    hipLaunchKernelGGL(vgl_k_pr_prepare, dim3(npart), dim3(VGL_BLOCK), 0, c->stream, V, indeg, rdeg, ranks, contrib, c->d_partials);
    hipLaunchKernelGGL(vgl_k_pr_dangling, dim3(1), dim3(VGL_BLOCK), 0, c->stream, npart, c->d_partials, dangling);
    const int hub_blocks = g->out.hub_blocks;
    {
        vgl_timed_launch tl(c, "pr_pull");
        const vgl_pr_epilogue epi{dangling, k, d, ranks_out};
        const int giant_blocks = g->out.giant_blocks;
        hipLaunchKernelGGL((vgl_k_pull_sum<float, true, false, vgl_pr_epilogue>), dim3(nblk + hub_blocks + giant_blocks), dim3(VGL_BLOCK), 0, c->stream, g->nrows,
                           g->row_begin, g->out.rowptr, g->out.adj, (const float *)contrib, epi, hub_blocks, (const int32_t *)g->out.hub_rows,
                           (const int32_t *)(g->out.hub_rows + g->out.nhubs), (double *)nullptr, (const int32_t *)g->out.pull_blk_row,
                           (const int32_t *)nullptr, 0, (float *)nullptr, giant_blocks, (const int32_t *)g->out.giant_rows,
                           (const int32_t *)(g->out.giant_rows + g->out.ngiants));
    }
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" {

int vgl_hip_indegree_noloops_add(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t *d_indeg)
{
    if (!c || !g || !d_indeg) VGL_FAIL("indegree_noloops_add: null argument");
    if (g->out.ntiles == 0) return 0;
    hipLaunchKernelGGL(vgl_k_indeg_noloops, dim3((unsigned)g->out.ntiles), dim3(VGL_BLOCK), 0, c->stream, g->out.rowptr, g->out.adj,
                       g->out.tile_row, g->out.edges, g->row_begin, d_indeg);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_sum_over_edges_f32(vgl_hip_ctx *c, vgl_hip_graph *g, const float *d_values, float sum_bound, float *d_sums)
{
    if (!c || !g || !d_values || !d_sums) VGL_FAIL("sum_over_edges: null argument");
    if (d_values == d_sums) VGL_FAIL("sum_over_edges: the sums cannot be written over the values they are formed from");
    if (!(sum_bound > 0.0f) || !(sum_bound <= 0x1p60f)) VGL_FAIL("sum_over_edges: the bound of a per-vertex sum must be positive (and at most 2^60)");
    int bound_exp = 0;
    (void)frexpf(sum_bound, &bound_exp);                    // sum_bound = m * 2^bound_exp, 0.5 <= m < 1: sums < 2^bound_exp, one spare bit on top
    bound_exp += 1;
    if (!g->blk_pr) VGL_TRY(vgl_blocked_plan_build(c, g->out, g->nrows, g->row_begin, g->V, 0, 1, nullptr, VGL_BLK_BITS - 1, &g->blk_pr));
    const vgl_sum_blk_op op{d_values, d_sums, g->row_begin, bound_exp};
    return vgl_blocked_pass<vgl_sum_blk_op, false, true>(c, g->blk_pr, op, "sum_blk_gather", "sum_blk_accumulate");
}

int vgl_hip_pr_setup(vgl_hip_ctx *c, int32_t V, const int32_t *d_indeg, float *d_ranks, float *d_rdeg)
{
    if (!c || !d_indeg || !d_ranks || !d_rdeg) VGL_FAIL("pr_setup: null argument");
    hipLaunchKernelGGL(vgl_k_pr_setup, dim3(vgl_grid3(V, 8192)), dim3(VGL_BLOCK), 0, c->stream, V, d_indeg, d_ranks, d_rdeg);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_pr_iteration_owned(vgl_hip_ctx *c, vgl_hip_graph *g, const int32_t *d_indeg, const float *d_rdeg, float *d_ranks,
                               float *d_contrib_scratch)
{
    if (!c || !g || !d_indeg || !d_rdeg || !d_ranks || !d_contrib_scratch) VGL_FAIL("pr_iteration_owned: null argument");
    // in place is safe: the pull kernel reads only contrib/dangling (both produced from the old ranks) and writes owned rows
    return vgl_pr_iteration(c, g, d_indeg, d_rdeg, d_ranks, d_contrib_scratch, d_ranks, VGL_HIP_PR_AUTO);
}

int vgl_hip_pr_prepare(vgl_hip_ctx *c, vgl_hip_graph *g, int mode, int *resolved_mode)
{
    if (!c || !g) VGL_FAIL("pr_prepare: null argument");
    if (mode < VGL_HIP_PR_EXACT_ORDER || mode > VGL_HIP_PR_AUTO) VGL_FAIL("pr_prepare: unknown mode");
    VGL_TRY(vgl_pr_mode_auto(c, g, &mode));
    if (mode == VGL_HIP_PR_BLOCKED) {
        if (!g->blk_pr) VGL_TRY(vgl_blocked_plan_build(c, g->out, g->nrows, g->row_begin, g->V, 0, 1, nullptr, VGL_BLK_BITS - 1, &g->blk_pr));
    } else VGL_TRY(vgl_pull_find_hubs(c, g, g->out));
    VGL_HIP_TRY(hipStreamSynchronize(c->stream));
    if (resolved_mode) *resolved_mode = mode;
    return 0;
}

int vgl_hip_pr_run(vgl_hip_ctx *c, vgl_hip_graph *g, const int32_t *d_indeg_noloops, int iterations, float *d_ranks,
                   vgl_hip_pr_stats *stats)
{
    return vgl_hip_pr_run_mode(c, g, d_indeg_noloops, iterations, VGL_HIP_PR_AUTO, d_ranks, stats);
}

int vgl_hip_pr_run_mode(vgl_hip_ctx *c, vgl_hip_graph *g, const int32_t *d_indeg_noloops, int iterations, int mode, float *d_ranks,
                        vgl_hip_pr_stats *stats)
{
    if (!c || !g || !d_ranks) VGL_FAIL("pr_run: null argument");
    if (mode < VGL_HIP_PR_EXACT_ORDER || mode > VGL_HIP_PR_AUTO) VGL_FAIL("pr_run: unknown mode");
    if (g->row_begin != 0 || g->row_end != g->V) VGL_FAIL("pr_run: graph handle must own all rows (use the step API for shards)");
    if (iterations < 0) VGL_FAIL("pr_run: negative iteration count");
    const int32_t V = g->V;
    const int32_t *indeg = d_indeg_noloops;
    if (!indeg) {
        if (g->in.rowptr) {
            hipLaunchKernelGGL(vgl_k_indeg_from_in_rows, dim3(vgl_grid3(V, 8192)), dim3(VGL_BLOCK), 0, c->stream, V, g->in.rowptr, g->iscratch);
            if (g->out.ntiles > 0)
                hipLaunchKernelGGL(vgl_k_indeg_sub_loops, dim3((unsigned)g->out.ntiles), dim3(VGL_BLOCK), 0, c->stream, g->out.rowptr, g->out.adj,
                                   g->out.tile_row, g->out.edges, g->row_begin, g->iscratch);
            VGL_HIP_TRY(hipGetLastError());
        } else {
            VGL_HIP_TRY(hipMemsetAsync(g->iscratch, 0, sizeof(int32_t) * (size_t)V, c->stream));
            VGL_TRY(vgl_hip_indegree_noloops_add(c, g, g->iscratch));
        }
        indeg = g->iscratch;
    }
    float *rdeg = g->fscratch2, *contrib = g->fscratch;
    VGL_TRY(vgl_hip_pr_setup(c, V, indeg, d_ranks, rdeg));
    vgl_hip_pr_stats st = {0, 0.0, 0};
    for (int it = 0; it < iterations; it++)
        VGL_TRY(vgl_pr_iteration(c, g, indeg, rdeg, d_ranks, contrib, d_ranks, mode));
    st.iterations = iterations;
    {   // reduce_ranks_sum (pr.hpp:130-134): deterministic two-stage f64 sum of the final ranks
        const int nb = (int)vgl_grid3(V, 1024);
        VGL_TRY(vgl_ensure_partials(c, (size_t)nb + 1));
        hipLaunchKernelGGL(vgl_k_pr_sum_partial, dim3(nb), dim3(VGL_BLOCK), 0, c->stream, V, d_ranks, c->d_partials);
        std::vector<double> h((size_t)nb);
        VGL_TRY(vgl_hip_memcpy_d2h(c, h.data(), c->d_partials, sizeof(double) * (size_t)nb));
        for (double x : h) st.ranks_sum += x;
    }
    st.algorithmic_bytes = (8 * g->out.edges + 28 * (int64_t)V) * iterations;
    if (stats) *stats = st;
    return 0;
}

}  // extern "C"
