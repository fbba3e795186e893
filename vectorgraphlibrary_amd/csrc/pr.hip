// pr.hip -- PageRank (PR::vgl_page_rank, algorithms/pr/pr.hpp:7-149; f32, d = 0.85, fixed iteration count).
//
// Per iteration (pr.hpp:83-136):
//   prepare : old = rank; contrib[v] = old[v] * rdeg[v]  (the per-edge product dst_rank * reversed_dst_links_num of
//             pr.hpp:111-116, hoisted: the f32 product is the same value for every edge that reads it);
//             dangling = sum over v with (indeg-loops)==0 of old[v] / V  -- accumulated in f64 in a fixed tree and rounded
//             to f32 once (the reference's f32 OpenMP reduction is thread-order dependent)
//   pull    : rank[src] = k + d * (sum_{src->dst, dst != src} contrib[dst] + dangling), the sum taken IN ADJACENCY ORDER
//             in f32 exactly like the reference's `+=` chain, so the result is bit-identical to seq_page_rank's
//             evaluation order (seq_pr.hpp:81-96).  Workgroup = 256 consecutive rows, one thread per row; the rows' edges are
//             staged through LDS in tiles of 2048 (coalesced adjacency read + contrib gather by all threads), then every
//             thread adds its own row's slice sequentially from LDS.
// Algorithmic bytes per iteration: 8*E + 28*V.
#include "vgl_hip_internal.h"
#include <queue>

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

// Rows with at least VGL_PR_HUB_DEGREE edges are "hubs": their strictly sequential f32 sum (deg dependent adds) would stall a
// whole 256-row workgroup (a 7*10^5-edge RMAT hub took 1.7 s per iteration that way).  They are listed once per graph, largest
// first, and summed by the first `hub_blocks` workgroups of the pull kernel (one workgroup per CU, i.e. one wavefront per SIMD,
// raised issue priority) while the remaining workgroups pull the ordinary rows on the same CUs (a separate kernel on a side
// stream overlapped worse: 5.0 vs 4.2 ms per RMAT-24 iteration): a wavefront takes its hubs
// from a precomputed longest-first schedule, gathers 512 values per batch (the next batch's gathers and the adjacency of the one after in flight; 1024-value batches cost
// 112 VGPRs and the ordinary rows' occupancy), parks them in LDS and folds them IN
// ADJACENCY ORDER with one dependent v_add per value (all lanes compute the same chain; broadcast LDS reads).  The critical
// path of an iteration is the chain of the largest hub (~10 cycles per edge measured), not the sum over hubs.
constexpr int VGL_PR_HUB_DEGREE = 512;
constexpr int VGL_PR_HUB_BATCH = 512;
constexpr int VGL_PR_HUB_BLOCKS = 256;      // one per CU

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_pr_find_hubs(int32_t nrows, const int64_t *rowptr, int32_t *hub_rows, int32_t *hub_deg,
                                                                int32_t *hub_count)
{
    for (int32_t r = blockIdx.x * VGL_BLOCK + threadIdx.x; r < nrows; r += gridDim.x * VGL_BLOCK) {
        const int64_t dg = rowptr[r + 1] - rowptr[r];
        if (dg >= VGL_PR_HUB_DEGREE) {                       // few thousand rows, once per graph
            const int slot = atomicAdd(hub_count, 1);
            hub_rows[slot] = r;
            hub_deg[slot] = (int32_t)min(dg, (int64_t)INT32_MAX);
        }
    }
}

__device__ __forceinline__ void vgl_pr_hub_waves(float *s_buf, const int32_t *hub_rows, const int32_t *hub_off, int32_t row_base,
                                                 const int64_t *rowptr, const int32_t *adj, const float *contrib, const float *dangling_ptr,
                                                 float k, float d, float *ranks_out)
{
    __builtin_amdgcn_s_setprio(3);
    const int lane = vgl_lane();
    float *cur = s_buf + vgl_wave() * VGL_PR_HUB_BATCH;                  // this wavefront's batch (consumed before the next is parked)
    constexpr int U = VGL_PR_HUB_BATCH / 64;
    constexpr int B = VGL_PR_HUB_BATCH;
    // No lane-dependent control flow anywhere below: the LDS hand-over relies on the wavefront staying converged (a ticket fetched
    // under `if (lane == 0)` let the compiler unswitch the loop on the lane id and the lanes ran apart).  Each wavefront owns a
    // precomputed list of hubs (longest-processing-time schedule built on the host, vgl_pr_find_hubs).
    const int32_t w = __builtin_amdgcn_readfirstlane((int32_t)blockIdx.x * VGL_WAVES + vgl_wave());    // scalar: loops are uniform
    const int32_t h_end = hub_off[w + 1];
    for (int32_t h = hub_off[w]; h < h_end; h++) {
        const int32_t r = hub_rows[h];
        const int64_t b = rowptr[r];
        const uint32_t n = (uint32_t)(rowptr[r + 1] - b);               // a row has fewer than 2^31 edges
        const int32_t *adj_h = adj + b;                                  // scalar base + 32-bit lane offsets
        const int32_t self = row_base + r;
        float acc = 0.0f;
        float val[U];
        int32_t dst[U];
        auto load_adj = [&](uint32_t base) {
#pragma unroll
            for (int u = 0; u < U; u++) { const uint32_t q = base + u * 64 + lane; dst[u] = q < n ? adj_h[q] : self; }
        };
        auto gather = [&]() {
#pragma unroll
            for (int u = 0; u < U; u++) val[u] = (dst[u] != self) ? contrib[(uint32_t)dst[u]] : 0.0f;   // x + 0.0f == x: exact no-op
        };
        // three batches in flight: fold(i) from LDS | gathers of batch i+1 | adjacency of batch i+2
        load_adj(0);
        gather();
        if (B < n) load_adj(B);
        for (uint32_t base = 0; base < n; base += B) {
#pragma unroll
            for (int u = 0; u < U; u++) cur[u * 64 + lane] = val[u];
            if (base + B < n) {
                gather();
                if (base + 2 * B < n) load_adj(base + 2 * B);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int n4 = (int)min((uint32_t)(B / 4), (n - base + 3) / 4);   // the tail of the batch is zero-filled
            const float4 *p = reinterpret_cast<const float4 *>(cur);
#pragma unroll 8
            for (int i = 0; i < n4; i++) {
                const float4 v = p[i];                                  // same address in every lane: LDS broadcast
                acc = __fadd_rn(acc, v.x);
                acc = __fadd_rn(acc, v.y);
                acc = __fadd_rn(acc, v.z);
                acc = __fadd_rn(acc, v.w);
            }
        }
        ranks_out[self] = __fadd_rn(k, __fmul_rn(d, __fadd_rn(acc, *dangling_ptr)));     // all lanes: same value, same address
    }
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_pr_pull(int32_t nrows, int32_t row_base, const int64_t *rowptr, const int32_t *adj,
                                                           const float *contrib, const float *dangling_ptr, float k, float d,
                                                           float *ranks_out, int hub_blocks, const int32_t *hub_rows, const int32_t *hub_off)
{
    __shared__ float s_buf[2 * VGL_TILE];               // pull: values | destinations; hub wavefronts: 4 x 512 values
    __shared__ int64_t s_jump;
    if ((int)blockIdx.x < hub_blocks) {                 // dispatched first: one workgroup per CU runs the hub schedule
        vgl_pr_hub_waves(s_buf, hub_rows, hub_off, row_base, rowptr, adj, contrib, dangling_ptr, k, d, ranks_out);
        return;
    }
    float *s_val = s_buf;
    int32_t *s_dst = reinterpret_cast<int32_t *>(s_buf + VGL_TILE);
    const int32_t blk = (int32_t)blockIdx.x - hub_blocks;
    const int32_t r = blk * VGL_BLOCK + threadIdx.x;
    const int32_t r_lo = blk * VGL_BLOCK;
    const int32_t r_hi = min(nrows, r_lo + VGL_BLOCK);
    const int64_t E0 = rowptr[r_lo], E1 = rowptr[r_hi];
    int64_t seg_b = 0, seg_e = 0;
    if (r < nrows) { seg_b = rowptr[r]; seg_e = rowptr[r + 1]; }
    const bool hub = (seg_e - seg_b) >= VGL_PR_HUB_DEGREE;      // summed by the hub wavefronts
    const int32_t self = row_base + r;
    float acc = 0.0f;
    int64_t base = E0;
    while (base < E1) {
        // a tile that starts inside a hub's edge range is skipped wholesale: jump to the end of that range
        if (threadIdx.x == 0) s_jump = -1;
        __syncthreads();
        if (hub && seg_b <= base && base < seg_e) s_jump = seg_e;      // at most one row contains `base`
        __syncthreads();
        const int64_t jump = s_jump;
        __syncthreads();                               // everyone has read s_jump before thread 0 resets it
        if (jump >= 0) { base = jump; continue; }
        const int n = (int)min((int64_t)VGL_TILE, E1 - base);
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) {
            const int i = threadIdx.x + j * VGL_BLOCK;
            if (i < n) {
                const int32_t dst = adj[base + i];
                s_dst[i] = dst;
                s_val[i] = contrib[dst];
            }
        }
        __syncthreads();
        if (!hub) {
            const int lo = (int)(max(seg_b, base) - base);
            const int hi = (int)(min(seg_e, base + n) - base);
            for (int i = lo; i < hi; i++)
                if (s_dst[i] != self) acc = __fadd_rn(acc, s_val[i]);     // if(src_id != dst_id) rank += ... (pr.hpp:115-116)
        }
        base += n;
    }
    if (r < nrows && !hub)
        ranks_out[self] = __fadd_rn(k, __fmul_rn(d, __fadd_rn(acc, *dangling_ptr)));   // k + d * (rank + dangling) (pr.hpp:121)
}

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

static inline unsigned vgl_grid3(int64_t n, int64_t cap) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cap, vgl_ceil_div(n, VGL_BLOCK))); }

static int vgl_pr_find_hubs(vgl_hip_ctx *c, vgl_hip_graph *g)
{
    if (g->pr_hub_rows) return 0;
    int32_t *d_rows = nullptr, *d_deg = nullptr, *d_count = nullptr;
    const size_t cap = (size_t)std::max<int32_t>(g->nrows, 1);
    VGL_HIP_TRY(hipMalloc((void **)&d_rows, sizeof(int32_t) * cap));
    VGL_HIP_TRY(hipMalloc((void **)&d_deg, sizeof(int32_t) * cap));
    VGL_HIP_TRY(hipMalloc((void **)&d_count, sizeof(int32_t)));
    VGL_HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(int32_t), c->stream));
    hipLaunchKernelGGL(vgl_k_pr_find_hubs, dim3(vgl_grid3(g->nrows, 4096)), dim3(VGL_BLOCK), 0, c->stream, g->nrows, g->out.rowptr, d_rows,
                       d_deg, d_count);
    VGL_HIP_TRY(hipGetLastError());
    VGL_TRY(vgl_hip_memcpy_d2h(c, &g->pr_nhubs, d_count, sizeof(int32_t)));
    const size_t n = (size_t)g->pr_nhubs;
    g->pr_hub_blocks = n ? (int)std::min<int64_t>(VGL_PR_HUB_BLOCKS, vgl_ceil_div((int64_t)n, VGL_WAVES)) : 0;
    const int W = g->pr_hub_blocks * VGL_WAVES;
    // device layout: [n hub rows grouped by wavefront][W+1 offsets]
    VGL_HIP_TRY(hipMalloc((void **)&g->pr_hub_rows, sizeof(int32_t) * (n + (size_t)W + 1)));
    if (n > 0) {
        std::vector<int32_t> rows(n), deg(n), order(n);
        VGL_TRY(vgl_hip_memcpy_d2h(c, rows.data(), d_rows, sizeof(int32_t) * n));
        VGL_TRY(vgl_hip_memcpy_d2h(c, deg.data(), d_deg, sizeof(int32_t) * n));
        for (size_t i = 0; i < n; i++) order[i] = (int32_t)i;
        std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return deg[x] != deg[y] ? deg[x] > deg[y] : rows[x] < rows[y]; });
        // longest-processing-time list scheduling: the next-largest hub goes to the least loaded wavefront, so the critical path is
        // max(largest hub, total / W) edges; a fixed per-hub cost stands for the un-overlapped first batch
        typedef std::pair<int64_t, int> slot;
        std::priority_queue<slot, std::vector<slot>, std::greater<slot>> heap;
        for (int wv = 0; wv < W; wv++) heap.push(slot(0, wv));
        std::vector<std::vector<int32_t>> lists((size_t)W);
        for (size_t i = 0; i < n; i++) {
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
        VGL_TRY(vgl_hip_memcpy_h2d(c, g->pr_hub_rows, packed.data(), sizeof(int32_t) * packed.size()));
    }
    VGL_HIP_TRY(hipFree(d_rows));
    VGL_HIP_TRY(hipFree(d_deg));
    VGL_HIP_TRY(hipFree(d_count));
    return 0;
}

static int vgl_pr_iteration(vgl_hip_ctx *c, vgl_hip_graph *g, const int32_t *indeg, const float *rdeg, float *ranks, float *contrib,
                            float *ranks_out)
{
    const int32_t V = g->V;
    const float d = 0.85f;
    const float k = (float)((1.0 - (double)d) / (double)((float)V));       // pr.hpp:37-38
    const int npart = (int)vgl_grid3(V, 1024);
    const unsigned nblk = (unsigned)vgl_ceil_div(g->nrows, VGL_BLOCK);
    VGL_TRY(vgl_pr_find_hubs(c, g));
    VGL_TRY(vgl_ensure_partials(c, (size_t)npart + 8));
    float *dangling = reinterpret_cast<float *>(c->d_partials + npart);    // one slot after the prepare partials
    hipLaunchKernelGGL(vgl_k_pr_prepare, dim3(npart), dim3(VGL_BLOCK), 0, c->stream, V, indeg, rdeg, ranks, contrib, c->d_partials);
    hipLaunchKernelGGL(vgl_k_pr_dangling, dim3(1), dim3(VGL_BLOCK), 0, c->stream, npart, c->d_partials, dangling);
    const int hub_blocks = g->pr_hub_blocks;
    {
        vgl_timed_launch tl(c, "pr_pull");
        hipLaunchKernelGGL(vgl_k_pr_pull, dim3(nblk + hub_blocks), dim3(VGL_BLOCK), 0, c->stream, g->nrows, g->row_begin, g->out.rowptr,
                           g->out.adj, contrib, dangling, k, d, ranks_out, hub_blocks, g->pr_hub_rows, g->pr_hub_rows + g->pr_nhubs);
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
    return vgl_pr_iteration(c, g, d_indeg, d_rdeg, d_ranks, d_contrib_scratch, d_ranks);
}

int vgl_hip_pr_run(vgl_hip_ctx *c, vgl_hip_graph *g, const int32_t *d_indeg_noloops, int iterations, float *d_ranks,
                   vgl_hip_pr_stats *stats)
{
    if (!c || !g || !d_ranks) VGL_FAIL("pr_run: null argument");
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
        VGL_TRY(vgl_pr_iteration(c, g, indeg, rdeg, d_ranks, contrib, d_ranks));
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
