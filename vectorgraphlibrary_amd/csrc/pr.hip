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
// whole 256-row workgroup (a 7*10^5-edge RMAT hub took 1.7 s per iteration that way), so they are listed once per graph and
// summed by vgl_k_pr_pull_hubs, one wavefront per hub: 64 coalesced loads + gathers, then the 64 values are folded into the
// running sum in lane order (same order as the reference's loop), next chunk's loads already in flight.
constexpr int VGL_PR_HUB_DEGREE = 512;

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_pr_find_hubs(int32_t nrows, const int64_t *rowptr, int32_t *hub_rows, int32_t *hub_count)
{
    for (int32_t r = blockIdx.x * VGL_BLOCK + threadIdx.x; r < nrows; r += gridDim.x * VGL_BLOCK)
        if (rowptr[r + 1] - rowptr[r] >= VGL_PR_HUB_DEGREE) hub_rows[atomicAdd(hub_count, 1)] = r;    // few thousand rows, once per graph
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_pr_pull_hubs(int32_t nhubs, const int32_t *hub_rows, int32_t row_base, const int64_t *rowptr,
                                                                const int32_t *adj, const float *contrib, const float *dangling_ptr,
                                                                float k, float d, float *ranks_out)
{
    const int lane = vgl_lane();
    constexpr int U = 8;                                    // 512 values per batch, next batch's loads in flight during the fold
    for (int32_t h = blockIdx.x * VGL_WAVES + vgl_wave(); h < nhubs; h += gridDim.x * VGL_WAVES) {
        const int32_t r = hub_rows[h];
        const int64_t b = rowptr[r], e = rowptr[r + 1];
        const int32_t self = row_base + r;
        float acc = 0.0f;
        float val[U], nxt[U];
        auto load = [&](int64_t base, float *out) {
            int32_t dst[U];
#pragma unroll
            for (int u = 0; u < U; u++) { const int64_t q = base + u * 64 + lane; dst[u] = q < e ? adj[q] : self; }
#pragma unroll
            for (int u = 0; u < U; u++) out[u] = (dst[u] != self) ? contrib[dst[u]] : 0.0f;   // x + 0.0f == x: exact no-op
        };
        load(b, val);
        for (int64_t base = b; base < e; base += 64 * U) {
            if (base + 64 * U < e) load(base + 64 * U, nxt);
#pragma unroll
            for (int u = 0; u < U; u++) {
#pragma unroll
                for (int l = 0; l < 64; l++) acc = __fadd_rn(acc, __shfl(val[u], l));       // lane order == adjacency order
            }
#pragma unroll
            for (int u = 0; u < U; u++) val[u] = nxt[u];
        }
        if (lane == 0) ranks_out[self] = __fadd_rn(k, __fmul_rn(d, __fadd_rn(acc, *dangling_ptr)));
    }
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_pr_pull(int32_t nrows, int32_t row_base, const int64_t *rowptr, const int32_t *adj,
                                                           const float *contrib, const float *dangling_ptr, float k, float d,
                                                           float *ranks_out)
{
    __shared__ float s_val[VGL_TILE];
    __shared__ int32_t s_dst[VGL_TILE];
    __shared__ int64_t s_jump;
    const int32_t r = blockIdx.x * VGL_BLOCK + threadIdx.x;
    const int32_t r_lo = blockIdx.x * VGL_BLOCK;
    const int32_t r_hi = min(nrows, r_lo + VGL_BLOCK);
    const int64_t E0 = rowptr[r_lo], E1 = rowptr[r_hi];
    int64_t seg_b = 0, seg_e = 0;
    if (r < nrows) { seg_b = rowptr[r]; seg_e = rowptr[r + 1]; }
    const bool hub = (seg_e - seg_b) >= VGL_PR_HUB_DEGREE;      // summed by vgl_k_pr_pull_hubs
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

static inline unsigned vgl_grid3(int64_t n, int64_t cap) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cap, vgl_ceil_div(n, VGL_BLOCK))); }

static int vgl_pr_find_hubs(vgl_hip_ctx *c, vgl_hip_graph *g)
{
    if (g->pr_hub_rows) return 0;
    VGL_HIP_TRY(hipMalloc((void **)&g->pr_hub_rows, sizeof(int32_t) * ((size_t)g->nrows + 1)));
    int32_t *d_count = g->pr_hub_rows + g->nrows;
    VGL_HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(int32_t), c->stream));
    hipLaunchKernelGGL(vgl_k_pr_find_hubs, dim3(vgl_grid3(g->nrows, 4096)), dim3(VGL_BLOCK), 0, c->stream, g->nrows, g->out.rowptr, g->pr_hub_rows, d_count);
    VGL_HIP_TRY(hipGetLastError());
    VGL_TRY(vgl_hip_memcpy_d2h(c, &g->pr_nhubs, d_count, sizeof(int32_t)));
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
    if (g->pr_nhubs > 0) {
        vgl_timed_launch tl(c, "pr_pull_hubs");
        hipLaunchKernelGGL(vgl_k_pr_pull_hubs, dim3(vgl_grid3((int64_t)g->pr_nhubs * 64, 2048)), dim3(VGL_BLOCK), 0, c->stream, g->pr_nhubs,
                           g->pr_hub_rows, g->row_begin, g->out.rowptr, g->out.adj, contrib, dangling, k, d, ranks_out);
    }
    {
        vgl_timed_launch tl(c, "pr_pull");
        hipLaunchKernelGGL(vgl_k_pr_pull, dim3(nblk), dim3(VGL_BLOCK), 0, c->stream, g->nrows, g->row_begin, g->out.rowptr, g->out.adj,
                           contrib, dangling, k, d, ranks_out);
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
        VGL_HIP_TRY(hipMemsetAsync(g->iscratch, 0, sizeof(int32_t) * (size_t)V, c->stream));
        VGL_TRY(vgl_hip_indegree_noloops_add(c, g, g->iscratch));
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
