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

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_pr_pull(int32_t nrows, int32_t row_base, const int64_t *rowptr, const int32_t *adj,
                                                           const float *contrib, const float *dangling_ptr, float k, float d,
                                                           float *ranks_out, double *sum_partials)
{
    __shared__ float s_val[VGL_TILE];
    __shared__ int32_t s_dst[VGL_TILE];
    __shared__ double s_red[VGL_WAVES];
    const int32_t r = blockIdx.x * VGL_BLOCK + threadIdx.x;
    const int32_t r_lo = blockIdx.x * VGL_BLOCK;
    const int32_t r_hi = min(nrows, r_lo + VGL_BLOCK);
    const int64_t E0 = rowptr[r_lo], E1 = rowptr[r_hi];
    int64_t seg_b = 0, seg_e = 0;
    if (r < nrows) { seg_b = rowptr[r]; seg_e = rowptr[r + 1]; }
    const int32_t self = row_base + r;
    float acc = 0.0f;
    for (int64_t base = E0; base < E1; base += VGL_TILE) {
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
        const int lo = (int)(max(seg_b, base) - base);
        const int hi = (int)(min(seg_e, base + n) - base);
        for (int i = lo; i < hi; i++)
            if (s_dst[i] != self) acc = __fadd_rn(acc, s_val[i]);     // if(src_id != dst_id) rank += ... (pr.hpp:115-116)
        __syncthreads();
    }
    double mine = 0.0;
    if (r < nrows) {
        const float res = __fadd_rn(k, __fmul_rn(d, __fadd_rn(acc, *dangling_ptr)));   // k + d * (rank + dangling) (pr.hpp:121)
        ranks_out[self] = res;
        mine = (double)res;
    }
    mine = vgl_block_reduce_add(mine, s_red);
    if (threadIdx.x == 0 && sum_partials) sum_partials[blockIdx.x] = mine;
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

static int vgl_pr_iteration(vgl_hip_ctx *c, vgl_hip_graph *g, const int32_t *indeg, const float *rdeg, float *ranks, float *contrib,
                            float *ranks_out, bool want_sum)
{
    const int32_t V = g->V;
    const float d = 0.85f;
    const float k = (float)((1.0 - (double)d) / (double)((float)V));       // pr.hpp:37-38
    const int npart = (int)vgl_grid3(V, 1024);
    const unsigned nblk = (unsigned)vgl_ceil_div(g->nrows, VGL_BLOCK);
    VGL_TRY(vgl_ensure_partials(c, (size_t)npart + nblk + 8));
    float *dangling = reinterpret_cast<float *>(c->d_partials + npart);    // one slot after the prepare partials
    double *sum_partials = c->d_partials + npart + 2;
    hipLaunchKernelGGL(vgl_k_pr_prepare, dim3(npart), dim3(VGL_BLOCK), 0, c->stream, V, indeg, rdeg, ranks, contrib, c->d_partials);
    hipLaunchKernelGGL(vgl_k_pr_dangling, dim3(1), dim3(VGL_BLOCK), 0, c->stream, npart, c->d_partials, dangling);
    {
        vgl_timed_launch tl(c, "pr_pull");
        hipLaunchKernelGGL(vgl_k_pr_pull, dim3(nblk), dim3(VGL_BLOCK), 0, c->stream, g->nrows, g->row_begin, g->out.rowptr, g->out.adj,
                           contrib, dangling, k, d, ranks_out, want_sum ? sum_partials : (double *)nullptr);
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
    return vgl_pr_iteration(c, g, d_indeg, d_rdeg, d_ranks, d_contrib_scratch, d_ranks, false);
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
        VGL_TRY(vgl_pr_iteration(c, g, indeg, rdeg, d_ranks, contrib, d_ranks, it == iterations - 1));
    st.iterations = iterations;
    if (iterations > 0) {
        const int npart = (int)vgl_grid3(V, 1024);
        const unsigned nblk = (unsigned)vgl_ceil_div(g->nrows, VGL_BLOCK);
        std::vector<double> h(nblk);
        VGL_TRY(vgl_hip_memcpy_d2h(c, h.data(), c->d_partials + npart + 2, sizeof(double) * nblk));
        for (double x : h) st.ranks_sum += x;
    } else {
        VGL_TRY(vgl_hip_ctx_sync(c));
    }
    st.algorithmic_bytes = (8 * g->out.edges + 28 * (int64_t)V) * iterations;
    if (stats) *stats = st;
    return 0;
}

}  // extern "C"
