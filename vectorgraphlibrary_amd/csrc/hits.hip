// hits.hip -- HITS (HITS::vgl_hits, algorithms/hits/hits.hpp:5-100; f64 like apps/hits/hits.cpp:13).
//
// Per step (hits.hpp:32-91):
//   auth[v] = sum over the INCOMING neighbours u of hub[u]        (gather direction; pre-op zeroes, edge-op adds)
//   auth   /= sqrt(sum_v auth[v]^2)                               (reduce<double> REDUCE_SUM + compute)
//   hub[v]  = sum over the OUTGOING neighbours w of auth[w]       (scatter direction)
//   hub    /= sqrt(sum_v hub[v]^2)
// The per-vertex sums are vgl_k_pull_sum<double> (vgl_pull.h): f64 `+=` chains in adjacency order, the same order as the
// reference's sequential checker (hits.hpp:117-160), for rows below 512 edges; longer rows are summed in 4096-entry chunks (per-lane
// partial sums, fixed butterfly, chunk sums added in order): deterministic, within ~1e-15 of the sequential order.  The sum of squares
// is accumulated per workgroup by the same launch and folded in a fixed order (the reference's OpenMP reduction order is
// unspecified), so a whole run is deterministic and agrees with seq_hits to ~1e-15 relative; the tests use 1e-12.
// Nothing is read back by the host between steps.  Algorithmic bytes per step: 2 * (12 B/edge + 8 + 16 + 16 B/vertex).
#include "vgl_pull.h"

struct vgl_hits_epilogue {
    double *out;
    __device__ __forceinline__ void operator()(int32_t v, double acc) const { out[v] = acc; }
};

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_hits_init(int32_t V, double *auth, double *hub)
{
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK) { auth[v] = 1.0; hub[v] = 1.0; }
}

// norm = sqrt(sum of the per-workgroup partial sums of squares), one workgroup, fixed order
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_hits_norm(int nparts, const double *partials, double *norm_out)
{
    __shared__ double s[VGL_WAVES];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += VGL_BLOCK) acc += partials[i];
    acc = vgl_block_reduce_add(acc, s);
    if (threadIdx.x == 0) *norm_out = sqrt(acc);
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_hits_scale(int32_t V, double *x, const double *norm)
{
    const double nrm = *norm;
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK) x[v] = __ddiv_rn(x[v], nrm);   // _auth[src_id] /= norm
}

static int vgl_hits_half_step(vgl_hip_ctx *c, vgl_hip_graph *g, vgl_dir_csr &dir, const double *x, double *out)
{
    VGL_TRY(vgl_pull_find_hubs(c, g, dir));
    const unsigned nblk = (unsigned)dir.pull_nblk;
    const int nfinish = (int)vgl_ceil_div((int64_t)dir.n_hub_list, VGL_BLOCK);
    const int hub_blocks = (int)vgl_ceil_div((int64_t)dir.n_hub_chunks, VGL_WAVES);      // unordered hub sums: one chunk per wavefront, as many workgroups as it takes
    const int nparts = (int)nblk + hub_blocks + nfinish;
    VGL_TRY(vgl_ensure_partials(c, (size_t)nparts + 2));
    double *norm = c->d_partials + nparts;
    {
        vgl_timed_launch tl(c, "hits_pull");
        const vgl_hits_epilogue epi{out};
        hipLaunchKernelGGL((vgl_k_pull_sum<double, false, true, vgl_hits_epilogue, false>), dim3(nblk + hub_blocks), dim3(VGL_BLOCK), 0, c->stream,
                           g->nrows, g->row_begin, dir.rowptr, dir.adj, x, epi, hub_blocks, (const int32_t *)dir.hub_rows,
                           (const int32_t *)(dir.hub_rows + dir.nhubs), c->d_partials, (const int32_t *)dir.pull_blk_row,
                           (const int32_t *)dir.hub_chunks, dir.n_hub_chunks, dir.hub_chunk_sums);
        if (nfinish > 0)                      // the hubs: chunk sums in order, results, their share of the sum of squares
            hipLaunchKernelGGL((vgl_k_pull_hub_finish<double, true, vgl_hits_epilogue>), dim3(nfinish), dim3(VGL_BLOCK), 0, c->stream, dir.n_hub_list,
                               (const int32_t *)(dir.hub_chunks + 2 * (size_t)dir.n_hub_chunks), g->row_begin, (const double *)dir.hub_chunk_sums, epi,
                               c->d_partials + nblk + hub_blocks);
    }
    hipLaunchKernelGGL(vgl_k_hits_norm, dim3(1), dim3(VGL_BLOCK), 0, c->stream, nparts, c->d_partials, norm);
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, vgl_ceil_div(g->V, VGL_BLOCK)));
    hipLaunchKernelGGL(vgl_k_hits_scale, dim3(grid), dim3(VGL_BLOCK), 0, c->stream, g->V, out, norm);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

// ---- the halves of a step over the rows one rank owns (sharded.hip: vgl_hip_hits_run_sharded) ----
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_hits_sumsq(int nparts, const double *partials, double *sum_out)
{
    __shared__ double s[VGL_WAVES];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += VGL_BLOCK) acc += partials[i];
    acc = vgl_block_reduce_add(acc, s);
    if (threadIdx.x == 0) *sum_out = acc;
}
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_hits_scale_rows(int32_t lo, int32_t hi, double *x, const double *sumsq)
{
    const double nrm = sqrt(*sumsq);
    for (int32_t v = lo + blockIdx.x * VGL_BLOCK + threadIdx.x; v < hi; v += gridDim.x * VGL_BLOCK) x[v] = __ddiv_rn(x[v], nrm);
}
// out[v] = sum over the neighbours (incoming lists when `incoming`) of x, for the owned rows; *d_sumsq = this rank's share of sum_v out[v]^2
int vgl_hits_pull_owned(vgl_hip_ctx *c, vgl_hip_graph *g, bool incoming, const double *x, double *out, double *d_sumsq)
{
    vgl_dir_csr &dir = incoming ? g->in : g->out;
    VGL_TRY(vgl_pull_find_hubs(c, g, dir));
    const unsigned nblk = (unsigned)dir.pull_nblk;
    const int nfinish = (int)vgl_ceil_div((int64_t)dir.n_hub_list, VGL_BLOCK);
    const int hub_blocks = (int)vgl_ceil_div((int64_t)dir.n_hub_chunks, VGL_WAVES);
    const int nparts = (int)nblk + hub_blocks + nfinish;
    VGL_TRY(vgl_ensure_partials(c, (size_t)nparts + 2));
    {
        vgl_timed_launch tl(c, "hits_pull");
        const vgl_hits_epilogue epi{out};
        if (nblk + hub_blocks > 0)
            hipLaunchKernelGGL((vgl_k_pull_sum<double, false, true, vgl_hits_epilogue, false>), dim3(nblk + hub_blocks), dim3(VGL_BLOCK), 0, c->stream,
                               g->nrows, g->row_begin, dir.rowptr, dir.adj, x, epi, hub_blocks, (const int32_t *)dir.hub_rows,
                               (const int32_t *)(dir.hub_rows + dir.nhubs), c->d_partials, (const int32_t *)dir.pull_blk_row,
                               (const int32_t *)dir.hub_chunks, dir.n_hub_chunks, dir.hub_chunk_sums);
        if (nfinish > 0)
            hipLaunchKernelGGL((vgl_k_pull_hub_finish<double, true, vgl_hits_epilogue>), dim3(nfinish), dim3(VGL_BLOCK), 0, c->stream, dir.n_hub_list,
                               (const int32_t *)(dir.hub_chunks + 2 * (size_t)dir.n_hub_chunks), g->row_begin, (const double *)dir.hub_chunk_sums, epi,
                               c->d_partials + nblk + hub_blocks);
    }
    hipLaunchKernelGGL(vgl_k_hits_sumsq, dim3(1), dim3(VGL_BLOCK), 0, c->stream, nparts, c->d_partials, d_sumsq);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}
int vgl_hits_scale_owned(vgl_hip_ctx *c, vgl_hip_graph *g, double *x, const double *d_sumsq)
{
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, vgl_ceil_div(g->nrows, VGL_BLOCK)));
    hipLaunchKernelGGL(vgl_k_hits_scale_rows, dim3(grid), dim3(VGL_BLOCK), 0, c->stream, g->row_begin, g->row_end, x, d_sumsq);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}
int vgl_hits_init(vgl_hip_ctx *c, int32_t V, double *d_auth, double *d_hub)
{
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, vgl_ceil_div(V, VGL_BLOCK)));
    hipLaunchKernelGGL(vgl_k_hits_init, dim3(grid), dim3(VGL_BLOCK), 0, c->stream, V, d_auth, d_hub);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" {

int vgl_hip_hits_run(vgl_hip_ctx *c, vgl_hip_graph *g, int steps, double *d_auth, double *d_hub)
{
    if (!c || !g || !d_auth || !d_hub) VGL_FAIL("hits_run: null argument");
    if (g->row_begin != 0 || g->row_end != g->V) VGL_FAIL("hits_run: graph handle must own all rows");
    if (!g->in.rowptr) VGL_FAIL("hits_run: the incoming CSR is required (authorities are sums over in-neighbours)");
    if (steps < 0) VGL_FAIL("hits_run: negative step count");
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, vgl_ceil_div(g->V, VGL_BLOCK)));
    hipLaunchKernelGGL(vgl_k_hits_init, dim3(grid), dim3(VGL_BLOCK), 0, c->stream, g->V, d_auth, d_hub);
    VGL_HIP_TRY(hipGetLastError());
    for (int step = 0; step < steps; step++) {
        VGL_TRY(vgl_hits_half_step(c, g, g->in, d_hub, d_auth));     // authorities from the hubs of the in-neighbours
        VGL_TRY(vgl_hits_half_step(c, g, g->out, d_auth, d_hub));    // hubs from the authorities of the out-neighbours
    }
    return 0;
}

}  // extern "C"
