// sharded.hip -- the super-step loops of the edge-cut multi-GPU path behind the C ABI: Bellman-Ford / widest paths, Shiloach-Vishkin
// and PageRank over the rows one rank owns, with ONE exchange per super-step issued on the context's stream (comm.hip).  They are the
// MPI flavour of the reference's algorithm loops: algorithms/sssp/shortest_paths.hpp:112-154 (exchange at :136-141),
// algorithms/cc/shiloach_vishkin.hpp:29-79, algorithms/pr/pr.hpp:83-136 (exchange at :58 and :127).  Vertex arrays are replicated
// (V entries on every rank), the graph handle holds rows [row_begin, row_end).  Kernels and collectives are stream-ordered; the host
// waits once per super-step, for the P change counts that decide the exchange form and the loop condition.
#include "vgl_comm.h"

// a world of one without RCCL (comm == NULL in the *_run_sharded calls): same code path, no exchange
struct vgl_solo_comm {
    vgl_hip_comm *m = nullptr;
    bool owned = false;
    int init(vgl_hip_ctx *c, vgl_hip_comm *given)
    {
        if (given) { m = given; return 0; }
        m = new vgl_hip_comm();
        m->ctx = c; m->rank = 0; m->world = 1;
        owned = true;
        if (hipMalloc((void **)&m->d_small, sizeof(int64_t) * VGL_COMM_SMALL) != hipSuccess ||
            hipHostMalloc((void **)&m->h_small, sizeof(int64_t) * (VGL_COMM_SMALL + 8), hipHostMallocDefault) != hipSuccess)
            VGL_FAIL("sharded run: cannot allocate the hand-over buffers");
        for (int i = 0; i < VGL_COMM_SMALL + 8; i++) m->h_small[i] = 0;
        return 0;
    }
    ~vgl_solo_comm() { if (owned && m) vgl_hip_comm_destroy(m); }
};

static int vgl_check_shard(vgl_hip_ctx *c, vgl_hip_comm *m, vgl_hip_graph *g, const char *who)
{
    if (m->ctx != c) return vgl_set_error(__FILE__, __LINE__, (std::string(who) + ": the communicator belongs to another context").c_str());
    if (m->world == 1 && (g->row_begin != 0 || g->row_end != g->V))
        return vgl_set_error(__FILE__, __LINE__, (std::string(who) + ": a world of one must own all rows").c_str());
    m->stats = {0, 0, 0, 0, 0, 0};
    return 0;
}

// Bellman-Ford / widest paths: every rank relaxes the out-edges of its rows into its copy, the copies are merged with min / max
template <bool WIDEST>
static int vgl_path_run_sharded(vgl_hip_ctx *c, vgl_hip_comm *given, vgl_hip_graph *g, const float *d_w, int32_t source, float *d_val,
                                vgl_hip_sssp_stats *stats, const char *who)
{
    if (!c || !g || !d_val || (!d_w && g->out.edges > 0)) return vgl_set_error(__FILE__, __LINE__, (std::string(who) + ": null argument").c_str());     // (a shard may own no edges)
    if (source < 0 || source >= g->V) return vgl_set_error(__FILE__, __LINE__, (std::string(who) + ": source vertex out of range").c_str());
    vgl_solo_comm solo;
    VGL_TRY(solo.init(c, given));
    vgl_hip_comm *m = solo.m;
    VGL_TRY(vgl_check_shard(c, m, g, who));
    const int32_t V = g->V;
    VGL_TRY(WIDEST ? vgl_hip_sswp_init(c, V, source, d_val) : vgl_hip_sssp_init(c, V, source, d_val));
    void *before = nullptr;
    VGL_TRY(vgl_comm_scratch(m, 4, sizeof(float) * (size_t)V, &before));
    vgl_hip_sssp_stats st = {0, 0, 0, 0, 0};
    for (;;) {
        VGL_HIP_TRY(hipMemcpyAsync(before, d_val, sizeof(float) * (size_t)V, hipMemcpyDeviceToDevice, c->stream));
        VGL_TRY(vgl_sssp_relax_enqueue(c, g, d_w, d_val, WIDEST));
        st.iterations++; st.push_steps++;
        st.edges_relaxed += g->out.edges;
        int changed = 0;
        VGL_TRY(vgl_hip_exchange_changed_u32(m, V, before, d_val, WIDEST ? 0 : 1, &changed));
        if (!changed) break;                          // do { ... } while (changes) over ALL ranks (shortest_paths.hpp:112-154)
    }
    st.algorithmic_bytes = 12 * st.edges_relaxed + 28 * (int64_t)(g->row_end - g->row_begin) * st.iterations;
    if (stats) *stats = st;
    return 0;
}

extern "C" {

int vgl_hip_sssp_run_sharded(vgl_hip_ctx *c, vgl_hip_comm *m, vgl_hip_graph *g, const float *d_weights, int32_t source, float *d_dist,
                             vgl_hip_sssp_stats *stats)
{
    return vgl_path_run_sharded<false>(c, m, g, d_weights, source, d_dist, stats, "sssp_run_sharded");
}

int vgl_hip_sswp_run_sharded(vgl_hip_ctx *c, vgl_hip_comm *m, vgl_hip_graph *g, const float *d_capacities, int32_t source, float *d_widths,
                             vgl_hip_sssp_stats *stats)
{
    return vgl_path_run_sharded<true>(c, m, g, d_capacities, source, d_widths, stats, "sswp_run_sharded");
}

// Shiloach-Vishkin: hook over the owned rows, labels merged with min, pointer jumping on the merged (replicated) labels -- every rank
// jumps the same array, so no exchange follows the jump
int vgl_hip_cc_run_sharded(vgl_hip_ctx *c, vgl_hip_comm *given, vgl_hip_graph *g, int32_t *d_comp, vgl_hip_cc_stats *stats)
{
    if (!c || !g || !d_comp) VGL_FAIL("cc_run_sharded: null argument");
    vgl_solo_comm solo;
    VGL_TRY(solo.init(c, given));
    vgl_hip_comm *m = solo.m;
    VGL_TRY(vgl_check_shard(c, m, g, "cc_run_sharded"));
    const int32_t V = g->V;
    VGL_TRY(vgl_hip_cc_init(c, V, d_comp));
    void *before = nullptr;
    VGL_TRY(vgl_comm_scratch(m, 4, sizeof(int32_t) * (size_t)V, &before));
    vgl_hip_cc_stats st = {0, 0};
    for (;;) {
        VGL_HIP_TRY(hipMemcpyAsync(before, d_comp, sizeof(int32_t) * (size_t)V, hipMemcpyDeviceToDevice, c->stream));
        VGL_TRY(vgl_cc_hook_launch(c, g, d_comp));
        st.hook_passes++;
        st.algorithmic_bytes += 8 * g->out.edges + 12 * (int64_t)(g->row_end - g->row_begin);
        int changed = 0;
        VGL_TRY(vgl_hip_exchange_changed_u32(m, V, before, d_comp, 1, &changed));
        if (!changed) break;                          // while (hook_changes)  (shiloach_vishkin.hpp:29)
        VGL_TRY(vgl_hip_cc_jump(c, V, d_comp));
        st.algorithmic_bytes += 12 * (int64_t)V;
    }
    if (stats) *stats = st;
    return 0;
}

// PageRank: every rank pulls the new ranks of the rows it owns from the replicated old ranks; the owned slices are all-gathered in
// place (EXCHANGE_PRIVATE_DATA, pr.hpp:127).  The in-degrees are summed over the shards once (pr.hpp:58).
int vgl_hip_pr_run_sharded(vgl_hip_ctx *c, vgl_hip_comm *given, vgl_hip_graph *g, int iterations, int mode, float *d_ranks,
                           vgl_hip_pr_stats *stats)
{
    if (!c || !g || !d_ranks) VGL_FAIL("pr_run_sharded: null argument");
    if (mode < VGL_HIP_PR_EXACT_ORDER || mode > VGL_HIP_PR_AUTO) VGL_FAIL("pr_run_sharded: unknown mode");
    if (iterations < 0) VGL_FAIL("pr_run_sharded: negative iteration count");
    vgl_solo_comm solo;
    VGL_TRY(solo.init(c, given));
    vgl_hip_comm *m = solo.m;
    VGL_TRY(vgl_check_shard(c, m, g, "pr_run_sharded"));
    const int32_t V = g->V;
    const int P = m->world;
    // AUTO is resolved ONCE from global numbers, so that every rank takes the same evaluation whatever its shard looks like
    VGL_TRY(vgl_pr_env_mode(c, mode, &mode));
    int64_t global_edges = g->out.edges;
    if (mode == VGL_HIP_PR_AUTO) {
        int64_t longest = 0;
        VGL_TRY(vgl_pr_longest_row(c, g, &longest));
        VGL_TRY(vgl_comm_allreduce_host_i64(m, &global_edges, 1, VGL_OP_SUM));
        VGL_TRY(vgl_comm_allreduce_host_i64(m, &longest, 1, VGL_OP_MAX));
        mode = (global_edges >= (1LL << 25) && longest <= 256) ? VGL_HIP_PR_BLOCKED : VGL_HIP_PR_EXACT_ORDER;
    }
    // in-degrees minus self loops of all vertices: a property of the graph (pr.hpp:31-65 recounts them in every run; here they are counted
    // over the shard's edges -- one atomic per edge without incoming lists -- and summed over the ranks ONCE per graph handle, like a plan)
    // (the flag is per graph handle: a rank that recreated its handle, or an earlier run that failed on some ranks only, must not leave the
    // ranks disagreeing on whether the all-reduce below is issued -- they agree on the minimum first)
    int64_t all_ready = g->pr_indeg_ready ? 1 : 0;
    VGL_TRY(vgl_comm_allreduce_host_i64(m, &all_ready, 1, VGL_OP_MIN));
    if (!all_ready) g->pr_indeg_ready = false;
    if (!g->pr_indeg_ready) {
        if (!g->pr_indeg) VGL_HIP_TRY(hipMalloc((void **)&g->pr_indeg, sizeof(int32_t) * (size_t)std::max(V, 1)));
        VGL_HIP_TRY(hipMemsetAsync(g->pr_indeg, 0, sizeof(int32_t) * (size_t)V, c->stream));
        VGL_TRY(vgl_hip_indegree_noloops_add(c, g, g->pr_indeg));
        VGL_TRY(vgl_comm_allreduce(m, g->pr_indeg, V, VGL_DT_I32, VGL_OP_SUM));
        g->pr_indeg_ready = true;
    }
    const int32_t *indeg = g->pr_indeg;
    float *rdeg = g->fscratch2, *contrib = g->fscratch;
    const int64_t *bounds = nullptr;
    std::vector<int64_t> bb((size_t)P + 1, 0);
    const bool active = vgl_comm_active(m);
    if (active) {
        VGL_TRY(vgl_comm_row_bounds(m, g, &bounds));
        for (int p = 0; p <= P; p++) bb[(size_t)p] = bounds[p] * (int64_t)sizeof(float);
    }
    VGL_TRY(vgl_hip_pr_setup(c, V, indeg, d_ranks, rdeg));
    for (int it = 0; it < iterations; it++) {
        // in place is safe: the pull reads only contrib / dangling (both produced from the old ranks) and writes the owned rows
        VGL_TRY(vgl_pr_iteration(c, g, indeg, rdeg, d_ranks, contrib, d_ranks, mode));
        if (active) VGL_TRY(vgl_comm_allgatherv_inplace(m, d_ranks, bb.data()));
    }
    VGL_HIP_TRY(hipStreamSynchronize(c->stream));
    if (active) VGL_TRY(vgl_comm_check(m));              // (a peer that never arrived: the ranks are void, not a silent 0)
    if (stats) {
        stats->iterations = iterations;
        stats->ranks_sum = 0.0;
        stats->algorithmic_bytes = (8 * g->out.edges + 28 * (int64_t)(g->row_end - g->row_begin)) * iterations;
    }
    return 0;
}

// HITS (algorithms/hits/hits.hpp:32-91; the reference distributes it with EXCHANGE_PRIVATE_DATA after each half step, :52 and :79): every
// rank sums the hub values of the in-neighbours (then the authority values of the out-neighbours) for the rows it owns, the ranks add up
// their shares of the sum of squares (one f64), every rank scales its rows and the owned slices are all-gathered in place.  A world of
// one is bit-identical to vgl_hip_hits_run; over several ranks the sum of squares is folded rank by rank instead of workgroup by workgroup,
// which moves the norm by an ulp (the reference's own OpenMP reduction order is unspecified; its runs differ by ~4e-15).
int vgl_hip_hits_run_sharded(vgl_hip_ctx *c, vgl_hip_comm *given, vgl_hip_graph *g, int steps, double *d_auth, double *d_hub)
{
    if (!c || !g || !d_auth || !d_hub) VGL_FAIL("hits_run_sharded: null argument");
    if (!g->in.rowptr) VGL_FAIL("hits_run_sharded: the incoming lists of the owned rows are required (authorities are sums over in-neighbours)");
    if (steps < 0) VGL_FAIL("hits_run_sharded: negative step count");
    vgl_solo_comm solo;
    VGL_TRY(solo.init(c, given));
    vgl_hip_comm *m = solo.m;
    VGL_TRY(vgl_check_shard(c, m, g, "hits_run_sharded"));
    const int P = m->world;
    const bool active = vgl_comm_active(m);
    std::vector<int64_t> bb((size_t)P + 1, 0);
    if (active) {
        const int64_t *bounds = nullptr;
        VGL_TRY(vgl_comm_row_bounds(m, g, &bounds));
        for (int p = 0; p <= P; p++) bb[(size_t)p] = bounds[p] * (int64_t)sizeof(double);
    }
    double *d_sumsq = reinterpret_cast<double *>(m->d_small + VGL_COMM_SMALL - 2);      // (a spare word of the hand-over buffer)
    VGL_TRY(vgl_hits_init(c, g->V, d_auth, d_hub));
    for (int step = 0; step < steps; step++)
        for (int half = 0; half < 2; half++) {
            const double *x = half == 0 ? d_hub : d_auth;
            double *out = half == 0 ? d_auth : d_hub;
            VGL_TRY(vgl_hits_pull_owned(c, g, half == 0, x, out, d_sumsq));
            if (active) VGL_TRY(vgl_comm_allreduce(m, d_sumsq, 1, VGL_DT_F64, VGL_OP_SUM));
            VGL_TRY(vgl_hits_scale_owned(c, g, out, d_sumsq));
            if (active) VGL_TRY(vgl_comm_allgatherv_inplace(m, out, bb.data()));
        }
    VGL_HIP_TRY(hipStreamSynchronize(c->stream));
    if (active) VGL_TRY(vgl_comm_check(m));
    return 0;
}

}  // extern "C"
