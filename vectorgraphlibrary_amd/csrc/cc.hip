// cc.hip -- Shiloach-Vishkin connected components (CC::vgl_shiloach_vishkin, algorithms/cc/shiloach_vishkin.hpp:7-88).
//
// hook : for every edge (src,dst): if comp[src] < comp[dst] then comp[dst] = comp[src]   (edge_op, lines 37-49); done with an
//        integer atomic-min so no update is lost.  Edge-balanced like the SSSP relax kernel: workgroup = 2048 CSR edges.
// jump : comp[v] = comp[comp[v]] repeated to the root (jump_op loop, lines 56-76) in a single pass per vertex.
// The loop ends when a hook pass changes nothing; the fixed point comp[v] = min{u : u reaches v} is unique, so labels are
// bit-identical to the reference under identity (CSR) numbering.  Algorithmic bytes: 8*E + 12*V per hook, 12*V per jump.
#include "vgl_hip_internal.h"
#include "vgl_comm.h"
#include "vgl_blocked.h"
#include <cstdlib>

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_cc_init(int32_t V, int32_t *comp)
{
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK) comp[v] = v;
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_cc_hook(const int64_t *rowptr, const int32_t *adj, const int32_t *tile_row,
                                                           int64_t E, int32_t row_base, int32_t *comp, int64_t *counters)
{
    __shared__ int s_map[VGL_TILE];
    __shared__ int s_w[VGL_WAVES];
    const int64_t e0 = (int64_t)blockIdx.x * VGL_TILE;
    const int n = (int)min((int64_t)VGL_TILE, E - e0);
    const int r_first = tile_row[blockIdx.x];
    const int r_last = tile_row[blockIdx.x + 1];
    vgl_tile_row_map(s_map, s_w, rowptr, e0, r_first, r_last);
    const int i0 = threadIdx.x * VGL_EPT;
    int changed = 0;
    if (i0 < n) {
        int32_t dsts[VGL_EPT];
        if (i0 + VGL_EPT <= n) {
            const int4 a0 = *reinterpret_cast<const int4 *>(adj + e0 + i0);
            const int4 a1 = *reinterpret_cast<const int4 *>(adj + e0 + i0 + 4);
            dsts[0] = a0.x; dsts[1] = a0.y; dsts[2] = a0.z; dsts[3] = a0.w; dsts[4] = a1.x; dsts[5] = a1.y; dsts[6] = a1.z; dsts[7] = a1.w;
        } else {
#pragma unroll
            for (int j = 0; j < VGL_EPT; j++) dsts[j] = (i0 + j < n) ? adj[e0 + i0 + j] : 0;
        }
        // loads unconditional and in rounds (a slot past the tile's end repeats the thread's first edge; guarded loads are compiled as
        // branches that are awaited one after the other), the atomics issued together and their return values read afterwards
        int32_t csrc[VGL_EPT], cdst[VGL_EPT];
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) csrc[j] = comp[row_base + r_first + s_map[i0 + j < n ? i0 + j : i0]];
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) cdst[j] = comp[dsts[j]];
        int32_t before[VGL_EPT];
        bool tried[VGL_EPT];
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) {
            tried[j] = i0 + j < n && csrc[j] < cdst[j];
            before[j] = 0;
            if (tried[j]) before[j] = atomicMin(comp + dsts[j], csrc[j]);
        }
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++)
            if (tried[j] && before[j] > csrc[j]) changed = 1;
    }
    if (__syncthreads_or(changed) && threadIdx.x == 0) counters[C_CHANGED] = 1;
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_cc_jump(int32_t V, int32_t *comp)
{
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK) {
        int32_t c = comp[v];
        int32_t cc = comp[c];
        if (c != cc) {
            while (cc != c) { c = cc; cc = comp[c]; }       // labels only decrease and comp[x] <= x: terminates at a root
            comp[v] = c;
        }
    }
}

// ---- symmetric graphs: min-id union-find (the same fixed point as the hook/jump loop when every edge has its reverse) ----
// comp[] is a parent forest with comp[x] <= x at all times; a root is x with comp[x] == x.  link() hooks the larger of the two
// roots under the smaller one with a CAS and climbs when it loses a race, so the root of a finished tree is its smallest id.
__device__ __forceinline__ void vgl_cc_link_from(int32_t p1, int32_t p2, int32_t *comp)      // p1 = comp[u], p2 = comp[v], read by the caller
{
    while (p1 != p2) {
        const int32_t high = max(p1, p2), low = min(p1, p2);
        const int32_t ph = comp[high];
        if (ph == low) break;
        if (ph == high && atomicCAS(comp + high, high, low) == high) break;
        p1 = comp[comp[high]];
        p2 = comp[low];
    }
}
__device__ __forceinline__ void vgl_cc_link(int32_t u, int32_t v, int32_t *comp)
{
    int32_t p1 = comp[u], p2 = comp[v];
    while (p1 != p2) {
        const int32_t high = max(p1, p2), low = min(p1, p2);
        const int32_t ph = comp[high];
        if (ph == low) break;
        if (ph == high && atomicCAS(comp + high, high, low) == high) break;
        p1 = comp[comp[high]];
        p2 = comp[low];
    }
}

// sampling round k: every owned row links with its k-th neighbour (V links instead of E)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_cc_link_sample(int32_t nrows, int32_t row_base, const int64_t *rowptr, const int32_t *adj,
                                                                  int k, int32_t *comp)
{
    for (int32_t r = blockIdx.x * VGL_BLOCK + threadIdx.x; r < nrows; r += gridDim.x * VGL_BLOCK) {
        const int64_t b = rowptr[r];
        if (rowptr[r + 1] - b > k) vgl_cc_link(row_base + r, adj[b + k], comp);
    }
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_cc_sample_labels(int32_t V, const int32_t *comp, int nsamples, int32_t *out)
{
    const int i = blockIdx.x * VGL_BLOCK + threadIdx.x;
    if (i < nsamples) out[i] = comp[(int32_t)(((uint64_t)(uint32_t)i * 2654435761ull) % (uint64_t)V)];
}

// remaining edges, edge-balanced: rows already in the tree `giant` skip their edges without loading them (their edges towards
// other trees are seen from the other endpoint -- the graph is symmetric); the first `skip` edges of a row were linked by the
// sampling rounds.  A workgroup whose rows are all in `giant` touches no adjacency at all.
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_cc_link_rest(const int64_t *rowptr, const int32_t *adj, const int32_t *tile_row, int64_t E,
                                                                int32_t row_base, int32_t giant, int skip, int32_t *comp)
{
    __shared__ int s_map[VGL_TILE];
    __shared__ int s_w[VGL_WAVES];
    const int64_t e0 = (int64_t)blockIdx.x * VGL_TILE;
    const int n = (int)min((int64_t)VGL_TILE, E - e0);
    const int r_first = tile_row[blockIdx.x];
    const int r_last = tile_row[blockIdx.x + 1];
    vgl_tile_row_map(s_map, s_w, rowptr, e0, r_first, r_last);
    // the rows' parents (and row starts), then the neighbours, then the neighbours' parents: three rounds of loads with nothing conditional
    // about the INSTRUCTIONS (guarded, each load was a branch awaited before the next was issued) -- a slot that skips its edge (row in
    // `giant`, edge linked by the sampling rounds, past the tile's end) reads entry 0 of the tile / of comp instead, one shared line, so the
    // adjacency of the rows in `giant` still is not fetched; a wavefront with nothing to link leaves after the first round
    int32_t pu[VGL_EPT], vs[VGL_EPT], pv[VGL_EPT];
    int64_t rb[VGL_EPT];
    bool take[VGL_EPT];
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) {
        const int i = threadIdx.x + j * VGL_BLOCK;
        const int32_t r = r_first + s_map[i < n ? i : 0];
        pu[j] = comp[row_base + r];
        rb[j] = rowptr[r];
    }
    bool any = false;
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) {
        const int i = threadIdx.x + j * VGL_BLOCK;
        take[j] = i < n && pu[j] != giant && e0 + i - rb[j] >= skip;
        any = any || take[j];
    }
    if (!__any(any)) return;
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) vs[j] = adj[take[j] ? e0 + threadIdx.x + j * VGL_BLOCK : e0];
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) pv[j] = comp[take[j] ? vs[j] : 0];
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++)
        if (take[j]) vgl_cc_link_from(pu[j], pv[j], comp);
}

static inline unsigned vgl_grid2(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, vgl_ceil_div(n, VGL_BLOCK))); }

// The hook as a blocked pass (vgl_blocked.h) over the outgoing CSR with the rows as the gather side: comp[src] is read from a
// 128 KiB LDS window, travels to the destination's block and is folded with an LDS integer minimum; one compare + store per
// vertex ends the pass.  12 B/edge of streamed traffic and no random L2 line per edge (the kernel above: 3.4 ms per pass on the
// symmetrised RMAT-24, 17 % of the HBM peak on algorithmic bytes).  The pass sees the labels as they were when it started, so a
// label moves one hop per pass before the pointer jump spreads it -- the fixed point, hence every label, is the same.
struct vgl_cc_blk_op {
    typedef uint32_t acc_t;
    static constexpr bool MARK = false;
    int32_t *comp;
    int32_t g_base;
    int64_t *counters;
    __device__ __forceinline__ uint32_t load(int32_t i) const { return (uint32_t)comp[g_base + i]; }
    __device__ __forceinline__ uint32_t edge(uint32_t x, float) const { return x; }
    __device__ __forceinline__ uint32_t identity() const { return 0x7FFFFFFFu; }
    __device__ __forceinline__ void accumulate(uint32_t *p, uint32_t v) const
    {
        __hip_atomic_fetch_min(reinterpret_cast<int *>(p), (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ uint32_t combine(uint32_t a, uint32_t b) const { return min(a, b); }
    __device__ __forceinline__ void finish(int32_t v, uint32_t acc) const
    {
        if ((int32_t)acc < comp[v]) { comp[v] = (int32_t)acc; counters[C_CHANGED] = 1; }
    }
    __device__ __forceinline__ bool partial(int32_t v, uint32_t acc) const
    {
        if ((int32_t)acc < comp[v] && atomicMin(comp + v, (int32_t)acc) > (int32_t)acc) counters[C_CHANGED] = 1;
        return true;
    }
};

// blocked from 2^25 stored edges (below that the labels sit in L2 and the atomic kernel needs fewer passes); VGL_CC_BLOCKED=0|1 overrides
static bool vgl_cc_use_blocked(vgl_hip_ctx *c, const vgl_hip_graph *g)
{
    const char *s = vgl_env(c, "VGL_CC_BLOCKED");
    if (s && *s) return atoi(s) != 0;
    return g->out.edges >= (1LL << 25);
}

extern "C" int vgl_hip_cc_prepare(vgl_hip_ctx *c, vgl_hip_graph *g);

int vgl_cc_hook_launch(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t *comp)
{
    if (g->out.ntiles == 0) return 0;
    if (vgl_cc_use_blocked(c, g)) {
        if (!g->blk_cc) VGL_TRY(vgl_hip_cc_prepare(c, g));      // dense pairs of 16384-id blocks as fused tiles (vgl_blocked.h): 4 B per edge streamed instead of 12
        const vgl_cc_blk_op op{comp, g->row_begin, c->d_counters};
        return vgl_blocked_pass<vgl_cc_blk_op, false, false>(c, g->blk_cc, op, "cc_hook_gather", "cc_hook_accumulate", false, "cc_hook_fused");
    }
    vgl_timed_launch tl(c, "cc_hook");
    hipLaunchKernelGGL(vgl_k_cc_hook, dim3((unsigned)g->out.ntiles), dim3(VGL_BLOCK), 0, c->stream, g->out.rowptr, g->out.adj,
                       g->out.tile_row, g->out.edges, g->row_begin, comp, c->d_counters);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" {

int vgl_hip_cc_prepare(vgl_hip_ctx *c, vgl_hip_graph *g)
{
    if (!c || !g) VGL_FAIL("cc_prepare: null argument");
    if (vgl_cc_use_blocked(c, g) && !g->blk_cc && g->out.ntiles > 0) {
        const char *fm = vgl_env(c, "VGL_BLK_FUSE_MIN");
        const int fuse_min = (fm && *fm) ? atoi(fm) : 16384;
        VGL_TRY(vgl_blocked_plan_build(c, g->out, g->nrows, g->row_begin, g->V, 1, 0, nullptr, VGL_BLK_BITS, &g->blk_cc, 32, fuse_min));
    }
    VGL_HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int vgl_hip_cc_init(vgl_hip_ctx *c, int32_t V, int32_t *d_comp)
{
    if (!c || !d_comp) VGL_FAIL("cc_init: null argument");
    hipLaunchKernelGGL(vgl_k_cc_init, dim3(vgl_grid2(V)), dim3(VGL_BLOCK), 0, c->stream, V, d_comp);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_cc_hook_owned(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t *d_comp, int *changed)
{
    if (!c || !g || !d_comp) VGL_FAIL("cc_hook_owned: null argument");
    VGL_TRY(vgl_zero_counters(c, C_CHANGED, 1));
    VGL_TRY(vgl_cc_hook_launch(c, g, d_comp));
    VGL_TRY(vgl_read_counters(c));
    if (changed) *changed = (int)c->h_counters[C_CHANGED];
    return 0;
}

int vgl_hip_cc_jump(vgl_hip_ctx *c, int32_t V, int32_t *d_comp)
{
    if (!c || !d_comp) VGL_FAIL("cc_jump: null argument");
    vgl_timed_launch tl(c, "cc_jump");
    hipLaunchKernelGGL(vgl_k_cc_jump, dim3(vgl_grid2(V)), dim3(VGL_BLOCK), 0, c->stream, V, d_comp);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_cc_run(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t *d_comp, vgl_hip_cc_stats *stats)
{
    if (!c || !g || !d_comp) VGL_FAIL("cc_run: null argument");
    if (g->row_begin != 0 || g->row_end != g->V) VGL_FAIL("cc_run: graph handle must own all rows (use the step API for shards)");
    VGL_TRY(vgl_hip_cc_init(c, g->V, d_comp));
    vgl_hip_cc_stats st = {0, 0};
    for (;;) {
        int changed = 0;
        VGL_TRY(vgl_hip_cc_hook_owned(c, g, d_comp, &changed));
        st.hook_passes++;
        st.algorithmic_bytes += 8 * g->out.edges + 12 * (int64_t)g->V;
        if (!changed) break;                         // while(hook_changes)  (shiloach_vishkin.hpp:29)
        VGL_TRY(vgl_hip_cc_jump(c, g->V, d_comp));
        st.algorithmic_bytes += 12 * (int64_t)g->V;
    }
    if (stats) *stats = st;
    return 0;
}

int vgl_hip_cc_run_symmetric(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t *d_comp, vgl_hip_cc_stats *stats)
{
    if (!c || !g || !d_comp) VGL_FAIL("cc_run_symmetric: null argument");
    if (g->row_begin != 0 || g->row_end != g->V) VGL_FAIL("cc_run_symmetric: graph handle must own all rows");
    const int32_t V = g->V;
    constexpr int ROUNDS = 2, NSAMPLES = 1024;
    VGL_TRY(vgl_hip_cc_init(c, V, d_comp));
    vgl_hip_cc_stats st = {0, 0};
    for (int k = 0; k < ROUNDS; k++) {
        {
            vgl_timed_launch tl(c, "cc_hook");
            hipLaunchKernelGGL(vgl_k_cc_link_sample, dim3(vgl_grid2(g->nrows)), dim3(VGL_BLOCK), 0, c->stream, g->nrows, g->row_begin,
                               g->out.rowptr, g->out.adj, k, d_comp);
        }
        VGL_TRY(vgl_hip_cc_jump(c, V, d_comp));
        st.hook_passes++;
        st.algorithmic_bytes += (int64_t)V * (16 + 4 + 8) + 12 * (int64_t)V;      // row offsets, one neighbour, two labels; compress
    }
    // most frequent label among a fixed sample = the largest tree so far (any label is a correct choice; it only decides what is skipped)
    int32_t giant = -1;
    if (V > 0) {
        int32_t *d_samples = reinterpret_cast<int32_t *>(g->iscratch);
        const int ns = std::min<int64_t>(NSAMPLES, V);
        hipLaunchKernelGGL(vgl_k_cc_sample_labels, dim3(vgl_ceil_div(ns, VGL_BLOCK)), dim3(VGL_BLOCK), 0, c->stream, V, d_comp, ns, d_samples);
        VGL_HIP_TRY(hipGetLastError());
        std::vector<int32_t> h((size_t)ns);
        VGL_TRY(vgl_hip_memcpy_d2h(c, h.data(), d_samples, sizeof(int32_t) * (size_t)ns));
        std::sort(h.begin(), h.end());
        int best = 0;
        for (int i = 0; i < ns;) {
            int j = i;
            while (j < ns && h[(size_t)j] == h[(size_t)i]) j++;
            if (j - i > best) { best = j - i; giant = h[(size_t)i]; }
            i = j;
        }
    }
    if (g->out.ntiles > 0) {
        vgl_timed_launch tl(c, "cc_hook");
        hipLaunchKernelGGL(vgl_k_cc_link_rest, dim3((unsigned)g->out.ntiles), dim3(VGL_BLOCK), 0, c->stream, g->out.rowptr, g->out.adj,
                           g->out.tile_row, g->out.edges, g->row_begin, giant, ROUNDS, d_comp);
        VGL_HIP_TRY(hipGetLastError());
    }
    VGL_TRY(vgl_hip_cc_jump(c, V, d_comp));
    st.hook_passes++;
    st.algorithmic_bytes += 12 * (int64_t)V + 12 * (int64_t)V;                    // row offsets + label per row (edges of skipped rows are not read); compress
    if (stats) *stats = st;
    return 0;
}

}  // extern "C"
