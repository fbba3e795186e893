// bfs.hip -- fused BFS: BFS::fast_vgl_top_down (algorithms/bfs/bfs.hpp:6-51) as edge-balanced HIP kernels plus the
// bottom-up step of the direction-optimising variant (hardwired_do_bfs.hpp is dead code in the reference; its switch
// rule change_state.hpp:100-141 is kept, and the result is required to equal the top-down levels).
//
// Data in HBM: levels int32[V] (result), three bitmaps of V bits (visited / current frontier / next frontier; 2 MiB
// each at scale 24 => resident in every XCD's 4 MiB L2), frontier ids int32 + exclusive edge offsets int64.
//
// Kernels and what bounds them (all HBM/L2-latency bound gathers, no MFMA):
//   gnf count/scan/write (vgl_gnf.h)   : V*4 B streamed per pass
//   vgl_k_td_expand                    : per examined edge 4 B adjacency (coalesced) + bitmap probe (L2) [+4 B levels]
//   vgl_k_bu_probe / vgl_k_bu_heavy    : per unvisited vertex 16 B row offsets + up to 8 adjacency probes (thread-serial),
//                                        remaining long rows strip-mined 64-wide by one wavefront per vertex
#include "vgl_hip_internal.h"
#include "vgl_gnf.h"

constexpr int VGL_BU_PROBES = 8;       // thread-serial probes before a vertex is deferred to the wavefront pass
constexpr int VGL_DO_ALPHA = 15;       // change_state.hpp:5
constexpr int VGL_DO_BETA = 18;        // change_state.hpp:6

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_bfs_init(int32_t V, int32_t source, int32_t *levels)
{
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK)
        levels[v] = (v == source) ? 1 : -1;      // FIRST_LEVEL_VERTEX / UNVISITED_VERTEX (change_state.h:21-23)
}

// tile_first[t] = frontier position whose edge range contains edge t*VGL_TILE
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_tile_first(int32_t F, const int64_t *offs, int32_t *tile_first)
{
    for (int32_t p = blockIdx.x * VGL_BLOCK + threadIdx.x; p < F; p += gridDim.x * VGL_BLOCK) {
        const int64_t t0 = (offs[p] + VGL_TILE - 1) / VGL_TILE;
        const int64_t t1 = (offs[p + 1] + VGL_TILE - 1) / VGL_TILE;
        for (int64_t t = t0; t < t1; t++) tile_first[t] = p;
    }
}

// top-down advance over a sparse frontier, edge-balanced: workgroup = 2048 consecutive frontier edges.
// edge_op of bfs.hpp:28-36: if levels[dst] == UNVISITED then levels[dst] = cur+1 (benign race, same value).
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_td_expand(const int32_t *ids, const int64_t *offs, const int32_t *tile_first,
                                                             int32_t F, int64_t M, const int64_t *rowptr, const int32_t *adj,
                                                             int32_t row_base, const uint64_t *visited, int32_t *levels,
                                                             int32_t next_level)
{
    __shared__ int s_map[VGL_TILE];
    __shared__ int s_w[VGL_WAVES];
    const int64_t e0 = (int64_t)blockIdx.x * VGL_TILE;
    const int n = (int)min((int64_t)VGL_TILE, M - e0);
    const int p_first = tile_first[blockIdx.x];
    const int p_last = (e0 + VGL_TILE < M) ? tile_first[blockIdx.x + 1] : F - 1;
    vgl_tile_row_map(s_map, s_w, offs, e0, p_first, p_last);
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) {
        const int i = threadIdx.x + j * VGL_BLOCK;          // strided slots => coalesced adjacency reads
        if (i < n) {
            const int p = p_first + s_map[i];
            const int32_t v = ids[p];
            const int64_t e = rowptr[v - row_base] + (e0 + i - offs[p]);
            const int32_t dst = adj[e];
            if (!((visited[dst >> 6] >> (dst & 63)) & 1ULL)) {
                if (levels[dst] == -1) levels[dst] = next_level;
            }
        }
    }
}

// bottom-up, pass 1: one thread per owned vertex; unvisited vertices probe their first incoming neighbours against
// the frontier bitmap.  Writes whole words of the next-frontier bitmap (wave = 64 consecutive vertices).
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_bu_probe(int32_t nrows, int32_t row_base, const int64_t *in_rowptr,
                                                            const int32_t *in_adj, const uint64_t *visited,
                                                            const uint64_t *front, uint64_t *next, int32_t *levels,
                                                            int32_t next_level, int32_t *heavy, int64_t *counters)
{
    __shared__ int64_t s64[VGL_WAVES];
    int64_t found_cnt = 0, probes = 0;
    const int32_t nround = (nrows + 63) & ~63;
    for (int32_t r = blockIdx.x * VGL_BLOCK + threadIdx.x; r < nround; r += gridDim.x * VGL_BLOCK) {
        const int32_t v = row_base + r;
        bool found = false, defer = false;
        if (r < nrows && !((visited[v >> 6] >> (v & 63)) & 1ULL)) {
            const int64_t b = in_rowptr[r], e = in_rowptr[r + 1];
            const int n = (int)min((int64_t)VGL_BU_PROBES, e - b);
            int i = 0;
            for (; i < n; i++) {
                const int32_t u = in_adj[b + i];
                if ((front[u >> 6] >> (u & 63)) & 1ULL) { found = true; i++; break; }
            }
            probes += i;
            defer = !found && (e - b) > VGL_BU_PROBES;
            if (found) levels[v] = next_level;
        }
        const unsigned long long fm = __ballot(found);
        if (vgl_lane() == 0) next[v >> 6] = fm;
        found_cnt += found;
        const unsigned long long dm = __ballot(defer);
        if (dm) {                                   // wave-aggregated append to the heavy list
            int base = 0;
            if (vgl_lane() == 0) base = (int)atomicAdd((unsigned long long *)&counters[C_HEAVY], (unsigned long long)__popcll(dm));
            base = __shfl(base, 0);
            if (defer) heavy[base + __popcll(dm & ((1ULL << vgl_lane()) - 1ULL))] = r;
        }
    }
    found_cnt = vgl_block_reduce_add(found_cnt, s64);
    probes = vgl_block_reduce_add(probes, s64);
    if (threadIdx.x == 0) {
        if (found_cnt) atomicAdd((unsigned long long *)&counters[C_BU_FOUND], (unsigned long long)found_cnt);
        if (probes) atomicAdd((unsigned long long *)&counters[C_BU_EDGES], (unsigned long long)probes);
    }
}

// bottom-up, pass 2: one wavefront per deferred vertex, 64 incoming neighbours per step, early exit on the first hit
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_bu_heavy(int32_t row_base, const int64_t *in_rowptr, const int32_t *in_adj,
                                                            const uint64_t *front, uint64_t *next, int32_t *levels,
                                                            int32_t next_level, const int32_t *heavy, int64_t *counters)
{
    const int64_t nheavy = counters[C_HEAVY];
    const int wave_global = blockIdx.x * VGL_WAVES + vgl_wave();
    const int nwaves = gridDim.x * VGL_WAVES;
    int64_t found_cnt = 0, probes = 0;
    for (int64_t h = wave_global; h < nheavy; h += nwaves) {
        const int32_t r = heavy[h];
        const int64_t b = in_rowptr[r] + VGL_BU_PROBES, e = in_rowptr[r + 1];
        bool hit_any = false;
        for (int64_t p = b; p < e && !hit_any; p += 64) {
            const int64_t q = p + vgl_lane();
            bool hit = false;
            if (q < e) { const int32_t u = in_adj[q]; hit = (front[u >> 6] >> (u & 63)) & 1ULL; }
            hit_any = __ballot(hit) != 0ULL;
            probes += min((int64_t)64, e - p);
        }
        if (hit_any && vgl_lane() == 0) {
            const int32_t v = row_base + r;
            levels[v] = next_level;
            atomicOr((unsigned long long *)&next[v >> 6], 1ULL << (v & 63));
            found_cnt++;
        }
    }
    if (vgl_lane() == 0) {
        if (found_cnt) atomicAdd((unsigned long long *)&counters[C_BU_FOUND], (unsigned long long)found_cnt);
        if (probes) atomicAdd((unsigned long long *)&counters[C_BU_EDGES], (unsigned long long)probes);
    }
}

// visited |= next; front = next   (one word per thread)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_bm_advance(int64_t words, uint64_t *visited, uint64_t *front, const uint64_t *next)
{
    for (int64_t w = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; w < words; w += (int64_t)gridDim.x * VGL_BLOCK) {
        const uint64_t n = next[w];
        visited[w] |= n;
        front[w] = n;
    }
}

// bit v = (levels[v] == level), or (levels[v] != level) when NOT_EQUAL (visited bitmap: level = -1)
template <bool NOT_EQUAL>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_levels_to_bitmap(int32_t V, const int32_t *levels, int32_t level, uint64_t *bits)
{
    const int32_t vround = (V + 63) & ~63;
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < vround; v += gridDim.x * VGL_BLOCK) {
        const bool on = v < V && ((levels[v] == level) != NOT_EQUAL);
        const unsigned long long m = __ballot(on);
        if (vgl_lane() == 0) bits[v >> 6] = m;
    }
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_apply_bitmaps(int32_t V, int parts, int64_t words, const uint64_t *bits_all,
                                                                 int32_t *levels, int32_t level, int64_t *counters)
{
    __shared__ int64_t s64[VGL_WAVES];
    int64_t cnt = 0;
    const int32_t vround = (V + 63) & ~63;
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < vround; v += gridDim.x * VGL_BLOCK) {
        uint64_t w = 0;
        for (int p = 0; p < parts; p++) w |= bits_all[(int64_t)p * words + (v >> 6)];   // wave-uniform loads
        if (v < V) {
            int32_t l = levels[v];
            if (((w >> (v & 63)) & 1ULL) && l == -1) { levels[v] = level; l = level; }
            cnt += (l == level);
        }
    }
    cnt = vgl_block_reduce_add(cnt, s64);
    if (threadIdx.x == 0 && cnt) atomicAdd((unsigned long long *)&counters[C_TMP0], (unsigned long long)cnt);
}

static inline unsigned vgl_grid(int64_t n, int64_t cap = 8192) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cap, vgl_ceil_div(n, VGL_BLOCK))); }

// expand frontier (ids/offs with F vertices, M edges already produced by the GNF write pass)
static int vgl_bfs_td_launch(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t F, int64_t M, int32_t *levels, int32_t next_level)
{
    if (F <= 0 || M <= 0) return 0;
    hipLaunchKernelGGL(vgl_k_tile_first, dim3(vgl_grid(F)), dim3(VGL_BLOCK), 0, c->stream, F, g->offs, g->tile_first);
    const int64_t nt = vgl_ceil_div(M, VGL_TILE);
    {
        vgl_timed_launch tl(c, "bfs_top_down");
        hipLaunchKernelGGL(vgl_k_td_expand, dim3((unsigned)nt), dim3(VGL_BLOCK), 0, c->stream, g->ids, g->offs, g->tile_first, F, M,
                           g->out.rowptr, g->out.adj, g->row_begin, g->bm_visited, levels, next_level);
    }
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" {

int vgl_hip_bfs_init(vgl_hip_ctx *c, int32_t V, int32_t source, int32_t *d_levels)
{
    if (!c || !d_levels) VGL_FAIL("bfs_init: null argument");
    if (source < 0 || source >= V) VGL_FAIL("bfs_init: source vertex out of range");
    hipLaunchKernelGGL(vgl_k_bfs_init, dim3(vgl_grid(V)), dim3(VGL_BLOCK), 0, c->stream, V, source, d_levels);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_bfs_run(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t source, int mode, int32_t *d_levels, vgl_hip_bfs_stats *stats)
{
    if (!c || !g || !d_levels) VGL_FAIL("bfs_run: null argument");
    if (g->row_begin != 0 || g->row_end != g->V) VGL_FAIL("bfs_run: graph handle must own all rows (use the step API for shards)");
    if (mode != VGL_HIP_BFS_TOP_DOWN && mode != VGL_HIP_BFS_DIRECTION_OPT) VGL_FAIL("bfs_run: unknown mode");
    if (mode == VGL_HIP_BFS_DIRECTION_OPT && !g->in.rowptr) VGL_FAIL("bfs_run: direction-optimising mode needs the incoming CSR");
    const int32_t V = g->V;
    const int64_t E = g->out.edges;
    const int64_t words = vgl_ceil_div(V, 64);
    VGL_TRY(vgl_hip_bfs_init(c, V, source, d_levels));

    vgl_hip_bfs_stats st = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int32_t cur = 1;
    bool bottom_up = false;          // state used to PROCESS level `cur`
    bool have_bitmaps = false;       // bm_front / bm_visited describe level `cur`
    int64_t F = 0, M = 0, prevF = 0, visited_total = 0;
    const int64_t factor = std::max<int64_t>(1, (E / V) / 2);     // change_state.hpp:104
    for (;;) {
        if (!bottom_up) {
            // frontier of level cur from the levels array: counts + bitmaps, then ids + edge offsets
            vgl_pred_equal_i32 pred{d_levels, cur};
            VGL_TRY(vgl_gnf_run(c, g, pred, g->ids, g->offs, (uint8_t *)g->bm_front, (uint8_t *)g->bm_visited, d_levels,
                                nullptr, false, true));
            F = c->h_counters[C_FRONT]; M = c->h_counters[C_NEIGH];
            have_bitmaps = true;
        }
        if (F == 0) break;
        visited_total += F;
        st.levels++; st.frontier_total += F;
        // direction for this level (gpu_change_state, change_state.hpp:100-141, evaluated with the frontier about to be expanded)
        if (mode == VGL_HIP_BFS_DIRECTION_OPT) {
            if (!bottom_up) {
                if (F > prevF && M >= ((V - visited_total) * factor + V) / VGL_DO_ALPHA) bottom_up = true;
            } else {
                if (F < prevF && F < ((V - visited_total) * factor + V) / (factor * VGL_DO_BETA)) {
                    bottom_up = false;
                    // need ids/offs of level cur: regenerate from levels
                    vgl_pred_equal_i32 pred{d_levels, cur};
                    VGL_TRY(vgl_gnf_run(c, g, pred, g->ids, g->offs, (uint8_t *)g->bm_front, (uint8_t *)g->bm_visited, d_levels,
                                        nullptr, false, true));
                    F = c->h_counters[C_FRONT]; M = c->h_counters[C_NEIGH];
                    have_bitmaps = true;
                }
            }
        }
        prevF = F;
        if (!bottom_up) {
            vgl_pred_equal_i32 pred{d_levels, cur};
            {
                vgl_timed_launch tl(c, "gnf");
                hipLaunchKernelGGL(vgl_k_gnf_write<vgl_pred_equal_i32>, dim3((unsigned)g->nvtiles), dim3(VGL_BLOCK), 0, c->stream, pred,
                                   g->nrows, g->row_begin, g->out.rowptr, g->vt_cnt_off, g->vt_deg_off, g->ids, g->offs);
            }
            VGL_TRY(vgl_bfs_td_launch(c, g, (int32_t)F, M, d_levels, cur + 1));
            st.td_steps++; st.edges_examined += M; st.td_edges += M; st.td_frontier += F;
        } else {
            if (!have_bitmaps) VGL_FAIL("bfs_run: internal error (bitmaps missing)");
            VGL_TRY(vgl_zero_counters(c, C_BU_FOUND, 3));        // C_BU_FOUND, C_BU_EDGES, C_HEAVY
            {
                vgl_timed_launch tl(c, "bfs_bottom_up");
                hipLaunchKernelGGL(vgl_k_bu_probe, dim3(vgl_grid(g->nrows, 1 << 20)), dim3(VGL_BLOCK), 0, c->stream, g->nrows, g->row_begin,
                                   g->in.rowptr, g->in.adj, g->bm_visited, g->bm_front, g->bm_next, d_levels, cur + 1, g->heavy,
                                   c->d_counters);
            }
            {
                vgl_timed_launch tl(c, "bfs_bottom_up_heavy");
                hipLaunchKernelGGL(vgl_k_bu_heavy, dim3(2048), dim3(VGL_BLOCK), 0, c->stream, g->row_begin, g->in.rowptr, g->in.adj,
                                   g->bm_front, g->bm_next, d_levels, cur + 1, g->heavy, c->d_counters);
            }
            hipLaunchKernelGGL(vgl_k_bm_advance, dim3(vgl_grid(words)), dim3(VGL_BLOCK), 0, c->stream, words, g->bm_visited,
                               g->bm_front, g->bm_next);
            VGL_HIP_TRY(hipGetLastError());
            VGL_TRY(vgl_read_counters(c));
            st.bu_steps++; st.edges_examined += c->h_counters[C_BU_EDGES];
            st.bu_edges += c->h_counters[C_BU_EDGES]; st.bu_found += c->h_counters[C_BU_FOUND];
            F = c->h_counters[C_BU_FOUND]; M = 0;      // next frontier; bitmaps now describe level cur+1
            have_bitmaps = true;
        }
        cur++;
    }
    st.discovered = visited_total;
    st.algorithmic_bytes = 8 * st.edges_examined + 20 * st.frontier_total + 4 * st.discovered + 4 * (int64_t)V +
                           (int64_t)st.bu_steps * (V / 8);
    if (stats) *stats = st;
    return 0;
}

int vgl_hip_bfs_step_top_down(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t *d_levels, int32_t level, int64_t *local_frontier,
                              int64_t *local_edges)
{
    if (!c || !g || !d_levels) VGL_FAIL("bfs_step_top_down: null argument");
    // visited bitmap over ALL vertices (destinations may live in any shard), then the owned part of the frontier
    hipLaunchKernelGGL(vgl_k_levels_to_bitmap<true>, dim3(vgl_grid(g->V)), dim3(VGL_BLOCK), 0, c->stream, g->V, d_levels, -1,
                       g->bm_visited);
    vgl_pred_equal_i32 pred{d_levels, level};
    VGL_TRY(vgl_gnf_run(c, g, pred, g->ids, g->offs, nullptr, nullptr, nullptr, nullptr, true, true));
    const int64_t F = c->h_counters[C_FRONT], M = c->h_counters[C_NEIGH];
    if (local_frontier) *local_frontier = F;
    if (local_edges) *local_edges = M;
    VGL_TRY(vgl_bfs_td_launch(c, g, (int32_t)F, M, d_levels, level + 1));
    return vgl_hip_ctx_sync(c);
}

int vgl_hip_levels_to_bitmap(vgl_hip_ctx *c, int32_t V, const int32_t *d_levels, int32_t level, uint64_t *d_bits)
{
    if (!c || !d_levels || !d_bits) VGL_FAIL("levels_to_bitmap: null argument");
    hipLaunchKernelGGL(vgl_k_levels_to_bitmap<false>, dim3(vgl_grid(V)), dim3(VGL_BLOCK), 0, c->stream, V, d_levels, level, d_bits);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_bfs_apply_bitmaps(vgl_hip_ctx *c, int32_t V, int parts, const uint64_t *d_bits_all, int32_t *d_levels, int32_t level,
                              int64_t *newly)
{
    if (!c || !d_bits_all || !d_levels) VGL_FAIL("bfs_apply_bitmaps: null argument");
    if (parts < 1) VGL_FAIL("bfs_apply_bitmaps: parts must be >= 1");
    VGL_TRY(vgl_zero_counters(c, C_TMP0, 1));
    hipLaunchKernelGGL(vgl_k_apply_bitmaps, dim3(vgl_grid(V)), dim3(VGL_BLOCK), 0, c->stream, V, parts, vgl_ceil_div(V, 64), d_bits_all,
                       d_levels, level, c->d_counters);
    VGL_HIP_TRY(hipGetLastError());
    VGL_TRY(vgl_read_counters(c));
    if (newly) *newly = c->h_counters[C_TMP0];
    return 0;
}

}  // extern "C"
