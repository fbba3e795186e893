// sssp.hip -- Bellman-Ford push relaxation (SSSP::vgl_dijkstra_all_active_push, algorithms/sssp/shortest_paths.hpp:85-163)
// as an edge-balanced HIP kernel.
//
// One launch = one super-step over the owned out-edges.  Workgroup = one static tile of 2048 consecutive CSR edges
// (thread = 8 consecutive edges: two 16-byte loads of adjacency and of weights, fully coalesced), rows recovered with
// the LDS marker + max-scan map.  Per edge: dist[src] (re-read only when the row changes), gather dist[dst] (4 B random,
// served by L2 / Infinity Cache for a 64 MiB array), f32 add + compare, integer atomic-min on the f32 bits when it
// improves (all distances are non-negative, so the int order equals the float order).  The fixed point
// d[v] = min_u (d[u] (+) w(u,v)) is unique, so any relaxation order gives bit-identical distances.
//
// VGL_HIP_SSSP_ACTIVE_TILES additionally keeps epoch[v] = last super-step in which d[v] decreased and skips a whole
// tile (no adjacency / weight traffic) when none of its rows changed in the current or previous super-step.
// Algorithmic bytes per streamed edge: 4 (adj) + 4 (weight) + 4 (dist[dst]) = 12; per vertex and super-step 28.
//
// The same kernel runs single-source WIDEST paths (SSWP::vgl_dijkstra, algorithms/sswp/widest_paths.hpp:5-76): the path algebra
// is a template parameter -- (min, +) with "smaller is better" for SSSP, (max, min) with "larger is better" for SSWP.  Widths
// are non-negative too, so the integer atomic-max on the f32 bits is exact; only min / max of inputs occur (no rounding).
#include "vgl_hip_internal.h"
#include "vgl_comm.h"
#include "vgl_blocked.h"
#include "vgl_gnf.h"
#include <cfloat>
#include <cstdlib>
#include <string>

struct vgl_path_shortest {                        // shortest_paths.hpp:99-133
    static __device__ __forceinline__ float source_value() { return 0.0f; }
    static __device__ __forceinline__ float other_value() { return FLT_MAX; }     // inf_val = FLT_MAX - MAX_WEIGHT == FLT_MAX in f32
    static __device__ __forceinline__ bool live(float d) { return d < FLT_MAX; }
    static __device__ __forceinline__ float dead_value() { return FLT_MAX; }
    static __device__ __forceinline__ float extend(float d, float w) { return __fadd_rn(d, w); }      // src_weight + weight
    static __device__ __forceinline__ bool better(float cand, float old) { return old > cand; }
    static __device__ __forceinline__ int update(float *p, float cand) { return atomicMin(reinterpret_cast<int *>(p), __float_as_int(cand)); }
    static __device__ __forceinline__ bool improved(int before, float cand) { return before > __float_as_int(cand); }
};
struct vgl_path_widest {                          // widest_paths.hpp:22-55
    static __device__ __forceinline__ float source_value() { return FLT_MAX; }    // numeric_limits<float>::max() - MAX_WEIGHT == FLT_MAX
    static __device__ __forceinline__ float other_value() { return 0.0f; }
    static __device__ __forceinline__ bool live(float d) { return d > 0.0f; }     // min(0, capacity) = 0 never beats a width >= 0
    static __device__ __forceinline__ float dead_value() { return 0.0f; }
    static __device__ __forceinline__ float extend(float d, float w) { return fminf(d, w); }          // vect_min(widths[src], edge_width)
    static __device__ __forceinline__ bool better(float cand, float old) { return old < cand; }
    static __device__ __forceinline__ int update(float *p, float cand) { return atomicMax(reinterpret_cast<int *>(p), __float_as_int(cand)); }
    static __device__ __forceinline__ bool improved(int before, float cand) { return before < __float_as_int(cand); }
};

template <class Path>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_sssp_init(int32_t V, int32_t source, float *dist, int32_t *epoch)
{
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK) {
        dist[v] = (v == source) ? Path::source_value() : Path::other_value();
        if (epoch) epoch[v] = (v == source) ? 0 : -4;
    }
}

template <bool ACTIVE_FILTER, class Path>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_sssp_relax(const int64_t *rowptr, const int32_t *adj, const float *w,
                                                              const int32_t *tile_row, int64_t E, int32_t row_base,
                                                              float *dist, int32_t *epoch, int32_t iter, int64_t *counters, int64_t *shards)
{
    __shared__ int s_map[VGL_TILE];
    __shared__ int s_w[VGL_WAVES];
    const int64_t e0 = (int64_t)blockIdx.x * VGL_TILE;
    const int n = (int)min((int64_t)VGL_TILE, E - e0);
    const int r_first = tile_row[blockIdx.x];
    const int r_last = tile_row[blockIdx.x + 1];
    if (ACTIVE_FILTER) {
        int any = 0;
        for (int r = r_first + threadIdx.x; r <= r_last; r += VGL_BLOCK) any |= (epoch[row_base + r] >= iter - 1);
        if (!__syncthreads_or(any)) return;              // nothing in this tile can relax: skip its 16 KB of edge data
    }
    vgl_tile_row_map(s_map, s_w, rowptr, e0, r_first, r_last);

    const int i0 = threadIdx.x * VGL_EPT;
    int changed = 0;
    if (i0 < n) {
        int32_t dsts[VGL_EPT];
        float ws[VGL_EPT];
        if (i0 + VGL_EPT <= n) {
            const int4 a0 = *reinterpret_cast<const int4 *>(adj + e0 + i0);
            const int4 a1 = *reinterpret_cast<const int4 *>(adj + e0 + i0 + 4);
            const float4 w0 = *reinterpret_cast<const float4 *>(w + e0 + i0);
            const float4 w1 = *reinterpret_cast<const float4 *>(w + e0 + i0 + 4);
            dsts[0] = a0.x; dsts[1] = a0.y; dsts[2] = a0.z; dsts[3] = a0.w; dsts[4] = a1.x; dsts[5] = a1.y; dsts[6] = a1.z; dsts[7] = a1.w;
            ws[0] = w0.x; ws[1] = w0.y; ws[2] = w0.z; ws[3] = w0.w; ws[4] = w1.x; ws[5] = w1.y; ws[6] = w1.z; ws[7] = w1.w;
        } else {
#pragma unroll
            for (int j = 0; j < VGL_EPT; j++) {
                const bool ok = i0 + j < n;
                dsts[j] = ok ? adj[e0 + i0 + j] : 0;
                ws[j] = ok ? w[e0 + i0 + j] : 0.0f;
            }
        }
        int rows[VGL_EPT];
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) rows[j] = s_map[i0 + j < n ? i0 + j : i0];       // (i0 < n here)
        // gather phase: every load unconditional and issued before any dependent work (a slot of the same row as its neighbour re-reads
        // the same cached word; a slot that cannot relax reads dist[0]) -- loads under per-lane conditions are compiled as branches that
        // are awaited one after the other, eight dependent round trips per thread in the first version of this kernel
        float olds[VGL_EPT];
        float dsrc[VGL_EPT];
        int32_t ep[VGL_EPT];
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) dsrc[j] = dist[row_base + r_first + rows[j]];
        if (ACTIVE_FILTER) {
#pragma unroll
            for (int j = 0; j < VGL_EPT; j++) ep[j] = epoch[row_base + r_first + rows[j]];
        }
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) {
            bool ok = Path::live(dsrc[j]) && (i0 + j < n);
            if (ACTIVE_FILTER) ok = ok && ep[j] >= iter - 1;
            if (!ok) dsrc[j] = Path::dead_value();
        }
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) olds[j] = dist[Path::live(dsrc[j]) ? dsts[j] : 0];
        // the atomics of a thread's edges are issued together, their return values read afterwards
        float nd[VGL_EPT];
        int before[VGL_EPT];
        bool tried[VGL_EPT];
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) {
            nd[j] = Path::extend(dsrc[j], ws[j]);                    // shortest_paths.hpp:126-130 / widest_paths.hpp:45-50
            tried[j] = Path::live(dsrc[j]) && Path::better(nd[j], olds[j]);
            before[j] = 0;
            if (tried[j]) before[j] = Path::update(dist + dsts[j], nd[j]);
        }
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++)
            if (tried[j] && Path::improved(before[j], nd[j])) {
                changed = 1;
                if (ACTIVE_FILTER) epoch[dsts[j]] = iter;
            }
    }
    const int any_changed = __syncthreads_or(changed);
    if (threadIdx.x == 0) {
        if (any_changed) counters[C_CHANGED] = 1;
        atomicAdd((unsigned long long *)&shards[blockIdx.x & (VGL_NSHARD - 1)], (unsigned long long)n);   // edges streamed (stats)
    }
}

static inline unsigned vgl_grid1(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, vgl_ceil_div(n, VGL_BLOCK))); }

template <class Path>
static int vgl_sssp_launch(vgl_hip_ctx *c, vgl_hip_graph *g, const float *w, float *dist, bool filter, int32_t iter)
{
    if (g->out.ntiles == 0) return 0;
    vgl_timed_launch tl(c, "sssp_relax");
    if (filter)
        hipLaunchKernelGGL((vgl_k_sssp_relax<true, Path>), dim3((unsigned)g->out.ntiles), dim3(VGL_BLOCK), 0, c->stream, g->out.rowptr, g->out.adj, w,
                           g->out.tile_row, g->out.edges, g->row_begin, dist, g->epoch, iter, c->d_counters, c->d_shards);
    else
        hipLaunchKernelGGL((vgl_k_sssp_relax<false, Path>), dim3((unsigned)g->out.ntiles), dim3(VGL_BLOCK), 0, c->stream, g->out.rowptr, g->out.adj, w,
                           g->out.tile_row, g->out.edges, g->row_begin, dist, g->epoch, iter, c->d_counters, c->d_shards);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Pull steps (SSSP::vgl_dijkstra_all_active_pull, shortest_paths.hpp:169-292: every vertex takes the minimum over its incoming
// edges, no scattered stores) as a blocked pass (vgl_blocked.h) over the OUTGOING CSR with the rows as the gather side: dist[src]
// is read from a 128 KiB LDS window, dist[src] (+) w travels to the destination's block, where the minimum is an LDS integer
// atomic; one plain compare + store per vertex ends the step (blocks cut into several units: one global atomic per improved
// vertex and unit).  No random L2 line per edge and no global atomics per edge.  A pull step sees the distances as they were
// when it started (Jacobi), so it moves information one hop per step; a push over the compacted frontier of the rows that changed
// (vgl_k_sssp_relax_sparse: asynchronous, work proportional to their edges) is better when few rows changed.  DIRECTION_OPT switches
// between the two per super-step on the share of the edges whose source changed in the step before -- the fixed point, hence every
// bit of the result, is the same.
// ---------------------------------------------------------------------------------------------------------------------------
template <class Path>
struct vgl_path_blk_op {
    typedef uint32_t acc_t;
    static constexpr bool MARK = true;
    float *dist;
    uint64_t *changed;           // bitmap of the vertices this step improved = the frontier of the next step
    int32_t g_base;
    __device__ __forceinline__ uint32_t load(int32_t i) const { return __float_as_uint(dist[g_base + i]); }
    __device__ __forceinline__ uint32_t edge(uint32_t x, float w) const
    {
        const float d = __uint_as_float(x);
        return __float_as_uint(Path::live(d) ? Path::extend(d, w) : Path::dead_value());
    }
    __device__ __forceinline__ uint32_t identity() const { return __float_as_uint(Path::other_value()); }
    __device__ __forceinline__ void accumulate(uint32_t *p, uint32_t v) const
    {
        if (Path::better(1.0f, 2.0f)) __hip_atomic_fetch_min(reinterpret_cast<int *>(p), (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_max(reinterpret_cast<int *>(p), (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ uint32_t combine(uint32_t a, uint32_t b) const { return Path::better(__uint_as_float(b), __uint_as_float(a)) ? b : a; }
    __device__ __forceinline__ bool finish_m(int32_t v, uint32_t acc) const          // the only unit of v's block: plain compare + store
    {
        const float cand = __uint_as_float(acc);
        if (!Path::better(cand, dist[v])) return false;
        dist[v] = cand;
        return true;
    }
    __device__ __forceinline__ bool partial_m(int32_t v, uint32_t acc) const         // other workgroups may improve v too: one atomic per improvement
    {
        const float cand = __uint_as_float(acc);
        if (!Path::better(cand, dist[v])) return false;
        return Path::improved(Path::update(dist + v, cand), cand);
    }
    __device__ __forceinline__ void mark(int32_t v0, unsigned long long mask, bool shared) const
    {
        if (shared) atomicOr(reinterpret_cast<unsigned long long *>(changed + (v0 >> 6)), mask);
        else changed[v0 >> 6] |= mask;               // (a whole-block unit is the only writer of its words in this launch; fused tiles run after it)
    }
    // (the non-marking interface of vgl_blocked.h, unused: MARK is set)
    __device__ __forceinline__ void finish(int32_t v, uint32_t acc) const { (void)finish_m(v, acc); }
    __device__ __forceinline__ bool partial(int32_t v, uint32_t acc) const { (void)partial_m(v, acc); return true; }
};

__global__ void vgl_k_sssp_seed_bits(int64_t words, int32_t source, uint64_t *front, uint64_t *next)
{
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (int64_t)gridDim.x * blockDim.x) {
        front[w] = (w == (source >> 6)) ? (1ULL << (source & 63)) : 0ULL;
        next[w] = 0;
    }
}

// tile_first[t] = frontier position whose edge range contains edge t * VGL_TILE (defined in bfs.hip, shared by the sparse advances)
__global__ void vgl_k_tile_first(int32_t F, const int64_t *offs, int32_t *tile_first);

// Push over a compacted frontier (the rows whose value changed in the step before: ids + exclusive edge offsets from the GNF),
// edge-balanced like vgl_k_td_expand: workgroup = 2048 consecutive frontier edges.  Work is proportional to the frontier's edges;
// the tile-filtered kernel above still walks every tile's row epochs (>= 0.7 ms per step on RMAT-24 however little changed).
template <class Path>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_sssp_relax_sparse(const int32_t *ids, const int64_t *offs, const int32_t *tile_first, int32_t F, int64_t M,
                                                                     const int64_t *rowptr, const int32_t *adj, const float *w, int32_t row_base,
                                                                     float *dist, uint64_t *changed)
{
    __shared__ int s_map[VGL_TILE];
    __shared__ int64_t s_base[VGL_TILE];
    __shared__ float s_d[VGL_TILE];
    __shared__ int s_w[VGL_WAVES];
    const int64_t e0 = (int64_t)blockIdx.x * VGL_TILE;
    const int n = (int)min((int64_t)VGL_TILE, M - e0);
    const int p_first = tile_first[blockIdx.x];
    const int p_last = tile_first[blockIdx.x + 1];      // last tile: owner of the last edge
    const int np = p_last - p_first + 1;
    const bool staged = np <= VGL_TILE;                 // (more than 2048 frontier positions in one tile: thousands of empty rows)
    if (staged)
        for (int k = threadIdx.x; k < np; k += VGL_BLOCK) {
            const int p = p_first + k;
            const int32_t u = ids[p];
            s_base[k] = rowptr[u - row_base] - offs[p];
            s_d[k] = dist[u];
        }
    vgl_tile_row_map(s_map, s_w, offs, e0, p_first, p_last);      // ends with a barrier: s_base / s_d are visible too
    // Loads are issued in rounds of VGL_EPT with nothing conditional about them (a slot past the tile's end reads the tile's first edge, a
    // slot without a candidate reads dist[0]): a load under a per-lane condition is compiled as a branch whose result is awaited before
    // the next one is issued -- the first version of this kernel made 8 x (adjacency, weight) dependent round trips per thread.
    int32_t dsts[VGL_EPT];
    float cand[VGL_EPT], olds[VGL_EPT];
    if (staged) {
        int64_t es[VGL_EPT];
        float ds[VGL_EPT], wv[VGL_EPT];
        bool ok[VGL_EPT];
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) {
            const int i = threadIdx.x + j * VGL_BLOCK;      // strided slots => coalesced adjacency / weight reads
            ok[j] = i < n;
            const int ii = ok[j] ? i : 0;
            const int k = s_map[ii];
            es[j] = s_base[k] + e0 + ii;
            ds[j] = s_d[k];
        }
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) dsts[j] = adj[es[j]];
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) wv[j] = w[es[j]];
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) {
            cand[j] = ok[j] && Path::live(ds[j]) ? Path::extend(ds[j], wv[j]) : Path::dead_value();
            if (!ok[j]) dsts[j] = -1;
        }
    } else {
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) {
            const int i = threadIdx.x + j * VGL_BLOCK;
            dsts[j] = -1;
            cand[j] = Path::dead_value();
            if (i < n) {
                const int k = s_map[i];
                const int32_t u = ids[p_first + k];
                const int64_t e = rowptr[u - row_base] - offs[p_first + k] + e0 + i;
                const float d = dist[u];
                dsts[j] = adj[e];
                if (Path::live(d)) cand[j] = Path::extend(d, w[e]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) olds[j] = dist[dsts[j] >= 0 && Path::live(cand[j]) ? dsts[j] : 0];
    // the atomics of a thread's edges are issued together, their return values read afterwards
    int before[VGL_EPT];
    bool tried[VGL_EPT];
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) {
        tried[j] = dsts[j] >= 0 && Path::live(cand[j]) && Path::better(cand[j], olds[j]);
        before[j] = 0;
        if (tried[j]) before[j] = Path::update(dist + dsts[j], cand[j]);
    }
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++)
        if (tried[j] && Path::improved(before[j], cand[j]))
            atomicOr(reinterpret_cast<unsigned long long *>(changed + (dsts[j] >> 6)), 1ULL << (dsts[j] & 63));
}

struct vgl_hip_sssp_pull_plan {
    vgl_blocked_plan *blk = nullptr;
    const float *weights = nullptr;
    vgl_hip_graph *g = nullptr;
    uint64_t g_uid = 0;              // the handle's uid: a destroyed graph whose address was reused is not mistaken for the plan's graph
};

template <class Path>
static int vgl_path_run_pull(vgl_hip_ctx *c, vgl_hip_graph *g, const float *d_weights, vgl_hip_sssp_pull_plan *plan, int32_t source, int mode,
                             float *d_dist, vgl_hip_sssp_stats *stats, const char *who)
{
    auto fail = [&](const char *what) { return vgl_set_error(__FILE__, __LINE__, (std::string(who) + ": " + what).c_str()); };
    if (!c || !g || !d_weights || !d_dist || !plan || !plan->blk) return fail("null argument");
    if (plan->g != g || plan->g_uid != g->uid || plan->weights != d_weights) return fail("the pull plan was built for another graph or weight array");
    if (g->row_begin != 0 || g->row_end != g->V) return fail("graph handle must own all rows");
    if (source < 0 || source >= g->V) return fail("source vertex out of range");
    if (mode != VGL_HIP_SSSP_PULL && mode != VGL_HIP_SSSP_DIRECTION_OPT) return fail("unknown mode");
    const char *env = vgl_env(c, "VGL_SSSP_PULL_SHARE");
    const double share = (env && *env) ? atof(env) : 0.2;       // pull when the rows that changed own more than this share of the edges (with the fused-tile
                                                                // pass, RMAT-24, mean of 5 sources: 0.1 15.5 ms, 0.15 15.4, 0.2 15.35, 0.25 15.6, 0.35 15.9, 0.5 17.6)
    hipLaunchKernelGGL(vgl_k_sssp_init<Path>, dim3(vgl_grid1(g->V)), dim3(VGL_BLOCK), 0, c->stream, g->V, source, d_dist, (int32_t *)nullptr);
    vgl_hip_sssp_stats st = {0, 0, 0, 0, 0};
    int64_t pull_edges = 0, push_edges = 0;
    // Every step starts from the frontier of the step before -- the rows whose value changed in it -- kept as a BITMAP that the step's
    // kernels fill (one word per wavefront in the blocked epilogues, one atomicOr per improvement in the push): sizing it reads V / 8 bytes
    // instead of the V * 4 of an epoch array (two 50 us scans per step of RMAT-24 before, 2.8 of 17.7 ms).  An empty frontier ends the run
    // (do { ... } while(changes), shortest_paths.hpp:112-154); its edge share picks the direction of a DIRECTION_OPT step.
    const int64_t words = vgl_ceil_div(g->V, 64);
    const bool debug = vgl_env(c, "VGL_HIP_DEBUG") != nullptr;
    uint64_t *front = g->bm_front, *next = g->bm_next;
    hipLaunchKernelGGL(vgl_k_sssp_seed_bits, dim3(vgl_grid1(words)), dim3(VGL_BLOCK), 0, c->stream, words, source, front, next);
    for (int32_t iter = 1;; iter++) {
        VGL_TRY(vgl_bfs_bm_gnf(c, g, front, true, false));
        const int64_t F = c->h_counters[C_FRONT], M = c->h_counters[C_NEIGH];
        if (F == 0) break;
        st.iterations = iter;
        const bool pull = mode == VGL_HIP_SSSP_PULL || (double)M > share * (double)g->out.edges;
        if (debug) fprintf(stderr, "[vgl_hip] %s step %d: %lld changed rows, %lld edges (%.3f of all) -> %s\n", who, iter, (long long)F, (long long)M,
                           (double)M / (double)std::max<int64_t>(g->out.edges, 1), pull ? "pull" : "push");
        if (pull) {
            const vgl_path_blk_op<Path> op{d_dist, next, g->row_begin};
            VGL_TRY((vgl_blocked_pass<vgl_path_blk_op<Path>, true, false>(c, plan->blk, op, "sssp_pull_gather", "sssp_pull_accumulate", false, "sssp_pull_fused")));
            pull_edges += vgl_blocked_plan_edges(plan->blk);
            st.pull_steps++;
        } else if (M > 0) {
            VGL_TRY(vgl_bfs_bm_gnf(c, g, front, false, true, M));       // ids + edge offsets + tile table of the frontier
            vgl_timed_launch tl(c, "sssp_relax");
            hipLaunchKernelGGL((vgl_k_sssp_relax_sparse<Path>), dim3((unsigned)vgl_ceil_div(M, VGL_TILE)), dim3(VGL_BLOCK), 0, c->stream, (const int32_t *)g->ids,
                               (const int64_t *)g->offs, (const int32_t *)g->tile_first, (int32_t)F, M, g->out.rowptr, g->out.adj, d_weights, g->row_begin, d_dist,
                               next);
            push_edges += M;
            st.push_steps++;
        } else st.push_steps++;                                 // the frontier has no outgoing edges: nothing to relax, the next frontier is empty
        VGL_HIP_TRY(hipGetLastError());
        std::swap(front, next);
        VGL_HIP_TRY(hipMemsetAsync(next, 0, sizeof(uint64_t) * (size_t)words, c->stream));
    }
    st.edges_relaxed = push_edges + pull_edges;
    st.algorithmic_bytes = 12 * st.edges_relaxed + 28 * (int64_t)g->V * st.iterations;
    if (stats) *stats = st;
    return 0;
}

// the path structure of a graph (vgl_hip_graph::blk_path), built on first use and rebuilt when a layout switch changed since
static int vgl_path_structure(vgl_hip_ctx *c, vgl_hip_graph *g)
{
    // pairs of 16384-id blocks with at least VGL_BLK_FUSE_MIN (16384) edges become fused tiles (vgl_blocked.h): on a degree-sorted RMAT graph
    // 82 % of the edges, streamed at 8 instead of 16 bytes each (RMAT-24 pull pass 1.70 -> 1.10 ms; 2048: 94 % fused but 1.78 ms,
    // the sweep over a tile's 16384 accumulators then costs more than its edges; 65536: 71 %, 1.11 ms; 262144: 52 %, 1.30 ms).  Graphs below 2^22 edges stay two-pass unless the variable is set.
    const char *fm = vgl_env(c, "VGL_BLK_FUSE_MIN");
    const int fuse_min = (fm && *fm) ? atoi(fm) : (g->out.edges >= (1LL << 22) ? 16384 : 0);
    std::string key = std::to_string(fuse_min);
    for (const char *name : {"VGL_BLK_GATHER_UNIT", "VGL_BLK_ACCUM_UNIT", "VGL_BLK_FUSED_UNIT", "VGL_BLK_PIECE_EDGES"}) {
        const char *v = vgl_env(c, name);
        key += "|"; key += v ? v : "";
    }
    if (g->blk_path && g->blk_path_key == key) return 0;
    if (g->blk_path) { VGL_HIP_TRY(hipStreamSynchronize(c->stream)); vgl_blocked_plan_destroy(g->blk_path); g->blk_path = nullptr; }
    VGL_TRY(vgl_blocked_plan_build_indexed(c, g->out, g->nrows, g->row_begin, g->V, 1, 0, VGL_BLK_BITS, &g->blk_path, fuse_min));
    g->blk_path_key = key;
    return 0;
}

static int vgl_pull_plan_create(vgl_hip_ctx *c, vgl_hip_graph *g, const float *d_weights, vgl_hip_sssp_pull_plan **out)
{
    if (!c || !g || !d_weights || !out) VGL_FAIL("sssp_pull_plan_create: null argument");
    vgl_hip_sssp_pull_plan *p = new vgl_hip_sssp_pull_plan();
    p->g = g; p->g_uid = g->uid; p->weights = d_weights;
    // Round 5: the layout is a per-GRAPH structure (one radix sort of the edges by block pair, the CSR position behind every value slot kept -- the
    // role of the reference's edges_reorder_indexes, csr_edges_array.hpp:31-40) plus per-WEIGHTS value arrays filled by one gather pass: a second
    // weights array on the same graph costs ~3 ms instead of the 37 ms of a full build (RMAT-24)
    int rc = vgl_path_structure(c, g);
    if (!rc) rc = vgl_blocked_plan_share(c, g->blk_path, &p->blk);
    if (!rc) rc = vgl_blocked_plan_load_weights(c, p->blk, d_weights);
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = vgl_set_error(__FILE__, __LINE__, "sssp_pull_plan_create: the weights pass failed");
    if (rc) { if (p->blk) vgl_blocked_plan_destroy(p->blk); delete p; return rc; }
    *out = p;
    return 0;
}

// shared driver of vgl_hip_sssp_run / vgl_hip_sswp_run: super-steps until a pass changes nothing
template <class Path>
static int vgl_path_run(vgl_hip_ctx *c, vgl_hip_graph *g, const float *d_weights, int32_t source, int mode, float *d_dist, vgl_hip_sssp_stats *stats,
                        const char *who)
{
    auto fail = [&](const char *what) { return vgl_set_error(__FILE__, __LINE__, (std::string(who) + ": " + what).c_str()); };
    if (!c || !g || !d_weights || !d_dist) return fail("null argument");
    if (g->row_begin != 0 || g->row_end != g->V) return fail("graph handle must own all rows (use the *_relax_owned step for shards)");
    if (source < 0 || source >= g->V) return fail("source vertex out of range");
    // the reference's own schedule -- every edge relaxed in every super-step until one changes nothing (shortest_paths.hpp:112-154) -- runs as blocked
    // passes once the graph carries the path structure (vgl_hip_sssp_prepare, or any pull plan built before): the same fixed point, hence the same
    // bits, at the rate of the LDS-window pass instead of one atomic and one random line per edge.  VGL_SSSP_ALL_ACTIVE_PUSH=1 keeps the atomic kernel.
    if (mode == VGL_HIP_SSSP_ALL_ACTIVE && g->blk_path) {
        const char *push = vgl_env(c, "VGL_SSSP_ALL_ACTIVE_PUSH");
        if (!(push && push[0] == '1')) mode = VGL_HIP_SSSP_PULL;
    }
    if (mode == VGL_HIP_SSSP_PULL || mode == VGL_HIP_SSSP_DIRECTION_OPT) {      // one-off: the layout's value arrays for these weights (the structure stays with the graph)
        vgl_hip_sssp_pull_plan *plan = nullptr;
        VGL_TRY(vgl_pull_plan_create(c, g, d_weights, &plan));
        const int rc = vgl_path_run_pull<Path>(c, g, d_weights, plan, source, mode, d_dist, stats, who);
        hipStreamSynchronize(c->stream);
        vgl_blocked_plan_destroy(plan->blk);
        delete plan;
        return rc;
    }
    if (mode != VGL_HIP_SSSP_ALL_ACTIVE && mode != VGL_HIP_SSSP_ACTIVE_TILES) return fail("unknown mode");
    const bool filter = mode == VGL_HIP_SSSP_ACTIVE_TILES;
    hipLaunchKernelGGL(vgl_k_sssp_init<Path>, dim3(vgl_grid1(g->V)), dim3(VGL_BLOCK), 0, c->stream, g->V, source, d_dist, g->epoch);
    VGL_TRY(vgl_zero_counters(c, C_EDGES, 1));
    vgl_hip_sssp_stats st = {0, 0, 0, 0, 0};
    for (int32_t iter = 1;; iter++) {
        VGL_TRY(vgl_zero_counters(c, C_CHANGED, 1));
        VGL_TRY(vgl_sssp_launch<Path>(c, g, d_weights, d_dist, filter, iter));
        st.push_steps++;
        VGL_TRY(vgl_read_counters(c));
        st.iterations = iter;
        if (!c->h_counters[C_CHANGED]) break;      // do { ... } while(changes)  (shortest_paths.hpp:112-154, widest_paths.hpp:34-64)
    }
    st.edges_relaxed = c->h_counters[C_EDGES];
    st.algorithmic_bytes = 12 * st.edges_relaxed + 28 * (int64_t)g->V * st.iterations;
    if (stats) *stats = st;
    return 0;
}

// one all-active push relaxation over the owned rows, enqueued only (the sharded loop learns what changed from its exchange)
int vgl_sssp_relax_enqueue(vgl_hip_ctx *c, vgl_hip_graph *g, const float *d_weights, float *d_values, bool widest)
{
    return widest ? vgl_sssp_launch<vgl_path_widest>(c, g, d_weights, d_values, false, 1)
                  : vgl_sssp_launch<vgl_path_shortest>(c, g, d_weights, d_values, false, 1);
}

extern "C" {

int vgl_hip_sssp_prepare(vgl_hip_ctx *c, vgl_hip_graph *g)
{
    if (!c || !g) VGL_FAIL("sssp_prepare: null argument");
    if (g->row_begin != 0 || g->row_end != g->V) VGL_FAIL("sssp_prepare: graph handle must own all rows");
    VGL_TRY(vgl_path_structure(c, g));
    VGL_HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int vgl_hip_sssp_pull_plan_create(vgl_hip_ctx *c, vgl_hip_graph *g, const float *d_weights, vgl_hip_sssp_pull_plan **out)
{
    return vgl_pull_plan_create(c, g, d_weights, out);
}
int vgl_hip_sssp_pull_plan_info(vgl_hip_sssp_pull_plan *p, int64_t *edges, int64_t *fused_edges, int64_t *streamed_bytes_per_pass, int64_t *plan_bytes)
{
    if (!p || !p->blk) VGL_FAIL("sssp_pull_plan_info: null plan");
    int64_t two_pass_slots = 0, fused_slots = 0, e_all = 0, f_all = 0, nch = 0;
    for (const vgl_blocked_plan *b = p->blk; b; b = b->next) {          // (a direction with 2^32 edges or more is laid out in pieces)
        two_pass_slots += (int64_t)b->nchunks * VGL_CHUNK; fused_slots += (int64_t)b->f_nchunks * VGL_CHUNK;
        e_all += b->edges; f_all += b->f_edges; nch += b->nchunks;
    }
    if (edges) *edges = e_all;
    if (fused_edges) *fused_edges = f_all;
    // what one pass streams, pad entries included: two-pass 2 + 4 + 4 (gather) + 2 + 4 (accumulate) per slot, fused 2 + 2 + 4 per slot
    if (streamed_bytes_per_pass) *streamed_bytes_per_pass = 16 * two_pass_slots + 8 * fused_slots;
    if (plan_bytes) *plan_bytes = 12 * two_pass_slots + 4 * nch + 8 * fused_slots;
    return 0;
}
int vgl_hip_sssp_pull_plan_destroy(vgl_hip_ctx *c, vgl_hip_sssp_pull_plan *p)
{
    if (!p) return 0;
    if (c) hipStreamSynchronize(c->stream);
    vgl_blocked_plan_destroy(p->blk);
    delete p;
    return 0;
}
// ONE all-edges relaxation as a blocked pass over the plan (dist[dst] = min(dist[dst], dist[src] + w) for every edge, the values of the pass
// start): the step behind the operator class's declared relax.  *changed = 1 when a distance decreased.  Synchronises.
int vgl_hip_sssp_pull_pass(vgl_hip_ctx *c, vgl_hip_graph *g, vgl_hip_sssp_pull_plan *plan, float *d_dist, int *changed)
{
    if (!c || !g || !plan || !plan->blk || !d_dist) VGL_FAIL("sssp_pull_pass: null argument");
    if (plan->g != g || plan->g_uid != g->uid) VGL_FAIL("sssp_pull_pass: the pull plan was built for another graph");
    if (g->row_begin != 0 || g->row_end != g->V) VGL_FAIL("sssp_pull_pass: graph handle must own all rows");
    const int64_t words = vgl_ceil_div(g->V, 64);
    VGL_TRY(vgl_zero_words(c, g->bm_next, words));
    const vgl_path_blk_op<vgl_path_shortest> op{d_dist, g->bm_next, g->row_begin};
    VGL_TRY((vgl_blocked_pass<vgl_path_blk_op<vgl_path_shortest>, true, false>(c, plan->blk, op, "sssp_pull_gather", "sssp_pull_accumulate", false, "sssp_pull_fused")));
    VGL_TRY(vgl_bfs_bm_gnf(c, g, g->bm_next, true, false));          // (the improved vertices were marked in the bitmap: its size says whether any were)
    if (changed) *changed = c->h_counters[C_FRONT] != 0;
    return 0;
}
int vgl_hip_sssp_run_pull(vgl_hip_ctx *c, vgl_hip_graph *g, const float *d_weights, vgl_hip_sssp_pull_plan *plan, int32_t source, int mode,
                          float *d_dist, vgl_hip_sssp_stats *stats)
{
    return vgl_path_run_pull<vgl_path_shortest>(c, g, d_weights, plan, source, mode, d_dist, stats, "sssp_run_pull");
}
int vgl_hip_sswp_run_pull(vgl_hip_ctx *c, vgl_hip_graph *g, const float *d_capacities, vgl_hip_sssp_pull_plan *plan, int32_t source, int mode,
                          float *d_widths, vgl_hip_sssp_stats *stats)
{
    return vgl_path_run_pull<vgl_path_widest>(c, g, d_capacities, plan, source, mode, d_widths, stats, "sswp_run_pull");
}

int vgl_hip_sssp_init(vgl_hip_ctx *c, int32_t V, int32_t source, float *d_dist)
{
    if (!c || !d_dist) VGL_FAIL("sssp_init: null argument");
    if (source < 0 || source >= V) VGL_FAIL("sssp_init: source vertex out of range");
    hipLaunchKernelGGL(vgl_k_sssp_init<vgl_path_shortest>, dim3(vgl_grid1(V)), dim3(VGL_BLOCK), 0, c->stream, V, source, d_dist, (int32_t *)nullptr);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_sssp_run(vgl_hip_ctx *c, vgl_hip_graph *g, const float *d_weights, int32_t source, int mode, float *d_dist,
                     vgl_hip_sssp_stats *stats)
{
    return vgl_path_run<vgl_path_shortest>(c, g, d_weights, source, mode, d_dist, stats, "sssp_run");
}

int vgl_hip_sswp_run(vgl_hip_ctx *c, vgl_hip_graph *g, const float *d_capacities, int32_t source, int mode, float *d_widths,
                     vgl_hip_sssp_stats *stats)
{
    return vgl_path_run<vgl_path_widest>(c, g, d_capacities, source, mode, d_widths, stats, "sswp_run");
}

int vgl_hip_sswp_init(vgl_hip_ctx *c, int32_t V, int32_t source, float *d_widths)
{
    if (!c || !d_widths) VGL_FAIL("sswp_init: null argument");
    if (source < 0 || source >= V) VGL_FAIL("sswp_init: source vertex out of range");
    hipLaunchKernelGGL(vgl_k_sssp_init<vgl_path_widest>, dim3(vgl_grid1(V)), dim3(VGL_BLOCK), 0, c->stream, V, source, d_widths, (int32_t *)nullptr);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_sswp_relax_owned(vgl_hip_ctx *c, vgl_hip_graph *g, const float *d_capacities, float *d_widths, int *changed)
{
    if (!c || !g || !d_capacities || !d_widths) VGL_FAIL("sswp_relax_owned: null argument");
    VGL_TRY(vgl_zero_counters(c, C_CHANGED, 1));
    VGL_TRY(vgl_sssp_launch<vgl_path_widest>(c, g, d_capacities, d_widths, false, 1));
    VGL_TRY(vgl_read_counters(c));
    if (changed) *changed = (int)c->h_counters[C_CHANGED];
    return 0;
}

int vgl_hip_sssp_relax_owned(vgl_hip_ctx *c, vgl_hip_graph *g, const float *d_weights, float *d_dist, int *changed)
{
    if (!c || !g || !d_weights || !d_dist) VGL_FAIL("sssp_relax_owned: null argument");
    VGL_TRY(vgl_zero_counters(c, C_CHANGED, 1));
    VGL_TRY(vgl_sssp_launch<vgl_path_shortest>(c, g, d_weights, d_dist, false, 1));
    VGL_TRY(vgl_read_counters(c));
    if (changed) *changed = (int)c->h_counters[C_CHANGED];
    return 0;
}

}  // extern "C"
