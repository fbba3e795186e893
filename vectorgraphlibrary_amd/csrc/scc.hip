// scc.hip -- strongly connected components (SCC::vgl_forward_backward, algorithms/scc/scc.hpp; checker seq_tarjan,
// seq_scc.hpp).  Output: comp[v] = smallest vertex id of v's component -- the canonical form of the PARTITION, which is what
// the reference's test compares (equal_components) and the only thing its counter-style labels define.
//
// Schedule (same ingredients as the reference: trimming, one forward/backward reach for the big component; the recursion on
// the three remaining sets is replaced by colour propagation, which handles thousands of small components per round):
//   1. trim      : a vertex without active in- or out-neighbours is its own component; removing it lowers the counters of its
//                  neighbours (worklist-free: every round scans the vertices, a removed vertex walks its own edges once);
//   2. giant     : pivot = active vertex with the largest min(in, out) count; FW = direction-optimising BFS on the graph, BW =
//                  the same on the transposed graph handle (cached); FW /\ BW over the WHOLE graph is exactly the pivot's
//                  component (the reference does not mask its reach either, scc.hpp:128-172);
//   3. leftovers : repeat { colour[v] = v; push the MAXIMUM colour along active edges to a fixed point (a vertex's colour is the
//                  largest id that reaches it); roots are the vertices that kept their own colour; within a colour class the
//                  vertices that reach the root form the root's component (pull along out-edges to a fixed point); remove them;
//                  recount and trim } until nothing is active.
// Edge passes are the usual 2048-edge tiles; tiles whose rows are all inactive are skipped before touching the adjacency.
// NOTE: the reference's forward-backward code itself returns wrong partitions on sparse inputs with thousands of small
// components (recorded in tests/golden/scc_ru_s12_e1_seed5.npz); its Tarjan checker is the ground truth used here.
#include "vgl_hip_internal.h"
#include <climits>

static inline unsigned scc_grid(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, vgl_ceil_div(n, VGL_BLOCK))); }

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_scc_init(int32_t V, int32_t *act, int32_t *comp)
{
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK) { act[v] = 1; comp[v] = -1; }
}

// cnt[r] = number of ACTIVE neighbours x != r of every active row r of one CSR direction (out: successors, in: predecessors).
// ALL (everything is active, the first count of a run): cnt was preset to the row lengths, this pass only takes the self loops off
// again (rare atomics).  Otherwise the counts of a tile are gathered in LDS first: rows that lie inside the tile are stored, only
// the two boundary rows (shared with the neighbouring tiles) are added atomically -- a hub row spanning hundreds of tiles used to
// receive an atomic from every thread, and in a degree-sorted graph those counters share cache lines (2x slower than unsorted).
template <bool ALL>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_scc_count_active(const int64_t *rowptr, const int32_t *adj, const int32_t *tile_row, int64_t E,
                                                                    const int32_t *act, int32_t *cnt)
{
    __shared__ int s_map[VGL_TILE];
    __shared__ int s_cnt[VGL_TILE];
    __shared__ int s_w[VGL_WAVES];
    const int64_t e0 = (int64_t)blockIdx.x * VGL_TILE;
    const int n = (int)min((int64_t)VGL_TILE, E - e0);
    const int r_first = tile_row[blockIdx.x], r_last = tile_row[blockIdx.x + 1];
    if (!ALL) {
        int any = 0;
        for (int r = r_first + threadIdx.x; r <= r_last; r += VGL_BLOCK) any |= act[r];
        if (!__syncthreads_or(any)) return;
    }
    const int nrows_here = min(r_last - r_first + 1, VGL_TILE);          // rows that START at or before the tile's last edge
    for (int i = threadIdx.x; i < VGL_TILE; i += VGL_BLOCK) s_cnt[i] = 0;
    vgl_tile_row_map(s_map, s_w, rowptr, e0, r_first, r_last);          // ends with a barrier
    const int i0 = threadIdx.x * VGL_EPT;
    int prev_row = -1, run = 0;
    bool live = false;
    // a tile of 2048 edges can span MORE than 2048 rows when many of them are empty: rows beyond the LDS table go straight to memory
    auto flush = [&](int row, int value) {
        if (!value) return;
        if (row < VGL_TILE) atomicAdd(&s_cnt[row], value);
        else if (ALL) atomicSub(cnt + r_first + row, value);
        else atomicAdd(cnt + r_first + row, value);
    };
    for (int j = 0; j < VGL_EPT && i0 + j < n; j++) {
        const int row = s_map[i0 + j];
        if (row != prev_row) {
            flush(prev_row, run);
            prev_row = row; run = 0;
            live = ALL || act[r_first + row] != 0;
        }
        if (live) {
            const int32_t x = adj[e0 + i0 + j];
            run += ALL ? (x == r_first + row) : ((x != r_first + row) && act[x]);          // ALL: count the self loops
        }
    }
    flush(prev_row, run);
    __syncthreads();
    // flush: s_map rows are 0 .. (number of rows with an edge in this tile - 1) relative to r_first; rows without edges here have 0
    for (int k = threadIdx.x; k < nrows_here; k += VGL_BLOCK) {
        const int v = s_cnt[k];
        if (!v) continue;
        const int32_t r = r_first + k;
        const bool inside = rowptr[r] >= e0 && rowptr[r + 1] <= e0 + n;          // no other tile sees this row
        if (ALL) atomicSub(cnt + r, v);
        else if (inside) cnt[r] = v;
        else atomicAdd(cnt + r, v);
    }
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_scc_row_lengths(int32_t V, const int64_t *rowptr, int32_t *cnt)
{
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK) cnt[v] = (int32_t)(rowptr[v + 1] - rowptr[v]);
}

// one trimming round: active vertices with no active predecessor or no active successor become singleton components; each
// removed vertex takes itself out of its neighbours' counters
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_scc_trim(int32_t V, const int64_t *out_rowptr, const int32_t *out_adj, const int64_t *in_rowptr,
                                                            const int32_t *in_adj, int32_t *act, int32_t *od, int32_t *id, int32_t *comp,
                                                            int64_t *counters)
{
    int changed = 0;
    const int32_t vround = (V + 63) & ~63;
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < vround; v += gridDim.x * VGL_BLOCK) {
        const bool removed = v < V && act[v] && (od[v] <= 0 || id[v] <= 0) && atomicExch(act + v, 0) == 1;
        bool wide = false;
        if (removed) {
            comp[v] = v;
            changed = 1;
            // short rows: the thread walks its own edges; long rows (a removed hub, e.g. a vertex with thousands of out-edges and no
            // in-edge) are walked by the whole wavefront below -- in a degree-sorted graph they sit next to each other
            wide = (out_rowptr[v + 1] - out_rowptr[v]) + (in_rowptr[v + 1] - in_rowptr[v]) >= 256;
            if (!wide) {
                for (int64_t p = out_rowptr[v]; p < out_rowptr[v + 1]; p++) { const int32_t w = out_adj[p]; if (w != v && act[w]) atomicSub(id + w, 1); }
                for (int64_t p = in_rowptr[v]; p < in_rowptr[v + 1]; p++) { const int32_t u = in_adj[p]; if (u != v && act[u]) atomicSub(od + u, 1); }
            }
        }
        unsigned long long todo = __ballot(wide);
        while (todo) {
            const int l = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int32_t x = __shfl(v, l);
            for (int64_t p = out_rowptr[x] + vgl_lane(); p < out_rowptr[x + 1]; p += 64) { const int32_t w = out_adj[p]; if (w != x && act[w]) atomicSub(id + w, 1); }
            for (int64_t p = in_rowptr[x] + vgl_lane(); p < in_rowptr[x + 1]; p += 64) { const int32_t u = in_adj[p]; if (u != x && act[u]) atomicSub(od + u, 1); }
        }
    }
    if (__syncthreads_or(changed) && threadIdx.x == 0) counters[C_CHANGED] = 1;
}

// counters[C_TMP0] = number of active vertices; counters[C_TMP1] = max over active v of (min(od, id) << 32 | v)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_scc_survey(int32_t V, const int32_t *act, const int32_t *od, const int32_t *id, int64_t *counters)
{
    __shared__ int64_t s64[VGL_WAVES];
    int64_t n = 0;
    unsigned long long best = 0;
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK)
        if (act[v]) {
            n++;
            const unsigned long long key = ((unsigned long long)(uint32_t)max(0, min(od[v], id[v])) << 32) | (uint32_t)v;
            best = max(best, key);
        }
    n = vgl_block_reduce_add(n, s64);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(best, o); best = max(best, t); }
    __syncthreads();
    if (vgl_lane() == 0) s64[vgl_wave()] = (int64_t)best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 0; w < VGL_WAVES; w++) best = max(best, (unsigned long long)s64[w]);
        if (n) atomicAdd((unsigned long long *)&counters[C_TMP0], (unsigned long long)n);      // <= 8192 workgroups, once per phase
        if (best) atomicMax((unsigned long long *)&counters[C_TMP1], best);
    }
}

// members of the pivot's component: active, reached forwards and backwards.  pass 0: smallest id -> counters[C_JUMP] (atomicMin);
// pass 1: label + deactivate
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_scc_intersect(int32_t V, const int32_t *fw, const int32_t *bw, int32_t *act, int32_t *comp,
                                                                 int pass, int64_t *counters)
{
    long long mn = LLONG_MAX;
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK)
        if (act[v] && fw[v] != -1 && bw[v] != -1) {
            if (pass == 0) mn = min(mn, (long long)v);
            else { comp[v] = (int32_t)counters[C_JUMP]; act[v] = 0; }
        }
    if (pass == 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mn = min(mn, __shfl_xor(mn, o));
        if (vgl_lane() == 0 && mn != LLONG_MAX) atomicMin((long long *)&counters[C_JUMP], mn);
    }
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_scc_colour_init(int32_t V, const int32_t *act, int32_t *colour, int32_t *cnt_a, int32_t *cnt_b)
{
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK) {
        colour[v] = act[v] ? v : -1;
        if (cnt_a) { cnt_a[v] = 0; cnt_b[v] = 0; }
    }
}

// MODE 0: colour[w] = max(colour[w], colour[u]) along active edges u -> w.
// MODE 1: reach[u] = 1 if an active successor w of the same colour is already marked (u can get back to the root of its class).
template <int MODE>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_scc_edge_pass(const int64_t *rowptr, const int32_t *adj, const int32_t *tile_row, int64_t E,
                                                                 const int32_t *act, int32_t *colour, int32_t *reach, int64_t *counters)
{
    __shared__ int s_map[VGL_TILE];
    __shared__ int s_w[VGL_WAVES];
    const int64_t e0 = (int64_t)blockIdx.x * VGL_TILE;
    const int n = (int)min((int64_t)VGL_TILE, E - e0);
    const int r_first = tile_row[blockIdx.x], r_last = tile_row[blockIdx.x + 1];
    int any = 0;
    for (int r = r_first + threadIdx.x; r <= r_last; r += VGL_BLOCK) any |= act[r] && (MODE == 0 || !reach[r]);
    if (!__syncthreads_or(any)) return;
    vgl_tile_row_map(s_map, s_w, rowptr, e0, r_first, r_last);
    int changed = 0;
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) {
        const int i = threadIdx.x + j * VGL_BLOCK;
        if (i < n) {
            const int32_t u = r_first + s_map[i];
            if (act[u] && (MODE == 0 || !reach[u])) {
                const int32_t w = adj[e0 + i];
                if (w != u && act[w]) {
                    if (MODE == 0) {
                        const int32_t cu = colour[u];
                        if (colour[w] < cu) { atomicMax(colour + w, cu); changed = 1; }
                    } else if (reach[w] && colour[w] == colour[u]) { reach[u] = 1; changed = 1; }
                }
            }
        }
    }
    if (__syncthreads_or(changed) && threadIdx.x == 0) counters[C_CHANGED] = 1;
}

// pass 0: roots mark themselves (reach) and reset their slot of minrep; pass 1: marked vertices lower minrep[colour] to their id;
// pass 2: marked vertices take the label and leave
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_scc_classes(int32_t V, int32_t *act, const int32_t *colour, int32_t *reach, int32_t *minrep,
                                                               int32_t *comp, int pass)
{
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK) {
        if (pass == 0) { const bool root = act[v] && colour[v] == v; reach[v] = root; if (root) minrep[v] = v; }
        else if (act[v] && reach[v]) {
            if (pass == 1) atomicMin(minrep + colour[v], v);
            else { comp[v] = minrep[colour[v]]; act[v] = 0; }
        }
    }
}

extern "C" {

int vgl_hip_scc_run(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t *d_comp, vgl_hip_scc_stats *stats)
{
    if (!c || !g || !d_comp) VGL_FAIL("scc_run: null argument");
    if (g->row_begin != 0 || g->row_end != g->V) VGL_FAIL("scc_run: graph handle must own all rows");
    if (!g->in.rowptr) VGL_FAIL("scc_run: the incoming CSR is required (backward reach, predecessor counts)");
    const int32_t V = g->V;
    hipStream_t st = c->stream;
    if (!g->transposed)
        VGL_TRY(vgl_hip_graph_create(c, V, 0, V, g->in.rowptr, g->in.adj, g->in.edges, g->out.rowptr, g->out.adj, g->out.edges, &g->transposed));
    int32_t *buf = nullptr;                      // act | od | id | colour | reach | minrep | fw | bw
    VGL_HIP_TRY(hipMalloc((void **)&buf, sizeof(int32_t) * 8 * (size_t)V));
    int32_t *act = buf, *od = buf + (size_t)V, *id = buf + 2 * (size_t)V, *colour = buf + 3 * (size_t)V, *reach = buf + 4 * (size_t)V,
            *minrep = buf + 5 * (size_t)V, *fw = buf + 6 * (size_t)V, *bw = buf + 7 * (size_t)V;
    vgl_hip_scc_stats s = {0, 0, 0, 0};
    int rc = 0;
    auto fail = [&](int code) { hipStreamSynchronize(st); hipFree(buf); return code; };
#define SCC_TRY(expr) do { rc = (expr); if (rc != 0) return fail(rc); } while (0)
    auto recount = [&](bool all) -> int {
        hipLaunchKernelGGL(vgl_k_scc_colour_init, dim3(scc_grid(V)), dim3(VGL_BLOCK), 0, st, V, act, colour, od, id);      // zeroes od / id too
        for (int k = 0; k < 2; k++) {
            const vgl_dir_csr &d = k == 0 ? g->out : g->in;
            int32_t *cnt = k == 0 ? od : id;
            if (d.ntiles == 0) continue;
            if (all) {
                hipLaunchKernelGGL(vgl_k_scc_row_lengths, dim3(scc_grid(V)), dim3(VGL_BLOCK), 0, st, V, d.rowptr, cnt);
                hipLaunchKernelGGL(vgl_k_scc_count_active<true>, dim3((unsigned)d.ntiles), dim3(VGL_BLOCK), 0, st, d.rowptr, d.adj, d.tile_row, d.edges, act, cnt);
            }
            else hipLaunchKernelGGL(vgl_k_scc_count_active<false>, dim3((unsigned)d.ntiles), dim3(VGL_BLOCK), 0, st, d.rowptr, d.adj, d.tile_row, d.edges, act, cnt);
        }
        VGL_HIP_TRY(hipGetLastError());
        return 0;
    };
    auto trim = [&]() -> int {
        for (;;) {
            VGL_TRY(vgl_zero_counters(c, C_CHANGED, 1));
            hipLaunchKernelGGL(vgl_k_scc_trim, dim3(scc_grid(V)), dim3(VGL_BLOCK), 0, st, V, g->out.rowptr, g->out.adj, g->in.rowptr, g->in.adj, act,
                               od, id, d_comp, c->d_counters);
            VGL_HIP_TRY(hipGetLastError());
            VGL_TRY(vgl_read_counters(c, false));
            s.trim_rounds++;
            if (!c->h_counters[C_CHANGED]) return 0;
        }
    };
    auto survey = [&](int64_t *active, int32_t *pivot, int32_t *pivot_key) -> int {
        VGL_TRY(vgl_zero_counters(c, C_TMP0, 2));
        hipLaunchKernelGGL(vgl_k_scc_survey, dim3(scc_grid(V)), dim3(VGL_BLOCK), 0, st, V, act, od, id, c->d_counters);
        VGL_HIP_TRY(hipGetLastError());
        VGL_TRY(vgl_read_counters(c, false));
        *active = c->h_counters[C_TMP0];
        *pivot = (int32_t)(uint32_t)(c->h_counters[C_TMP1] & 0xffffffffLL);
        *pivot_key = (int32_t)((uint64_t)c->h_counters[C_TMP1] >> 32);
        return 0;
    };
    hipLaunchKernelGGL(vgl_k_scc_init, dim3(scc_grid(V)), dim3(VGL_BLOCK), 0, st, V, act, d_comp);
    SCC_TRY(recount(true));
    SCC_TRY(trim());
    int64_t active = 0;
    int32_t pivot = 0, key = 0;
    SCC_TRY(survey(&active, &pivot, &key));
    if (active > 0 && key > 0) {                 // the big component: forward and backward reach of the best-connected vertex
        SCC_TRY(vgl_hip_bfs_run(c, g, pivot, VGL_HIP_BFS_DIRECTION_OPT, fw, nullptr));
        SCC_TRY(vgl_hip_bfs_run(c, g->transposed, pivot, VGL_HIP_BFS_DIRECTION_OPT, bw, nullptr));
        const int64_t big = LLONG_MAX;
        SCC_TRY(vgl_hip_memcpy_h2d(c, c->d_counters + C_JUMP, &big, sizeof(int64_t)));
        hipLaunchKernelGGL(vgl_k_scc_intersect, dim3(scc_grid(V)), dim3(VGL_BLOCK), 0, st, V, fw, bw, act, d_comp, 0, c->d_counters);
        hipLaunchKernelGGL(vgl_k_scc_intersect, dim3(scc_grid(V)), dim3(VGL_BLOCK), 0, st, V, fw, bw, act, d_comp, 1, c->d_counters);
        s.forward_backward_steps++;
        SCC_TRY(recount(false));
        SCC_TRY(trim());
        SCC_TRY(survey(&active, &pivot, &key));
    }
    while (active > 0) {                         // leftovers: colour classes
        hipLaunchKernelGGL(vgl_k_scc_colour_init, dim3(scc_grid(V)), dim3(VGL_BLOCK), 0, st, V, act, colour, (int32_t *)nullptr, (int32_t *)nullptr);
        for (int mode = 0; mode < 2; mode++) {
            if (mode == 1) hipLaunchKernelGGL(vgl_k_scc_classes, dim3(scc_grid(V)), dim3(VGL_BLOCK), 0, st, V, act, colour, reach, minrep, d_comp, 0);
            for (;;) {
                SCC_TRY(vgl_zero_counters(c, C_CHANGED, 1));
                if (g->out.ntiles > 0) {
                    if (mode == 0)
                        hipLaunchKernelGGL(vgl_k_scc_edge_pass<0>, dim3((unsigned)g->out.ntiles), dim3(VGL_BLOCK), 0, st, g->out.rowptr, g->out.adj,
                                           g->out.tile_row, g->out.edges, act, colour, reach, c->d_counters);
                    else
                        hipLaunchKernelGGL(vgl_k_scc_edge_pass<1>, dim3((unsigned)g->out.ntiles), dim3(VGL_BLOCK), 0, st, g->out.rowptr, g->out.adj,
                                           g->out.tile_row, g->out.edges, act, colour, reach, c->d_counters);
                }
                if (hipGetLastError() != hipSuccess) return fail(vgl_set_error(__FILE__, __LINE__, "scc_run: kernel launch failed"));
                SCC_TRY(vgl_read_counters(c, false));
                s.edge_passes++;
                if (!c->h_counters[C_CHANGED]) break;
            }
        }
        hipLaunchKernelGGL(vgl_k_scc_classes, dim3(scc_grid(V)), dim3(VGL_BLOCK), 0, st, V, act, colour, reach, minrep, d_comp, 1);
        hipLaunchKernelGGL(vgl_k_scc_classes, dim3(scc_grid(V)), dim3(VGL_BLOCK), 0, st, V, act, colour, reach, minrep, d_comp, 2);
        s.colour_rounds++;
        SCC_TRY(recount(false));
        SCC_TRY(trim());
        SCC_TRY(survey(&active, &pivot, &key));
    }
#undef SCC_TRY
    VGL_HIP_TRY(hipStreamSynchronize(st));
    hipFree(buf);
    if (stats) *stats = s;
    return 0;
}

}  // extern "C"
