// sssp_delta.hip -- SSSP with bucketed scheduling (delta-stepping with a light / heavy edge split) on the same
// edge-tile relax machinery as sssp.hip.  Same operators as SSSP::vgl_dijkstra_* (algorithms/sssp/shortest_paths.hpp:
// relax d[dst] = min(d[dst], d[src] + w) in f32), different SCHEDULE: because the fixed point of the relaxation is unique,
// the distances are bit-identical to the reference's Bellman-Ford / Dijkstra results (tests assert this).
//
// Why: an all-edges relax pass costs ~3.4 ms on RMAT-24 of which ~0.7 ms is the coalesced adjacency+weight stream and the
// rest is the per-edge 4-byte gather of dist[dst] (L2 request bound).  Plain Bellman-Ford re-relaxes every out-edge of a
// vertex each time its distance improves (hubs improve many times): ~5 E gathers.  Here a vertex's LIGHT edges (w < delta)
// are relaxed whenever it improves inside the current bucket [.., T), its HEAVY edges only once the bucket has settled:
// ~1.2 E gathers (measured by simulation on RMAT-18), the stream is re-read but that is the cheap part.
//
// Robustness rule (no reliance on bucket theory for correctness): EVERY improvement of d[v] sets both dirty[v] (light
// edges pending) and heavy[v] (heavy edges pending); a flag is cleared only by the select pass that schedules the row, so
// every improvement is eventually followed by a relaxation of all out-edges => the loop ends exactly at the fixed point.
//
// Per step: vgl_k_ds_select (V scan: 6 B/vertex; builds the active-row bitmap + active-tile bytes, clears scheduled flags,
// reduces counters without same-address atomics) -> vgl_k_ds_relax (persistent workgroups stride over the tiles, skip
// inactive ones after a 1-byte probe) -> one host read of 4 counters.
#include "vgl_hip_internal.h"
#include <cfloat>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>

constexpr int VGL_DS_BLOCKS = 1024;       // persistent grid of both kernels

// state[v]: bit 0 = light edges pending ("dirty"), bit 1 = heavy edges pending
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_ds_init(int32_t V, int32_t source, float *dist, uint8_t *state)
{
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK) {
        const bool s = v == source;
        dist[v] = s ? 0.0f : FLT_MAX;
        state[v] = s ? 3 : 0;
    }
}

// mode 0: schedule rows with (state & 1) && d < T (light pass); mode 1: rows with (state & 2) && d < T (heavy pass).
// Thread = 8 consecutive vertices: one 8-byte load of their state bytes (all zero for most vertices most of the time), one
// byte of the active-row bitmap written back.  partials[b*4 + {0,1,2}] = scheduled rows, rows with heavy pending and d < T,
// min d over flagged rows with d >= T (as int bits; non-negative floats order like ints).
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_ds_select(int32_t nrows, int32_t row_base, const int64_t *__restrict__ rowptr,
                                                             const float *__restrict__ dist, uint8_t *__restrict__ state, float T, int mode,
                                                             uint8_t *__restrict__ active_bytes, uint8_t *__restrict__ tile_active,
                                                             int64_t *__restrict__ partials)
{
    __shared__ int64_t s64[VGL_WAVES];
    __shared__ int s32[VGL_WAVES];
    int64_t n_sched = 0, n_heavy_near = 0;
    int min_far = __float_as_int(FLT_MAX);
    const uint8_t bit = mode == 0 ? 1 : 2;
    const int32_t ngroups = (nrows + 7) >> 3;
    for (int32_t gidx = blockIdx.x * VGL_BLOCK + threadIdx.x; gidx < ngroups; gidx += gridDim.x * VGL_BLOCK) {
        const int32_t r0 = gidx << 3;
        const int32_t v0 = row_base + r0;
        const int nvalid = min(8, nrows - r0);
        uint64_t st8 = 0;
        if (nvalid == 8) st8 = *reinterpret_cast<const uint64_t *>(state + v0);
        else for (int j = 0; j < nvalid; j++) st8 |= (uint64_t)state[v0 + j] << (8 * j);
        uint32_t act = 0;
        if (st8) {
            uint64_t st_new = st8;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint8_t f = (uint8_t)(st8 >> (8 * j));
                if (f) {
                    const float d = dist[v0 + j];
                    if (d < T) {
                        if (f & bit) {
                            act |= 1u << j;
                            st_new &= ~((uint64_t)bit << (8 * j));
                            const int64_t b = rowptr[r0 + j], e = rowptr[r0 + j + 1];
                            if (e > b) for (int64_t t = b / VGL_TILE; t * VGL_TILE < e; t++) tile_active[t] = 1;
                        }
                        n_heavy_near += (mode == 0) && (f & 2);
                    } else {
                        min_far = min(min_far, __float_as_int(d));
                    }
                }
            }
            if (st_new != st8) {
                if (nvalid == 8) *reinterpret_cast<uint64_t *>(state + v0) = st_new;
                else for (int j = 0; j < nvalid; j++) state[v0 + j] = (uint8_t)(st_new >> (8 * j));
            }
            n_sched += __popc(act);
        }
        active_bytes[v0 >> 3] = (uint8_t)act;
    }
    n_sched = vgl_block_reduce_add(n_sched, s64);
    n_heavy_near = vgl_block_reduce_add(n_heavy_near, s64);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) min_far = min(min_far, __shfl_xor(min_far, o));
    __syncthreads();
    if (vgl_lane() == 0) s32[vgl_wave()] = min_far;
    __syncthreads();
    if (threadIdx.x == 0) {
        int m = s32[0];
        for (int w = 1; w < VGL_WAVES; w++) m = min(m, s32[w]);
        partials[blockIdx.x * 4 + 0] = n_sched;
        partials[blockIdx.x * 4 + 1] = n_heavy_near;
        partials[blockIdx.x * 4 + 2] = m;
    }
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_ds_fold(const int64_t *partials, int64_t *counters)
{
    __shared__ int64_t s64[VGL_WAVES];
    __shared__ int s32[VGL_WAVES];
    int64_t a = 0, b = 0;
    int m = __float_as_int(FLT_MAX);
    for (int i = threadIdx.x; i < VGL_DS_BLOCKS; i += VGL_BLOCK) {
        a += partials[i * 4 + 0]; b += partials[i * 4 + 1];
        m = min(m, (int)partials[i * 4 + 2]);
    }
    a = vgl_block_reduce_add(a, s64);
    b = vgl_block_reduce_add(b, s64);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
    __syncthreads();
    if (vgl_lane() == 0) s32[vgl_wave()] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < VGL_WAVES; w++) m = min(m, s32[w]);
        counters[C_TMP0] = a; counters[C_TMP1] = b; counters[C_JUMP] = min(m, s32[0]);
    }
}

// persistent relax: HEAVY = false relaxes edges with w < delta of the scheduled rows, HEAVY = true the others
template <bool HEAVY>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_ds_relax(const int64_t *rowptr, const int32_t *adj, const float *w, const int32_t *tile_row,
                                                            int64_t E, int64_t ntiles, int32_t row_base, float delta, float *dist,
                                                            const uint64_t *active_bm, uint8_t *tile_active, uint8_t *state,
                                                            int64_t *shards)
{
    __shared__ int s_map[VGL_TILE];
    __shared__ int s_w[VGL_WAVES];
    __shared__ int s_list[VGL_BLOCK];
    __shared__ int s_count;
    int64_t streamed = 0;
    // the workgroup owns tiles blockIdx.x + k*gridDim.x; probe up to 256 of their flags at once (one byte per thread),
    // compact the active ones through LDS, then walk only those (a serial 1-byte probe per tile cost ~1 us each)
    for (int64_t k0 = 0; blockIdx.x + k0 * gridDim.x < ntiles; k0 += VGL_BLOCK) {
        if (threadIdx.x == 0) s_count = 0;
        __syncthreads();
        const int64_t mine = blockIdx.x + (k0 + threadIdx.x) * gridDim.x;
        if (mine < ntiles && tile_active[mine]) { tile_active[mine] = 0; s_list[atomicAdd(&s_count, 1)] = (int)(k0 + threadIdx.x); }
        __syncthreads();
        const int cnt = s_count;
    for (int li = 0; li < cnt; li++) {
        __syncthreads();          // every thread is done reading s_map of the previous tile before it is rebuilt
        const int64_t tile = blockIdx.x + (int64_t)s_list[li] * gridDim.x;
        const int64_t e0 = tile * VGL_TILE;
        const int n = (int)min((int64_t)VGL_TILE, E - e0);
        const int r_first = tile_row[tile];
        const int r_last = tile_row[tile + 1];
        vgl_tile_row_map(s_map, s_w, rowptr, e0, r_first, r_last);
        streamed += n;
        const int i0 = threadIdx.x * VGL_EPT;
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
            float olds[VGL_EPT], dsrc[VGL_EPT];
            int prev_row = -1;
            float d = 0.0f;
            bool live = false;
#pragma unroll
            for (int j = 0; j < VGL_EPT; j++) {
                const int row = s_map[i0 + j];
                if (row != prev_row) {
                    prev_row = row;
                    const int32_t u = row_base + r_first + row;
                    live = (active_bm[u >> 6] >> (u & 63)) & 1ULL;
                    d = live ? dist[u] : FLT_MAX;
                }
                const bool ok = live && (i0 + j < n) && ((ws[j] < delta) != HEAVY);
                dsrc[j] = ok ? d : FLT_MAX;
                olds[j] = ok ? dist[dsts[j]] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < VGL_EPT; j++) {
                if (dsrc[j] < FLT_MAX) {
                    const float nd = __fadd_rn(dsrc[j], ws[j]);
                    if (olds[j] > nd) {
                        const int before = atomicMin(reinterpret_cast<int *>(dist + dsts[j]), __float_as_int(nd));
                        if (before > __float_as_int(nd)) state[dsts[j]] = 3;     // light and heavy edges pending again
                    }
                }
            }
        }
    }
        __syncthreads();          // s_list / s_count are reused by the next probe round
    }
    if (threadIdx.x == 0 && streamed) atomicAdd((unsigned long long *)&shards[blockIdx.x & (VGL_NSHARD - 1)], (unsigned long long)streamed);
}

extern "C" {

int vgl_hip_sssp_run_delta(vgl_hip_ctx *c, vgl_hip_graph *g, const float *d_weights, int32_t source, float delta, float *d_dist,
                           vgl_hip_sssp_stats *stats)
{
    if (!c || !g || !d_weights || !d_dist) VGL_FAIL("sssp_run_delta: null argument");
    if (g->row_begin != 0 || g->row_end != g->V) VGL_FAIL("sssp_run_delta: graph handle must own all rows");
    if (source < 0 || source >= g->V) VGL_FAIL("sssp_run_delta: source vertex out of range");
    if (!(delta > 0.0f)) VGL_FAIL("sssp_run_delta: delta must be positive");
    const int32_t V = g->V;
    hipStream_t st = c->stream;
    uint8_t *state = reinterpret_cast<uint8_t *>(g->epoch);          // V bytes carved from the int32[V] epoch scratch
    if (!g->ds_tile_active) {
        VGL_HIP_TRY(hipMalloc((void **)&g->ds_tile_active, (size_t)g->out.ntiles + 1));
        VGL_HIP_TRY(hipMalloc((void **)&g->ds_partials, sizeof(int64_t) * VGL_DS_BLOCKS * 4));
        VGL_HIP_TRY(hipMemsetAsync(g->ds_tile_active, 0, (size_t)g->out.ntiles + 1, st));
    }
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, vgl_ceil_div(V, VGL_BLOCK)));
    hipLaunchKernelGGL(vgl_k_ds_init, dim3(grid), dim3(VGL_BLOCK), 0, st, V, source, d_dist, state);
    VGL_TRY(vgl_zero_counters(c, C_EDGES, 1));
    vgl_hip_sssp_stats s = {0, 0, 0};
    float T = delta;
    const bool debug = getenv("VGL_HIP_DEBUG") != nullptr;
    auto select = [&](int mode) -> int {
        vgl_timed_launch tl(c, "sssp_select");
        hipLaunchKernelGGL(vgl_k_ds_select, dim3(VGL_DS_BLOCKS), dim3(VGL_BLOCK), 0, st, g->nrows, g->row_begin, g->out.rowptr,
                           d_dist, state, T, mode, (uint8_t *)g->bm_front, g->ds_tile_active, g->ds_partials);
        hipLaunchKernelGGL(vgl_k_ds_fold, dim3(1), dim3(VGL_BLOCK), 0, st, g->ds_partials, c->d_counters);
        VGL_HIP_TRY(hipGetLastError());
        return vgl_read_counters(c, false);
    };
    auto relax = [&](bool heavy_pass) -> int {
        if (g->out.ntiles == 0) return 0;
        vgl_timed_launch tl(c, "sssp_relax");
        if (heavy_pass)
            hipLaunchKernelGGL(vgl_k_ds_relax<true>, dim3(VGL_DS_BLOCKS * 2), dim3(VGL_BLOCK), 0, st, g->out.rowptr, g->out.adj, d_weights,
                               g->out.tile_row, g->out.edges, g->out.ntiles, g->row_begin, delta, d_dist, g->bm_front, g->ds_tile_active,
                               state, c->d_shards);
        else
            hipLaunchKernelGGL(vgl_k_ds_relax<false>, dim3(VGL_DS_BLOCKS * 2), dim3(VGL_BLOCK), 0, st, g->out.rowptr, g->out.adj, d_weights,
                               g->out.tile_row, g->out.edges, g->out.ntiles, g->row_begin, delta, d_dist, g->bm_front, g->ds_tile_active,
                               state, c->d_shards);
        VGL_HIP_TRY(hipGetLastError());
        s.iterations++;
        return 0;
    };
    for (;;) {
        VGL_TRY(select(0));
        const int64_t n_light = c->h_counters[C_TMP0], n_heavy_near = c->h_counters[C_TMP1];
        const int min_far_bits = (int)c->h_counters[C_JUMP];
        if (debug) fprintf(stderr, "ds step %d: T=%g light=%lld heavy_near=%lld\n", s.iterations, T, (long long)n_light, (long long)n_heavy_near);
        if (n_light > 0) { VGL_TRY(relax(false)); continue; }
        if (n_heavy_near > 0) {                       // bucket settled: heavy edges of everything below T
            VGL_TRY(select(1));
            if (c->h_counters[C_TMP0] > 0) VGL_TRY(relax(true));
            continue;
        }
        float min_far;
        memcpy(&min_far, &min_far_bits, sizeof(float));
        if (!(min_far < FLT_MAX)) break;              // nothing flagged anywhere: fixed point
        T = std::max(min_far + delta, std::nextafter(min_far, FLT_MAX));   // always admits the nearest flagged vertex
    }
    VGL_TRY(vgl_read_counters(c, true));
    s.edges_relaxed = c->h_counters[C_EDGES];
    s.algorithmic_bytes = 12 * s.edges_relaxed + 6 * (int64_t)V * s.iterations;
    if (stats) *stats = s;
    return 0;
}

}  // extern "C"
