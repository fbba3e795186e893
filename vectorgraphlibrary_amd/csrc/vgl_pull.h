// vgl_pull.h -- out[v] = epilogue( sum over the adjacency of row v, IN ADJACENCY ORDER, of x[adj] ) for one CSR direction.
// Shared by PageRank (f32, self loops skipped, pr.hpp:105-124) and HITS (f64, both directions, hits.hpp:43-78): both reference
// loops are a strictly sequential `+=` chain per vertex, and keeping that order makes the sums bit-identical to the reference's
// sequential checkers (seq_pr.hpp:81-96, hits.hpp:117-160).
//
// Workgroup = 256 consecutive rows, one thread per row; the rows' edges are staged through LDS in tiles of 2048 (coalesced
// adjacency read + gather by all threads), then every thread adds its own row's slice sequentially from LDS.
// Rows with at least VGL_PULL_HUB_DEGREE edges are "hubs": their chain (deg dependent adds) would stall a whole 256-row workgroup
// (a 7*10^5-edge RMAT hub took 1.7 s per PageRank iteration that way).  They are listed once per graph and direction, dealt to
// 1024 wavefronts by a longest-processing-time schedule (vgl_pull_find_hubs), and summed by the first `hub_blocks` workgroups of
// the same launch (one workgroup per CU, i.e. one wavefront per SIMD, raised issue priority) while the remaining workgroups pull
// the ordinary rows on the same CUs (a separate kernel on a side stream overlapped worse: 5.0 vs 4.2 ms per RMAT-24 PageRank
// iteration): a wavefront gathers 512 values per batch (the next batch's gathers and the adjacency of the one after in flight;
// 1024-value batches cost 112 VGPRs and the ordinary rows' occupancy), parks them in LDS and folds them with one dependent add
// per value (all lanes compute the same chain; broadcast LDS reads).  The critical path of a launch is the chain of the largest
// hub (~10 cycles per edge measured), not the sum over hubs.
#pragma once
#include "vgl_hip_internal.h"

constexpr int VGL_PULL_HUB_DEGREE = 512;
constexpr int VGL_PULL_HUB_BATCH = 512;
constexpr int VGL_PULL_HUB_BLOCKS = 256;      // one per CU
constexpr int VGL_PULL_CHUNK = 4096;         // unordered hub sums: entries per chunk (one wavefront, 16 trips of 4 x 64 gathers)
constexpr int64_t VGL_PULL_BLOCK_EDGES = 16384; // edges per ordinary workgroup (8 tiles) before its 256 rows are split further

// lists the hubs of direction `d` (rows with >= VGL_PULL_HUB_DEGREE edges) grouped per wavefront; lazy, once per graph + direction
int vgl_pull_find_hubs(vgl_hip_ctx *c, vgl_hip_graph *g, vgl_dir_csr &d);

#ifdef __HIPCC__
__device__ __forceinline__ float vgl_add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double vgl_add_rn(double a, double b) { return __dadd_rn(a, b); }

// Epi: struct with  __device__ void operator()(int32_t v, T sum) const  (writes the result of vertex v)
template <class T, bool SKIP_SELF, bool SUMSQ, class Epi>
__device__ __forceinline__ void vgl_pull_hub_waves(T *s_vals, const int32_t *hub_rows, const int32_t *hub_off, int32_t row_base,
                                                   const int64_t *rowptr, const int32_t *adj, const T *x, Epi epi, double &sumsq, int hub_block = -1)
{
    __builtin_amdgcn_s_setprio(3);
    const int lane = vgl_lane();
    T *cur = s_vals + vgl_wave() * VGL_PULL_HUB_BATCH;                   // this wavefront's batch (consumed before the next is parked)
    constexpr int U = VGL_PULL_HUB_BATCH / 64;
    constexpr int B = VGL_PULL_HUB_BATCH;
    // No lane-dependent control flow anywhere below: the LDS hand-over relies on the wavefront staying converged (a ticket fetched
    // under `if (lane == 0)` let the compiler unswitch the loop on the lane id and the lanes ran apart).  Each wavefront owns a
    // precomputed list of hubs.
    const int32_t w = __builtin_amdgcn_readfirstlane((hub_block >= 0 ? hub_block : (int32_t)blockIdx.x) * VGL_WAVES + vgl_wave());    // scalar: loops are uniform
    const int32_t h_end = hub_off[w + 1];
    for (int32_t h = hub_off[w]; h < h_end; h++) {
        const int32_t r = hub_rows[h];
        const int64_t b = rowptr[r];
        const uint32_t n = (uint32_t)(rowptr[r + 1] - b);               // a row has fewer than 2^31 edges
        const int32_t *adj_h = adj + b;                                  // scalar base + 32-bit lane offsets
        const int32_t self = row_base + r;
        T acc = (T)0;
        T val[U];
        int32_t dst[U];
        // entries past the end of the row (and self loops when they are skipped) are marked with -1 and contribute +0: exact no-op
        auto load_adj = [&](uint32_t base) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t q = base + u * 64 + lane;
                int32_t t = q < n ? adj_h[q] : -1;
                if (SKIP_SELF && t == self) t = -1;
                dst[u] = t;
            }
        };
        auto gather = [&]() {
#pragma unroll
            for (int u = 0; u < U; u++) val[u] = dst[u] >= 0 ? x[(uint32_t)dst[u]] : (T)0;
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
#pragma unroll 8
            for (int i = 0; i < n4; i++) {                              // same addresses in every lane: LDS broadcast
                const T v0 = cur[4 * i], v1 = cur[4 * i + 1], v2 = cur[4 * i + 2], v3 = cur[4 * i + 3];
                acc = vgl_add_rn(acc, v0);
                acc = vgl_add_rn(acc, v1);
                acc = vgl_add_rn(acc, v2);
                acc = vgl_add_rn(acc, v3);
            }
        }
        epi(self, acc);                                                  // all lanes: same value, same address
        if (SUMSQ && lane == 0) sumsq += (double)acc * (double)acc;
    }
}

// GIANT hubs (ORDERED kernels, round 3).  A hub's adjacency-order chain is one dependent add per entry; the per-wavefront scheme above
// runs it at ~10 cycles per entry because the same wavefront also issues and waits for its gathers (one 512-value batch per ~2 us of memory
// latency).  For the few hubs whose chain alone exceeds what a wavefront's whole list costs -- RMAT-24: the largest row has 7.4e5 entries,
// ~3 of the 4.3 ms of a PageRank iteration -- the WORKGROUP shares the job: wavefronts 0-2 gather (three batches in flight each:
// adjacency of batch i + 2, values of batch i + 1, LDS store of batch i), wavefront 3 only adds, reading 16-byte LDS words; rounds of three
// batches are handed over through two sets of three LDS slots with one workgroup barrier per round.  Same order of additions, same bits.
constexpr int VGL_PULL_GIANT_DEGREE = 32768;      // rows at least this long take the workgroup scheme (RMAT-24 x 32: 554 rows, 11 % of the hub entries;
                                                  // PageRank iteration 4.19 ms without, 3.66 with the two longest rows only, 3.50 from 32768 down:
                                                  // below that the launch sits on its 537 M gathers of a 64 MiB table, ~3.3 ms -- profiles/README.md, gather floor)
constexpr int VGL_PULL_GIANT_BLOCKS = 64;
template <class T, bool SKIP_SELF, bool SUMSQ, class Epi>
__device__ __forceinline__ void vgl_pull_giant_hubs(T *s_a, T *s_b, const int32_t *giant_rows, const int32_t *giant_off, int gb, int32_t row_base,
                                                    const int64_t *rowptr, const int32_t *adj, const T *x, Epi epi, double &sumsq)
{
    constexpr int B = VGL_PULL_HUB_BATCH, U = B / 64, NP = 3;          // NP producer wavefronts, wavefront NP is the consumer
    const int lane = vgl_lane(), wave = vgl_wave();
    auto slot = [&](int i) -> T * { return i < 4 ? s_a + i * B : s_b + (i - 4) * B; };     // 6 slots: two sets of three
    if (wave == NP) __builtin_amdgcn_s_setprio(3);
    const int32_t h_end = giant_off[gb + 1];
    for (int32_t h = giant_off[gb]; h < h_end; h++) {
        const int32_t r = giant_rows[h];
        const int64_t b = rowptr[r];
        const uint32_t n = (uint32_t)(rowptr[r + 1] - b);
        const int32_t *adj_h = adj + b;
        const int32_t self = row_base + r;
        const uint32_t nbatches = (n + B - 1) / B, rounds = (nbatches + NP - 1) / NP;
        T acc = (T)0;
        T val[U];
        int32_t dst[U];
        auto load_adj = [&](uint32_t batch) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t q = batch * B + u * 64 + lane;
                int32_t t = (batch < nbatches && q < n) ? adj_h[q] : -1;
                if (SKIP_SELF && t == self) t = -1;
                dst[u] = t;
            }
        };
        auto gather = [&]() {
#pragma unroll
            for (int u = 0; u < U; u++) val[u] = dst[u] >= 0 ? x[(uint32_t)dst[u]] : (T)0;      // (+0 is an exact no-op of the chain)
        };
        // producer `wave` owns batches wave, wave + NP, ...: its k-th batch belongs to round k and goes to slot (k & 1) * NP + wave
        if (wave < NP) { load_adj((uint32_t)wave); gather(); load_adj((uint32_t)wave + NP); }
        for (uint32_t k = 0; k <= rounds; k++) {
            if (wave < NP) {
                if (k < rounds) {
                    T *cur = slot((int)(k & 1) * NP + wave);
#pragma unroll
                    for (int u = 0; u < U; u++) cur[u * 64 + lane] = val[u];
                    gather();                                           // batch wave + (k + 1) NP (entries past the row: zeros)
                    load_adj((uint32_t)wave + (k + 2) * NP);
                }
            } else if (k > 0) {                                         // consumer: the three batches of round k - 1, in order
                const uint32_t first = (k - 1) * NP;
                for (int p = 0; p < NP; p++) {
                    if (first + p >= nbatches) break;
                    const T *cur = slot((int)((k - 1) & 1) * NP + p);
                    const uint32_t left = n - (first + p) * B;
                    const int n4 = (int)min((uint32_t)(B / 4), (left + 3) / 4);
#pragma unroll 8
                    for (int i = 0; i < n4; i++) {
                        const T v0 = cur[4 * i], v1 = cur[4 * i + 1], v2 = cur[4 * i + 2], v3 = cur[4 * i + 3];
                        acc = vgl_add_rn(acc, v0);
                        acc = vgl_add_rn(acc, v1);
                        acc = vgl_add_rn(acc, v2);
                        acc = vgl_add_rn(acc, v3);
                    }
                }
            }
            __syncthreads();                                            // round k is parked, round k - 1 is folded: the sets swap
        }
        if (wave == NP) {
            epi(self, acc);
            if (SUMSQ && lane == 0) sumsq += (double)acc * (double)acc;
        }
    }
}

// Ordinary workgroups own the rows [blk_row[b], blk_row[b + 1]): at most 256 of them and -- the point -- about VGL_PULL_BLOCK_EDGES
// edges.  A workgroup walks its rows' edges tile by tile and in every tile the longest row slice is a serial chain, so with fixed
// 256-row workgroups the first ones of a degree-sorted graph (256 rows of just under VGL_PULL_HUB_DEGREE edges: 64 tiles, each with
// a ~500-step chain) were the launch's critical path below ~2^21 vertices (RMAT-18: 0.7 ms degree-sorted against 0.3 ms unsorted).
// ORDERED = false (f64 HITS: the bar is 1e-12 relative, the reference's own runs differ by 4e-15): the hubs are cut into chunks of
// VGL_PULL_CHUNK entries, the hub wavefronts take chunks round-robin, sum each per lane (lane l: entries l, l + 64, ...) and fold the
// 64 partial sums by a fixed butterfly into chunk_sums[chunk]; vgl_k_pull_hub_finish adds a hub's chunk sums in order.  No chain over
// a whole row (the 7.4e5-entry hub of RMAT-24 was a 4 ms f64 chain per sweep), still deterministic.
template <class T, bool SKIP_SELF>
__device__ __forceinline__ void vgl_pull_hub_chunks(const int32_t *chunks, int n_chunks, int32_t row_base, const int64_t *rowptr, const int32_t *adj,
                                                    const T *x, T *chunk_sums, int ci)
{
    const int lane = vgl_lane();
    if (ci < n_chunks) {                                        // one chunk per wavefront (the caller sizes the grid)
        const int32_t r = chunks[2 * ci];
        const int64_t b = rowptr[r] + (int64_t)chunks[2 * ci + 1] * VGL_PULL_CHUNK;
        const uint32_t n = (uint32_t)min((int64_t)VGL_PULL_CHUNK, rowptr[r + 1] - b);
        const int32_t *adj_h = adj + b;
        const int32_t self = row_base + r;
        T part[4] = {(T)0, (T)0, (T)0, (T)0};
        for (uint32_t q = lane; q < n; q += 256) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t qq = q + u * 64;
                int32_t t = qq < n ? adj_h[qq] : -1;
                if (SKIP_SELF && t == self) t = -1;
                part[u] = vgl_add_rn(part[u], t >= 0 ? x[(uint32_t)t] : (T)0);
            }
        }
        T acc = vgl_add_rn(vgl_add_rn(part[0], part[1]), vgl_add_rn(part[2], part[3]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc = vgl_add_rn(acc, __shfl_xor(acc, o));
        if (lane == 0) chunk_sums[ci] = acc;
    }
}
// one thread per hub: its chunk sums in order, the epilogue, and (SUMSQ) this workgroup's share of the sum of squares
template <class T, bool SUMSQ, class Epi>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_pull_hub_finish(int n_hubs, const int32_t *hub_list, int32_t row_base, const T *chunk_sums, Epi epi,
                                                                   double *sumsq_partials)
{
    __shared__ double s_sq[VGL_WAVES];
    const int h = blockIdx.x * VGL_BLOCK + threadIdx.x;
    double sumsq = 0.0;
    if (h < n_hubs) {
        const int32_t r = hub_list[3 * h], c0 = hub_list[3 * h + 1], nc = hub_list[3 * h + 2];
        T acc = (T)0;
        for (int32_t k = 0; k < nc; k++) acc = vgl_add_rn(acc, chunk_sums[c0 + k]);
        epi(row_base + r, acc);
        if (SUMSQ) sumsq = (double)acc * (double)acc;
    }
    if (SUMSQ) {
        const double tot = vgl_block_reduce_add(sumsq, s_sq);
        if (threadIdx.x == 0) sumsq_partials[blockIdx.x] = tot;
    }
}

// grid = hub_blocks + (number of row blocks) workgroups.  SUMSQ: sumsq_partials[blockIdx.x] = sum over this workgroup's rows of sum^2 (f64)
template <class T, bool SKIP_SELF, bool SUMSQ, class Epi, bool ORDERED = true>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_pull_sum(int32_t nrows, int32_t row_base, const int64_t *rowptr, const int32_t *adj,
                                                            const T *x, Epi epi, int hub_blocks, const int32_t *hub_rows,
                                                            const int32_t *hub_off, double *sumsq_partials, const int32_t *blk_row,
                                                            const int32_t *hub_chunks = nullptr, int n_hub_chunks = 0, T *hub_chunk_sums = nullptr,
                                                            int giant_blocks = 0, const int32_t *giant_rows = nullptr, const int32_t *giant_off = nullptr)
{
    __shared__ T s_val[VGL_TILE];                       // ordinary rows: staged values; hub wavefronts: 4 x 512 values
    __shared__ int32_t s_dst[VGL_TILE];
    __shared__ int64_t s_jump;
    __shared__ double s_sq[VGL_WAVES];
    double sumsq = 0.0;
    // ORDERED: the hub workgroups are dispatched first (one per CU runs the hub schedule: the longest chain starts at once).
    // Unordered: the chunk workgroups come LAST -- they are short, and behind the row blocks they fill the tail of the launch.
    // ORDERED: the giant-hub workgroups come first of all (giant_blocks of them; their chains are the launch's critical path)
    const int first_hub = ORDERED ? giant_blocks : (int)gridDim.x - hub_blocks;
    if (ORDERED && (int)blockIdx.x < giant_blocks) {
        if constexpr (ORDERED)
            vgl_pull_giant_hubs<T, SKIP_SELF, SUMSQ>(s_val, reinterpret_cast<T *>(s_dst), giant_rows, giant_off, (int)blockIdx.x, row_base, rowptr, adj, x, epi, sumsq);
    } else if ((int)blockIdx.x >= first_hub && (int)blockIdx.x < first_hub + hub_blocks) {
        if (ORDERED) vgl_pull_hub_waves<T, SKIP_SELF, SUMSQ>(s_val, hub_rows, hub_off, row_base, rowptr, adj, x, epi, sumsq, (int)blockIdx.x - first_hub);
        else {
            const int cw = ((int)blockIdx.x - first_hub) * VGL_WAVES + vgl_wave();
            vgl_pull_hub_chunks<T, SKIP_SELF>(hub_chunks, n_hub_chunks, row_base, rowptr, adj, x, hub_chunk_sums, cw);
        }
    } else {
        const int32_t blk = ORDERED ? (int32_t)blockIdx.x - hub_blocks - giant_blocks : (int32_t)blockIdx.x;
        const int32_t r_lo = blk_row[blk];
        const int32_t r_hi = blk_row[blk + 1];                          // <= r_lo + VGL_BLOCK
        const int32_t r = r_lo + threadIdx.x;
        const int64_t E0 = rowptr[r_lo], E1 = rowptr[r_hi];
        int64_t seg_b = 0, seg_e = 0;
        if (r < r_hi) { seg_b = rowptr[r]; seg_e = rowptr[r + 1]; }
        const bool hub = (seg_e - seg_b) >= VGL_PULL_HUB_DEGREE;      // summed by the hub wavefronts
        const int32_t self = row_base + r;
        T acc = (T)0;
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
                    s_val[i] = x[dst];
                }
            }
            __syncthreads();
            if (!hub) {
                const int lo = (int)(max(seg_b, base) - base);
                const int hi = (int)(min(seg_e, base + n) - base);
                // if(src_id != dst_id) rank += ... (pr.hpp:115-116).  Eight LDS reads are issued before the eight dependent adds: one
                // element per trip waits for a full LDS round trip per add, and on degree-sorted graphs the first workgroups own 256
                // rows of almost VGL_PULL_HUB_DEGREE edges each -- their chains are the launch's critical path below ~2^21 vertices
                // (RMAT-18 degree-sorted: 2.26 -> 0.7 ms per PageRank iteration; handing rows of 128+ edges to the hub wavefronts
                // instead made those the critical path: 20 % slower on unsorted graphs).
                int i = lo;
                for (; i + 8 <= hi; i += 8) {
                    T v[8]; int32_t d[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { v[u] = s_val[i + u]; d[u] = SKIP_SELF ? s_dst[i + u] : -1; }
#pragma unroll
                    for (int u = 0; u < 8; u++) { const T t = vgl_add_rn(acc, v[u]); acc = (SKIP_SELF && d[u] == self) ? acc : t; }
                }
                for (; i < hi; i++)
                    if (!SKIP_SELF || s_dst[i] != self) acc = vgl_add_rn(acc, s_val[i]);
            }
            base += n;
        }
        if (r < r_hi && !hub) {
            epi(self, acc);
            if (SUMSQ) sumsq = (double)acc * (double)acc;
        }
    }
    if (SUMSQ) {
        const double tot = vgl_block_reduce_add(sumsq, s_sq);
        if (threadIdx.x == 0) sumsq_partials[blockIdx.x] = tot;
    }
}
#endif
