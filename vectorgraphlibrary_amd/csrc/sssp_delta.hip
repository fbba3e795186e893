// sssp_delta.hip -- SSSP with a bucketed schedule (delta-stepping with a light / heavy edge split).
// Same operators as SSSP::vgl_dijkstra_* (algorithms/sssp/shortest_paths.hpp: relax d[dst] = min(d[dst], d[src] + w) in
// f32), different SCHEDULE: the fixed point of the relaxation is unique, so the distances are bit-identical to the
// reference's Bellman-Ford / Dijkstra results (tests assert this for several deltas).
//
// Why (measured on RMAT-24, degree-sorted ids): one all-edges relax pass = 3.4 ms, of which 0.7 ms is the coalesced
// adjacency + weight stream and 2.7 ms the per-edge 4-byte gather of dist[dst] (L2-request bound).  Bellman-Ford re-relaxes
// every out-edge of a vertex each time its distance improves and hubs improve many times (~5 E gathers, ~18 sweeps).  Here
//   * vertices are processed in distance buckets [.., T), T advancing by delta from the nearest pending vertex;
//   * a vertex's LIGHT edges (w < delta) are relaxed whenever it improves inside the bucket, its HEAVY edges once, after the
//     bucket has settled  (~1.3-1.6 E gathers in total);
//   * a per-(graph, weights, delta) plan stores the light and the heavy edges as two CSRs over the same rows (stable split), so a
//     step streams exactly the kind of edges it relaxes (the tile-granular first version re-streamed ~9 E per run);
//   * a SPARSE step works on a compacted, ascending-id FRONTIER of scheduled rows and walks only their segments of the part;
//     a DENSE step (most of the part is scheduled anyway: the light rounds at the peak of the first bucket, its heavy step)
//     marks the scheduled rows and sweeps the whole part as static tiles with the all-edges kernel.  The choice is made from a
//     prediction (improvements below T seen by the previous relax / rows with heavy edges pending), so it only affects speed.
//
// Robustness rule (no reliance on bucket theory for correctness): EVERY improvement of d[v] sets both pending bits of
// state[v]; a bit is cleared only when the row is scheduled, so every improvement is eventually followed by a relaxation of
// all out-edges => the loop ends exactly at the fixed point.
//
// Sparse step = count / scan / write (frontier ids + exclusive edge offsets, 8 rows per thread like the GNF) -> tile_first ->
// persistent relax over the frontier's edge tiles; dense step = mark / scan -> static relax.  Both end with vgl_k_ds_min_pending,
// whose last workgroup folds the partials and hands F, M, heavy-pending, near-improvements and the smallest pending distance
// to the host (one wait per step); everything else is sized from DEVICE counters.
#include "vgl_hip_internal.h"
#include <cfloat>
#include <cmath>
#include <cstring>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <rocprim/rocprim.hpp>
#include "vgl_blocked.h"

constexpr int VGL_DS_BLOCKS = 2048;       // persistent grid of the relax kernel

struct vgl_hip_sssp_plan {
    float delta = 0.0f;
    hipStream_t stream = nullptr;        // the stream whose pool owns the part arrays
    // the edges split into two CSRs over the same rows, both in original relative order: [0] light (w < delta), [1] heavy.
    // Own row offsets / adjacency / weights / tile table each, so a step can either walk a compacted frontier's segments or
    // sweep the whole part as static tiles.
    vgl_dir_csr part[2];
    int64_t *prow[2] = {nullptr, nullptr};
    int32_t *padj[2] = {nullptr, nullptr};
    float *pw[2] = {nullptr, nullptr};
    int64_t rows_nonempty[2] = {0, 0};   // rows that have edges in the part (what "most of the part is scheduled" is measured against)
    uint8_t *state = nullptr;     // V: bit0 light edges pending, bit1 heavy edges pending
    uint8_t *active = nullptr;    // V: rows scheduled by the current DENSE step (static sweep over a whole part)
    int32_t *vt_aux = nullptr;    // per vertex tile: rows with heavy pending below T
    int64_t *partials = nullptr;  // min-pending reduction (1024) followed by the relax kernels' near-improvement counts (+ 1 flag word)
    uint32_t *tickets = nullptr;  // arrival counters of the min_pending kernel ([1]; [0] unused)
    // round 4: the two parts laid out for the blocked advance as well (vgl_blocked.h, fused tiles for the dense block pairs): a DENSE step then
    // streams its part at ~5 TB/s with every random access in LDS instead of one L2 line per gather (the heavy step of the first bucket of an
    // RMAT-24 run: 465 M edges, 3.4 ms as a static sweep -- the push relax kernel at 0.21 of the HBM peak -- a quarter of the run)
    vgl_blocked_plan *blk[2] = {nullptr, nullptr};
};

// the relax of a DENSE step as a blocked pass over a part: rows that are not scheduled load +inf and contribute nothing, every improvement
// sets both pending bits of its vertex (the robustness rule above) and the improvements below T are counted per wavefront step
struct vgl_ds_blk_op {
    typedef uint32_t acc_t;
    static constexpr bool MARK = true;
    float *dist;
    uint8_t *state;
    const uint8_t *active;
    float T;
    int64_t *near_partials;
    int32_t g_base;
    __device__ __forceinline__ uint32_t load(int32_t i) const { return active[g_base + i] ? __float_as_uint(dist[g_base + i]) : __float_as_uint(FLT_MAX); }
    __device__ __forceinline__ uint32_t edge(uint32_t x, float w) const
    {
        const float d = __uint_as_float(x);
        return __float_as_uint(d < FLT_MAX ? __fadd_rn(d, w) : FLT_MAX);        // src_weight + weight (shortest_paths.hpp:126-130)
    }
    __device__ __forceinline__ uint32_t identity() const { return __float_as_uint(FLT_MAX); }
    __device__ __forceinline__ void accumulate(uint32_t *p, uint32_t v) const
    { __hip_atomic_fetch_min(reinterpret_cast<int *>(p), (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __device__ __forceinline__ uint32_t combine(uint32_t a, uint32_t b) const { return b < a ? b : a; }
    __device__ __forceinline__ bool finish_m(int32_t v, uint32_t acc) const
    {
        const float cand = __uint_as_float(acc);
        if (!(cand < dist[v])) return false;
        dist[v] = cand;
        state[v] = 3;
        return cand < T;
    }
    __device__ __forceinline__ bool partial_m(int32_t v, uint32_t acc) const
    {
        const float cand = __uint_as_float(acc);
        if (!(cand < dist[v])) return false;
        if (atomicMin(reinterpret_cast<int *>(dist + v), (int)acc) <= (int)acc) return false;
        state[v] = 3;
        return cand < T;
    }
    __device__ __forceinline__ void mark(int32_t v0, unsigned long long mask, bool) const
    {
        atomicAdd((unsigned long long *)&near_partials[(v0 >> 6) & (VGL_DS_BLOCKS - 1)], (unsigned long long)__popcll(mask));
        near_partials[VGL_DS_BLOCKS] = 1;
    }
    __device__ __forceinline__ void finish(int32_t v, uint32_t acc) const { (void)finish_m(v, acc); }
    __device__ __forceinline__ bool partial(int32_t v, uint32_t acc) const { (void)partial_m(v, acc); return true; }
};

// ---------------------------------------------------------------------------------------------------------------------
// plan construction
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_ds_light_flags(int64_t E, const float *w, float delta, uint32_t *flags)
{
    for (int64_t e = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; e < E; e += (int64_t)gridDim.x * VGL_BLOCK) flags[e] = w[e] < delta;
    if (blockIdx.x == 0 && threadIdx.x == 0) flags[E] = 0;
}
// S = exclusive scan of the light flags (E+1 entries): light edge e goes to position S[e] of the light part, heavy edge e to
// position e - S[e] of the heavy part (stable, rows stay contiguous in both parts)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_ds_partition(int64_t E, const int32_t *adj, const float *w, const uint32_t *S, float delta,
                                                                int32_t *adj_l, float *w_l, int32_t *adj_h, float *w_h)
{
    for (int64_t e = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; e < E; e += (int64_t)gridDim.x * VGL_BLOCK) {
        const float we = w[e];
        const int64_t l = S[e];
        if (we < delta) { adj_l[l] = adj[e]; w_l[l] = we; }
        else { adj_h[e - l] = adj[e]; w_h[e - l] = we; }
    }
}
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_ds_count_nonempty(int32_t nrows, const int64_t *prow, unsigned long long *out)
{
    __shared__ int s32[VGL_WAVES];
    int cnt = 0;
    for (int32_t r = blockIdx.x * VGL_BLOCK + threadIdx.x; r < nrows; r += gridDim.x * VGL_BLOCK) cnt += prow[r + 1] > prow[r];
    cnt = vgl_block_reduce_add(cnt, s32);
    if (threadIdx.x == 0 && cnt) atomicAdd(out, (unsigned long long)cnt);      // once per plan, <= 1024 workgroups
}
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_ds_split_rows(int32_t nrows, const int64_t *rowptr, const uint32_t *S, int64_t *row_l, int64_t *row_h)
{
    for (int32_t r = blockIdx.x * VGL_BLOCK + threadIdx.x; r <= nrows; r += gridDim.x * VGL_BLOCK) {
        const int64_t e = rowptr[r], l = S[e];
        row_l[r] = l;
        row_h[r] = e - l;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// per-step kernels
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_ds_init(int32_t V, int32_t source, float *dist, uint8_t *state)
{
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK) {
        const bool s = v == source;
        dist[v] = s ? 0.0f : FLT_MAX;
        state[v] = s ? 3 : 0;
    }
}

// bits of the 8 rows at v0 that are scheduled: (state & bit) && dist < T; *aux = rows with heavy pending and dist < T
__device__ __forceinline__ uint32_t vgl_ds_bits8(const uint8_t *state, const float *dist, int32_t v0, int nvalid, uint8_t bit, float T,
                                                 uint32_t *aux, uint64_t *st_out)
{
    uint64_t st8 = 0;
    if (nvalid == 8) st8 = *reinterpret_cast<const uint64_t *>(state + v0);
    else for (int j = 0; j < nvalid; j++) st8 |= (uint64_t)state[v0 + j] << (8 * j);
    uint32_t act = 0, hv = 0;
    if (st8) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint8_t f = (uint8_t)(st8 >> (8 * j));
            if (f && dist[v0 + j] < T) {
                if (f & bit) act |= 1u << j;
                if (f & 2) hv |= 1u << j;
            }
        }
    }
    *aux = hv; *st_out = st8;
    return act;
}
__device__ __forceinline__ int64_t vgl_ds_degree(const int64_t *prow, int32_t r) { return prow[r + 1] - prow[r]; }

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_ds_count(int32_t nrows, int32_t row_base, const int64_t *prow,
                                                            const uint8_t *state, const float *dist, uint8_t bit, float T,
                                                            int32_t *vt_cnt, int64_t *vt_deg, int32_t *vt_aux)
{
    __shared__ int64_t s64[VGL_WAVES];
    __shared__ int s32[VGL_WAVES];
    const int32_t r0 = blockIdx.x * VGL_TILE + threadIdx.x * VGL_EPT;
    int cnt = 0, aux_cnt = 0;
    int64_t deg = 0;
    if (r0 < nrows) {
        const int nvalid = min(VGL_EPT, nrows - r0);
        uint32_t aux; uint64_t st8;
        const uint32_t bits = vgl_ds_bits8(state, dist, row_base + r0, nvalid, bit, T, &aux, &st8);
        cnt = __popc(bits); aux_cnt = __popc(aux);
        if (bits)
            for (int j = 0; j < nvalid; j++)
                if ((bits >> j) & 1) deg += vgl_ds_degree(prow, r0 + j);
    }
    const int tc = vgl_block_reduce_add(cnt, s32);
    const int ta = vgl_block_reduce_add(aux_cnt, s32);
    const int64_t td = vgl_block_reduce_add(deg, s64);
    if (threadIdx.x == 0) { vt_cnt[blockIdx.x] = tc; vt_deg[blockIdx.x] = td; vt_aux[blockIdx.x] = ta; }
}

// exclusive offsets per vertex tile; counters[C_FRONT] = F, [C_NEIGH] = M, [C_TMP1] = rows with heavy pending below T; offs[F] = M
// one workgroup: exclusive offsets of the per-tile counts, totals into counters[C_FRONT] = F, [C_NEIGH] = M, [C_TMP1] = rows with
// heavy pending below T, and offs[F] = M.  (Doing this in the last workgroup of the count kernel was tried: with 8192 small
// workgroups the per-workgroup arrival atomics cost more than this launch.)
constexpr int VGL_DS_SCAN_THREADS = 1024;     // 16 wavefronts: 8 tiles per thread for the 8192 vertex tiles of a 2^24-vertex graph (256 threads: 36 us)
__global__ __launch_bounds__(VGL_DS_SCAN_THREADS) void vgl_k_ds_scan(int ntiles, const int32_t *vt_cnt, const int64_t *vt_deg, const int32_t *vt_aux,
                                                                     int32_t *vt_cnt_off, int64_t *vt_deg_off, int64_t *counters, int64_t *offs)
{
    constexpr int NW = VGL_DS_SCAN_THREADS / 64;
    __shared__ int s_c[NW], s_a[NW];
    __shared__ int64_t s_d[NW];
    const int per = (ntiles + VGL_DS_SCAN_THREADS - 1) / VGL_DS_SCAN_THREADS;
    const int lo = min(ntiles, (int)threadIdx.x * per), hi = min(ntiles, lo + per);
    int c = 0, a = 0;
    int64_t d = 0;
    for (int t = lo; t < hi; t++) { c += vt_cnt[t]; d += vt_deg[t]; a += vt_aux[t]; }
    const int ci = vgl_wave_incl_add(c), ai = vgl_wave_incl_add(a);
    const int64_t di = vgl_wave_incl_add(d);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) { s_c[wave] = ci; s_a[wave] = ai; s_d[wave] = di; }
    __syncthreads();
    int cbase = 0, ctot = 0, atot = 0;
    int64_t dbase = 0, dtot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        if (w < wave) { cbase += s_c[w]; dbase += s_d[w]; }
        ctot += s_c[w]; dtot += s_d[w]; atot += s_a[w];
    }
    int cpre = cbase + ci - c;
    int64_t dpre = dbase + di - d;
    for (int t = lo; t < hi; t++) {
        vt_cnt_off[t] = cpre; vt_deg_off[t] = dpre;
        cpre += vt_cnt[t]; dpre += vt_deg[t];
    }
    if (threadIdx.x == 0) {
        counters[C_FRONT] = ctot; counters[C_NEIGH] = dtot; counters[C_TMP1] = atot;
        offs[ctot] = dtot;
    }
}

// ids / offs of the scheduled rows (ascending) and the scheduled bit is cleared (each row is owned by exactly one thread)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_ds_write(int32_t nrows, int32_t row_base, const int64_t *prow,
                                                            uint8_t *state, const float *dist, uint8_t bit, float T,
                                                            const int32_t *vt_cnt_off, const int64_t *vt_deg_off, int32_t *ids, int64_t *offs)
{
    __shared__ int64_t s64[VGL_WAVES];
    __shared__ int s32[VGL_WAVES];
    const int32_t r0 = blockIdx.x * VGL_TILE + threadIdx.x * VGL_EPT;
    uint32_t bits = 0;
    int nvalid = 0;
    int64_t degs[VGL_EPT], deg = 0;
    uint64_t st8 = 0;
    if (r0 < nrows) {
        nvalid = min(VGL_EPT, nrows - r0);
        uint32_t aux;
        bits = vgl_ds_bits8(state, dist, row_base + r0, nvalid, bit, T, &aux, &st8);
        if (bits) {
#pragma unroll
            for (int j = 0; j < VGL_EPT; j++) {
                degs[j] = 0;
                if (j < nvalid && ((bits >> j) & 1)) { degs[j] = vgl_ds_degree(prow, r0 + j); deg += degs[j]; }
            }
        }
    }
    int ctot; int64_t dtot;
    int pos = vt_cnt_off[blockIdx.x] + vgl_block_excl_add((int)__popc(bits), s32, &ctot);
    int64_t eoff = vt_deg_off[blockIdx.x] + vgl_block_excl_add(deg, s64, &dtot);
    if (bits) {
        uint64_t st_new = st8;
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) {
            if ((bits >> j) & 1) {
                ids[pos] = row_base + r0 + j;
                offs[pos] = eoff; eoff += degs[j];
                pos++;
                st_new &= ~((uint64_t)bit << (8 * j));
            }
        }
        const int32_t v0 = row_base + r0;
        if (nvalid == 8) *reinterpret_cast<uint64_t *>(state + v0) = st_new;
        else for (int j = 0; j < nvalid; j++) state[v0 + j] = (uint8_t)(st_new >> (8 * j));
    }
}

// SMALL step: count + scan + write in one pass.  Workgroups that schedule rows reserve their slice of ids / offs with ONE packed
// atomic (rows << 40 | edges), so both cursors advance together and offs stays ascending along ids; the order of the slices is
// the order of arrival, which the relax does not care about (atomic minima).  Same-address atomics serialise (~12 ns each), so the
// host only takes this path when it expects few scheduled rows; a wrong guess costs at most one atomic per vertex tile.
constexpr int VGL_DS_EDGE_BITS = 40;
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_ds_select(int32_t nrows, int32_t row_base, const int64_t *prow, uint8_t *state, const float *dist,
                                                             uint8_t bit, float T, unsigned long long *cursor, int32_t *vt_aux, int32_t *ids,
                                                             int64_t *offs)
{
    __shared__ int64_t s64[VGL_WAVES];
    __shared__ int s32[VGL_WAVES];
    __shared__ unsigned long long s_base;
    const int32_t r0 = blockIdx.x * VGL_TILE + threadIdx.x * VGL_EPT;
    uint32_t bits = 0, aux = 0;
    int nvalid = 0;
    int64_t degs[VGL_EPT], deg = 0;
    uint64_t st8 = 0;
    if (r0 < nrows) {
        nvalid = min(VGL_EPT, nrows - r0);
        bits = vgl_ds_bits8(state, dist, row_base + r0, nvalid, bit, T, &aux, &st8);
        if (bits) {
#pragma unroll
            for (int j = 0; j < VGL_EPT; j++) {
                degs[j] = 0;
                if (j < nvalid && ((bits >> j) & 1)) { degs[j] = vgl_ds_degree(prow, r0 + j); deg += degs[j]; }
            }
        }
    }
    const int ta = vgl_block_reduce_add((int)__popc(aux), s32);
    int ctot; int64_t dtot;
    int pos = vgl_block_excl_add((int)__popc(bits), s32, &ctot);
    int64_t eoff = vgl_block_excl_add(deg, s64, &dtot);
    if (threadIdx.x == 0) {
        vt_aux[blockIdx.x] = ta;
        s_base = ctot ? atomicAdd(cursor, ((unsigned long long)ctot << VGL_DS_EDGE_BITS) | (unsigned long long)dtot) : 0ULL;
    }
    __syncthreads();
    if (ctot == 0) return;                                   // the same value in every thread
    pos += (int)(s_base >> VGL_DS_EDGE_BITS);
    eoff += (int64_t)(s_base & ((1ULL << VGL_DS_EDGE_BITS) - 1ULL));
    if (bits) {
        uint64_t st_new = st8;
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) {
            if ((bits >> j) & 1) {
                ids[pos] = row_base + r0 + j;
                offs[pos] = eoff; eoff += degs[j];
                pos++;
                st_new &= ~((uint64_t)bit << (8 * j));
            }
        }
        const int32_t v0 = row_base + r0;
        if (nvalid == 8) *reinterpret_cast<uint64_t *>(state + v0) = st_new;
        else for (int j = 0; j < nvalid; j++) state[v0 + j] = (uint8_t)(st_new >> (8 * j));
    }
}
// The same for steps of ANY size: a workgroup owns 8 consecutive vertex tiles (1024 workgroups for 2^24 vertices, so at most that
// many same-address atomics, ~12 us even when every workgroup schedules rows), counts them, reserves once and writes ids / offs
// and -- every scheduled row knows its own edge range -- the tile owners, which saves the tile_first launch too (the owner of the
// last edge is found by vgl_k_ds_select_finish).  Replaces count + scan + write + tile_first: two launches instead of four and one
// pass over state / dist instead of two.
constexpr int VGL_DS_WIDE_TILES = 4;           // vertex tiles per workgroup: 2048 workgroups for 2^24 vertices (8: too few wavefronts in flight, 2: twice the atomics)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_ds_select_wide(int32_t nrows, int32_t row_base, const int64_t *prow, uint8_t *state, const float *dist,
                                                                  uint8_t bit, float T, unsigned long long *cursor, int32_t *blk_aux, int32_t *ids,
                                                                  int64_t *offs, int32_t *tile_first, int ntiles)
{
    __shared__ int64_t s64[VGL_WAVES];
    __shared__ int s32[VGL_WAVES];
    __shared__ unsigned long long s_base;
    // a thread takes the same 8 rows of each of the workgroup's tiles; the output order inside the workgroup is thread-major (any
    // order will do, the order between workgroups is arbitrary anyway), so ONE scan over the workgroup places everything
    const int t_lo = blockIdx.x * VGL_DS_WIDE_TILES;
    uint32_t bits[VGL_DS_WIDE_TILES];
    uint64_t st8[VGL_DS_WIDE_TILES];
    int cnt = 0, aux_cnt = 0;
    int64_t deg = 0;
#pragma unroll
    for (int k = 0; k < VGL_DS_WIDE_TILES; k++) {
        const int32_t r0 = (t_lo + k) * VGL_TILE + threadIdx.x * VGL_EPT;
        bits[k] = 0; st8[k] = 0;
        if (t_lo + k < ntiles && r0 < nrows) {
            uint32_t aux;
            bits[k] = vgl_ds_bits8(state, dist, row_base + r0, min(VGL_EPT, nrows - r0), bit, T, &aux, &st8[k]);
            cnt += __popc(bits[k]); aux_cnt += __popc(aux);
        }
    }
#pragma unroll
    for (int k = 0; k < VGL_DS_WIDE_TILES; k++)
        if (bits[k]) {
            const int32_t r0 = (t_lo + k) * VGL_TILE + threadIdx.x * VGL_EPT;
            for (int j = 0; j < VGL_EPT; j++)
                if ((bits[k] >> j) & 1) deg += vgl_ds_degree(prow, r0 + j);
        }
    const int a_blk = vgl_block_reduce_add(aux_cnt, s32);
    int c_blk; int64_t d_blk;
    int pos = vgl_block_excl_add(cnt, s32, &c_blk);
    int64_t eoff = vgl_block_excl_add(deg, s64, &d_blk);
    if (threadIdx.x == 0) {
        blk_aux[blockIdx.x] = a_blk;
        s_base = c_blk ? atomicAdd(cursor, ((unsigned long long)c_blk << VGL_DS_EDGE_BITS) | (unsigned long long)d_blk) : 0ULL;
    }
    __syncthreads();
    if (c_blk == 0) return;                                    // the same value in every thread
    pos += (int)(s_base >> VGL_DS_EDGE_BITS);
    eoff += (int64_t)(s_base & ((1ULL << VGL_DS_EDGE_BITS) - 1ULL));
#pragma unroll
    for (int k = 0; k < VGL_DS_WIDE_TILES; k++) {
        if (!bits[k]) continue;
        const int32_t r0 = (t_lo + k) * VGL_TILE + threadIdx.x * VGL_EPT;
        const int nvalid = min(VGL_EPT, nrows - r0);
        uint64_t st_new = st8[k];
        for (int j = 0; j < VGL_EPT; j++) {
            if ((bits[k] >> j) & 1) {
                ids[pos] = row_base + r0 + j;
                offs[pos] = eoff;
                const int64_t eend = eoff + vgl_ds_degree(prow, r0 + j);
                for (int64_t tt = (eoff + VGL_TILE - 1) / VGL_TILE; tt < (eend + VGL_TILE - 1) / VGL_TILE; tt++) tile_first[tt] = pos;
                eoff = eend;
                pos++;
                st_new &= ~((uint64_t)bit << (8 * j));
            }
        }
        const int32_t v0 = row_base + r0;
        if (nvalid == 8) *reinterpret_cast<uint64_t *>(state + v0) = st_new;
        else for (int j = 0; j < nvalid; j++) state[v0 + j] = (uint8_t)(st_new >> (8 * j));
    }
}
// totals of a SMALL step where the scan kernel leaves them, and the cursor back to zero
__global__ __launch_bounds__(VGL_DS_SCAN_THREADS) void vgl_k_ds_select_finish(int ntiles, const int32_t *vt_aux, unsigned long long *cursor, int64_t *counters,
                                                                              int64_t *offs, int32_t *tile_first)
{
    __shared__ int s_a[VGL_DS_SCAN_THREADS / 64];
    int a = 0;
    for (int t = threadIdx.x; t < ntiles; t += VGL_DS_SCAN_THREADS) a += vt_aux[t];
    a = vgl_wave_incl_add(a);
    if ((threadIdx.x & 63) == 63) s_a[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        int atot = 0;
        for (int w = 0; w < VGL_DS_SCAN_THREADS / 64; w++) atot += s_a[w];
        const unsigned long long cur = *cursor;
        const int64_t F = (int64_t)(cur >> VGL_DS_EDGE_BITS), M = (int64_t)(cur & ((1ULL << VGL_DS_EDGE_BITS) - 1ULL));
        counters[C_FRONT] = F; counters[C_NEIGH] = M; counters[C_TMP1] = atot;
        offs[F] = M;
        *cursor = 0ULL;
        if (tile_first && M > 0) {                             // owner of the last edge: the last position whose range is not empty
            int64_t lo = 0, hi = F;                            // offs is ascending over [0, F]; largest p with offs[p] < M
            while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (offs[mid] < M) lo = mid; else hi = mid; }
            tile_first[(M + VGL_TILE - 1) / VGL_TILE] = (int32_t)lo;
        }
    }
}

// tile_first[t] = frontier position owning frontier edge t*VGL_TILE; entry [#tiles] = owner of the last edge.  F from the device.
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_ds_tile_first(const int64_t *counters, const int64_t *offs, int32_t *tile_first)
{
    const int32_t F = (int32_t)counters[C_FRONT];
    const int64_t M = counters[C_NEIGH];
    for (int32_t p = blockIdx.x * VGL_BLOCK + threadIdx.x; p < F; p += gridDim.x * VGL_BLOCK) {
        const int64_t t0 = (offs[p] + VGL_TILE - 1) / VGL_TILE;
        const int64_t t1 = (offs[p + 1] + VGL_TILE - 1) / VGL_TILE;
        for (int64_t t = t0; t < t1; t++) tile_first[t] = p;
        if (offs[p] < offs[p + 1] && offs[p + 1] == M) tile_first[(M + VGL_TILE - 1) / VGL_TILE] = p;
    }
}

// persistent relax over the frontier's edge tiles; prow / adj_p / w_p are the part (light or heavy) the step walks
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_ds_relax(const int64_t *counters, const int32_t *ids, const int64_t *offs, const int32_t *tile_first,
                                                            const int64_t *prow, const int32_t *adj_p, const float *w_p,
                                                            int32_t row_base, float T, float *dist, uint8_t *state, int64_t *near_partials)
{
    constexpr int STAGE = 1024;                             // rows staged per tile: 8 + 8 + 4 KB of LDS => 8 workgroups per CU
    __shared__ int s_map[VGL_TILE];
    __shared__ int64_t s_base[STAGE];
    __shared__ float s_dsrc[STAGE];
    __shared__ int s_w[VGL_WAVES];
    const int64_t M = counters[C_NEIGH];
    const int64_t ntiles = (M + VGL_TILE - 1) / VGL_TILE;
    int near = 0;                                           // improvements that fall inside the current bucket
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                    // LDS of the previous tile is no longer read
        const int64_t e0 = tile * VGL_TILE;
        const int n = (int)min((int64_t)VGL_TILE, M - e0);
        const int p_first = tile_first[tile];
        const int p_last = tile_first[tile + 1];
        const int np = p_last - p_first + 1;
        const bool staged = np <= STAGE;
        if (staged)
            for (int k = threadIdx.x; k < np; k += VGL_BLOCK) {
                const int p = p_first + k;
                const int32_t u = ids[p];
                const int32_t r = u - row_base;
                s_base[k] = prow[r] - offs[p];
                s_dsrc[k] = dist[u];
            }
        vgl_tile_row_map(s_map, s_w, offs, e0, p_first, p_last);
        int32_t dsts[VGL_EPT];
        float nds[VGL_EPT], olds[VGL_EPT];
        if (staged) {                                          // loads unconditional and in rounds (see vgl_k_sssp_relax_sparse)
            int64_t es[VGL_EPT];
            float ds[VGL_EPT], wv[VGL_EPT];
            bool ok[VGL_EPT];
#pragma unroll
            for (int j = 0; j < VGL_EPT; j++) {
                const int i = threadIdx.x + j * VGL_BLOCK;
                ok[j] = i < n;
                const int ii = ok[j] ? i : 0;
                const int k = s_map[ii];
                es[j] = s_base[k] + e0 + ii;
                ds[j] = s_dsrc[k];
            }
#pragma unroll
            for (int j = 0; j < VGL_EPT; j++) dsts[j] = adj_p[es[j]];
#pragma unroll
            for (int j = 0; j < VGL_EPT; j++) wv[j] = w_p[es[j]];
#pragma unroll
            for (int j = 0; j < VGL_EPT; j++) {
                nds[j] = __fadd_rn(ds[j], wv[j]);               // src_weight + weight (shortest_paths.hpp:126-130)
                if (!ok[j]) dsts[j] = -1;
            }
        } else {
#pragma unroll
            for (int j = 0; j < VGL_EPT; j++) {
                const int i = threadIdx.x + j * VGL_BLOCK;
                dsts[j] = -1;
                nds[j] = 0.0f;
                if (i < n) {
                    const int k = s_map[i];
                    const int p = p_first + k; const int32_t u = ids[p]; const int32_t r = u - row_base;
                    const int64_t e = prow[r] - offs[p] + e0 + i;
                    dsts[j] = adj_p[e];
                    nds[j] = __fadd_rn(dist[u], w_p[e]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) olds[j] = dist[max(dsts[j], 0)];
        int before[VGL_EPT];
        bool tried[VGL_EPT];
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) {
            tried[j] = dsts[j] >= 0 && olds[j] > nds[j];
            before[j] = 0;
            if (tried[j]) before[j] = atomicMin(reinterpret_cast<int *>(dist + dsts[j]), __float_as_int(nds[j]));
        }
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++)
            if (tried[j] && before[j] > __float_as_int(nds[j])) {
                state[dsts[j]] = 3;                             // light and heavy edges pending again
                near += nds[j] < T;
            }
    }
    __shared__ int s_near[VGL_WAVES];
    const int n_near = vgl_block_reduce_add(near, s_near);  // improvements that fall inside the current bucket (>= rows of the next light step)
    if (threadIdx.x == 0) { near_partials[blockIdx.x] = n_near; if (n_near) near_partials[VGL_DS_BLOCKS] = 1; }   // flag: same value from everyone
}

// DENSE step, pass 1: what count + write do for a compacted frontier, without the compaction -- every scheduled row gets
// active[v] = 1 and its pending bit cleared; per-vertex-tile counts feed the same scan kernel (totals for the host)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_ds_mark(int32_t nrows, int32_t row_base, const int64_t *prow, uint8_t *state, const float *dist,
                                                           uint8_t bit, float T, uint8_t *active, int32_t *vt_cnt, int64_t *vt_deg, int32_t *vt_aux)
{
    __shared__ int64_t s64[VGL_WAVES];
    __shared__ int s32[VGL_WAVES];
    const int32_t r0 = blockIdx.x * VGL_TILE + threadIdx.x * VGL_EPT;
    int cnt = 0, aux_cnt = 0;
    int64_t deg = 0;
    if (r0 < nrows) {
        const int nvalid = min(VGL_EPT, nrows - r0);
        uint32_t aux; uint64_t st8;
        const uint32_t bits = vgl_ds_bits8(state, dist, row_base + r0, nvalid, bit, T, &aux, &st8);
        cnt = __popc(bits); aux_cnt = __popc(aux);
        uint64_t act8 = 0, st_new = st8;
        if (bits)
            for (int j = 0; j < nvalid; j++)
                if ((bits >> j) & 1) {
                    deg += vgl_ds_degree(prow, r0 + j);
                    act8 |= 1ULL << (8 * j);
                    st_new &= ~((uint64_t)bit << (8 * j));
                }
        const int32_t v0 = row_base + r0;
        if (nvalid == 8) {
            *reinterpret_cast<uint64_t *>(active + v0) = act8;
            if (bits) *reinterpret_cast<uint64_t *>(state + v0) = st_new;
        } else {
            for (int j = 0; j < nvalid; j++) { active[v0 + j] = (uint8_t)(act8 >> (8 * j)); if (bits) state[v0 + j] = (uint8_t)(st_new >> (8 * j)); }
        }
    }
    const int tc = vgl_block_reduce_add(cnt, s32);
    const int ta = vgl_block_reduce_add(aux_cnt, s32);
    const int64_t td = vgl_block_reduce_add(deg, s64);
    if (threadIdx.x == 0) { vt_cnt[blockIdx.x] = tc; vt_deg[blockIdx.x] = td; vt_aux[blockIdx.x] = ta; }
}

// DENSE step, pass 2: static sweep over ALL edge tiles of the part (the all-edges kernel of sssp.hip: thread = 8 consecutive
// edges, two 16-byte loads of adjacency and of weights); rows that are not scheduled contribute nothing, tiles without a
// scheduled row are skipped before any edge data is read.  Worth it when most of the part is scheduled anyway: 158 G edges/s
// against ~65 G/s for the compacted walk (strided slots, staged row bases).
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_ds_relax_static(const int64_t *prow, const int32_t *adj_p, const float *w_p, const int32_t *tile_row,
                                                                   int64_t E, int32_t row_base, const uint8_t *active, float T, float *dist,
                                                                   uint8_t *state, int64_t *near_partials)
{
    __shared__ int s_map[VGL_TILE];
    __shared__ int s_w[VGL_WAVES];
    __shared__ int s_near[VGL_WAVES];
    const int64_t e0 = (int64_t)blockIdx.x * VGL_TILE;
    const int n = (int)min((int64_t)VGL_TILE, E - e0);
    const int r_first = tile_row[blockIdx.x];
    const int r_last = tile_row[blockIdx.x + 1];
    int any = 0;
    for (int r = r_first + threadIdx.x; r <= r_last; r += VGL_BLOCK) any |= active[row_base + r];
    if (!__syncthreads_or(any)) return;
    vgl_tile_row_map(s_map, s_w, prow, e0, r_first, r_last);
    const int i0 = threadIdx.x * VGL_EPT;
    int near = 0;
    if (i0 < n) {
        int32_t dsts[VGL_EPT];
        float ws[VGL_EPT];
        if (i0 + VGL_EPT <= n && ((e0 + i0) & 3) == 0) {
            const int4 a0 = *reinterpret_cast<const int4 *>(adj_p + e0 + i0);
            const int4 a1 = *reinterpret_cast<const int4 *>(adj_p + e0 + i0 + 4);
            const float4 w0 = *reinterpret_cast<const float4 *>(w_p + e0 + i0);
            const float4 w1 = *reinterpret_cast<const float4 *>(w_p + e0 + i0 + 4);
            dsts[0] = a0.x; dsts[1] = a0.y; dsts[2] = a0.z; dsts[3] = a0.w; dsts[4] = a1.x; dsts[5] = a1.y; dsts[6] = a1.z; dsts[7] = a1.w;
            ws[0] = w0.x; ws[1] = w0.y; ws[2] = w0.z; ws[3] = w0.w; ws[4] = w1.x; ws[5] = w1.y; ws[6] = w1.z; ws[7] = w1.w;
        } else {
#pragma unroll
            for (int j = 0; j < VGL_EPT; j++) {
                const bool ok = i0 + j < n;
                dsts[j] = ok ? adj_p[e0 + i0 + j] : 0;
                ws[j] = ok ? w_p[e0 + i0 + j] : 0.0f;
            }
        }
        // every load unconditional and issued in rounds (rows' flags, rows' distances, destinations' distances, then the atomics): loads
        // under per-lane conditions are compiled as branches that are awaited one after the other (see vgl_k_sssp_relax)
        float olds[VGL_EPT], nds[VGL_EPT], ds[VGL_EPT];
        bool ok[VGL_EPT];
        int32_t us[VGL_EPT];
        uint8_t act[VGL_EPT];
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) {
            us[j] = row_base + r_first + s_map[i0 + j < n ? i0 + j : i0];
            act[j] = active[us[j]];
        }
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) ds[j] = dist[us[j]];
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) {
            ok[j] = act[j] != 0 && (i0 + j < n);
            nds[j] = __fadd_rn(ds[j], ws[j]);                  // src_weight + weight (shortest_paths.hpp:126-130)
        }
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) olds[j] = dist[ok[j] ? dsts[j] : 0];
        int before[VGL_EPT];
        bool tried[VGL_EPT];
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) {
            tried[j] = ok[j] && olds[j] > nds[j];
            before[j] = 0;
            if (tried[j]) before[j] = atomicMin(reinterpret_cast<int *>(dist + dsts[j]), __float_as_int(nds[j]));
        }
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++)
            if (tried[j] && before[j] > __float_as_int(nds[j])) {
                state[dsts[j]] = 3;
                near += nds[j] < T;
            }
    }
    const int n_near = vgl_block_reduce_add(near, s_near);
    if (threadIdx.x == 0 && n_near) {
        atomicAdd((unsigned long long *)&near_partials[blockIdx.x & (VGL_DS_BLOCKS - 1)], (unsigned long long)n_near);
        near_partials[VGL_DS_BLOCKS] = 1;
    }
}

// min distance over all vertices with a pending bit (needed only when a bucket is exhausted)
// End of a step, one launch: (1) the smallest distance among vertices with a pending bit -- only when the host will need it:
// if the relax produced improvements below T the bucket continues, and after a light step with heavy edges pending below T the
// heavy step comes next, so the scan of state / dist is skipped; (2) the LAST workgroup folds the partial minima and the
// near-improvement counts, resets them, and hands the five numbers the host decides on straight to the pinned mirror.
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_ds_min_pending(int32_t V, const uint8_t *state, const float *dist, int64_t *partials,
                                                                  int64_t *near_partials, int light_step, int64_t *counters, uint32_t *ticket,
                                                                  volatile int64_t *host, int64_t seq)
{
    __shared__ int s32[VGL_WAVES];
    __shared__ int64_t s64[VGL_WAVES];
    const bool skip = near_partials[VGL_DS_BLOCKS] != 0 || (light_step && counters[C_TMP1] > 0);
    int m = __float_as_int(FLT_MAX);
    if (!skip) {
        const int32_t ngroups = V >> 3;                          // state has 8 bytes of slack; the tail is handled below
        for (int32_t gi = blockIdx.x * VGL_BLOCK + threadIdx.x; gi < ngroups; gi += gridDim.x * VGL_BLOCK) {
            const uint64_t st8 = *reinterpret_cast<const uint64_t *>(state + ((int64_t)gi << 3));
            if (st8) {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if ((st8 >> (8 * j)) & 0xff) m = min(m, __float_as_int(dist[(gi << 3) + j]));
            }
        }
        if (blockIdx.x == 0 && threadIdx.x < (V & 7)) { const int32_t v = (V & ~7) + threadIdx.x; if (state[v]) m = min(m, __float_as_int(dist[v])); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
    if (vgl_lane() == 0) s32[vgl_wave()] = m;
    __syncthreads();
    uint32_t dep = 0;
    if (threadIdx.x == 0) { for (int w = 1; w < VGL_WAVES; w++) m = min(m, s32[w]); dep = vgl_put_agent(partials + blockIdx.x, (int64_t)min(m, s32[0])); }
    if (!vgl_last_block(ticket, dep)) return;
    int64_t near = 0;
    for (int i = threadIdx.x; i < VGL_DS_BLOCKS; i += VGL_BLOCK) { near += near_partials[i]; near_partials[i] = 0; }
    near = vgl_block_reduce_add(near, s64);
    int mm = __float_as_int(FLT_MAX);
    for (int i = threadIdx.x; i < (int)gridDim.x; i += VGL_BLOCK) mm = min(mm, (int)vgl_load_agent(partials + i));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mm = min(mm, __shfl_xor(mm, o));
    __syncthreads();
    if (vgl_lane() == 0) s32[vgl_wave()] = mm;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 0; w < VGL_WAVES; w++) mm = min(mm, s32[w]);
        near_partials[VGL_DS_BLOCKS] = 0;
        counters[C_TMP0] = near; counters[C_JUMP] = mm;
        host[C_FRONT] = counters[C_FRONT]; host[C_NEIGH] = counters[C_NEIGH]; host[C_TMP1] = counters[C_TMP1];
        host[C_TMP0] = near; host[C_JUMP] = mm;
        __threadfence_system();
        host[C_NSLOTS] = seq;
        __threadfence_system();
    }
}

static inline unsigned vgl_ds_grid(int64_t n, int64_t cap) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cap, vgl_ceil_div(n, VGL_BLOCK))); }

extern "C" {

int vgl_hip_sssp_plan_create(vgl_hip_ctx *c, vgl_hip_graph *g, const float *d_weights, float delta, vgl_hip_sssp_plan **out)
{
    if (!c || !g || !d_weights || !out) VGL_FAIL("sssp_plan_create: null argument");
    if (!(delta > 0.0f)) VGL_FAIL("sssp_plan_create: delta must be positive");
    const int64_t E = g->out.edges;
    if (E >= 0xFFFFFFF0LL) VGL_FAIL("sssp_plan_create: at most 2^32-16 edges per graph handle");
    hipStream_t st = c->stream;
    vgl_hip_sssp_plan *p = new vgl_hip_sssp_plan();
    p->delta = delta;
    p->stream = st;
    VGL_HIP_TRY(hipMalloc((void **)&p->state, (size_t)g->V + 8));
    VGL_HIP_TRY(hipMalloc((void **)&p->active, (size_t)g->V + 8));
    VGL_HIP_TRY(hipMemsetAsync(p->active, 0, (size_t)g->V + 8, st));
    VGL_HIP_TRY(hipMalloc((void **)&p->vt_aux, sizeof(int32_t) * (size_t)std::max<int64_t>(g->nvtiles, 1)));
    VGL_HIP_TRY(hipMalloc((void **)&p->partials, sizeof(int64_t) * (1024 + VGL_DS_BLOCKS + 2)));          // minima | near counts + flag | select cursor
    VGL_HIP_TRY(hipMemsetAsync(p->partials, 0, sizeof(int64_t) * (1024 + VGL_DS_BLOCKS + 2), st));
    VGL_HIP_TRY(hipMalloc((void **)&p->tickets, sizeof(uint32_t) * 2 * VGL_TICKET_WORDS));
    VGL_HIP_TRY(hipMemsetAsync(p->tickets, 0, sizeof(uint32_t) * 2 * VGL_TICKET_WORDS, st));
    uint32_t *flags = nullptr, *S = nullptr;
    void *temp = nullptr;
    size_t need = 0;
    // (the multi-gigabyte temporaries and part arrays come from the library's stream-ordered pool: fresh hipMalloc / hipFree calls of that size
    // stalled for hundreds of milliseconds in the bench, where another plan had just been released -- 357 ms for a 42 ms build)
    VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&flags, sizeof(uint32_t) * ((size_t)E + 1)));
    VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&S, sizeof(uint32_t) * ((size_t)E + 1)));
    hipLaunchKernelGGL(vgl_k_ds_light_flags, dim3(vgl_ds_grid(std::max<int64_t>(E, 1), 16384)), dim3(VGL_BLOCK), 0, st, E, d_weights, delta, flags);
    VGL_HIP_TRY(rocprim::exclusive_scan(nullptr, need, flags, S, 0u, (size_t)E + 1, rocprim::plus<uint32_t>(), st));
    VGL_HIP_TRY(vgl_pool_alloc(st, &temp, need ? need : 16));
    VGL_HIP_TRY(rocprim::exclusive_scan(temp, need, flags, S, 0u, (size_t)E + 1, rocprim::plus<uint32_t>(), st));
    uint32_t n_light = 0;                                   // S[E] = number of light edges: sizes of the two parts
    VGL_TRY(vgl_hip_memcpy_d2h(c, &n_light, S + E, sizeof(uint32_t)));
    const int64_t part_edges[2] = {(int64_t)n_light, E - (int64_t)n_light};
    for (int k = 0; k < 2; k++) {
        VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&p->prow[k], sizeof(int64_t) * ((size_t)g->nrows + 1)));
        VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&p->padj[k], sizeof(int32_t) * (size_t)std::max<int64_t>(part_edges[k], 1)));
        VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&p->pw[k], sizeof(float) * (size_t)std::max<int64_t>(part_edges[k], 1)));
    }
    if (E > 0)
        hipLaunchKernelGGL(vgl_k_ds_partition, dim3(vgl_ds_grid(E, 16384)), dim3(VGL_BLOCK), 0, st, E, g->out.adj, d_weights, S, delta, p->padj[0],
                           p->pw[0], p->padj[1], p->pw[1]);
    hipLaunchKernelGGL(vgl_k_ds_split_rows, dim3(vgl_ds_grid((int64_t)g->nrows + 1, 8192)), dim3(VGL_BLOCK), 0, st, g->nrows, g->out.rowptr, S, p->prow[0],
                       p->prow[1]);
    VGL_HIP_TRY(hipGetLastError());
    VGL_HIP_TRY(hipMemsetAsync(p->partials, 0, 2 * sizeof(int64_t), st));
    for (int k = 0; k < 2; k++) {
        p->part[k].rowptr = p->prow[k]; p->part[k].adj = p->padj[k]; p->part[k].edges = part_edges[k];
        VGL_TRY(vgl_build_tile_rows(c, p->part[k], g->nrows));
        hipLaunchKernelGGL(vgl_k_ds_count_nonempty, dim3(vgl_ds_grid(g->nrows, 1024)), dim3(VGL_BLOCK), 0, st, g->nrows, p->prow[k],
                           reinterpret_cast<unsigned long long *>(p->partials + k));
    }
    VGL_TRY(vgl_hip_memcpy_d2h(c, p->rows_nonempty, p->partials, 2 * sizeof(int64_t)));
    VGL_HIP_TRY(hipMemsetAsync(p->partials, 0, 2 * sizeof(int64_t), st));
    VGL_HIP_TRY(hipStreamSynchronize(st));
    vgl_pool_free(st, temp); vgl_pool_free(st, flags); vgl_pool_free(st, S);
    // blocked layout of the HEAVY part for its dense steps (VGL_DS_BLOCKED: 0 never, 1 the heavy part, 2 both parts; default: the heavy part
    // from 2^25 edges -- the build costs about four static sweeps of the part, once per plan).  The light part keeps the static sweep: a
    // blocked pass streams the WHOLE part and sees the distances of the step's start (Jacobi), so the light rounds of the first bucket took
    // ten passes of 0.5 ms instead of five sweeps of 0.5 - 0.9 ms over the active rows' edges (RMAT-24: 11.5 ms per run with both parts
    // blocked against 11.85 with none); the heavy step runs once per bucket over nearly all of its part: 3.37 -> 1.27 ms.
    {
        const char *e0 = vgl_env(c, "VGL_DS_BLOCKED");
        const std::string e_keep = e0 ? e0 : "";
        const char *e = e0 ? e_keep.c_str() : nullptr;
        const int level = (e && *e) ? atoi(e) : (E >= (1LL << 25) ? 1 : 0);
        const bool want = level > 0;
        const char *fm0 = vgl_env(c, "VGL_BLK_FUSE_MIN");
        const std::string fm_keep = fm0 ? fm0 : "";
        const char *fm = fm0 ? fm_keep.c_str() : nullptr;
        for (int k = level >= 2 ? 0 : 1; k < 2 && want; k++) {
            if (part_edges[k] < (1LL << 20) && !(e && *e)) continue;
            const int fuse_min = (fm && *fm) ? atoi(fm) : (part_edges[k] >= (1LL << 22) ? 16384 : 0);
            if (vgl_blocked_plan_build(c, p->part[k], g->nrows, g->row_begin, g->V, 1, 0, p->pw[k], VGL_BLK_BITS, &p->blk[k], 32, fuse_min)) {
                vgl_hip_sssp_plan_destroy(c, p);
                return 1;
            }
        }
    }
    *out = p;
    return 0;
}

int vgl_hip_sssp_plan_destroy(vgl_hip_ctx *c, vgl_hip_sssp_plan *p)
{
    if (!p) return 0;
    if (c) hipStreamSynchronize(c->stream);
    for (int k = 0; k < 2; k++) { vgl_pool_free(p->stream, p->prow[k]); vgl_pool_free(p->stream, p->padj[k]); vgl_pool_free(p->stream, p->pw[k]); hipFree(p->part[k].tile_row); }
    hipFree(p->state); hipFree(p->active); hipFree(p->vt_aux); hipFree(p->partials); hipFree(p->tickets);
    for (int k = 0; k < 2; k++) if (p->blk[k]) vgl_blocked_plan_destroy(p->blk[k]);
    delete p;
    return 0;
}

int vgl_hip_sssp_run_plan(vgl_hip_ctx *c, vgl_hip_graph *g, vgl_hip_sssp_plan *p, int32_t source, float *d_dist, vgl_hip_sssp_stats *stats)
{
    if (!c || !g || !p || !d_dist) VGL_FAIL("sssp_run_plan: null argument");
    if (g->row_begin != 0 || g->row_end != g->V) VGL_FAIL("sssp_run_plan: graph handle must own all rows");
    if (source < 0 || source >= g->V) VGL_FAIL("sssp_run_plan: source vertex out of range");
    const int32_t V = g->V;
    hipStream_t st = c->stream;
    const bool debug = vgl_env(c, "VGL_HIP_DEBUG") != nullptr;
    hipLaunchKernelGGL(vgl_k_ds_init, dim3(vgl_ds_grid(V, 8192)), dim3(VGL_BLOCK), 0, st, V, source, d_dist, p->state);
    vgl_hip_sssp_stats s = {0, 0, 0};
    float T = p->delta;
    const unsigned nvt = (unsigned)g->nvtiles;
    int64_t *near_partials = p->partials + 1024;
    // one step: schedule rows with (state & bit) && dist < T, relax their light (bit 1) or heavy (bit 2) segment, then gather
    // everything the host needs for the next decision in ONE read: F, M, heavy-pending-below-T (before the relax), whether
    // the relax produced improvements below T, and the smallest pending distance after the relax.
    // dense = sweep the whole part as static tiles (rows marked in p->active) instead of walking a compacted frontier: chosen by
    // the caller from a PREDICTION of the step's size, so a wrong guess only costs time
    unsigned long long *cursor = reinterpret_cast<unsigned long long *>(p->partials + 1024 + VGL_DS_BLOCKS + 1);
    const bool wide_select = vgl_env(c, "VGL_DS_WIDE") ? atoi(vgl_env(c, "VGL_DS_WIDE")) != 0 : true;     // 0: count + scan + write (+ the small-step selection)
    auto step = [&](uint8_t bit, bool dense, bool small) -> int {
        const int k = bit == 1 ? 0 : 1;                     // which part this step walks
        if (dense && p->part[k].ntiles > 0) {
            hipLaunchKernelGGL(vgl_k_ds_mark, dim3(nvt), dim3(VGL_BLOCK), 0, st, g->nrows, g->row_begin, p->prow[k], p->state, d_dist, bit, T,
                               p->active, g->vt_cnt, g->vt_deg, p->vt_aux);
            hipLaunchKernelGGL(vgl_k_ds_scan, dim3(1), dim3(VGL_DS_SCAN_THREADS), 0, st, (int)g->nvtiles, g->vt_cnt, g->vt_deg, p->vt_aux, g->vt_cnt_off,
                               g->vt_deg_off, c->d_counters, g->offs);
            if (p->blk[k]) {
                const vgl_ds_blk_op op{d_dist, p->state, p->active, T, near_partials, g->row_begin};
                VGL_TRY((vgl_blocked_pass<vgl_ds_blk_op, true, false>(c, p->blk[k], op, "sssp_relax", "sssp_relax", false, "sssp_relax")));
            } else {
                vgl_timed_launch tl(c, "sssp_relax");
                hipLaunchKernelGGL(vgl_k_ds_relax_static, dim3((unsigned)p->part[k].ntiles), dim3(VGL_BLOCK), 0, st, p->prow[k], p->padj[k], p->pw[k],
                                   p->part[k].tile_row, p->part[k].edges, g->row_begin, p->active, T, d_dist, p->state, near_partials);
            }
        } else {
            if (wide_select) {
                const int nblk = (int)vgl_ceil_div((int64_t)nvt, VGL_DS_WIDE_TILES);
                hipLaunchKernelGGL(vgl_k_ds_select_wide, dim3(nblk), dim3(VGL_BLOCK), 0, st, g->nrows, g->row_begin, p->prow[k], p->state, d_dist, bit, T,
                                   cursor, p->vt_aux, g->ids, g->offs, g->tile_first, (int)nvt);
                hipLaunchKernelGGL(vgl_k_ds_select_finish, dim3(1), dim3(VGL_DS_SCAN_THREADS), 0, st, nblk, p->vt_aux, cursor, c->d_counters, g->offs,
                                   g->tile_first);
            } else if (small) {
                hipLaunchKernelGGL(vgl_k_ds_select, dim3(nvt), dim3(VGL_BLOCK), 0, st, g->nrows, g->row_begin, p->prow[k], p->state, d_dist, bit, T, cursor,
                                   p->vt_aux, g->ids, g->offs);
                hipLaunchKernelGGL(vgl_k_ds_select_finish, dim3(1), dim3(VGL_DS_SCAN_THREADS), 0, st, (int)g->nvtiles, p->vt_aux, cursor, c->d_counters, g->offs,
                                   (int32_t *)nullptr);
            } else {
                hipLaunchKernelGGL(vgl_k_ds_count, dim3(nvt), dim3(VGL_BLOCK), 0, st, g->nrows, g->row_begin, p->prow[k], p->state, d_dist,
                                   bit, T, g->vt_cnt, g->vt_deg, p->vt_aux);
                hipLaunchKernelGGL(vgl_k_ds_scan, dim3(1), dim3(VGL_DS_SCAN_THREADS), 0, st, (int)g->nvtiles, g->vt_cnt, g->vt_deg, p->vt_aux, g->vt_cnt_off,
                                   g->vt_deg_off, c->d_counters, g->offs);
                hipLaunchKernelGGL(vgl_k_ds_write, dim3(nvt), dim3(VGL_BLOCK), 0, st, g->nrows, g->row_begin, p->prow[k], p->state, d_dist,
                                   bit, T, g->vt_cnt_off, g->vt_deg_off, g->ids, g->offs);
            }
            if (!wide_select) hipLaunchKernelGGL(vgl_k_ds_tile_first, dim3(1024), dim3(VGL_BLOCK), 0, st, c->d_counters, g->offs, g->tile_first);
            vgl_timed_launch tl(c, "sssp_relax");
            hipLaunchKernelGGL(vgl_k_ds_relax, dim3(VGL_DS_BLOCKS), dim3(VGL_BLOCK), 0, st, c->d_counters, g->ids, g->offs, g->tile_first,
                               p->prow[k], p->padj[k], p->pw[k], g->row_begin, T, d_dist, p->state, near_partials);
        }
        const int64_t seq = vgl_next_seq(c);
        hipLaunchKernelGGL(vgl_k_ds_min_pending, dim3(1024), dim3(VGL_BLOCK), 0, st, V, p->state, d_dist, p->partials, near_partials, (int)(bit == 1),
                           c->d_counters, p->tickets + VGL_TICKET_WORDS, (volatile int64_t *)c->h_counters, seq);
        VGL_HIP_TRY(hipGetLastError());
        VGL_TRY(vgl_wait_counters(c, seq));
        if (c->h_counters[C_FRONT] > 0) { s.iterations++; s.edges_relaxed += c->h_counters[C_NEIGH]; }
        if (debug) { static double t_last = 0; timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); const double t_now = ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
                     fprintf(stderr, "[%7.0f us] ", t_now - t_last); t_last = t_now; }
        if (debug) fprintf(stderr, "ds %s%s%s T=%g rows=%lld edges=%lld heavy_pending_below_T=%lld near_improved=%lld\n", bit == 1 ? "light" : "heavy", dense ? " (dense)" : "", small && !dense ? " (small)" : "", T,
                           (long long)c->h_counters[C_FRONT], (long long)c->h_counters[C_NEIGH], (long long)c->h_counters[C_TMP1],
                           (long long)c->h_counters[C_TMP0]);
        return 0;
    };
    // bucket width: delta, multiplied by 16 after every bucket that relaxed fewer than E/16 edges (the sparse tail: fewer steps).  Narrower buckets in the dense core were measured
    // (W0 = 1/4 .. 1/16): 10 % fewer relaxed edges but 2-3x the steps, i.e. slower -- the per-step passes dominate.  The
    // light/heavy edge split always uses the plan's delta.  Any width is correct, it only changes the amount of re-relaxation.
    auto envf = [c](const char *n, double dflt) { const char *v = vgl_env(c, n); return v ? atof(v) : dflt; };
    float width = (float)(p->delta * envf("VGL_DS_W0", 1.0));
    const int64_t rows_hi = (int64_t)envf("VGL_DS_HI", 1.0e18), rows_lo = (int64_t)envf("VGL_DS_LO", 4096.0);
    T = width;
    const int64_t edges_lo = (int64_t)envf("VGL_DS_ELO", std::max<double>(1.0e6, (double)g->out.edges / 16.0));
    const double grow = envf("VGL_DS_GROW", 16.0);
    const double dense_frac = envf("VGL_DS_DENSE", 0.4);    // dense when the predicted rows exceed this share of the part's non-empty rows
    const double dense_frac_blk = envf("VGL_DS_DENSE_BLK", 0.4);       // ... of a part that has a blocked layout (a blocked pass streams the whole part whatever is scheduled: 0.25 9.7 ms, 0.4 8.8, 0.1 10.8 per RMAT-24 run)
    auto dense_share = [&](int k) { return p->blk[k] ? dense_frac_blk : dense_frac; };
    int64_t bucket_rows = 0, bucket_edges = 0;
    int64_t pred_light_rows = 0;                             // improvements below T seen by the last relax: the next light frontier, roughly
    // small = expected to schedule at most `small_rows` rows: one selection pass with a packed atomic cursor instead of count + scan +
    // write.  Light steps inside a bucket are predicted by the previous relax's improvements below T, heavy steps know their size;
    // the first step of a bucket has no prediction and takes the ordinary path.
    const int64_t small_rows = (int64_t)envf("VGL_DS_SMALL", 4096.0);
    bool fresh_bucket = true;
    for (;;) {
        VGL_TRY(step(1, (double)pred_light_rows > dense_share(0) * (double)p->rows_nonempty[0], !fresh_bucket && pred_light_rows <= small_rows));
        fresh_bucket = false;
        pred_light_rows = c->h_counters[C_TMP0];
        bucket_rows += c->h_counters[C_FRONT];
        bucket_edges += c->h_counters[C_NEIGH];
        // (leaving a bucket early once its light frontier is small and shrinking was measured: the tail reappears in the next
        // bucket, 1-10 % slower)
        const bool near = c->h_counters[C_FRONT] > 0 && c->h_counters[C_TMP0] != 0;
        if (near) continue;                                             // the bucket received improvements: light edges again
        if (c->h_counters[C_TMP1] > 0) {                                // bucket settled: its heavy edges, once
            VGL_TRY(step(2, (double)c->h_counters[C_TMP1] > dense_share(1) * (double)p->rows_nonempty[1], c->h_counters[C_TMP1] <= small_rows));
            pred_light_rows = c->h_counters[C_TMP0];
            bucket_edges += c->h_counters[C_NEIGH];
            if (c->h_counters[C_FRONT] > 0 && c->h_counters[C_TMP0] != 0) continue;
        }
        const int bits = (int)c->h_counters[C_JUMP];                    // smallest pending distance after the last relax
        float min_far;
        memcpy(&min_far, &bits, sizeof(float));
        if (!(min_far < FLT_MAX)) break;                                // nothing pending anywhere: fixed point
        if (min_far < T) continue;                                      // (defensive) something below T is still pending
        // a bucket that relaxed few edges was all per-step overhead (7 launches + a host read, ~0.1 ms): widen quickly
        if (bucket_rows > rows_hi && width > p->delta / 64.0f) width *= 0.5f;
        else if ((bucket_edges < edges_lo || bucket_rows < rows_lo) && width < 4096.0f * p->delta) width *= (float)grow;
        if (debug) fprintf(stderr, "ds bucket done: rows=%lld edges=%lld next width=%g\n", (long long)bucket_rows, (long long)bucket_edges, width);
        bucket_rows = bucket_edges = 0;
        T = std::max(min_far + width, std::nextafter(min_far, FLT_MAX));
        fresh_bucket = true;
    }
    s.algorithmic_bytes = 12 * s.edges_relaxed + 5 * (int64_t)V * s.iterations;
    if (stats) *stats = s;
    return 0;
}

int vgl_hip_sssp_run_delta(vgl_hip_ctx *c, vgl_hip_graph *g, const float *d_weights, int32_t source, float delta, float *d_dist,
                           vgl_hip_sssp_stats *stats)
{
    vgl_hip_sssp_plan *p = nullptr;
    VGL_TRY(vgl_hip_sssp_plan_create(c, g, d_weights, delta, &p));
    const int rc = vgl_hip_sssp_run_plan(c, g, p, source, d_dist, stats);
    vgl_hip_sssp_plan_destroy(c, p);
    return rc;
}

}  // extern "C"
