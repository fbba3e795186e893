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
#include <chrono>
#include "vgl_gnf.h"
#include "vgl_blocked.h"

#ifndef VGL_BU_HEAVY_LANES
#define VGL_BU_HEAVY_LANES 16       // lanes per deferred vertex in the second bottom-up pass
#endif       // thread-serial probes before a vertex is deferred to the wavefront pass
constexpr int VGL_DO_ALPHA = 15;       // change_state.hpp:5
constexpr int VGL_DO_BETA = 18;        // change_state.hpp:6

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_bfs_init(int32_t V, int32_t source, int32_t *levels)
{
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK)
        levels[v] = (v == source) ? 1 : -1;      // FIRST_LEVEL_VERTEX / UNVISITED_VERTEX (change_state.h:21-23)
}

// start of a fused traversal: levels, the three bitmaps (visited = front = {source}, next = 0) and the tickets in one launch
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_bfs_init_all(int32_t V, int32_t source, int32_t *levels, int64_t words, uint64_t *visited,
                                                                uint64_t *front, uint64_t *next, uint32_t *tickets, unsigned long long *list_count)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *list_count = 0ULL;
    for (int32_t v = blockIdx.x * VGL_BLOCK + threadIdx.x; v < V; v += gridDim.x * VGL_BLOCK)
        levels[v] = (v == source) ? 1 : -1;      // FIRST_LEVEL_VERTEX / UNVISITED_VERTEX (change_state.h:21-23)
    for (int64_t w = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; w < words; w += (int64_t)gridDim.x * VGL_BLOCK) {
        const uint64_t bit = (w == (source >> 6)) ? (1ULL << (source & 63)) : 0ULL;
        visited[w] = bit; front[w] = bit; next[w] = 0;
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < 4 * VGL_TICKET_WORDS; i += VGL_BLOCK) tickets[i] = 0;
}

// tile_first[t] = frontier position whose edge range contains edge t*VGL_TILE
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_tile_first(int32_t F, const int64_t *offs, int32_t *tile_first)
{
    for (int32_t p = blockIdx.x * VGL_BLOCK + threadIdx.x; p < F; p += gridDim.x * VGL_BLOCK) {
        const int64_t t0 = (offs[p] + VGL_TILE - 1) / VGL_TILE;
        const int64_t t1 = (offs[p + 1] + VGL_TILE - 1) / VGL_TILE;
        for (int64_t t = t0; t < t1; t++) tile_first[t] = p;
        // entry [#tiles] = the position that owns the LAST edge, so that trailing zero-degree frontier vertices (they sort to
        // the end of a degree-sorted graph) are not walked by the last tile
        if (offs[p] < offs[p + 1] && offs[p + 1] == offs[F]) tile_first[(offs[F] + VGL_TILE - 1) / VGL_TILE] = p;
    }
}

// top-down advance over a sparse frontier, edge-balanced: workgroup = 2048 consecutive frontier edges.
// edge_op of bfs.hpp:28-36: if levels[dst] == UNVISITED then levels[dst] = cur+1 (benign race, same value).
// EMIT: every newly discovered vertex also sets its bit in the next-frontier bitmap (idempotent atomicOr), so that the next
// level's frontier can be generated from 2 MiB of bitmap instead of a 64 MiB scan of levels.  Used for small frontiers only.
// (levels[dst] == -1 is what keeps the atomics few: asking the next-frontier bitmap instead -- 2 MiB in L2 against a random sector of
// a 64 MiB array per unvisited edge -- lets every edge that arrives before the bit is visible issue its own atomicOr: 252 us per level
// instead of 74.)
// COUNT (with EMIT): the vertex's one claimer -- the atomicOr whose return value lacks the bit -- also counts it and adds its out-degree; the
// last workgroup leaves the sums in counters[C_NEXT_F / C_NEXT_M] for the count launch that follows (vgl_k_bm_gnf_count: when they say
// "bottom-up next" it does not walk the new frontier's rows at all).
// FILTER = false (round 4): the visited-bitmap probe is left out.  While a traversal is still growing almost no destination has its bit set, so
// the probe is one scattered L2 request per edge that rules nothing out; `levels[dst] == -1` alone decides.  The caller keeps the probe once a
// sizeable part of the vertices has been visited (the shrinking phase after the bottom-up levels), where it saves the levels sector.  Worth
// 1.5 - 3 us per RMAT-24 traversal (profiles/r04_bfs_ab.log): this kernel does not follow its request count either.
template <bool EMIT, bool COUNT, bool FILTER = true>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_td_expand(const int32_t *ids, const int64_t *offs, const int32_t *tile_first,
                                                             int32_t F, int64_t M, const int64_t *rowptr, const int32_t *adj,
                                                             int32_t row_base, const uint64_t *visited, int32_t *levels,
                                                             int32_t next_level, uint64_t *next, int64_t *partials, uint32_t *ticket, int64_t *counters)
{
    __shared__ int s_map[VGL_TILE];
    __shared__ int64_t s_base[VGL_TILE];
    __shared__ int s_w[VGL_WAVES];
    const int64_t e0 = (int64_t)blockIdx.x * VGL_TILE;
    const int n = (int)min((int64_t)VGL_TILE, M - e0);
    const int p_first = tile_first[blockIdx.x];
    const int p_last = tile_first[blockIdx.x + 1];      // last tile: owner of the last edge
    // per frontier position of this tile: (first adjacency index of the vertex) - (its edge offset in the frontier), staged
    // in LDS so that the per-edge path is LDS lookups + one adjacency load (falls back to global reads when a tile spans
    // more than 2048 frontier positions, i.e. thousands of zero-degree frontier vertices)
    const int np = p_last - p_first + 1;
    const bool staged = np <= VGL_TILE;
    if (staged)
        for (int k = threadIdx.x; k < np; k += VGL_BLOCK) {
            const int p = p_first + k;
            s_base[k] = rowptr[ids[p] - row_base] - offs[p];
        }
    vgl_tile_row_map(s_map, s_w, offs, e0, p_first, p_last);      // ends with a barrier: s_base is visible too
    int32_t dsts[VGL_EPT];
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) {
        const int i = threadIdx.x + j * VGL_BLOCK;          // strided slots => coalesced adjacency reads
        dsts[j] = -1;
        if (i < n) {
            const int k = s_map[i];
            const int64_t base = staged ? s_base[k] : (rowptr[ids[p_first + k] - row_base] - offs[p_first + k]);
            dsts[j] = adj[base + e0 + i];
        }
    }
    // Every load below is UNCONDITIONAL (a slot without an edge, or a destination the bitmap already rules out, reads entry 0 instead): a
    // load under a per-lane condition is compiled as a branch whose result is awaited before the next one is issued -- eight visited
    // words and eight levels, sixteen dependent round trips per thread, is what the first version of this kernel waited for.  And the
    // levels are requested BEFORE the first store: a load after a store to the same array may alias it.
#ifndef VGL_TD_BATCH
#define VGL_TD_BATCH 3
#endif
    uint64_t vw[VGL_EPT];
    bool unvis[VGL_EPT];
    bool fresh[VGL_EPT];
#if VGL_TD_BATCH & 1
    if (FILTER) {
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) vw[j] = visited[max(dsts[j], 0) >> 6];
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) unvis[j] = dsts[j] >= 0 && !((vw[j] >> (dsts[j] & 63)) & 1ULL);
    } else {
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) unvis[j] = dsts[j] >= 0;
    }
#else
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) unvis[j] = dsts[j] >= 0 && (!FILTER || !((visited[dsts[j] >> 6] >> (dsts[j] & 63)) & 1ULL));
#endif
#if VGL_TD_BATCH & 2
    int32_t lv[VGL_EPT];
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) lv[j] = levels[unvis[j] ? dsts[j] : 0];
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) fresh[j] = unvis[j] && lv[j] == -1;
#else
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) fresh[j] = unvis[j] && levels[dsts[j]] == -1;
#endif
    if (!COUNT) {
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++)
            if (fresh[j]) {
                levels[dsts[j]] = next_level;
                if (EMIT) atomicOr((unsigned long long *)&next[dsts[j] >> 6], 1ULL << (dsts[j] & 63));
            }
        return;
    }
    // the atomics of a thread's edges are issued together, their return values read afterwards
    __shared__ int64_t s64[VGL_WAVES];
    unsigned long long old[VGL_EPT];
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) {
        old[j] = ~0ULL;
        if (fresh[j]) {
            levels[dsts[j]] = next_level;
            old[j] = atomicOr((unsigned long long *)&next[dsts[j] >> 6], 1ULL << (dsts[j] & 63));
        }
    }
    int64_t cnt = 0, deg = 0;
    int64_t lo[VGL_EPT], hi[VGL_EPT];
    bool claimed[VGL_EPT];
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) {                     // (unconditional loads again: a slot that claimed nothing reads row 0)
        claimed[j] = dsts[j] >= 0 && !((old[j] >> (dsts[j] & 63)) & 1ULL);
        const int64_t r = claimed[j] ? dsts[j] - row_base : 0;
        lo[j] = rowptr[r]; hi[j] = rowptr[r + 1];
    }
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++)
        if (claimed[j]) { cnt++; deg += hi[j] - lo[j]; }
    cnt = vgl_block_reduce_add(cnt, s64);
    deg = vgl_block_reduce_add(deg, s64);
    uint32_t dep = 0;
    if (threadIdx.x == 0) dep = vgl_put_agent(partials + 2 * blockIdx.x, cnt) ^ vgl_put_agent(partials + 2 * blockIdx.x + 1, deg);
    if (!vgl_last_block(ticket, dep)) return;
    int64_t a = 0, b = 0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += VGL_BLOCK) { a += vgl_load_agent(partials + 2 * i); b += vgl_load_agent(partials + 2 * i + 1); }
    a = vgl_block_reduce_add(a, s64);
    b = vgl_block_reduce_add(b, s64);
    if (threadIdx.x == 0) { counters[C_NEXT_F] = a; counters[C_NEXT_M] = b; }
}

// ---- a large top-down level as a blocked pass (vgl_blocked.h; plan built by vgl_hip_bfs_prepare_blocked) ----
// vgl_k_td_expand pays per frontier edge one adjacency entry, one probe of the visited bitmap (an L2 line) and one sector of `levels`:
// 4.7 us per million edges, 2.1 ms for the level of an RMAT-24 traversal that holds 83 % of the edges.  What has to travel from the
// source's side of an edge to the destination's is ONE BIT (source in the frontier?), so the blocked pass carries one 64-bit word per
// 64-edge chunk: the gather kernel keeps the frontier bits of its 32768 rows in LDS (4 KiB) and reads the uint16 row indices (2 B per
// edge), the accumulate kernel reads the uint16 destination indices of the chunks that hold a frontier edge (2 B per edge) and ORs
// discovery bits into an LDS window; its epilogue masks with `visited`, writes the next-frontier words and the levels.  4.2 B per edge
// of streamed traffic whatever the frontier -- the all-edges cost of ~0.5 ms on RMAT-24 pays from about a tenth of the edges (the sparse path adds a frontier
// generation per level on top of its 4.7 us per million edges).
__global__ __launch_bounds__(VGL_BTHREADS) void vgl_k_bfs_blk_gather(const vgl_blk_unit *units, const uint16_t *g_lo, const uint32_t *mid_to_a,
                                                                     uint64_t *bits, int32_t g_count, const uint64_t *front, int64_t word0)
{
    __shared__ uint32_t s_f[VGL_BLK / 32];
    const vgl_blk_unit u = units[blockIdx.x];
    const int32_t base = u.block << VGL_BLK_BITS;
    const int nw = (min(VGL_BLK, g_count - base) + 31) >> 5;            // 32-bit words of this block that exist
    const uint32_t *f32 = reinterpret_cast<const uint32_t *>(front + word0) + ((size_t)u.block << (VGL_BLK_BITS - 5));
    uint32_t mine = 0;
    if ((int)threadIdx.x < nw) mine = f32[threadIdx.x];                  // VGL_BTHREADS = VGL_BLK / 32: one word per thread
    s_f[threadIdx.x] = mine;
    const bool any = __syncthreads_or(mine != 0u);                       // no frontier row in this block: its chunks carry zeros
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 3, off = (lane & 7) * 8;                    // this lane: entries off..off+7 of chunk m0 + sub
    for (uint32_t m0 = u.chunk0 + wave * VGL_BGROUP; m0 < u.chunk1; m0 += VGL_BWAVES * VGL_BGROUP) {
        const uint32_t m = m0 + sub;                                    // (the eight lanes of a chunk agree on m)
        if (m >= u.chunk1) continue;
        uint32_t mask = 0;
        if (any) {
            const uint4 gl = *reinterpret_cast<const uint4 *>(g_lo + (size_t)m * VGL_CHUNK + off);
            const uint32_t g[4] = {gl.x, gl.y, gl.z, gl.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t lo = g[k] & 0xFFFFu, hi = g[k] >> 16;
                mask |= ((s_f[lo >> 5] >> (lo & 31)) & 1u) << (2 * k);
                mask |= ((s_f[hi >> 5] >> (hi & 31)) & 1u) << (2 * k + 1);
            }
        }
        // the chunk's word: byte (lane & 7) comes from this lane (pad entries read row 0 of the block: their destinations are dummies)
        uint32_t w0 = (lane & 4) ? 0u : mask << (8 * (lane & 3)), w1 = (lane & 4) ? mask << (8 * (lane & 3)) : 0u;
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) { w0 |= __shfl_xor(w0, d); w1 |= __shfl_xor(w1, d); }
        if ((lane & 7) == 0) bits[mid_to_a[m]] = (uint64_t)w0 | ((uint64_t)w1 << 32);
    }
}

// slab < 0: the block is this workgroup's alone (plain stores of the next-frontier words), else several units share it (atomicOr)
__global__ __launch_bounds__(VGL_BTHREADS) void vgl_k_bfs_blk_accumulate(const vgl_blk_unit *units, const uint16_t *a_lo, const uint64_t *bits,
                                                                         int32_t a_count, const uint64_t *visited, uint64_t *next, int32_t *levels,
                                                                         int32_t next_level)
{
    __shared__ uint32_t s_d[VGL_BLK / 32 + VGL_CHUNK / 32];              // + the dummy destinations of pad entries
    const vgl_blk_unit u = units[blockIdx.x];
    s_d[threadIdx.x] = 0;
    if (threadIdx.x < VGL_CHUNK / 32) s_d[VGL_BLK / 32 + threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 3, off = (lane & 7) * 8;
    for (uint32_t j0 = u.chunk0 + wave * VGL_BGROUP; j0 < u.chunk1; j0 += VGL_BWAVES * VGL_BGROUP) {
        const uint32_t j = j0 + sub;
        if (j >= u.chunk1) continue;
        const uint32_t mask = (uint32_t)(bits[j] >> off) & 0xFFu;       // this lane's eight entries
        if (mask == 0) continue;                                        // (their indices are not even read)
        const uint4 al = *reinterpret_cast<const uint4 *>(a_lo + (size_t)j * VGL_CHUNK + off);
        const uint32_t a[4] = {al.x, al.y, al.z, al.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t lo = a[k] & 0xFFFFu, hi = a[k] >> 16;
            if (mask & (1u << (2 * k))) atomicOr(&s_d[lo >> 5], 1u << (lo & 31));
            if (mask & (2u << (2 * k))) atomicOr(&s_d[hi >> 5], 1u << (hi & 31));
        }
    }
    __syncthreads();
    // one 32-bit word of the block per thread: discovered and not visited before -> next frontier + levels
    const int32_t base = u.block << VGL_BLK_BITS;
    const int32_t v0 = base + (int32_t)threadIdx.x * 32;
    if (v0 >= a_count) return;
    const size_t wi = ((size_t)u.block << (VGL_BLK_BITS - 5)) + threadIdx.x;
    uint32_t n = s_d[threadIdx.x] & ~reinterpret_cast<const uint32_t *>(visited)[wi];
    if (n == 0) return;
    if (u.slab < 0) reinterpret_cast<uint32_t *>(next)[wi] = n;
    else atomicOr(reinterpret_cast<uint32_t *>(next) + wi, n);
    while (n) {
        const int b = __ffs(n) - 1;
        n &= n - 1;
        levels[v0 + b] = next_level;
    }
}

// Degrees of the rows behind a wavefront's 64 bitmap words, LANE = BIT: a frontier of a degree-sorted graph sits in runs of consecutive ids (the
// hubs of an early level, the short rows of a late one), i.e. in a few FULL words -- with one thread walking the bits of its own word, 300
// threads did the row-offset reads of a 6 800-vertex level one after the other (82 us for the write pass) and a 5 M-vertex level took 198 us.
// Here the wavefront takes its non-empty words in turn (VGL_WW_BATCH at a time, so that their loads are in flight together): lane b reads the row
// offsets of bit b -- consecutive rows, two coalesced 512-byte reads per word -- and `visit(j, set, deg)` sees word j's bit of this lane.
#ifndef VGL_WW_BATCH_VALUE
#define VGL_WW_BATCH_VALUE 8
#endif
constexpr int VGL_WW_BATCH = VGL_WW_BATCH_VALUE;      // non-empty words whose row-offset loads are in flight together
template <class Visit>
__device__ __forceinline__ void vgl_wave_words_degrees(uint64_t w, int64_t first_row_of_wave, const int64_t *rowptr, Visit &&visit)
{
    const int lane = vgl_lane();
    unsigned long long nonempty = __ballot(w != 0);
    while (nonempty) {
        int j[VGL_WW_BATCH];
        uint64_t wj[VGL_WW_BATCH];
        int64_t lo[VGL_WW_BATCH], hi[VGL_WW_BATCH];
#pragma unroll
        for (int k = 0; k < VGL_WW_BATCH; k++) {
            j[k] = nonempty ? __ffsll((long long)nonempty) - 1 : -1;
            if (nonempty) nonempty &= nonempty - 1;
            wj[k] = j[k] >= 0 ? __shfl(w, j[k]) : 0ULL;
            lo[k] = hi[k] = 0;
            if ((wj[k] >> lane) & 1) {
                const int64_t r = first_row_of_wave + ((int64_t)j[k] << 6) + lane;
                lo[k] = rowptr[r]; hi[k] = rowptr[r + 1];
            }
        }
#pragma unroll
        for (int k = 0; k < VGL_WW_BATCH; k++)
            if (j[k] >= 0) visit(j[k], (bool)((wj[k] >> lane) & 1), hi[k] - lo[k]);
    }
}
// lane = bit pays when the wavefront's words are well filled; a sparse bitmap (a bit or two per non-empty word) is better served by every thread
// walking its own word, all words at once
__device__ __forceinline__ bool vgl_wave_words_dense(uint64_t w)
{
    const int words = __popcll(__ballot(w != 0));
    int bits = __popcll(w);
    for (int o = 32; o > 0; o >>= 1) bits += __shfl_xor(bits, o);
    return bits >= 8 * words && words > 0;
}
__device__ __forceinline__ int64_t vgl_wave_sum_i64(int64_t v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- frontier generation from the frontier BITMAP (small frontiers): one 64-bit word per thread, 256 words per workgroup ----
// The last workgroup to finish also does what used to be two more launches: the exclusive scan of the per-workgroup counts
// (<= a few thousand entries) and the hand-over of F and M to the host (counters + pinned mirror + sequence number).
// ADVANCE: the launch also turns the discoveries of the level before into the frontier it counts (what vgl_k_bm_advance does:
// visited |= next, front = next, next = 0) -- after a top-down level the count always follows, so the two passes over the same
// words share one launch.
template <bool ADVANCE>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_bm_gnf_count(int64_t nwords, int64_t word0, int32_t row_base, uint64_t *front,
                                                                const int64_t *rowptr, int32_t *vt_cnt, int64_t *vt_deg,
                                                                int32_t *vt_cnt_off, int64_t *vt_deg_off, int64_t *offs, int64_t *counters,
                                                                uint32_t *ticket, volatile int64_t *host, int64_t seq, uint64_t *visited,
                                                                uint64_t *next, int use_hint, vgl_do_hint hint)
{
    __shared__ int64_t s64[VGL_WAVES];
    __shared__ int s32[VGL_WAVES];
    const int64_t wi = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x;
    if (ADVANCE && use_hint) {
        // The launch before this one (an emitting top-down level, or the list kernel) left the size F and the out-degree sum M of the
        // frontier it produced.  When the direction rule -- the host evaluates the same integers right after this launch -- turns the
        // level bottom-up, nobody needs the rows' degrees or the compaction offsets: only the bitmaps move on (what vgl_k_bm_advance
        // does) and the first workgroup hands F and M over at once.
        const int64_t nf = counters[C_NEXT_F], nm = counters[C_NEXT_M];
        if (nf > hint.prev_f && nm >= ((hint.V - hint.visited_total - nf) * hint.factor + hint.V) / VGL_DO_ALPHA) {
            if (wi < nwords) {
                const uint64_t w = next[word0 + wi];
                front[word0 + wi] = w;
                if (w) { visited[word0 + wi] |= w; next[word0 + wi] = 0; }
            }
            if (blockIdx.x == 0 && threadIdx.x == 0) { host[C_SKIPPED] = 1; vgl_publish2(counters, host, seq, C_FRONT, nf, C_NEIGH, nm); }
            return;
        }
    }
    int cnt = 0;
    int64_t deg = 0;
    uint64_t wbits = 0;
    if (wi < nwords) {
        uint64_t w;
        if (ADVANCE) {
            w = next[word0 + wi];
            front[word0 + wi] = w;
            if (w) { visited[word0 + wi] |= w; next[word0 + wi] = 0; }
        } else w = front[word0 + wi];
        cnt = __popcll(w);
        wbits = w;
    }
    // (a partial sum per lane: lane b adds the degrees of bit b of every word of its wavefront; the block total is what counts)
    if (vgl_wave_words_dense(wbits))
        vgl_wave_words_degrees(wbits, ((word0 + wi - vgl_lane()) << 6) - row_base, rowptr, [&](int, bool set, int64_t d) { if (set) deg += d; });
    else
        for (uint64_t t = wbits; t; t &= t - 1) {
            const int64_t r = ((word0 + wi) << 6) + (__ffsll((long long)t) - 1) - row_base;
            deg += rowptr[r + 1] - rowptr[r];
        }
    const int tc = vgl_block_reduce_add(cnt, s32);
    const int64_t td = vgl_block_reduce_add(deg, s64);
    uint32_t dep = 0;
    if (threadIdx.x == 0) dep = vgl_put_agent(vt_cnt + blockIdx.x, tc) ^ vgl_put_agent(vt_deg + blockIdx.x, td);
    if (!vgl_last_block(ticket, dep)) return;
    const int nb = (int)gridDim.x;
    const int per = (nb + VGL_BLOCK - 1) / VGL_BLOCK;
    const int lo = min(nb, (int)threadIdx.x * per), hi = min(nb, lo + per);
    int c = 0;
    int64_t d = 0;
    for (int t = lo; t < hi; t++) { c += vgl_load_agent(vt_cnt + t); d += vgl_load_agent(vt_deg + t); }
    int ctot;
    int64_t dtot;
    int cpre = vgl_block_excl_add(c, s32, &ctot);
    int64_t dpre = vgl_block_excl_add(d, s64, &dtot);
    for (int t = lo; t < hi; t++) {
        vt_cnt_off[t] = cpre; vt_deg_off[t] = dpre;
        cpre += vgl_load_agent(vt_cnt + t);
        dpre += vgl_load_agent(vt_deg + t);
    }
    if (threadIdx.x == 0) {
        offs[ctot] = dtot;
        host[C_SKIPPED] = 0;
        vgl_publish2(counters, host, seq, C_FRONT, (int64_t)ctot, C_NEIGH, dtot);
    }
}
// ---- the frontier generation after a large (non-emitting) top-down level, when all the traversal may need of it is the BITMAPS ----
// Such a level leaves its discoveries in `levels` only, and vgl_k_gnf_count<vgl_pred_equal_i32> rebuilds bitmaps, F and M from them: 64 MiB
// of levels plus the row offsets of every frontier vertex (up to 128 MiB more) -- 52 us on RMAT-24, more than the level itself -- and right
// after it the direction rule usually turns the traversal bottom-up, which needs the bitmaps and F but of M only "at least the threshold".
// This pass reads the levels alone: bitmaps, F, and a LOWER BOUND of M from the per-tile frontier sizes times the smallest out-degree of
// the tile (vt_min_deg; on a degree-sorted graph the rows of a tile have similar degrees, so the bound is tight where the edges are).
// When the bound already satisfies the rule the traversal goes on bottom-up; otherwise the exact sizes are taken from the bitmap just built
// (vgl_k_bm_gnf_count), not from a second scan of the levels.  Frontier generations 24.5 -> 17.5 us per RMAT-24 traversal.
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_bfs_scan_bound(int32_t nrows, int32_t row_base, const int32_t *levels, int32_t level,
                                                                 const int32_t *vt_min_deg, uint8_t *front_bytes, uint8_t *visited_bytes,
                                                                 int64_t *partials, uint32_t *ticket, int64_t *counters,
                                                                 volatile int64_t *host, int64_t seq)
{
    __shared__ int s32[VGL_WAVES];
    __shared__ int64_t s64[VGL_WAVES];
    const int32_t r0 = blockIdx.x * VGL_TILE + threadIdx.x * VGL_EPT;
    int cnt = 0;
    if (r0 < nrows) {
        const int nvalid = min(VGL_EPT, nrows - r0);
        const int32_t v0 = row_base + r0;
        uint32_t aux;
        const vgl_pred_equal_i32 pred{levels, level};
        const uint32_t bits = pred.bits8(v0, nvalid, &aux);
        cnt = __popc(bits);
        front_bytes[v0 >> 3] = (uint8_t)bits;
        visited_bytes[v0 >> 3] = (uint8_t)aux;
    }
    const int tc = vgl_block_reduce_add(cnt, s32);
    uint32_t dep = 0;
    if (threadIdx.x == 0)
        dep = vgl_put_agent(partials + 2 * blockIdx.x, (int64_t)tc) ^ vgl_put_agent(partials + 2 * blockIdx.x + 1, (int64_t)tc * (int64_t)vt_min_deg[blockIdx.x]);
    if (!vgl_last_block(ticket, dep)) return;
    int64_t f = 0, m = 0;
    for (int t = threadIdx.x; t < (int)gridDim.x; t += VGL_BLOCK) { f += vgl_load_agent(partials + 2 * t); m += vgl_load_agent(partials + 2 * t + 1); }
    f = vgl_block_reduce_add(f, s64);
    m = vgl_block_reduce_add(m, s64);
    if (threadIdx.x == 0) vgl_publish2(counters, host, seq, C_FRONT, f, C_NEIGH, m);
}

// write pass; also fills tile_first (the frontier position that owns edge t*VGL_TILE, and, in entry [#tiles], the owner of the
// last edge) -- every frontier vertex knows its own edge range, so the separate vgl_k_tile_first launch is not needed
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_bm_gnf_write(int64_t nwords, int64_t word0, int32_t row_base, const uint64_t *front,
                                                                const int64_t *rowptr, const int32_t *vt_cnt_off, const int64_t *vt_deg_off,
                                                                int32_t *ids, int64_t *offs, int32_t *tile_first, int64_t M)
{
    __shared__ int64_t s64[VGL_WAVES];
    __shared__ int s32[VGL_WAVES];
    const int64_t wi = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x;
    const int lane = vgl_lane();
    const uint64_t w = wi < nwords ? front[word0 + wi] : 0ULL;
    const int64_t first_row = ((word0 + wi - lane) << 6) - row_base;           // row of bit 0 of the wavefront's first word
    // degree sum of every word (lane j ends up with the sum of word j), then the exclusive offsets of the words in bitmap order
    int64_t word_deg = 0;
    const bool dense = vgl_wave_words_dense(w);
    if (dense)
        vgl_wave_words_degrees(w, first_row, rowptr, [&](int j, bool set, int64_t d) {
            const int64_t sum = vgl_wave_sum_i64(set ? d : 0);
            if (lane == j) word_deg = sum;
        });
    else
        for (uint64_t t = w; t; t &= t - 1) {
            const int64_t r = first_row + ((int64_t)lane << 6) + (__ffsll((long long)t) - 1);
            word_deg += rowptr[r + 1] - rowptr[r];
        }
    int ctot; int64_t dtot;
    const int word_pos = vt_cnt_off[blockIdx.x] + vgl_block_excl_add((int)__popcll(w), s32, &ctot);
    const int64_t word_eoff = vt_deg_off[blockIdx.x] + vgl_block_excl_add(word_deg, s64, &dtot);
    if (!dense) {                            // every thread emits the vertices of its own word
        int pos = word_pos;
        int64_t eoff = word_eoff;
        for (uint64_t t = w; t; t &= t - 1) {
            const int64_t r = first_row + ((int64_t)lane << 6) + (__ffsll((long long)t) - 1);
            ids[pos] = (int32_t)(r + row_base); offs[pos] = eoff;
            const int64_t eend = eoff + (rowptr[r + 1] - rowptr[r]);
            for (int64_t q = (eoff + VGL_TILE - 1) / VGL_TILE; q < (eend + VGL_TILE - 1) / VGL_TILE; q++) tile_first[q] = pos;
            if (eoff < eend && eend == M) tile_first[(M + VGL_TILE - 1) / VGL_TILE] = pos;
            eoff = eend;
            pos++;
        }
        return;
    }
    // second walk (the row offsets now come from the cache): lane b of word j writes frontier position word_pos[j] + (set bits below b)
    vgl_wave_words_degrees(w, first_row, rowptr, [&](int j, bool set, int64_t d) {
        const uint64_t wj = __shfl(w, j);
        const int base_pos = __shfl(word_pos, j);
        const int64_t base_eoff = __shfl(word_eoff, j);
        const int64_t mine = set ? d : 0;
        const int64_t eoff = base_eoff + vgl_wave_incl_add(mine) - mine;
        if (!set) return;
        const int pos = base_pos + __popcll(wj & ((1ULL << lane) - 1ULL));
        ids[pos] = (int32_t)(first_row + row_base + ((int64_t)j << 6) + lane);
        offs[pos] = eoff;
        const int64_t eend = eoff + d;
        for (int64_t t = (eoff + VGL_TILE - 1) / VGL_TILE; t < (eend + VGL_TILE - 1) / VGL_TILE; t++) tile_first[t] = pos;
        if (eoff < eend && eend == M) tile_first[(M + VGL_TILE - 1) / VGL_TILE] = pos;
    });
}

// ---- small frontiers: several top-down levels in ONE workgroup ----
// The first and the last levels of a traversal hold a handful of vertices; through the ordinary path each of them costs a count
// launch, a host round trip, a write launch and an expand launch (~45 us) for microseconds of work.  This kernel keeps going from a
// list of at most VGL_SMALL_F frontier vertices for as long as the next frontier fits the same bounds (list entries, edges <= cap_m):
// the level's edges are dealt flat to the 1024 threads (owner by binary search in the LDS prefix of the degrees), four per thread
// and round so that their loads and atomics overlap -- one workgroup has little else to hide latency with.  A vertex is claimed by
// the atomicOr on the visited bitmap (its return value decides, so a stale cached word only costs an extra atomic); the claimer
// writes the level, ORs the bit into the next bitmap and appends the id to the next list.  On leaving with a frontier that does not
// fit (too many entries or edges) the state is exactly that after an emitting top-down level: discoveries in `next`, the ordinary
// loop continues with count<ADVANCE>.  Within a level of the lists `next` is cleared again for the vertices being expanded.
// source >= 0: the list is {source} (first level, nothing counted yet); if it has more than cap_m edges nothing is done (run = 0).
// Hand-over (counters + pinned mirror): C_FRONT = size of the frontier left (0: traversal finished), C_TMP0 = levels run,
// C_EDGES = edges examined, C_TMP1 = frontier vertices of the levels run AFTER the first one, C_JUMP = size of the last level run,
// C_NEIGH = edges of the first level (source mode).
constexpr int VGL_SMALL_F = 2048;
// One top-down level straight from the frontier BITMAP (one word per thread), for the switch from bottom-up back to top-down: the
// host knows the frontier is small-ish (its size is the bottom-up step's found count) but has neither ids nor edge counts, and the
// rows left that late are short.  Rows of at most 32 edges are walked by their thread, longer ones by the wavefront.  Discoveries
// are claimed by the atomicOr on `next` (state afterwards = after an emitting top-down level) and, while they are few, appended to
// `list` so that vgl_k_bfs_small_levels can carry on without a count / host / write round: *list_count may run past `cap`, which
// then means "too many".  edge_partials[block] = edges examined.
constexpr int VGL_BM_EXPAND_THREAD_ROW = 32;
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_bm_expand(int64_t nwords, const uint64_t *front, const int64_t *rowptr, const int32_t *adj,
                                                             const uint64_t *visited, uint64_t *next, int32_t *levels, int32_t next_level,
                                                             int32_t *list, unsigned long long *list_count, int32_t cap, int64_t *edge_partials)
{
    __shared__ int64_t s64[VGL_WAVES];
    const int64_t wi = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x;
    uint64_t w = wi < nwords ? front[wi] : 0ULL;
    uint64_t big = 0;
    int64_t my_edges = 0;
    auto visit = [&](int32_t dst) {
        const unsigned long long bit = 1ULL << (dst & 63);
        if (visited[dst >> 6] & bit) return;
        const unsigned long long old = atomicOr((unsigned long long *)&next[dst >> 6], bit);
        if (old & bit) return;
        levels[dst] = next_level;
        if (__hip_atomic_load(list_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= (unsigned long long)cap) {
            const unsigned long long pos = atomicAdd(list_count, 1ULL);
            if (pos < (unsigned long long)cap) list[pos] = dst;
        }
    };
    while (w) {
        const int b = __ffsll((long long)w) - 1;
        w &= w - 1;
        const int64_t v = (wi << 6) + b;
        const int64_t lo = rowptr[v], hi = rowptr[v + 1];
        my_edges += hi - lo;
        if (hi - lo > VGL_BM_EXPAND_THREAD_ROW) { big |= 1ULL << b; continue; }
        for (int64_t q = lo; q < hi; q++) visit(adj[q]);
    }
    // longer rows: the wavefront walks them one after the other (lanes stride over the row)
    unsigned long long todo = __ballot(big != 0);
    while (todo) {
        const int src_lane = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int64_t base = __shfl(wi, src_lane) << 6;
        uint64_t m = __shfl(big, src_lane);
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int64_t lo = rowptr[base + b], hi = rowptr[base + b + 1];
            for (int64_t q = lo + vgl_lane(); q < hi; q += 64) visit(adj[q]);
        }
    }
    const int64_t tot = vgl_block_reduce_add(my_edges, s64);
    if (threadIdx.x == 0) edge_partials[blockIdx.x] = tot;
}
constexpr int VGL_SMALL_THREADS = 1024;
constexpr int VGL_SMALL_UNROLL = 4;            // edges per thread and round: their loads / atomics are in flight together
__global__ __launch_bounds__(VGL_SMALL_THREADS) void vgl_k_bfs_small_levels(int32_t *ids_in, int32_t F0, int32_t source, const int64_t *rowptr,
                                                                            const int32_t *adj, uint64_t *visited, uint64_t *next, int32_t *levels,
                                                                            int32_t level0, int64_t cap_m, int64_t *counters, volatile int64_t *host,
                                                                            int64_t seq, unsigned long long *list_count, const int64_t *edge_partials,
                                                                            int n_partials, uint64_t *front, int64_t *offs_out, int32_t *tile_first)
{
    constexpr int NT = VGL_SMALL_THREADS, NW = NT / 64, U = VGL_SMALL_UNROLL;
    __shared__ int32_t s_list[2][VGL_SMALL_F];
    __shared__ int64_t s_beg[VGL_SMALL_F];                    // first adjacency entry of list entry i
    __shared__ int32_t s_off[VGL_SMALL_F + 2];                // exclusive prefix of the list's degrees (M <= cap_m < 2^31)
    __shared__ int32_t s_wave[NW];
    __shared__ int s_cnt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int cur = 0;
    int32_t F = source >= 0 ? 1 : F0;
    // list_count != nullptr: the list comes from vgl_k_bm_expand (length on the device, entries also set in `next` but not in
    // `visited` yet); its edge count is folded here so that the host gets everything in one read (C_CHANGED, list length in C_BU_FOUND)
    bool usable = true;
    int64_t n_list = 0, bm_edges = 0;
    if (list_count) {
        n_list = (int64_t)*list_count;
        usable = n_list > 0 && n_list <= VGL_SMALL_F;
        F = usable ? (int32_t)n_list : 0;
        for (int i = tid; i < n_partials; i += NT) bm_edges += edge_partials[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bm_edges += __shfl_xor(bm_edges, o);
        __shared__ int64_t s_bm[NW];
        if (lane == 0) s_bm[wave] = bm_edges;
        __syncthreads();                                       // (also: everyone has read *list_count)
        bm_edges = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) bm_edges += s_bm[w];
        if (tid == 0) *list_count = 0ULL;                      // ready for the next traversal
    }
    for (int i = tid; i < F; i += NT) {
        const int32_t v = source >= 0 ? source : ids_in[i];
        s_list[0][i] = v;
        if (list_count) atomicOr((unsigned long long *)&visited[v >> 6], 1ULL << (v & 63));
    }
    int32_t level = level0, run = 0, last_f = 0, handed = 0, hinted = 0;
    int64_t edges = 0, later_front = 0, first_m = 0, exit_m = 0, next_m = 0;
    __shared__ int64_t s_deg[NW];
    for (;;) {
        if (!usable) break;                                    // (uniform) empty or overflowed list: nothing is done
        __syncthreads();                                       // the list of this level is complete (and everyone has read the old s_cnt)
        // degrees of the (at most 2 per thread) list entries, exclusive prefix over the workgroup
        int64_t b0 = 0, b1 = 0, d0 = 0, d1 = 0;
        const int i0 = 2 * tid, i1 = 2 * tid + 1;
        if (i0 < F) { const int32_t v = s_list[cur][i0]; b0 = rowptr[v]; d0 = rowptr[v + 1] - b0; }
        if (i1 < F) { const int32_t v = s_list[cur][i1]; b1 = rowptr[v]; d1 = rowptr[v + 1] - b1; }
        const int64_t mine = d0 + d1;
        const int64_t capped = mine > cap_m ? cap_m + 1 : mine;          // keeps the 32-bit scan exact whenever the level is taken
        int inc = vgl_wave_incl_add((int)capped);
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        int64_t M = 0;
        int base = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) { if (w < wave) base += s_wave[w]; M += s_wave[w]; }
        if (run == 0) first_m = M;
        if (M > cap_m) {                                       // too many edges for one workgroup: the ordinary path takes this level
            // Hand-over: the list IS what that path's count + write passes would produce, so finish their job here -- ids, exact edge
            // offsets, tile owners, the frontier bitmap (the entry level's bits were taken out below), `next` empty again -- and the
            // host launches the expand (or the bottom-up step) without a count / host / write round.  Not when the entry frontier
            // came from vgl_k_bm_expand: its bits in `front` are unknown here.
            if (run > 0 && !list_count) {
                __shared__ int64_t s_w64[NW];
                const int64_t inc64 = vgl_wave_incl_add(mine);
                if (lane == 63) s_w64[wave] = inc64;
                __syncthreads();
                int64_t base64 = 0;
                exit_m = 0;
#pragma unroll
                for (int w = 0; w < NW; w++) { if (w < wave) base64 += s_w64[w]; exit_m += s_w64[w]; }
                int64_t eoff = base64 + inc64 - mine;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int i = 2 * tid + k;
                    const int64_t d = k ? d1 : d0;
                    if (i < F) {
                        const int32_t v = s_list[cur][i];
                        const unsigned long long bit = 1ULL << (v & 63);
                        ids_in[i] = v; offs_out[i] = eoff;
                        const int64_t eend = eoff + d;
                        for (int64_t t = (eoff + VGL_TILE - 1) / VGL_TILE; t < (eend + VGL_TILE - 1) / VGL_TILE; t++) tile_first[t] = i;
                        if (eoff < eend && eend == exit_m) tile_first[(exit_m + VGL_TILE - 1) / VGL_TILE] = i;
                        atomicOr((unsigned long long *)&front[v >> 6], bit);
                        atomicAnd((unsigned long long *)&next[v >> 6], ~bit);
                        eoff = eend;
                    }
                }
                if (tid == 0) offs_out[F] = exit_m;
                handed = 1;
            }
            break;
        }
        const int excl = base + inc - (int)capped;
        if (i0 < F) { s_beg[i0] = b0; s_off[i0] = excl; }
        if (i1 < F) { s_beg[i1] = b1; s_off[i1] = excl + (int)d0; }
        if (tid == 0) { s_cnt = 0; s_off[F] = (int)M; }
        if (run > 0 || list_count)                             // these vertices are the frontier now, not discoveries any more
            for (int i = tid; i < F; i += NT) { const int32_t v = s_list[cur][i]; atomicAnd((unsigned long long *)&next[v >> 6], ~(1ULL << (v & 63))); }
        else                                                   // entry level taken: its bits leave `front` (a hand-over rebuilds it from a list)
            for (int i = tid; i < F; i += NT) { const int32_t v = s_list[cur][i]; atomicAnd((unsigned long long *)&front[v >> 6], ~(1ULL << (v & 63))); }
        __syncthreads();                                       // offsets staged; (a discovery below can share a word with a bit cleared above)
        const int m = (int)M;
        int64_t found_deg = 0;                                 // out-degrees of the vertices this thread claims on this level
        for (int e0 = 0; e0 < m; e0 += NT * U) {               // uniform trip count
            int32_t dst[U];
            unsigned long long word[U], old[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int e = e0 + u * NT + tid;
                dst[u] = -1;
                if (e < m) {
                    int lo = 0, hi = F;                        // entry k with s_off[k] <= e < s_off[k + 1]
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off[mid] <= e) lo = mid; else hi = mid; }
                    dst[u] = adj[s_beg[lo] + (e - s_off[lo])];
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) word[u] = dst[u] >= 0 ? visited[dst[u] >> 6] : ~0ULL;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const unsigned long long bit = dst[u] >= 0 ? 1ULL << (dst[u] & 63) : 0ULL;
                old[u] = (word[u] & bit) || dst[u] < 0 ? ~0ULL : atomicOr((unsigned long long *)&visited[dst[u] >> 6], bit);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (dst[u] < 0) continue;
                const unsigned long long bit = 1ULL << (dst[u] & 63);
                if (old[u] & bit) continue;                    // the return value of the atomic decides: exactly one claimer per vertex
                levels[dst[u]] = level + 1;
                atomicOr((unsigned long long *)&next[dst[u] >> 6], bit);
                const int pos = atomicAdd(&s_cnt, 1);
                if (pos < VGL_SMALL_F) s_list[cur ^ 1][pos] = dst[u];
                else found_deg += rowptr[dst[u] + 1] - rowptr[dst[u]];       // (only a level that overflows the list gets here; the listed ones below)
            }
        }
        __syncthreads();                                       // every append of this level has happened
        const int32_t Fn = s_cnt;
        edges += M; run++; last_f = F;
        if (run > 1) later_front += F;
        F = Fn;
        if (Fn > VGL_SMALL_F) {                                // the next frontier does not fit the list (it is all in `next`): leave its
            {                                                  // size and out-degree sum for the count launch that follows (C_NEXT_F / C_NEXT_M)
                static_assert(VGL_SMALL_F == 2 * NT, "two listed vertices per thread");
                const int32_t v0 = s_list[cur ^ 1][tid], v1 = s_list[cur ^ 1][tid + NT];
                const int64_t a0 = rowptr[v0], a1 = rowptr[v0 + 1], c0 = rowptr[v1], c1 = rowptr[v1 + 1];
                found_deg += (a1 - a0) + (c1 - c0);
            }
            found_deg = vgl_wave_incl_add(found_deg);
            if (lane == 63) s_deg[wave] = found_deg;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < NW; w++) next_m += s_deg[w];
            hinted = 1;
            break;
        }
        if (Fn == 0) break;                                    // finished
        cur ^= 1; level++;
    }
    if (tid == 0) {
        counters[C_FRONT] = F; counters[C_TMP0] = run; counters[C_EDGES] = edges; counters[C_TMP1] = later_front; counters[C_JUMP] = last_f;
        counters[C_NEIGH] = first_m; counters[C_BU_FOUND] = n_list; counters[C_CHANGED] = bm_edges;
        if (hinted) { counters[C_NEXT_F] = F; counters[C_NEXT_M] = next_m; }
        host[C_HINT] = hinted;
        host[C_FRONT] = F; host[C_TMP0] = run; host[C_EDGES] = edges; host[C_TMP1] = later_front; host[C_JUMP] = last_f; host[C_NEIGH] = first_m;
        host[C_BU_FOUND] = n_list; host[C_CHANGED] = bm_edges; host[C_HEAVY] = handed; host[C_BU_EDGES] = exit_m;
        __threadfence_system();
        host[C_NSLOTS] = seq;
        __threadfence_system();
    }
}

// Bottom-up step.  No global atomics: a single same-address device atomic costs ~12 ns and serialises (65 536 blocks
// adding to one counter took 1.5 ms per launch in the first version); instead a fixed grid of VGL_BU_BLOCKS persistent
// workgroups each owns a contiguous vertex range, a private segment of the deferred-vertex list and a private slot of
// partial counters, which the last workgroup of the second pass sums in a fixed order.
constexpr int VGL_BU_BLOCKS = 2048;

// pass 1: one thread per owned vertex.  A vertex is a candidate when it is unvisited AND has incoming edges (in_nz
// bitmap, built once per graph: ~45 % of RMAT vertices have none and would otherwise re-read 16 B of row offsets in every
// bottom-up level).  The first VGL_BU_PROBES incoming neighbours are loaded together and their frontier bits tested
// together (two dependent memory round trips per vertex).  Writes whole words of the next-frontier bitmap.
typedef int vgl_int4_u __attribute__((ext_vector_type(4), aligned(4)));     // 16-byte load from a 4-byte aligned address

// Counters of this kernel before the candidate words were fetched 64 at a time (profiles/r02_pmc_bfs.json, per 60 us launch): 3 M L2
// requests and 67 MB fetched -- neither the L2 request rate (~2e11 /s) nor the bandwidth is near its limit; the wavefronts were parked
// on memory 77 % of their cycles.  Tried on top of that and dropped (no gain): non-temporal loads for the once-read streams (the hot
// frontier words are not being evicted from L1), the first 2048 words of the frontier bitmap in LDS (16 KiB per workgroup: 61.9 vs
// 62.5 us; 4096 words halve the occupancy: 94 us -- the kernel's rate follows the number of resident wavefronts), a two-phase form
// (uniform first probe for all candidates with four row groups in flight, misses compacted and probed densely: 78 us), requesting
// the next group's head records ahead (also as a true software pipeline: this group's first probe issued, then the next group's records
// issued unconditionally, s_waitcnt vmcnt(1) checked in the ISA: 49.1 vs 48 us -- a wavefront's dependent chain is not the limit either, the
// other resident wavefronts already cover it).  The head records hold each row's eight SMALLEST ids (= its best-connected in-neighbours
// under the degree renumbering; vgl_k_row_heads) instead of the first eight: on RMAT the first bottom-up level then sends half as many
// candidates to the second round of probes and a quarter to the deferred pass (simulated on scale 20, tests/studies/bu_head_order.py).
// On RMAT-24 that changes nothing in this kernel (49 us either way: its time does not follow the number of probe rounds) and the
// traversal is 1-2 % slower (the deferred pass now scans whole rows); on RMAT-27, where the deferred pass had grown to half of this
// kernel's time (301 us per launch against 594), the traversal goes from 3.13 to 2.11 ms (probe 408, deferred pass 91 us).
// What a divergent load costs was then measured on its own (profiles/microbench/ta_rate_bench.hip, profiles/r02_ta_rate_bench.log): 2.3
// clocks per ACTIVE LANE when it misses L1 (0.6 when it hits), never less than 8 (4-byte) / 17 (8-byte) clocks per instruction -- 2.65e11
// lane-loads/s for the chip whatever the occupancy; a random LDS read costs 0.1 clock per lane.  On that basis the kernel was rebuilt
// around an LDS copy of the frontier bits of the first 2^20 ids (one 1024-thread workgroup per CU, 128 KiB window, smallest-id heads so
// that 96 % of the first probes land in it, four groups per round, memory only for entries outside the window), in three stages, each
// traced per launch on RMAT-24 (first / second / third bottom-up level; this kernel: 80 / 30 / 14 us):
//   window + guarded loads, stores as here             80 / 43 / 18   (the compiler waits for each guarded load before issuing the next)
//   + unconditional loads issued together, NO store in the loop (found masks stored once per lane after it, levels written from the
//     bitmap by vgl_k_bm_advance: +9 / +5 / +0 us there)                                    64 / 37 / 15
//   + next round's head records requested before this round's second step                  64 / 41 / 16
// Removing every global gather did not change the first level; taking the stores out did (a wavefront cannot wait for a load without
// waiting for every store issued before it -- one in-order counter -- and a partial-line store takes ~1.4 us to be acknowledged), but the
// levels still have to be written somewhere; overlapping the dependent steps changed nothing.  With synthetic head records the same launch
// takes 36 us, without the loop 8: what is left is the head records themselves -- ~250 MB of 128-byte lines on the first level, i.e. the
// launch runs at ~4 TB/s of lines touched.  The traversal took 0.396-0.400 ms with the rebuilt kernel against 0.389-0.404 with this one, so
// this one (simpler, no assumption about where the hubs are numbered) stayed.
// Per-launch durations (profiles/microbench/bfs_launches.py on a kernel trace): first bottom-up level 58-120 us, second 20-52, third
// 12-18; an empty deferred pass costs 9.5 us, vgl_k_bm_advance 4.2.  Also tried: chaining the bottom-up levels on the device (the last workgroup of a level evaluates the
// switch rule, speculative launches of the next levels return at once when it says stop; one host wait per chain): 0.394 ms per
// traversal with three levels per wait against 0.389 with one -- the polled hand-over costs less than the extra launches.
// INLINE_HEAVY (round 3): the rows a workgroup defers are scanned by the same workgroup right after its probe loop, 16 lanes per row, and
// the last workgroup folds the counters and hands them to the host -- no second launch (vgl_k_bu_heavy: ~15 us per level even when it has
// nothing to scan, three levels per traversal).  The 64-row groups are dealt round-robin to the wavefronts of the grid, so the deferred
// rows -- the hubs, i.e. the first ids of a degree-sorted graph -- are spread over all workgroups by construction; the balanced second
// pass dates from the time of contiguous chunks.  VGL_BU_SPLIT=1 keeps the two-launch form.
template <bool INLINE_HEAVY>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_bu_probe(int32_t nrows, int32_t row_base, int32_t chunk, const int64_t *in_rowptr,
                                                            const int32_t *in_adj, int64_t in_edges, const uint64_t *visited, const uint64_t *in_nz,
                                                            const uint64_t *front, uint64_t *next, int32_t *levels,
                                                            int32_t next_level, int32_t *heavy, int32_t *heavy_cnt, int64_t *partials,
                                                            int32_t *heavy_off, uint32_t *ticket, const int4 *in_head, const uint64_t *in_long,
                                                            int64_t *counters, volatile int64_t *host, int64_t seq, const int32_t *nz_rank, int32_t nz_rows)
{
    __shared__ int64_t s64[VGL_WAVES];
    __shared__ int s_nheavy;
    if (threadIdx.x == 0) s_nheavy = 0;
    __syncthreads();
    auto in_front = [&](int32_t u) -> uint32_t { return (uint32_t)((front[u >> 6] >> (u & 63)) & 1ULL); };
    int64_t found_cnt = 0, probes = 0;
    // The 64-row groups are dealt round-robin to the wavefronts of the grid (candidates cluster -- in a degree-sorted graph the unvisited
    // vertices of the later levels are the low-degree tail -- so contiguous chunks would leave most workgroups idle), and a wavefront
    // fetches the candidate words of 64 of its groups with ONE load per lane, then walks the groups that hold candidates.  (One group
    // per iteration with its own candidate-word load was a chain of 32 dependent L2 round trips per thread: 31 us per launch even
    // for a level with no candidates at all -- the last bottom-up level of every traversal -- and three round trips per group otherwise.)
    const int64_t groups = ((int64_t)nrows + 63) >> 6;
    const int64_t word0 = (int64_t)row_base >> 6;                       // row_base is a multiple of 64 (shards own whole bitmap words)
    constexpr int64_t NW = (int64_t)VGL_BU_BLOCKS * VGL_WAVES;
    int32_t *my_heavy = heavy + (int64_t)blockIdx.x * chunk;            // at most `chunk` deferrals per workgroup
    const int lane = vgl_lane();
    for (int64_t g0 = (int64_t)blockIdx.x * VGL_WAVES + vgl_wave(); g0 < groups; g0 += NW * 64) {
        const int64_t mine = g0 + (int64_t)lane * NW;                   // lane l holds the candidate word of group g0 + l * NW
        uint64_t cw = 0, nzw = 0;
        int32_t nzr = 0;
        if (mine < groups) {
            nzw = in_nz[word0 + mine];
            cw = ~visited[word0 + mine] & nzw;
            if (cw == 0ULL) next[word0 + mine] = 0ULL;                  // nothing to find here
            else nzr = nz_rank[mine];
        }
        unsigned long long todo = __ballot(cw != 0ULL);
        while (todo) {
            const int j = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const uint64_t cand_word = __shfl(cw, j);
            // the row's head record: rows with incoming edges are numbered consecutively, so the candidates of a group -- every other
            // row of an RMAT graph has none -- read consecutive 16-byte records
            const int32_t rec = __shfl(nzr, j) + (int32_t)__popcll(__shfl(nzw, j) & ((1ULL << lane) - 1ULL));
            const int64_t grp = g0 + (int64_t)j * NW;
            const int32_t r = (int32_t)(grp << 6) + lane;
            const int32_t v = row_base + r;
            bool found = false, defer = false;
            if (r < nrows && ((cand_word >> lane) & 1ULL)) {
                // The row's first eight in-neighbours come from two planes of 16-byte head records (coalesced over the wavefront, no
                // row offsets, no dependent adjacency load): most candidates find their parent among the first four, the rest look
                // at the next four; a row that still misses and is longer than eight goes to the wavefront pass.  (Requesting the next
                // group's records before waiting for this group's probes was tried: no change.)
                const int4 h = in_head[rec];
                const int32_t u0[4] = {h.x, h.y, h.z, h.w};
                uint32_t hit = 0;
                int n = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) if (u0[q] >= 0) n = q + 1;
                // the first in-neighbour alone first: when it is in the frontier (the common case once the frontier is large) the
                // other three frontier words are never requested.  (Requesting those three -- and the second plane's four -- together
                // and unconditionally, the change that took 10 % off the top-down expansion, does nothing for THIS kernel: 60.2 against
                // 59.2 us per launch, RMAT-27 533 against 517 -- its time does not follow a wavefront's dependent chain, see above.)
                if (u0[0] >= 0) hit = in_front(u0[0]);
                if (hit == 0) {
#pragma unroll
                    for (int q = 1; q < 4; q++)
                        if (u0[q] >= 0) hit |= in_front(u0[q]) << q;
                }
                bool longer = false;
                if (hit == 0 && n == 4) {
                    const int4 k = in_head[(int64_t)nz_rows + rec];
                    const int32_t u1[4] = {k.x, k.y, k.z, k.w};
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (u1[q] >= 0) { n = 5 + q; hit |= in_front(u1[q]) << (4 + q); }
                    longer = (in_long[v >> 6] >> (v & 63)) & 1ULL;
                }
                found = hit != 0;
                probes += found ? __ffs(hit) : n;        // adjacency entries a sequential scan would have examined
                defer = !found && longer;
                if (found) levels[v] = next_level;
            }
            const unsigned long long fm = __ballot(found);
            if (lane == 0) next[word0 + grp] = fm;
            found_cnt += found;
            const unsigned long long dm = __ballot(defer);
            if (dm) {                                   // wave-aggregated append to this workgroup's segment (LDS counter)
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_nheavy, (int)__popcll(dm));
                base = __shfl(base, 0);
                if (defer) my_heavy[base + __popcll(dm & ((1ULL << lane) - 1ULL))] = r;
            }
        }
    }
    if (INLINE_HEAVY) {
        __syncthreads();                                    // the workgroup's deferred list is complete (and its stores to `next` have landed)
        const int nh = s_nheavy;
        constexpr int G = VGL_BU_HEAVY_LANES, NG = 64 / G;  // lanes per deferred row, rows per wavefront
        const int quarter = lane / G, ql = lane % G;
        for (int h0 = vgl_wave() * NG; h0 < nh; h0 += VGL_WAVES * NG) {
            const int h = h0 + quarter;
            bool done = h >= nh, hit_any = false;
            int32_t r = 0;
            int64_t p = 0, e = 0;
            if (!done) { r = my_heavy[h]; p = in_rowptr[r]; e = in_rowptr[r + 1]; }     // the whole row: its head records are a selection, not a prefix
            while (!__all(done)) {
                const int64_t q = p + ql;
                bool hit = false;
                if (!done && q < e) { const int32_t u = in_adj[q]; hit = (front[u >> 6] >> (u & 63)) & 1ULL; }
                const unsigned long long hm = __ballot(hit);
                const unsigned qm = (unsigned)(hm >> (quarter * G)) & (G >= 32 ? 0xFFFFFFFFu : ((1u << (G & 31)) - 1u));
                if (!done) {
                    if (qm) { hit_any = true; done = true; if (ql == 0) probes += __ffs(qm); }
                    else { if (ql == 0) probes += min((int64_t)G, e - p); p += G; if (p >= e) done = true; }
                }
            }
            if (hit_any && ql == 0) {
                const int32_t v = row_base + r;
                levels[v] = next_level;
                atomicOr((unsigned long long *)&next[v >> 6], 1ULL << (v & 63));
                found_cnt++;
            }
        }
        found_cnt = vgl_block_reduce_add(found_cnt, s64);
        probes = vgl_block_reduce_add(probes, s64);
        uint32_t dep2 = 0;
        if (threadIdx.x == 0) dep2 = vgl_put_agent(partials + blockIdx.x * 4 + 0, found_cnt) ^ vgl_put_agent(partials + blockIdx.x * 4 + 1, probes);
        if (!vgl_last_block(ticket, dep2)) return;
        int64_t f = 0, pr = 0;
        for (int b = threadIdx.x; b < (int)gridDim.x; b += VGL_BLOCK) { f += vgl_load_agent(partials + b * 4 + 0); pr += vgl_load_agent(partials + b * 4 + 1); }
        f = vgl_block_reduce_add(f, s64);
        pr = vgl_block_reduce_add(pr, s64);
        if (threadIdx.x == 0) {
            if (host) vgl_publish2(counters, host, seq, C_BU_FOUND, f, C_BU_EDGES, pr);
            else { counters[C_BU_FOUND] = f; counters[C_BU_EDGES] = pr; }
        }
        return;
    }
    found_cnt = vgl_block_reduce_add(found_cnt, s64);
    probes = vgl_block_reduce_add(probes, s64);
    uint32_t dep = 0;
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 4 + 0] = found_cnt;
        partials[blockIdx.x * 4 + 1] = probes;
        dep = vgl_put_agent(heavy_cnt + blockIdx.x, (int32_t)s_nheavy);
    }
    // last workgroup: exclusive prefix of the per-workgroup deferred counts (VGL_BU_BLOCKS entries -> VGL_BU_BLOCKS+1 offsets)
    if (!vgl_last_block(ticket, dep)) return;
    __shared__ int s32[VGL_WAVES];
    constexpr int PER = VGL_BU_BLOCKS / VGL_BLOCK;
    int local[PER], sum = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) { local[j] = vgl_load_agent(heavy_cnt + threadIdx.x * PER + j); sum += local[j]; }
    int total;
    int pre = vgl_block_excl_add(sum, s32, &total);
#pragma unroll
    for (int j = 0; j < PER; j++) { heavy_off[threadIdx.x * PER + j] = pre; pre += local[j]; }
    if (threadIdx.x == 0) heavy_off[VGL_BU_BLOCKS] = total;
}

// pass 2: all deferred vertices, a quarter wavefront per vertex, 16 incoming neighbours per step, early exit on the first hit.
// The deferred lists are per-workgroup segments (pass 1); wavefronts stride over the CONCATENATION of the segments so the
// work is balanced even when the deferred vertices cluster (in a degree-sorted graph they are the first ids).
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_bu_heavy(int32_t row_base, int32_t chunk, const int64_t *in_rowptr,
                                                            const int32_t *in_adj, const uint64_t *front, uint64_t *next,
                                                            int32_t *levels, int32_t next_level, const int32_t *heavy,
                                                            const int32_t *heavy_off, int64_t *partials, int64_t *counters, uint32_t *ticket,
                                                            volatile int64_t *host, int64_t seq)
{
    __shared__ int64_t s64[VGL_WAVES];
    __shared__ int s_off[VGL_BU_BLOCKS + 1];
    // most levels defer a few hundred rows or none: only the workgroups that will scan rows stage the 8 KiB offset table (every one of
    // the 2048 doing it made an empty pass cost 11 us)
    const int total = heavy_off[VGL_BU_BLOCKS];
    if ((int64_t)blockIdx.x * VGL_WAVES * (64 / VGL_BU_HEAVY_LANES) < total) {
        for (int i = threadIdx.x; i <= VGL_BU_BLOCKS; i += VGL_BLOCK) s_off[i] = heavy_off[i];
    }
    __syncthreads();
    int64_t found_cnt = 0, probes = 0;
    // Four deferred vertices per wavefront, 16 lanes each: most deferred rows have a few dozen entries left, a whole wavefront per
    // row left three quarters of the lanes idle.  A quarter scans 16 entries per step and stops at its first hit; the wavefront
    // moves on when all four are done.
    constexpr int G = VGL_BU_HEAVY_LANES, NG = 64 / G;       // lanes per vertex, vertices per wavefront
    const int quarter = vgl_lane() / G, ql = vgl_lane() % G;
    for (int h0 = (blockIdx.x * VGL_WAVES + vgl_wave()) * NG; h0 < total; h0 += gridDim.x * VGL_WAVES * NG) {
        const int h = h0 + quarter;
        bool done = h >= total, hit_any = false;
        int32_t r = 0;
        int64_t p = 0, e = 0;
        if (!done) {
            int lo = 0, hi = VGL_BU_BLOCKS;                 // segment s with s_off[s] <= h < s_off[s+1]
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off[mid] <= h) lo = mid; else hi = mid; }
            r = heavy[(int64_t)lo * chunk + (h - s_off[lo])];
            p = in_rowptr[r]; e = in_rowptr[r + 1];             // the whole row: its head records are a selection (the smallest ids), not a prefix
        }
        while (!__all(done)) {
            const int64_t q = p + ql;
            bool hit = false;
            if (!done && q < e) { const int32_t u = in_adj[q]; hit = (front[u >> 6] >> (u & 63)) & 1ULL; }
            const unsigned long long hm = __ballot(hit);
            const unsigned qm = (unsigned)(hm >> (quarter * G)) & (G >= 32 ? 0xFFFFFFFFu : ((1u << (G & 31)) - 1u));
            if (!done) {
                if (qm) { hit_any = true; done = true; if (ql == 0) probes += __ffs(qm); }
                else { if (ql == 0) probes += min((int64_t)G, e - p); p += G; if (p >= e) done = true; }
            }
        }
        if (hit_any && ql == 0) {
            const int32_t v = row_base + r;
            levels[v] = next_level;
            atomicOr((unsigned long long *)&next[v >> 6], 1ULL << (v & 63));
            found_cnt++;
        }
    }
    found_cnt = vgl_block_reduce_add(found_cnt, s64);
    probes = vgl_block_reduce_add(probes, s64);
    uint32_t dep = 0;
    if (threadIdx.x == 0) dep = vgl_put_agent(partials + blockIdx.x * 4 + 2, found_cnt) ^ vgl_put_agent(partials + blockIdx.x * 4 + 3, probes);
    // last workgroup: counters[C_BU_FOUND] / [C_BU_EDGES] = sums of the per-workgroup partials of both passes in a fixed order,
    // handed to the host (when it listens: host != nullptr)
    if (!vgl_last_block(ticket, dep)) return;
    int64_t f = 0, p = 0;
    for (int b = threadIdx.x; b < VGL_BU_BLOCKS; b += VGL_BLOCK) {      // (this pass may run with fewer workgroups than the probe pass)
        f += partials[b * 4 + 0];
        p += partials[b * 4 + 1];
        if (b < (int)gridDim.x) { f += vgl_load_agent(partials + b * 4 + 2); p += vgl_load_agent(partials + b * 4 + 3); }
    }
    f = vgl_block_reduce_add(f, s64);
    p = vgl_block_reduce_add(p, s64);
    if (threadIdx.x == 0) {
        if (host) vgl_publish2(counters, host, seq, C_BU_FOUND, f, C_BU_EDGES, p);
        else { counters[C_BU_FOUND] = f; counters[C_BU_EDGES] = p; }
    }
}

// visited |= next; front = next; next = 0 (ready for the next emitting step without a memset)   (one word per thread)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_bm_advance(int64_t words, uint64_t *visited, uint64_t *front, uint64_t *next)
{
    for (int64_t w = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; w < words; w += (int64_t)gridDim.x * VGL_BLOCK) {
        const uint64_t n = next[w];
        if (n) { visited[w] |= n; next[w] = 0; }
        front[w] = n;
    }
}

// ---- sparse exchange of tiny levels (multi-GPU): ids instead of V-bit bitmaps ----
// out[0] = number of set bits (may exceed cap), out[1 .. 1+cap) = the ids of the first `cap` positions handed out (id = 64 * (word_base + word) + bit).
// A workgroup owns a contiguous run of words: it counts its bits, reserves its range of the list with ONE atomic and writes in word
// order (one atomic per non-empty word on a single counter, the first version, serialises at ~12 ns each: fine for a few thousand ids,
// half a millisecond for the 10^5 of a late bottom-up level).
constexpr int VGL_B2I_BLOCKS = 2048;
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_bitmap_to_ids(int64_t words, const uint64_t *bits, int64_t word_base, int32_t cap, int32_t *out)
{
    __shared__ int s32[VGL_WAVES];
    __shared__ int s_base;
    const int64_t per = (((words + gridDim.x - 1) / gridDim.x + VGL_BLOCK - 1) / VGL_BLOCK) * VGL_BLOCK;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = min(words, lo + per);
    int n = 0;
    for (int64_t wi = lo + threadIdx.x; wi < hi; wi += VGL_BLOCK) n += (int)__popcll(bits[wi]);
    int total = 0;
    vgl_block_excl_add(n, s32, &total);
    if (total == 0) return;                                     // (the same in every thread of the workgroup)
    if (threadIdx.x == 0) s_base = atomicAdd(out, total);
    __syncthreads();
    int run = s_base;
    for (int64_t w0 = lo; w0 < hi && run < cap; w0 += VGL_BLOCK) {      // (run is uniform over the workgroup)
        const int64_t wi = w0 + threadIdx.x;
        uint64_t w = wi < hi ? bits[wi] : 0ULL;
        int step = 0;
        int pos = run + vgl_block_excl_add((int)__popcll(w), s32, &step);
        while (w) {
            const int b = __ffsll((long long)w) - 1;
            w &= w - 1;
            if (pos < cap) out[1 + pos] = (int32_t)(((word_base + wi) << 6) + b);
            pos++;
        }
        run += step;
    }
}
// `parts` lists of that layout (stride 1 + cap): a vertex reported by several ranks is taken once (claimed on the visited bitmap)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_apply_ids(int32_t V, int parts, int32_t cap, const int32_t *lists, int32_t *levels, int32_t level,
                                                             uint64_t *visited, uint64_t *front, const int32_t *degrees, int64_t *partials)
{
    __shared__ int64_t s64[VGL_WAVES];
    int64_t cnt = 0, deg = 0;
    const int64_t total = (int64_t)parts * cap;
    for (int64_t t = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; t < total; t += (int64_t)gridDim.x * VGL_BLOCK) {
        const int p = (int)(t / cap), i = (int)(t % cap);
        const int32_t *list = lists + (int64_t)p * (1 + cap);
        if (i >= list[0]) continue;
        const int32_t v = list[1 + i];
        if (v < 0 || v >= V) continue;
        const unsigned long long bit = 1ULL << (v & 63);
        const unsigned long long old = atomicOr((unsigned long long *)&visited[v >> 6], bit);
        if (old & bit) continue;
        levels[v] = level;
        atomicOr((unsigned long long *)&front[v >> 6], bit);
        cnt++;
        if (degrees) deg += degrees[v];
    }
    cnt = vgl_block_reduce_add(cnt, s64);
    deg = vgl_block_reduce_add(deg, s64);
    if (threadIdx.x == 0) { partials[blockIdx.x * 2] = cnt; partials[blockIdx.x * 2 + 1] = deg; }
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

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_bitmap_or_parts(int64_t words, int parts, const uint64_t *in, uint64_t *out)
{
    for (int64_t w = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; w < words; w += (int64_t)gridDim.x * VGL_BLOCK) {
        uint64_t acc = 0;
        for (int p = 0; p < parts; p++) acc |= in[(int64_t)p * words + w];
        out[w] = acc;
    }
}

// OR the per-rank discovery bitmaps, mark the newly discovered vertices in levels, and (optionally) keep the replicated
// visited / frontier bitmaps and the frontier's out-degree sum up to date for the direction decision of the next level
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_apply_bitmaps(int32_t V, int parts, int64_t words, const uint64_t *bits_all,
                                                                 int32_t *levels, int32_t level, uint64_t *visited, uint64_t *front,
                                                                 const int32_t *degrees, int64_t *partials, int32_t own_begin, int32_t own_end)
{
    // own_begin .. own_end (visited bitmap given): the bitmaps are updated for ALL vertices -- every rank probes any vertex's bits --
    // but levels / counts / degree sums only for the owned ones: that per-vertex part is the bulk of the pass and, done for all
    // vertices on every rank, it was the serial fraction of the sharded traversal (0.35 - 0.4 ms per large level at scale 27 whatever
    // the number of ranks).  The caller adds the counts over the ranks.
    // one bitmap word per lane (coalesced OR over the parts; almost all words are zero on small levels), then the wavefront
    // walks its non-zero words together so that the levels / degrees accesses of a word are one coalesced 256-byte row
    __shared__ int64_t s64[VGL_WAVES];
    int64_t cnt = 0, deg = 0;
    const int lane = vgl_lane();
    const int64_t wround = (words + 63) & ~(int64_t)63;
    for (int64_t wi = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; wi < wround; wi += (int64_t)gridDim.x * VGL_BLOCK) {
        uint64_t w = 0;
        if (wi < words) {
            for (int p = 0; p < parts; p++) w |= bits_all[(int64_t)p * words + wi];
            if (visited) {                       // replicated visited bitmap: new = reported and not yet visited
                const uint64_t vis = visited[wi];
                w &= ~vis;
                if (w) visited[wi] = vis | w;
            }
        }
        // (words wholly outside the owned range need no per-vertex walk: own ranges are multiples of 64)
        const bool walk = w != 0 && (!visited || (((wi << 6) + 63) >= own_begin && (wi << 6) < own_end));
        unsigned long long todo = __ballot(walk);
        uint64_t mine_new = visited ? w : 0;
        while (todo) {
            const int l = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const uint64_t ww = __shfl(w, l);
            const int64_t v = ((wi - lane + l) << 6) + lane;
            bool is_new = false;
            if (v < V && ((ww >> lane) & 1ULL) && (!visited || (v >= own_begin && v < own_end))) {
                if (visited) { levels[v] = level; is_new = true; }
                else {                            // no bitmap: levels decides (vertices of this level were marked by their finder)
                    const int32_t lv = levels[v];
                    if (lv == -1) levels[v] = level;
                    is_new = (lv == -1) || (lv == level);
                }
                if (is_new && degrees) deg += degrees[v];
                if (is_new && visited) cnt++;
            }
            if (!visited) { const unsigned long long nm = __ballot(is_new); if (lane == l) mine_new = nm; }
        }
        if (wi < words) {
            if (front) front[wi] = mine_new;
            if (!visited) cnt += __popcll(mine_new);
        }
    }
    cnt = vgl_block_reduce_add(cnt, s64);
    deg = vgl_block_reduce_add(deg, s64);
    if (threadIdx.x == 0) { partials[blockIdx.x * 2] = cnt; partials[blockIdx.x * 2 + 1] = deg; }
}
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_apply_fold(int n, const int64_t *partials, int64_t *counters)
{
    __shared__ int64_t s64[VGL_WAVES];
    int64_t a = 0, b = 0;
    for (int i = threadIdx.x; i < n; i += VGL_BLOCK) { a += partials[i * 2]; b += partials[i * 2 + 1]; }
    a = vgl_block_reduce_add(a, s64);
    b = vgl_block_reduce_add(b, s64);
    if (threadIdx.x == 0) { counters[C_TMP0] = a; counters[C_TMP1] = b; }
}

static inline unsigned vgl_grid(int64_t n, int64_t cap = 8192) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cap, vgl_ceil_div(n, VGL_BLOCK))); }

// expand frontier (ids/offs with F vertices, M edges already produced by a frontier-generation write pass)
constexpr int64_t VGL_TD_COUNT_TILES = 8192;           // (the counting level's partial sums live in g->bu_partials: 4 * 4096 slots)
static int vgl_bfs_td_launch(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t F, int64_t M, int32_t *levels, int32_t next_level, bool emit,
                             bool have_tile_first, bool count = false, bool filter = true)
{
    if (F <= 0 || M <= 0) return 0;
    if (!have_tile_first) hipLaunchKernelGGL(vgl_k_tile_first, dim3(vgl_grid(F)), dim3(VGL_BLOCK), 0, c->stream, F, g->offs, g->tile_first);
    const int64_t nt = vgl_ceil_div(M, VGL_TILE);
    {
        vgl_timed_launch tl(c, "bfs_top_down");
        int64_t *no_i64 = nullptr;
        uint32_t *no_u32 = nullptr;
#define VGL_TD_LAUNCH(E, C, F, PARTIALS, TICKET, COUNTERS)                                                                                          \
        hipLaunchKernelGGL((vgl_k_td_expand<E, C, F>), dim3((unsigned)nt), dim3(VGL_BLOCK), 0, c->stream, g->ids, g->offs, g->tile_first, F_, M, \
                           g->out.rowptr, g->out.adj, g->row_begin, g->bm_visited, levels, next_level, g->bm_next, PARTIALS, TICKET, COUNTERS)
        const int32_t F_ = F;
        if (emit && count) {
            if (filter) VGL_TD_LAUNCH(true, true, true, g->bu_partials, g->tickets + 3 * VGL_TICKET_WORDS, c->d_counters);
            else VGL_TD_LAUNCH(true, true, false, g->bu_partials, g->tickets + 3 * VGL_TICKET_WORDS, c->d_counters);
        } else if (emit) {
            if (filter) VGL_TD_LAUNCH(true, false, true, no_i64, no_u32, no_i64);
            else VGL_TD_LAUNCH(true, false, false, no_i64, no_u32, no_i64);
        } else {
            if (filter) VGL_TD_LAUNCH(false, false, true, no_i64, no_u32, no_i64);
            else VGL_TD_LAUNCH(false, false, false, no_i64, no_u32, no_i64);
        }
#undef VGL_TD_LAUNCH
    }
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

// frontier of the current level from a frontier bitmap (owned words).  count: per-workgroup counts, their scan and F / M in
// h_counters[C_FRONT] / [C_NEIGH] (one launch, the host waits for it); write: ids + edge offsets + tile_first (needs the M of
// the count pass)
int vgl_bfs_bm_gnf(vgl_hip_ctx *c, vgl_hip_graph *g, const uint64_t *front, bool count, bool write, int64_t M_known, bool advance, const vgl_do_hint *hint)
{
    const vgl_do_hint no_hint = {0, 0, 0, 0};
    const int64_t word0 = g->row_begin >> 6;
    const int64_t nwords = vgl_ceil_div(g->row_end, 64) - word0;
    const unsigned nb = (unsigned)vgl_ceil_div(nwords, VGL_BLOCK);
    if (count) {
        const int64_t seq = vgl_next_seq(c);
        {
            vgl_timed_launch tl(c, "gnf");
            if (advance)                 // front == g->bm_front: rebuilt from g->bm_next first (whole-graph handles only)
                hipLaunchKernelGGL(vgl_k_bm_gnf_count<true>, dim3(nb), dim3(VGL_BLOCK), 0, c->stream, nwords, word0, g->row_begin, g->bm_front,
                                   g->out.rowptr, g->vt_cnt, g->vt_deg, g->vt_cnt_off, g->vt_deg_off, g->offs, c->d_counters,
                                   g->tickets + 0 * VGL_TICKET_WORDS, (volatile int64_t *)c->h_counters, seq, g->bm_visited, g->bm_next,
                                   hint ? 1 : 0, hint ? *hint : no_hint);
            else
                hipLaunchKernelGGL(vgl_k_bm_gnf_count<false>, dim3(nb), dim3(VGL_BLOCK), 0, c->stream, nwords, word0, g->row_begin,
                                   const_cast<uint64_t *>(front), g->out.rowptr, g->vt_cnt, g->vt_deg, g->vt_cnt_off, g->vt_deg_off, g->offs, c->d_counters,
                                   g->tickets + 0 * VGL_TICKET_WORDS, (volatile int64_t *)c->h_counters, seq, (uint64_t *)nullptr, (uint64_t *)nullptr,
                                   0, no_hint);
        }
        VGL_HIP_TRY(hipGetLastError());
        VGL_TRY(vgl_wait_counters(c, seq));
    }
    if (write) {
        vgl_timed_launch tl(c, "gnf");
        hipLaunchKernelGGL(vgl_k_bm_gnf_write, dim3(nb), dim3(VGL_BLOCK), 0, c->stream, nwords, word0, g->row_begin, front,
                           g->out.rowptr, g->vt_cnt_off, g->vt_deg_off, g->ids, g->offs, g->tile_first, M_known >= 0 ? M_known : c->h_counters[C_NEIGH]);
        VGL_HIP_TRY(hipGetLastError());
    }
    return 0;
}

// one bottom-up step over the owned rows: probe (+ deferred-list offsets) and the balanced heavy pass (+ fold of the counters
// C_BU_FOUND / C_BU_EDGES, published to the host under sequence number *seq_out: vgl_wait_counters when they are needed)
// heavy_blocks: workgroups of the deferred pass.  A bottom-up phase defers rows on its first level (when the frontier is still small
// against the unvisited rows); on the levels after it the pass finds nothing to do in almost every launch and costs what its launch
// costs -- 9 us with 2048 workgroups, so those levels run it with a small grid (correct for any number of deferred rows, only slower).
static int vgl_bfs_bu_launch(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t *levels, int32_t next_level, const uint64_t *visited,
                             const uint64_t *front, uint64_t *next, int64_t *seq_out, int heavy_blocks = VGL_BU_BLOCKS)
{
    const int32_t chunk = (int32_t)(vgl_ceil_div(vgl_ceil_div(g->nrows, VGL_BU_BLOCKS), VGL_BLOCK) * VGL_BLOCK);
    const bool split = c->bfs.bu_split;
    const int64_t seq = vgl_next_seq(c);
    if (!split) {
        vgl_timed_launch tl(c, "bfs_bottom_up");
        hipLaunchKernelGGL(vgl_k_bu_probe<true>, dim3(VGL_BU_BLOCKS), dim3(VGL_BLOCK), 0, c->stream, g->nrows, g->row_begin, chunk,
                           g->in.rowptr, g->in.adj, g->in.edges, visited, g->bm_in_nz, front, next, levels, next_level,
                           g->heavy, g->heavy_cnt, g->bu_partials, g->heavy_off, g->tickets + 1 * VGL_TICKET_WORDS, reinterpret_cast<const int4 *>(g->in_head), g->bm_in_long,
                           c->d_counters, (volatile int64_t *)c->h_counters, seq, (const int32_t *)g->in_nz_rank, g->in_nz_rows);
        VGL_HIP_TRY(hipGetLastError());
        if (seq_out) *seq_out = seq;
        return 0;
    }
    {
        vgl_timed_launch tl(c, "bfs_bottom_up");
        hipLaunchKernelGGL(vgl_k_bu_probe<false>, dim3(VGL_BU_BLOCKS), dim3(VGL_BLOCK), 0, c->stream, g->nrows, g->row_begin, chunk,
                           g->in.rowptr, g->in.adj, g->in.edges, visited, g->bm_in_nz, front, next, levels, next_level,
                           g->heavy, g->heavy_cnt, g->bu_partials, g->heavy_off, g->tickets + 1 * VGL_TICKET_WORDS, reinterpret_cast<const int4 *>(g->in_head), g->bm_in_long,
                           c->d_counters, (volatile int64_t *)c->h_counters, seq, (const int32_t *)g->in_nz_rank, g->in_nz_rows);
    }
    {
        vgl_timed_launch tl(c, "bfs_bottom_up_heavy");
        hipLaunchKernelGGL(vgl_k_bu_heavy, dim3((unsigned)heavy_blocks), dim3(VGL_BLOCK), 0, c->stream, g->row_begin, chunk, g->in.rowptr,
                           g->in.adj, front, next, levels, next_level, g->heavy, g->heavy_off, g->bu_partials, c->d_counters, g->tickets + 2 * VGL_TICKET_WORDS,
                           (volatile int64_t *)c->h_counters, seq);
    }
    VGL_HIP_TRY(hipGetLastError());
    if (seq_out) *seq_out = seq;
    return 0;
}

// one top-down level from g->bm_front / g->bm_visited (both current) into g->bm_next (all zero before) and levels
static int vgl_bfs_blocked_level(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t *levels, int32_t next_level)
{
    const vgl_blocked_plan *p = g->blk_bfs;
    if (p->n_g_units > 0) {
        vgl_timed_launch tl(c, "bfs_blk_gather");
        hipLaunchKernelGGL(vgl_k_bfs_blk_gather, dim3((unsigned)p->n_g_units), dim3(VGL_BTHREADS), 0, c->stream, (const vgl_blk_unit *)p->g_units,
                           (const uint16_t *)p->g_lo, (const uint32_t *)p->mid_to_a, reinterpret_cast<uint64_t *>(p->vals), p->g_count,
                           (const uint64_t *)g->bm_front, (int64_t)g->row_begin >> 6);
    }
    if (p->n_a_units > 0) {
        vgl_timed_launch tl(c, "bfs_blk_accumulate");
        hipLaunchKernelGGL(vgl_k_bfs_blk_accumulate, dim3((unsigned)p->n_a_units), dim3(VGL_BTHREADS), 0, c->stream, (const vgl_blk_unit *)p->a_units,
                           (const uint16_t *)p->a_lo, reinterpret_cast<const uint64_t *>(p->vals), p->a_count, (const uint64_t *)g->bm_visited,
                           g->bm_next, levels, next_level);
    }
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

// a bitmap cleared with ONE launch (hipMemsetAsync of these 2 - 16 MiB buffers shows as three fill kernels of ~5 us each in the trace of the
// sharded traversal: 15 us per level for a 3 us job)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_zero_words(int64_t words, uint64_t *p)
{
    for (int64_t i = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; i < words; i += (int64_t)gridDim.x * VGL_BLOCK) p[i] = 0ULL;
}
int vgl_zero_words(vgl_hip_ctx *c, uint64_t *d_words, int64_t words)
{
    if (words > 0)
        hipLaunchKernelGGL(vgl_k_zero_words, dim3((unsigned)std::min<int64_t>(4096, vgl_ceil_div(words, VGL_BLOCK))), dim3(VGL_BLOCK), 0, c->stream, words, d_words);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_bitmap_to_ids(vgl_hip_ctx *c, int64_t words, const uint64_t *d_bits, int64_t word_base, int32_t cap, int32_t *d_out)
{
    VGL_HIP_TRY(hipMemsetAsync(d_out, 0, sizeof(int32_t), c->stream));
    if (words > 0) {
        vgl_timed_launch tl(c, "bfs_shard_resolve");             // (only the sharded traversal uses it: counted with its owner-side passes)
        hipLaunchKernelGGL(vgl_k_bitmap_to_ids, dim3((unsigned)std::min<int64_t>(VGL_B2I_BLOCKS, vgl_ceil_div(words, VGL_BLOCK))), dim3(VGL_BLOCK), 0, c->stream,
                           words, d_bits, word_base, cap, d_out);
    }
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" {

// Graph preparation for repeated top-down traversals (like the reference's offline import, vgl_graph.hpp:57-68): lays the outgoing
// edges out for the blocked level above (4 B per edge kept; a radix sort of the edges, ~40 ms for RMAT-24).  vgl_hip_bfs_run then
// takes the blocked pass for the levels that hold at least VGL_BFS_BLOCKED_SHARE (0.1) of the edges; levels are the same.
int vgl_hip_bfs_prepare_blocked(vgl_hip_ctx *c, vgl_hip_graph *g)
{
    if (!c || !g) VGL_FAIL("bfs_prepare_blocked: null argument");
    if (g->row_begin != 0 || g->row_end != g->V) VGL_FAIL("bfs_prepare_blocked: graph handle must own all rows");
    if (g->blk_bfs) return 0;
    static_assert(VGL_BTHREADS == VGL_BLK / 32, "one 32-bit frontier word per thread");
    return vgl_blocked_plan_build(c, g->out, g->nrows, g->row_begin, g->V, 1, 1, nullptr, VGL_BLK_BITS, &g->blk_bfs, 1);
}

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
    if (source < 0 || source >= V) VGL_FAIL("bfs_run: source vertex out of range");
    vgl_ctx_refresh_env(c);                                  // (a pass over the pointers of `environ`; the strings are parsed only when something was set since)
    const vgl_bfs_tunables &tn = c->bfs;
    hipLaunchKernelGGL(vgl_k_bfs_init_all, dim3(vgl_grid(V)), dim3(VGL_BLOCK), 0, c->stream, V, source, d_levels, words, g->bm_visited, g->bm_front,
                       g->bm_next, g->tickets, reinterpret_cast<unsigned long long *>(c->d_counters + C_HEAVY));

    vgl_hip_bfs_stats st = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int32_t cur = 1;
    bool bottom_up = false;          // state used to PROCESS level `cur`
    // Frontier of level `cur`: bm_front when front_valid (bm_visited is then current too); otherwise only `levels` knows it and
    // the levels-scanning GNF rebuilds both bitmaps.  Frontiers are kept as bitmaps whenever the level that produced them was
    // small (bottom-up steps always; top-down steps with at most VGL_TD_EMIT_EDGES edges, which OR their discoveries in).
    bool front_valid = true;
    bool counted = false;            // vt_cnt_off / vt_deg_off describe the frontier (needed by the write pass)
    bool counted_from_bitmap = false;
    bool advance_pending = false;    // bm_next holds the discoveries of the last (top-down, emitting) level: the count launch applies them
    int64_t F = 0, M = 0, prevF = 0, visited_total = 0;
    const int64_t factor = std::max<int64_t>(1, (E / V) / 2);     // change_state.hpp:104
    // An emitting level pays one device-scope atomicOr per discovery, a plain one a scan of `levels` (V * 4 bytes) by the next frontier
    // generation: the bitmap pays only while the level is small against V.  RMAT-24 (V = 16.8 M), direction-optimising traversal: bound
    // 16 M edges 0.392 ms, 4 M 0.377, 2 M 0.381, 1 M 0.366, 512 K 0.361, 256 K 0.364.
    // Round 4: 64 K edges on RMAT-24 (V / 256).  The emitting form is dearer than round 2 priced it -- its atomics go to the memory side at
    // ~2.6e10 /s when they scatter and SERIALISE at ~12 ns each when they meet (profiles/r04_atomic_scope_bench.log; every edge into a popular
    // vertex that arrives before the first store is visible issues its own: a 500 K-edge level took 170 us) -- and the scan it avoids is
    // cheaper (vgl_k_bfs_scan_bound: 13 us): V / 24 0.317 ms per traversal, V / 64 0.305, V / 256 0.3047, V / 1024 0.306, never 0.317.
    int64_t VGL_TD_EMIT_EDGES = std::max<int64_t>(65536, (int64_t)V / 256);
    if (tn.td_emit_edges >= 0) VGL_TD_EMIT_EDGES = tn.td_emit_edges;
    double td_filter_share = 0.125;
    if (tn.td_filter_share >= 0.0) td_filter_share = tn.td_filter_share;
    // Late levels: once all but a few of the vertices that CAN be discovered (rows with incoming edges; known when the incoming CSR is stored)
    // are visited, an emitting level is cheap whatever its edge count -- its visited-bitmap probe turns almost every edge away before the
    // atomic -- and it leaves the next frontier as a bitmap (2 MiB to count) instead of only in `levels` (64 MiB to scan).  VGL_TD_LATE_SHARE:
    // "few" as a share of V (0 = rule off).
    double td_late_share = 1.0 / 64;
    if (tn.td_late_share >= 0.0) td_late_share = tn.td_late_share;
    auto late = [&]() -> bool {          // evaluated when visited_total already holds the frontier about to be expanded
        if (g->in_nz_rows <= 0 || td_late_share <= 0.0) return false;
        const int64_t remain = std::max<int64_t>(0, (int64_t)g->in_nz_rows + 1 - visited_total);      // (+1: the source may have no incoming edge)
        return (double)remain <= td_late_share * (double)V;
    };
    int bu_in_a_row = 0;                                 // bottom-up levels since the last top-down one
    int later_heavy_blocks = 256;                        // RMAT-24 traversal: 0.376 ms with 2048, 0.368-0.370 with 512 / 256 / 128
    if (tn.later_heavy_blocks >= 0) later_heavy_blocks = std::max(1, std::min(VGL_BU_BLOCKS, tn.later_heavy_blocks));
    double blocked_share = 0.1;                          // top-down levels with at least this share of the edges take the blocked pass (when prepared;
                                                         // RMAT-24 top-down traversal: 1.83 ms at 0.2, 1.67 at 0.1, 1.66 at 0.05, 1.69 at 0.02)
    if (tn.blocked_share >= 0.0) blocked_share = tn.blocked_share;
    // hint_ready: the launch that produced the frontier about to be counted left its F and M on the device (a counting top-down level, or
    // the list kernel leaving with a frontier too long for its list): the count launch may then skip everything but the bitmaps when the
    // rule turns the level bottom-up (`skipped`; the rule below must -- and does, same integers -- come to the same conclusion)
    bool hint_ready = false, skipped = false;
    const bool use_hints = mode == VGL_HIP_BFS_DIRECTION_OPT && !tn.no_hint;
    const bool scan_bound = mode == VGL_HIP_BFS_DIRECTION_OPT && !tn.no_scan_bound;
    auto count_frontier = [&]() -> int {
        skipped = false;
        if (front_valid) {
            const vgl_do_hint hint = {prevF, visited_total, (int64_t)V, factor};
            const bool hinted = hint_ready && advance_pending && use_hints;
            VGL_TRY(vgl_bfs_bm_gnf(c, g, g->bm_front, true, false, -1, advance_pending, hinted ? &hint : nullptr));
            skipped = hinted && c->h_counters[C_SKIPPED] != 0;
            counted_from_bitmap = true; advance_pending = false;
        }
        else if (scan_bound && g->nvtiles <= 8192) {           // (per-tile partials live in g->bu_partials: 4 * 4096 slots)
            const int64_t seq = vgl_next_seq(c);
            {
                vgl_timed_launch tl(c, "gnf");
                hipLaunchKernelGGL(vgl_k_bfs_scan_bound, dim3((unsigned)g->nvtiles), dim3(VGL_BLOCK), 0, c->stream, g->nrows, g->row_begin, (const int32_t *)d_levels,
                                   cur, (const int32_t *)g->vt_min_deg, (uint8_t *)g->bm_front, (uint8_t *)g->bm_visited, g->bu_partials,
                                   g->tickets + 0 * VGL_TICKET_WORDS, c->d_counters, (volatile int64_t *)c->h_counters, seq);
            }
            VGL_HIP_TRY(hipGetLastError());
            VGL_TRY(vgl_wait_counters(c, seq));
            front_valid = true; advance_pending = false;
            const int64_t f = c->h_counters[C_FRONT], m_lb = c->h_counters[C_NEIGH];
            if (f > prevF && m_lb >= ((V - (visited_total + f)) * factor + V) / VGL_DO_ALPHA) {
                // bottom-up whatever the exact edge count is (the rule below sees M = the bound and comes to the same conclusion)
                F = f; M = m_lb; counted = false; skipped = true; hint_ready = false;
                return 0;
            }
            VGL_TRY(vgl_bfs_bm_gnf(c, g, g->bm_front, true, false, -1, false, nullptr));      // exact sizes off the bitmap
            counted_from_bitmap = true;
        }
        else {
            vgl_pred_equal_i32 pred{d_levels, cur};
            VGL_TRY(vgl_gnf_run(c, g, pred, g->ids, g->offs, (uint8_t *)g->bm_front, (uint8_t *)g->bm_visited, nullptr, false, true));
            front_valid = true; counted_from_bitmap = false;
        }
        F = c->h_counters[C_FRONT]; M = c->h_counters[C_NEIGH];
        counted = !skipped;
        hint_ready = false;
        return 0;
    };
    // small frontiers (the first and the last levels): several levels in one single-workgroup launch, vgl_k_bfs_small_levels.  The edge
    // bound also keeps the direction rule silent inside the kernel: it needs M >= ((V - visited) * factor + V) / ALPHA >= V / ALPHA.
    int64_t small_m = 8192;                              // 4096 .. 16384 measure the same on RMAT-24 (0.628 ms against 0.711 without, 0.646 at 65536)
    if (tn.small_m >= 0) small_m = std::min<int64_t>(tn.small_m, 1 << 20);     // (the kernel scans 32-bit degree sums)
    if (mode == VGL_HIP_BFS_DIRECTION_OPT) small_m = std::min<int64_t>(small_m, (int64_t)V / VGL_DO_ALPHA - 1);
    if (small_m < 64) small_m = 0;                       // not worth a launch of its own
    // VGL_BFS_TRACE=1: one line per dispatch (level, frontier, edges, path taken, milliseconds since the start; every line synchronises)
    const bool trace_on = tn.trace;
    const auto trace_t0 = std::chrono::steady_clock::now();
    auto trace = [&](const char *what) {
        if (!trace_on) return;
        hipStreamSynchronize(c->stream);
        fprintf(stderr, "[bfs trace] %7.3f ms  level %d  F %lld  M %lld  visited %lld  %s\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - trace_t0).count(), (int)cur, (long long)F, (long long)M, (long long)visited_total, what);
    };
    bool finished = false;
    bool precounted = false;         // F, M, ids, offs, tile_first of level cur were left by vgl_k_bfs_small_levels: no count, no write pass
    // runs the kernel on the list g->ids[0..F) of level `cur` (or on {source}); returns through C_* what it did
    unsigned long long *list_count = reinterpret_cast<unsigned long long *>(c->d_counters + C_HEAVY);      // slot unused by the fused traversal
    const unsigned bm_blocks = (unsigned)vgl_ceil_div(words, VGL_BLOCK);
    int64_t bm_expand_f = 262144;                        // bottom-up -> top-down switch: frontiers up to this size are expanded from the bitmap
    if (tn.bm_expand >= 0) bm_expand_f = tn.bm_expand;
    if (bm_blocks > 8192) bm_expand_f = 0;               // (edge partials live in g->bu_partials: 4 * VGL_BU_BLOCKS slots)
    // from_bitmap: level `cur` is expanded from bm_front by vgl_k_bm_expand first, the list kernel starts at level cur + 1
    auto small_levels = [&](int32_t listF, int32_t src, bool from_bitmap = false) -> int {
        if (from_bitmap) {
            vgl_timed_launch tl(c, "bfs_bitmap_expand");
            hipLaunchKernelGGL(vgl_k_bm_expand, dim3(bm_blocks), dim3(VGL_BLOCK), 0, c->stream, words, g->bm_front, g->out.rowptr, g->out.adj, g->bm_visited,
                               g->bm_next, d_levels, cur + 1, g->ids, list_count, (int32_t)VGL_SMALL_F, g->bu_partials);
        }
        const int64_t seq = vgl_next_seq(c);
        {
            vgl_timed_launch tl(c, "bfs_small_levels");
            hipLaunchKernelGGL(vgl_k_bfs_small_levels, dim3(1), dim3(VGL_SMALL_THREADS), 0, c->stream, g->ids, listF, src, g->out.rowptr, g->out.adj,
                               g->bm_visited, g->bm_next, d_levels, from_bitmap ? cur + 1 : cur, small_m, c->d_counters, (volatile int64_t *)c->h_counters,
                               seq, from_bitmap ? list_count : (unsigned long long *)nullptr, (const int64_t *)g->bu_partials, from_bitmap ? (int)bm_blocks : 0,
                               g->bm_front, g->offs, g->tile_first);
        }
        VGL_HIP_TRY(hipGetLastError());
        VGL_TRY(vgl_wait_counters(c, seq));
        return 0;
    };
    // books the levels the kernel ran beyond the first one (the caller has booked that one) and moves `cur` past all of them
    auto account_small = [&](int64_t firstF) {
        const int64_t run = c->h_counters[C_TMP0], later = c->h_counters[C_TMP1];
        st.td_steps += (int32_t)run; st.edges_examined += c->h_counters[C_EDGES]; st.td_edges += c->h_counters[C_EDGES];
        st.td_frontier += firstF + later;
        visited_total += later; st.levels += (int32_t)(run - 1); st.frontier_total += later;
        if (run > 1) prevF = c->h_counters[C_JUMP];
        cur += (int32_t)run;
        if (c->h_counters[C_FRONT] == 0) finished = true;                      // the last level run discovered nothing
        else if (c->h_counters[C_HEAVY] != 0) {                                // handed over: ids / offs / tile_first / bm_front describe level cur
            F = c->h_counters[C_FRONT]; M = c->h_counters[C_BU_EDGES];
            precounted = true; front_valid = true; advance_pending = false;
        } else { advance_pending = true; front_valid = true; hint_ready = c->h_counters[C_HINT] != 0; }      // its discoveries wait in bm_next
    };
    if (small_m > 0) {                                                         // level 1 = {source}
        VGL_TRY(small_levels(1, source));
        if (c->h_counters[C_TMP0] > 0) {
            visited_total += 1; st.levels++; st.frontier_total += 1; prevF = 1;
            account_small(1);
        }
    }
    for (;;) {
        if (finished) break;
        counted = false;
        const bool ids_ready = precounted;
        if (precounted) { counted = true; counted_from_bitmap = true; precounted = false; }
        else if (!bottom_up) { trace("-> count"); VGL_TRY(count_frontier()); trace(counted_from_bitmap ? "count done (bitmap)" : "count done (levels scan)"); }      // after a bottom-up step F is already known (M is not needed to stay bottom-up)
        if (F == 0) break;
        visited_total += F;
        st.levels++; st.frontier_total += F;
        // direction for this level (gpu_change_state, change_state.hpp:100-141, evaluated with the frontier about to be expanded)
        if (mode == VGL_HIP_BFS_DIRECTION_OPT) {
            if (!bottom_up) {
                if (F > prevF && M >= ((V - visited_total) * factor + V) / VGL_DO_ALPHA) { bottom_up = true; bu_in_a_row = 0; }
            } else if (F <= prevF && F < ((V - visited_total) * factor + V) / (factor * VGL_DO_BETA)) {      // "shrinking phase" = not growing (change_state.hpp:106,121)
                bottom_up = false;
                if (small_m > 0 && F <= bm_expand_f) {
                    // a modest frontier of short rows: expand it from the bitmap and let the list kernel run whatever follows -- no
                    // count / host / write rounds.  Level cur is booked here, the list kernel's levels by account_small.
                    VGL_TRY(small_levels(0, -1, true));
                    const int64_t m_level = c->h_counters[C_CHANGED], n_next = c->h_counters[C_BU_FOUND];
                    st.td_steps++; st.edges_examined += m_level; st.td_edges += m_level; st.td_frontier += F;
                    prevF = F;
                    cur++;
                    if (n_next == 0) break;                                     // nothing discovered: the traversal is complete
                    if (c->h_counters[C_TMP0] > 0) {                            // the list kernel ran level cur (and maybe more)
                        visited_total += n_next; st.levels++; st.frontier_total += n_next; prevF = n_next;
                        account_small(n_next);
                    } else { advance_pending = true; front_valid = true; }      // too many discoveries (or edges) for it: they wait in bm_next
                    continue;
                }
                VGL_TRY(count_frontier());              // ids / offsets of level cur are needed again (bm_front is valid: cheap)
            }
        }
        if (!bottom_up && !ids_ready && counted && counted_from_bitmap && small_m > 0 && F <= bm_expand_f && late()) {
            // the tail of a traversal whose frontier came as a bitmap (top-down mode, or a direction-optimising one that never went bottom-up):
            // expand it from the bitmap and let the list kernel run whatever follows -- the treatment the switch back from bottom-up gets above
            trace("counted -> bitmap expand + list kernel");
            VGL_TRY(small_levels(0, -1, true));
            trace("bitmap expand + list kernel done");
            const int64_t m_level = c->h_counters[C_CHANGED], n_next = c->h_counters[C_BU_FOUND];
            st.td_steps++; st.edges_examined += m_level; st.td_edges += m_level; st.td_frontier += F;
            prevF = F;
            cur++;
            if (n_next == 0) break;
            if (c->h_counters[C_TMP0] > 0) {
                visited_total += n_next; st.levels++; st.frontier_total += n_next; prevF = n_next;
                account_small(n_next);
            } else { advance_pending = true; front_valid = true; }
            continue;
        }
        prevF = F;
        if (skipped && !bottom_up) VGL_FAIL("bfs_run: internal error (the count launch and the host disagree on the direction rule)");
        if (!bottom_up && g->blk_bfs && front_valid && (double)M >= blocked_share * (double)E) {
            // a level that holds a large share of the edges: the blocked pass (bitmaps in, bitmap + levels out: the state afterwards is
            // that after an emitting top-down level)
            trace("counted -> blocked level");
            VGL_TRY(vgl_bfs_blocked_level(c, g, d_levels, cur + 1));
            trace("blocked level done");
            advance_pending = true;
            st.td_steps++; st.edges_examined += M; st.td_edges += M; st.td_frontier += F;
            cur++;
            continue;
        }
        if (!bottom_up) {
            if (!counted) VGL_FAIL("bfs_run: internal error (frontier not counted)");
            if (ids_ready) {}                                   // written by the list kernel
            else if (counted_from_bitmap) VGL_TRY(vgl_bfs_bm_gnf(c, g, g->bm_front, false, true, M));
            else {
                vgl_pred_equal_i32 pred{d_levels, cur};
                vgl_timed_launch tl(c, "gnf");
                hipLaunchKernelGGL(vgl_k_gnf_write<vgl_pred_equal_i32>, dim3((unsigned)g->nvtiles), dim3(VGL_BLOCK), 0, c->stream, pred,
                                   g->nrows, g->row_begin, g->out.rowptr, g->vt_cnt_off, g->vt_deg_off, g->ids, g->offs, (int32_t *)nullptr, (int64_t)0);
            }
            trace("ids written");
            if (small_m > 0 && counted_from_bitmap && F <= VGL_SMALL_F && M <= small_m) {
                trace("counted + ids -> list kernel");
                VGL_TRY(small_levels((int32_t)F, -1));
                trace("list kernel done");
                account_small(F);
                continue;
            }
            const bool is_late = late();
            const bool emit = M <= VGL_TD_EMIT_EDGES || is_late;        // bm_next is all zero here (init / vgl_k_bm_advance leave it so)
            const bool td_counts = emit && use_hints && vgl_ceil_div(M, VGL_TILE) <= VGL_TD_COUNT_TILES;
            // the visited-bitmap probe pays once a good part of the vertices is visited (VGL_TD_FILTER_SHARE of V; 0 = always, 2 = never)
            const bool filter = is_late || (double)visited_total >= td_filter_share * (double)V;
            trace(emit ? "counted + ids -> top-down (emitting)" : "counted + ids -> top-down (levels only)");
            VGL_TRY(vgl_bfs_td_launch(c, g, (int32_t)F, M, d_levels, cur + 1, emit, counted_from_bitmap, td_counts, filter));
            trace("top-down done");
            hint_ready = td_counts && F > 0 && M > 0;     // (a level without edges launches nothing: C_NEXT_* would be another traversal's)
            advance_pending = emit;          // a top-down level is always followed by count_frontier (or the loop ends below)
            front_valid = emit;
            st.td_steps++; st.edges_examined += M; st.td_edges += M; st.td_frontier += F;
        } else {
            if (!front_valid) VGL_FAIL("bfs_run: internal error (bitmaps missing)");
            int64_t seq = 0;
            trace("-> bottom-up");
            VGL_TRY(vgl_bfs_bu_launch(c, g, d_levels, cur + 1, g->bm_visited, g->bm_front, g->bm_next, &seq, bu_in_a_row > 0 ? later_heavy_blocks : VGL_BU_BLOCKS));
            bu_in_a_row++;
            hipLaunchKernelGGL(vgl_k_bm_advance, dim3(vgl_grid(words)), dim3(VGL_BLOCK), 0, c->stream, words, g->bm_visited,
                               g->bm_front, g->bm_next);
            VGL_HIP_TRY(hipGetLastError());
            VGL_TRY(vgl_wait_counters(c, seq));
            st.bu_steps++; st.edges_examined += c->h_counters[C_BU_EDGES];
            st.bu_edges += c->h_counters[C_BU_EDGES]; st.bu_found += c->h_counters[C_BU_FOUND];
            F = c->h_counters[C_BU_FOUND]; M = 0;      // next frontier; bitmaps now describe level cur+1
            front_valid = true;
        }
        cur++;
    }
    st.discovered = visited_total;
    st.algorithmic_bytes = 8 * st.edges_examined + 20 * st.frontier_total + 4 * st.discovered + 4 * (int64_t)V +
                           (int64_t)st.bu_steps * (V / 8);
    if (stats) *stats = st;
    return 0;
}

// `count` traversals one after the other (the rounds loop of apps/bfs/bfs.cpp:36-50 behind one call): d_levels is reused and holds the levels
// of the last source afterwards, stats[i] (optional) those of traversal i.  Stops at the first failing traversal.
int vgl_hip_bfs_run_batch(vgl_hip_ctx *c, vgl_hip_graph *g, const int32_t *sources, int32_t count, int mode, int32_t *d_levels, vgl_hip_bfs_stats *stats)
{
    if (!sources || count < 0) VGL_FAIL("bfs_run_batch: null argument");
    for (int32_t i = 0; i < count; i++) VGL_TRY(vgl_hip_bfs_run(c, g, sources[i], mode, d_levels, stats ? stats + i : nullptr));
    return 0;
}

int vgl_hip_bfs_step_top_down(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t *d_levels, int32_t level, const uint64_t *d_visited_bits,
                              int64_t *local_frontier, int64_t *local_edges)
{
    if (!c || !g || !d_levels) VGL_FAIL("bfs_step_top_down: null argument");
    // visited bitmap over ALL vertices (destinations may live in any shard): the caller's replicated one, or rebuilt here
    const uint64_t *visited = d_visited_bits;
    if (!visited) {
        hipLaunchKernelGGL(vgl_k_levels_to_bitmap<true>, dim3(vgl_grid(g->V)), dim3(VGL_BLOCK), 0, c->stream, g->V, d_levels, -1,
                           g->bm_visited);
        visited = g->bm_visited;
    }
    vgl_pred_equal_i32 pred{d_levels, level};
    VGL_TRY(vgl_gnf_run(c, g, pred, g->ids, g->offs, nullptr, nullptr, nullptr, true, true));
    const int64_t F = c->h_counters[C_FRONT], M = c->h_counters[C_NEIGH];
    if (local_frontier) *local_frontier = F;
    if (local_edges) *local_edges = M;
    if (F > 0 && M > 0) {
        hipLaunchKernelGGL(vgl_k_tile_first, dim3(vgl_grid(F)), dim3(VGL_BLOCK), 0, c->stream, (int32_t)F, g->offs, g->tile_first);
        vgl_timed_launch tl(c, "bfs_top_down");
        hipLaunchKernelGGL((vgl_k_td_expand<false, false>), dim3((unsigned)vgl_ceil_div(M, VGL_TILE)), dim3(VGL_BLOCK), 0, c->stream, g->ids, g->offs,
                           g->tile_first, (int32_t)F, M, g->out.rowptr, g->out.adj, g->row_begin, visited, d_levels, level + 1,
                           (uint64_t *)nullptr, (int64_t *)nullptr, (uint32_t *)nullptr, (int64_t *)nullptr);
    }
    VGL_HIP_TRY(hipGetLastError());
    return 0;                                    // enqueued; the caller's next call on this context orders after it
}

int vgl_hip_bfs_step_top_down_bits(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t *d_levels, int32_t level, const uint64_t *d_visited_bits,
                                   const uint64_t *d_front_bits, uint64_t *d_next_bits, int64_t *local_frontier, int64_t *local_edges)
{
    if (!c || !g || !d_levels || !d_visited_bits || !d_front_bits || !d_next_bits) VGL_FAIL("bfs_step_top_down_bits: null argument");
    if (g->row_begin & 63) VGL_FAIL("bfs_step_top_down_bits: the first owned row must be a multiple of 64");
    VGL_TRY(vgl_bfs_bm_gnf(c, g, d_front_bits, true, true));     // owned part of the frontier: ids + edge offsets
    const int64_t F = c->h_counters[C_FRONT], M = c->h_counters[C_NEIGH];
    if (local_frontier) *local_frontier = F;
    if (local_edges) *local_edges = M;
    // Small levels OR every discovery into the candidate bitmap (one device-scope atomic each); a level with many edges leaves its
    // discoveries in `levels` only -- the value level + 1 is stored by this level and nobody else -- and the bitmap is read off `levels`
    // afterwards: V * 4 bytes streamed against ~40 us per million edges of atomics (a 4.3 M-edge level of an RMAT-24 traversal: 180 us
    // with the atomics; the fused traversal draws the same line at V / 24 edges, VGL_TD_EMIT_EDGES)
    int64_t emit_edges = std::max<int64_t>(65536, (int64_t)g->V / 256);        // (round 4: as VGL_TD_EMIT_EDGES of the fused traversal)
    if (c->bfs.shard_td_emit_edges >= 0) emit_edges = c->bfs.shard_td_emit_edges;
    const bool emit = M <= emit_edges;
    const int64_t words = vgl_ceil_div(g->V, 64);
    if (emit) VGL_TRY(vgl_zero_words(c, d_next_bits, words));
    if (F > 0 && M > 0) {                                        // tile_first came with the write pass
        vgl_timed_launch tl(c, "bfs_top_down");
        if (emit)
            hipLaunchKernelGGL((vgl_k_td_expand<true, false>), dim3((unsigned)vgl_ceil_div(M, VGL_TILE)), dim3(VGL_BLOCK), 0, c->stream, g->ids, g->offs,
                               g->tile_first, (int32_t)F, M, g->out.rowptr, g->out.adj, g->row_begin, d_visited_bits, d_levels, level + 1,
                               d_next_bits, (int64_t *)nullptr, (uint32_t *)nullptr, (int64_t *)nullptr);
        else
            hipLaunchKernelGGL((vgl_k_td_expand<false, false>), dim3((unsigned)vgl_ceil_div(M, VGL_TILE)), dim3(VGL_BLOCK), 0, c->stream, g->ids, g->offs,
                               g->tile_first, (int32_t)F, M, g->out.rowptr, g->out.adj, g->row_begin, d_visited_bits, d_levels, level + 1,
                               d_next_bits, (int64_t *)nullptr, (uint32_t *)nullptr, (int64_t *)nullptr);
    }
    if (!emit) {
        vgl_timed_launch tl(c, "gnf");
        hipLaunchKernelGGL(vgl_k_levels_to_bitmap<false>, dim3(vgl_grid(g->V)), dim3(VGL_BLOCK), 0, c->stream, g->V, (const int32_t *)d_levels, level + 1, d_next_bits);
    }
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_bfs_step_bottom_up(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t *d_levels, int32_t level, const uint64_t *d_visited_bits,
                               const uint64_t *d_front_bits, uint64_t *d_next_bits, int64_t *found, int64_t *probed)
{
    if (!c || !g || !d_levels || !d_visited_bits || !d_front_bits || !d_next_bits) VGL_FAIL("bfs_step_bottom_up: null argument");
    if (!g->in.rowptr) VGL_FAIL("bfs_step_bottom_up: the incoming CSR of the owned rows is required");
    // only the owned words of d_next_bits are written by the kernels: clear the rest so the buffer can be exchanged as is
    VGL_TRY(vgl_zero_words(c, d_next_bits, vgl_ceil_div(g->V, 64)));
    int64_t seq = 0;
    VGL_TRY(vgl_bfs_bu_launch(c, g, d_levels, level + 1, d_visited_bits, d_front_bits, d_next_bits, &seq));
    if (found || probed) {
        VGL_TRY(vgl_wait_counters(c, seq));
        if (found) *found = c->h_counters[C_BU_FOUND];
        if (probed) *probed = c->h_counters[C_BU_EDGES];
    }
    return 0;
}

int vgl_hip_bitmap_or_parts(vgl_hip_ctx *c, int64_t words, int parts, const uint64_t *d_in, uint64_t *d_out)
{
    if (!c || !d_in || !d_out) VGL_FAIL("bitmap_or_parts: null argument");
    if (words < 0 || parts < 1) VGL_FAIL("bitmap_or_parts: bad size");
    if (words > 0) hipLaunchKernelGGL(vgl_k_bitmap_or_parts, dim3(vgl_grid(words)), dim3(VGL_BLOCK), 0, c->stream, words, parts, d_in, d_out);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_levels_to_bitmap(vgl_hip_ctx *c, int32_t V, const int32_t *d_levels, int32_t level, uint64_t *d_bits)
{
    if (!c || !d_levels || !d_bits) VGL_FAIL("levels_to_bitmap: null argument");
    hipLaunchKernelGGL(vgl_k_levels_to_bitmap<false>, dim3(vgl_grid(V)), dim3(VGL_BLOCK), 0, c->stream, V, d_levels, level, d_bits);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

int vgl_hip_bitmap_to_ids(vgl_hip_ctx *c, int64_t words, const uint64_t *d_bits, int32_t cap, int32_t *d_out)
{
    if (!c || !d_bits || !d_out) VGL_FAIL("bitmap_to_ids: null argument");
    if (words < 0 || cap < 1) VGL_FAIL("bitmap_to_ids: bad size");
    return vgl_bitmap_to_ids(c, words, d_bits, 0, cap, d_out);
}

int vgl_hip_bfs_apply_ids(vgl_hip_ctx *c, int32_t V, int parts, int32_t cap, const int32_t *d_lists, int32_t *d_levels, int32_t level,
                          uint64_t *d_visited_bits, uint64_t *d_front_bits, const int32_t *d_degrees, int64_t *newly, int64_t *newly_degree)
{
    if (!c || !d_lists || !d_levels || !d_visited_bits || !d_front_bits) VGL_FAIL("bfs_apply_ids: null argument");
    if (parts < 1 || cap < 1) VGL_FAIL("bfs_apply_ids: parts and cap must be >= 1");
    const int nb = (int)vgl_grid((int64_t)parts * cap, 256);
    VGL_TRY(vgl_ensure_partials(c, (size_t)nb * 2 + 2));
    int64_t *partials = reinterpret_cast<int64_t *>(c->d_partials);
    VGL_HIP_TRY(hipMemsetAsync(d_front_bits, 0, sizeof(uint64_t) * (size_t)vgl_ceil_div(V, 64), c->stream));      // the new frontier is these vertices only
    hipLaunchKernelGGL(vgl_k_apply_ids, dim3(nb), dim3(VGL_BLOCK), 0, c->stream, V, parts, cap, d_lists, d_levels, level, d_visited_bits, d_front_bits,
                       d_degrees, partials);
    hipLaunchKernelGGL(vgl_k_apply_fold, dim3(1), dim3(VGL_BLOCK), 0, c->stream, nb, partials, c->d_counters);
    VGL_HIP_TRY(hipGetLastError());
    VGL_TRY(vgl_read_counters(c, false));
    if (newly) *newly = c->h_counters[C_TMP0];
    if (newly_degree) *newly_degree = c->h_counters[C_TMP1];
    return 0;
}

static int vgl_bfs_apply_bitmaps_range(vgl_hip_ctx *c, int32_t V, int parts, const uint64_t *d_bits_all, int32_t *d_levels, int32_t level,
                                       uint64_t *d_visited_bits, uint64_t *d_front_bits, const int32_t *d_degrees, int64_t *newly,
                                       int64_t *newly_degree, int32_t own_begin, int32_t own_end);

int vgl_hip_bfs_apply_bitmaps(vgl_hip_ctx *c, int32_t V, int parts, const uint64_t *d_bits_all, int32_t *d_levels, int32_t level,
                              uint64_t *d_visited_bits, uint64_t *d_front_bits, const int32_t *d_degrees, int64_t *newly,
                              int64_t *newly_degree)
{
    return vgl_bfs_apply_bitmaps_range(c, V, parts, d_bits_all, d_levels, level, d_visited_bits, d_front_bits, d_degrees, newly, newly_degree, 0, V);
}

int vgl_hip_bfs_apply_bitmaps_owned(vgl_hip_ctx *c, int32_t V, int parts, const uint64_t *d_bits_all, int32_t *d_levels, int32_t level,
                                    uint64_t *d_visited_bits, uint64_t *d_front_bits, const int32_t *d_degrees, int32_t own_begin,
                                    int32_t own_end, int64_t *newly_owned, int64_t *newly_owned_degree)
{
    if (!d_visited_bits) VGL_FAIL("bfs_apply_bitmaps_owned: the replicated visited bitmap is required");
    if (own_begin < 0 || own_end > V || own_begin > own_end || (own_begin & 63) || (own_end != V && (own_end & 63)))
        VGL_FAIL("bfs_apply_bitmaps_owned: the owned range must lie in [0, V] and start / end on multiples of 64");
    return vgl_bfs_apply_bitmaps_range(c, V, parts, d_bits_all, d_levels, level, d_visited_bits, d_front_bits, d_degrees, newly_owned,
                                       newly_owned_degree, own_begin, own_end);
}

static int vgl_bfs_apply_bitmaps_range(vgl_hip_ctx *c, int32_t V, int parts, const uint64_t *d_bits_all, int32_t *d_levels, int32_t level,
                                       uint64_t *d_visited_bits, uint64_t *d_front_bits, const int32_t *d_degrees, int64_t *newly,
                                       int64_t *newly_degree, int32_t own_begin, int32_t own_end)
{
    if (!c || !d_bits_all || !d_levels) VGL_FAIL("bfs_apply_bitmaps: null argument");
    if (parts < 1) VGL_FAIL("bfs_apply_bitmaps: parts must be >= 1");
    const int nb = (int)vgl_grid(vgl_ceil_div(V, 64), 1024);
    VGL_TRY(vgl_ensure_partials(c, (size_t)nb * 2 + 2));
    int64_t *partials = reinterpret_cast<int64_t *>(c->d_partials);
    hipLaunchKernelGGL(vgl_k_apply_bitmaps, dim3(nb), dim3(VGL_BLOCK), 0, c->stream, V, parts, vgl_ceil_div(V, 64), d_bits_all,
                       d_levels, level, d_visited_bits, d_front_bits, d_degrees, partials, own_begin, own_end);
    hipLaunchKernelGGL(vgl_k_apply_fold, dim3(1), dim3(VGL_BLOCK), 0, c->stream, nb, partials, c->d_counters);
    VGL_HIP_TRY(hipGetLastError());
    VGL_TRY(vgl_read_counters(c, false));
    if (newly) *newly = c->h_counters[C_TMP0];
    if (newly_degree) *newly_degree = c->h_counters[C_TMP1];
    return 0;
}

}  // extern "C"
