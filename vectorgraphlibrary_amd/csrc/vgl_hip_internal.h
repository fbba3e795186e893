// vgl_hip_internal.h -- shared host/device internals of libvgl_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <map>
#include <vector>
#include "../../include/vgl_hip.h"

// ---------------------------------------------------------------------------------------------
// geometry: 256-thread workgroups (4 wavefronts of 64), edge tiles of 2048 entries (8 per thread)
// ---------------------------------------------------------------------------------------------
constexpr int VGL_BLOCK = 256;
constexpr int VGL_EPT = 8;
constexpr int VGL_TILE = VGL_BLOCK * VGL_EPT;      // 2048 edges (or vertices) per workgroup tile
constexpr int VGL_WAVES = VGL_BLOCK / 64;

int vgl_set_error(const char *file, int line, const char *msg);
#define VGL_FAIL(msg) return vgl_set_error(__FILE__, __LINE__, (msg))
#define VGL_HIP_TRY(expr)                                                          \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) return vgl_set_error(__FILE__, __LINE__, hipGetErrorString(_e)); \
    } while (0)
#define VGL_TRY(expr)            \
    do {                         \
        int _s = (expr);         \
        if (_s != 0) return _s;  \
    } while (0)

// device counters (int64 slots in ctx->d_counters, mirrored into pinned host memory on demand)
enum {
    C_FRONT = 0, C_NEIGH = 1, C_BU_FOUND = 2, C_BU_EDGES = 3, C_HEAVY = 4, C_CHANGED = 5,
    C_EDGES = 6, C_TMP0 = 7, C_TMP1 = 8, C_JUMP = 9,
    C_NEXT_F = 10, C_NEXT_M = 11,    // device only: size / out-degree sum of the frontier an emitting top-down level (or the list kernel) just produced
    C_SKIPPED = 12, C_HINT = 13,     // host mirror only: the bitmap count skipped its degree pass (the level turns bottom-up) / the list kernel left C_NEXT_*
    C_NSLOTS = 32
};
// sharded accumulators: kernels launched with very many workgroups add into shard (blockIdx & (VGL_NSHARD-1)) so that no
// single address receives more than a few hundred device atomics; vgl_read_counters folds them into the counter slots.
constexpr int VGL_NSHARD = 1024;

struct vgl_timing_slot {
    int64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0.0;
};

// Tunables of the fused traversal (DESIGN "Run-time switches"), parsed from the environment when the context is created and again only when the
// process environment has changed since (vgl_ctx_refresh_env: one pass over the pointers of `environ`, no string is looked at) -- not a dozen
// getenv() per traversal.  A negative / NaN value = "not set": the traversal derives its default from the graph.
struct vgl_bfs_tunables {
    int64_t td_emit_edges = -1, small_m = -1, bm_expand = -1, shard_td_emit_edges = -1;
    double td_filter_share = -1.0, td_late_share = -1.0, blocked_share = -1.0;
    int later_heavy_blocks = -1, shard_sparse_cap = -1;
    bool no_hint = false, no_scan_bound = false, trace = false, bu_split = false;
};

struct vgl_hip_ctx {
    std::map<std::string, std::string> env;      // the VGL_* variables as they stood at the last refresh
    uint64_t env_signature = 0;
    vgl_bfs_tunables bfs;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t *d_counters = nullptr;   // C_NSLOTS
    int64_t *d_shards = nullptr;     // VGL_NSHARD accumulators folded into d_counters[C_EDGES]
    int64_t *h_counters = nullptr;   // pinned mirror (+ publish sequence number at [C_NSLOTS])
    int64_t publish_seq = 0;
    double *d_partials = nullptr;    // reduction partials (f64), capacity partials_cap
    size_t partials_cap = 0;
    bool timing = false;
    std::string timing_only;          // when not empty: only launches under this name are bracketed by events
    int timing_stride = 1;            // ... and of those only every stride-th
    int64_t timing_seen = 0;
    std::map<std::string, vgl_timing_slot> slots;
    std::vector<hipEvent_t> event_pool;
};

// the environment as the context sees it: refresh at the entry of a run (cheap when nothing changed), then read from the snapshot
void vgl_ctx_refresh_env(vgl_hip_ctx *c);
const char *vgl_env(vgl_hip_ctx *c, const char *name);            // refreshes, then looks up; nullptr when unset (use the value at once)

struct vgl_dir_csr {                 // one direction of the graph (borrowed) + derived tile table (owned)
    const int64_t *rowptr = nullptr; // nrows+1, rebased to 0
    const int32_t *adj = nullptr;    // global vertex ids
    int64_t edges = 0;
    int32_t *tile_row = nullptr;     // ntiles+1: local row that contains edge t*VGL_TILE
    int64_t ntiles = 0;
    int32_t *hub_rows = nullptr;     // pull sums (PageRank, HITS): rows with >= 512 edges grouped per wavefront + offsets (lazy)
    int32_t nhubs = 0;
    int hub_blocks = 0;              // workgroups of the pull kernel that run the hub schedule
    int32_t *giant_rows = nullptr;   // rows of at least VGL_PULL_GIANT_DEGREE entries, grouped per workgroup + offsets (the workgroup scheme of vgl_pull.h)
    int ngiants = 0, giant_blocks = 0;
    int32_t *hub_chunks = nullptr;   // unordered hub sums (HITS): per chunk (row, first entry - row start in units of VGL_PULL_CHUNK); then, per hub in
                                     // row order, (row, first chunk, chunks) triples -- see vgl_pull_find_hubs
    int n_hub_chunks = 0, n_hub_list = 0;
    double *hub_chunk_sums = nullptr;
    int32_t *pull_blk_row = nullptr; // pull sums: first row of every ordinary workgroup (+ end): <= 256 rows and ~16 K edges each (lazy)
    int pull_nblk = 0;
    int64_t max_row = -1;            // longest row (lazy; PageRank's choice between the ordered and the blocked pull)
};

struct vgl_hip_graph {
    uint64_t uid = 0;                // unique per created handle (a freed handle's address may be reused: caches key on this, not on the pointer)
    int32_t V = 0, row_begin = 0, row_end = 0, nrows = 0;
    vgl_dir_csr out, in;
    // scratch shared by the fused algorithms (allocated at creation, sized by V / nrows / edges)
    uint64_t *bm_visited = nullptr, *bm_front = nullptr, *bm_next = nullptr; // ceil(V/64)+1 words each
    uint64_t *bm_in_nz = nullptr;    // bit v = owned vertex v has incoming edges (bottom-up candidates)
    int32_t *in_head = nullptr;      // two planes of 16-byte records (in-neighbours 0-3 and 4-7 of the eight smallest ids, -1 padded), ONE RECORD PER
                                     // OWNED ROW THAT HAS INCOMING EDGES, in row order: record index = in_nz_rank[group] + rank of the row among the set
                                     // bits of its in_nz word -- the rows without incoming edges (45 % of an RMAT graph) are never bottom-up
                                     // candidates and used to take half of every 128-byte line of the planes
    int32_t *in_nz_rank = nullptr;   // per 64-row group: number of owned rows with incoming edges before the group
    int32_t in_nz_rows = 0;          // owned rows with incoming edges = records per plane
    int32_t *pr_indeg = nullptr;     // sharded PageRank: in-degrees minus self loops of ALL vertices, summed over the ranks once per graph handle
    bool pr_indeg_ready = false;
    uint64_t *bm_in_long = nullptr;  // bit v = owned vertex v has more than 8 incoming edges (deferred to the wavefront pass when it misses)
    int32_t *ids = nullptr;          // nrows
    int64_t *offs = nullptr;         // nrows+1
    int32_t *vt_cnt = nullptr, *vt_cnt_off = nullptr;   // per vertex tile
    int64_t *vt_deg = nullptr, *vt_deg_off = nullptr;
    int64_t nvtiles = 0;
    int32_t *vt_min_deg = nullptr;   // per vertex tile: smallest out-degree of its rows (lower bound of a frontier's edge count from its per-tile sizes)
    uint8_t *gnf_bits = nullptr;     // generate_new_frontier of the operator classes: the predicate's bits (byte v >> 3, bit v & 7; lazy, V / 8 bytes) -- the
                                     // compaction of a SPARSE result reads these 2 MiB instead of 64 MiB of int32 flags (vgl_hip_gnf_begin / _complete)
    int32_t *tile_first = nullptr;   // out.ntiles + 2
    int32_t *heavy = nullptr;        // nrows + slack: per-workgroup segments of deferred bottom-up vertices
    int32_t *heavy_cnt = nullptr;    // one count per bottom-up workgroup
    int32_t *heavy_off = nullptr;    // exclusive prefix of heavy_cnt (+ total)
    int64_t *bu_partials = nullptr;  // 4 partial counters per bottom-up workgroup
    uint32_t *tickets = nullptr;     // arrival counters of the "last workgroup finishes the job" kernels (reset by that workgroup)
    int32_t *epoch = nullptr;        // V (SSSP active filter)
    float *fscratch = nullptr;       // V (PR contrib)
    float *fscratch2 = nullptr;      // V (PR rdeg)
    float *fscratch3 = nullptr;      // V (PR new ranks)
    int32_t *iscratch = nullptr;     // V (PR indeg when not supplied)
    uint8_t *ds_tile_active = nullptr;   // delta-stepping SSSP: one byte per out-edge tile (lazy)
    vgl_hip_graph *transposed = nullptr; // SCC: handle with the two directions swapped (backward reach = BFS on it), lazy, owned
    int64_t *ds_partials = nullptr;
    struct vgl_blocked_plan *blk_pr = nullptr;  // PageRank's blocked pull over the outgoing CSR (lazy, owned; vgl_blocked.h)
    struct vgl_blocked_plan *blk_cc = nullptr;  // the Shiloach-Vishkin hook as a blocked pass (lazy, owned)
    struct vgl_blocked_plan *blk_bfs = nullptr; // the large top-down BFS levels as a blocked pass (vgl_hip_bfs_prepare_blocked, owned)
    struct vgl_blocked_plan *blk_path = nullptr; // STRUCTURE of the path layouts (Bellman-Ford / widest-path pull: rows gather, edge values; lazy, owned): built once
                                                 // per graph, every vgl_hip_sssp_pull_plan shares it and loads its own weights with one gather pass
    std::string blk_path_key;                    // the layout switches it was built under (fuse threshold, unit sizes, piece bound: the tests vary them per plan)
};

struct vgl_hip_frontier {
    vgl_hip_graph *g = nullptr;
    int32_t *flags = nullptr;        // V
    int32_t *ids = nullptr;          // V
    int32_t size = 0;
    int64_t neighbours = 0;
    int sparsity = VGL_HIP_FRONTIER_ALL_ACTIVE;
    // advance plan of a SPARSE frontier (built on demand by vgl_hip_frontier_advance_plan)
    int64_t *offs = nullptr;         // V+1 exclusive edge offsets of ids[] in the planned direction
    int32_t *tile_first = nullptr;   // ceil(E/VGL_TILE)+2
    int64_t *blk_sum = nullptr, *blk_off = nullptr;   // per 2048-id block degree sums / offsets
    bool borrowed = false;           // flags / ids belong to the caller (vgl_hip_frontier_create_on)
    int plan_dir = -1;               // direction whose edge offsets `offs` AND tile table `tile_first` describe the current ids (-1: none; set by
                                     // vgl_hip_gnf_complete and vgl_hip_frontier_advance_plan, voided by every change of the ids)
    int64_t plan_edges = 0;          // edges of the ids in that direction
};

// timing helpers (no-ops unless ctx->timing)
struct vgl_timed_launch {
    vgl_hip_ctx *ctx; vgl_timing_slot *slot; hipEvent_t a, b;
    vgl_timed_launch(vgl_hip_ctx *c, const char *name);
    ~vgl_timed_launch();
};

int vgl_read_counters(vgl_hip_ctx *ctx, bool fold_shards = true);   // D2H all slots into h_counters, synchronises
// kernels that publish their own results (last workgroup writes the slots it produced + `seq` into the pinned mirror):
// seq = vgl_next_seq() is passed to the kernel, vgl_wait_counters(seq) spins until it shows up
int64_t vgl_next_seq(vgl_hip_ctx *ctx);
int vgl_wait_counters(vgl_hip_ctx *ctx, int64_t seq);
int vgl_zero_counters(vgl_hip_ctx *ctx, int first, int count);
int vgl_ensure_partials(vgl_hip_ctx *ctx, size_t n);
int vgl_build_tile_rows(vgl_hip_ctx *ctx, struct vgl_dir_csr &d, int32_t nrows);
// frontier from a bitmap over the owned words (bfs.hip): count = sizes to h_counters[C_FRONT / C_NEIGH] (waits), write = ids + edge offsets + tile table
// what the direction rule (gpu_change_state) needs besides the new frontier's F and M: with it the count launch may find on the device that
// the level turns bottom-up and leave out everything a top-down level would need (degree sums, compaction offsets)
struct vgl_do_hint { int64_t prev_f, visited_total, V, factor; };
int vgl_bfs_bm_gnf(vgl_hip_ctx *c, struct vgl_hip_graph *g, const uint64_t *front, bool count, bool write, int64_t M_known = -1, bool advance = false,
                   const vgl_do_hint *hint = nullptr);   // d.tile_row / d.ntiles from d.rowptr / d.edges (owned by the caller)
// set bits of `words` 64-bit words as ids (64 * (word_base + word) + bit): d_out[0] = their number, d_out[1 .. 1 + cap) = the first cap handed out
int vgl_bitmap_to_ids(vgl_hip_ctx *c, int64_t words, const uint64_t *d_bits, int64_t word_base, int32_t cap, int32_t *d_out);
int vgl_zero_words(vgl_hip_ctx *c, uint64_t *d_words, int64_t words);      // one launch (bfs.hip)

static inline int64_t vgl_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Large, short-lived or plan-sized device buffers come from a stream-ordered memory pool OWNED BY THIS LIBRARY (one per device, created on
// first use: hipMemPoolCreate + hipMallocFromPoolAsync on the context's stream; the device's default pool, which other hipMallocAsync
// users of the process share, is left as it was found).  The pool keeps what is freed up to VGL_POOL_KEEP_GB (default 48 GiB): a plain
// hipMalloc right after tens of GB were hipFree'd was measured to stall for 0.5 - 1.3 s on this driver, and fresh mappings cost ~14 ms
// per GB on first touch; pool memory that has been used once costs neither.  vgl_hip_ctx_trim / vgl_hip_ctx_destroy hand it back.
hipMemPool_t vgl_lib_pool(int device);          // context.hip; nullptr when the pool could not be created (callers fall back to hipMallocAsync)
// LARGE blocks do not come from the pool (round 5, late).  Kernels that touched a freshly grown multi-GB pool block ended in a GPU memory fault in
// 2 - 12 % of the processes that built an RMAT-24 graph (apps/bin/*_hip -s 24: `GPU coredump`, exit by SIGPIPE, the bench's operator leg lost a row;
// profiles/r05_pool_fault_ab.log: 7 of 60 runs with the scratch of vgl_hip_coo_to_csr from the pool, 0 of 60 with it from hipMalloc; ROCm 7.2; the
// round's first workaround drew the line at 2^33 bytes, where the fault was first seen).  Blocks of at least vgl_pool_block_limit() bytes
// (VGL_POOL_MAX_MB, default 64 MiB) are plain hipMalloc blocks, remembered so that vgl_pool_free hands them to hipFree after a stream
// synchronisation; the pool keeps what it is good at, the many small buffers of a plan build.
size_t vgl_pool_block_limit();                  // context.hip
void vgl_big_block_remember(void *p);           // context.hip: p came from hipMalloc
bool vgl_big_block_forget(void *p);             // context.hip: true (and forgotten) when p came from hipMalloc
static inline hipError_t vgl_pool_alloc(hipStream_t st, void **p, size_t bytes)
{
    if (bytes >= vgl_pool_block_limit()) {
        const hipError_t e = hipMalloc(p, bytes);
        if (e == hipSuccess) vgl_big_block_remember(*p);
        return e;
    }
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess)
        if (hipMemPool_t pool = vgl_lib_pool(dev)) return hipMallocFromPoolAsync(p, bytes ? bytes : 16, pool, st);
    return hipMallocAsync(p, bytes ? bytes : 16, st);
}
static inline void vgl_pool_free(hipStream_t st, void *p)
{
    if (!p) return;
    if (vgl_big_block_forget(p)) { (void)hipStreamSynchronize(st); (void)hipFree(p); }      // (kernels of this stream may still use it)
    else (void)hipFreeAsync(p, st);
}
// scratch of the graph builders (a block and where it came from: kept as a type of its own for the cleanup guards of gen.hip)
struct vgl_scratch {
    void *p = nullptr;
    bool pooled = false;
};
static inline hipError_t vgl_scratch_alloc(hipStream_t st, vgl_scratch *b, size_t bytes) { b->pooled = true; return vgl_pool_alloc(st, &b->p, bytes); }
static inline void vgl_scratch_free(hipStream_t st, vgl_scratch *b)
{
    if (!b->p) return;
    vgl_pool_free(st, b->p);
    b->p = nullptr;
}

// ---------------------------------------------------------------------------------------------
// device helpers (wave = 64 lanes)
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ int vgl_lane() { return threadIdx.x & 63; }
__device__ __forceinline__ int vgl_wave() { return threadIdx.x >> 6; }

// true in exactly one workgroup per launch: the last one to get here.  Protocol (cheap on a multi-XCD part: a device-scope
// release FENCE writes back the whole L2 of the XCD -- every workgroup doing that made a BFS 2.4x slower):
//   * what the last workgroup must see is written by THREAD 0 with vgl_put_agent (device-scope atomic exchange) BEFORE the
//     call, and the values the exchanges RETURN are folded into `dep`: the ticket increment is made data-dependent on them, so
//     it cannot be issued before the exchanges have completed at the coherence point (no fence, no cache write-back);
//   * the last workgroup reads those values with vgl_load_agent (device-scope atomic load).
// Call from all threads of every workgroup (dep only matters in thread 0); the ticket is left at 0 for the next launch.
__device__ __forceinline__ uint32_t vgl_put_agent(int32_t *p, int32_t v) { return (uint32_t)atomicExch(p, v); }
__device__ __forceinline__ uint32_t vgl_put_agent(int64_t *p, int64_t v)
{
    return (uint32_t)atomicExch(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v);
}
template <class T> __device__ __forceinline__ void vgl_store_agent(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ T vgl_load_agent(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Tickets are two-level: same-address device atomics serialise (~12 ns each: 2048 workgroups on one counter cost 25 us, more than
// the launches this saves), so workgroup b arrives at sub-counter b % 32 (its own cache line) and only the last arrival of a
// sub-counter moves on to the top counter: ~64 + 32 serialised atomics instead of 2048.
constexpr int VGL_TICKET_FAN = 32;
constexpr int VGL_TICKET_STRIDE = 16;                                  // uint32 per counter: one 64-byte line each
constexpr int VGL_TICKET_WORDS = (VGL_TICKET_FAN + 1) * VGL_TICKET_STRIDE;
__device__ __forceinline__ bool vgl_last_block(uint32_t *ticket, uint32_t dep)
{
    __shared__ int s_is_last;
    if (threadIdx.x == 0) {
        uint32_t one = 1u;
        asm volatile("" : "+v"(one) : "v"(dep));                     // `one` now depends on the returned values
        const uint32_t nb = gridDim.x, sub = blockIdx.x % VGL_TICKET_FAN;
        const uint32_t expect = (nb - sub + VGL_TICKET_FAN - 1) / VGL_TICKET_FAN;           // workgroups with this residue
        uint32_t *sub_ticket = ticket + (1 + sub) * VGL_TICKET_STRIDE;
        bool last = false;
        if (__hip_atomic_fetch_add(sub_ticket, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == expect - 1) {
            vgl_store_agent(sub_ticket, 0u);
            const uint32_t groups = nb < (uint32_t)VGL_TICKET_FAN ? nb : (uint32_t)VGL_TICKET_FAN;
            if (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1) {
                vgl_store_agent(ticket, 0u);
                last = true;
            }
        }
        s_is_last = last;
    }
    __syncthreads();
    return s_is_last != 0;
}
// results of a last workgroup go to the device counters and straight into the pinned host mirror, then the sequence number
__device__ __forceinline__ void vgl_publish2(int64_t *counters, volatile int64_t *host, int64_t seq, int slot_a, int64_t a, int slot_b, int64_t b)
{
    counters[slot_a] = a; counters[slot_b] = b;
    host[slot_a] = a; host[slot_b] = b;
    __threadfence_system();
    host[C_NSLOTS] = seq;
    __threadfence_system();
}

template <class T>
__device__ __forceinline__ T vgl_wave_incl_add(T v)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        T n = __shfl_up(v, o);
        if (vgl_lane() >= o) v += n;
    }
    return v;
}
__device__ __forceinline__ int vgl_wave_incl_max(int v)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int n = __shfl_up(v, o);
        if (vgl_lane() >= o) v = max(v, n);
    }
    return v;
}
template <class T>
__device__ __forceinline__ T vgl_wave_reduce_add(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// exclusive block scan (add); smem must hold VGL_WAVES elements of T; returns exclusive prefix, *total = block sum
template <class T>
__device__ __forceinline__ T vgl_block_excl_add(T v, T *smem, T *total)
{
    T inc = vgl_wave_incl_add(v);
    __syncthreads();                       // protect smem reuse across calls
    if (vgl_lane() == 63) smem[vgl_wave()] = inc;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < VGL_WAVES; w++) {
        T s = smem[w];
        if (w < vgl_wave()) base += s;
        tot += s;
    }
    *total = tot;
    return base + inc - v;
}
// exclusive block scan (max) for non-negative ints, identity 0
__device__ __forceinline__ int vgl_block_excl_max(int v, int *smem)
{
    int inc = vgl_wave_incl_max(v);
    __syncthreads();
    if (vgl_lane() == 63) smem[vgl_wave()] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < VGL_WAVES; w++)
        if (w < vgl_wave()) base = max(base, smem[w]);
    int up = __shfl_up(inc, 1);
    if (vgl_lane() == 0) up = 0;
    return max(base, up);
}
template <class T>
__device__ __forceinline__ T vgl_block_reduce_add(T v, T *smem)
{
    v = vgl_wave_reduce_add(v);
    __syncthreads();
    if (vgl_lane() == 0) smem[vgl_wave()] = v;
    __syncthreads();
    T tot = 0;
#pragma unroll
    for (int w = 0; w < VGL_WAVES; w++) tot += smem[w];
    return tot;
}

// Row map of an edge tile.  Positions [0, n) of the tile belong to consecutive "rows" (CSR rows or frontier
// positions) r_first..r_last whose start offsets are starts[r] (int64, monotone).  On return s_map[i] holds
// (row of slot i) - r_first.  Empty rows are skipped naturally (the last row starting at a slot owns it).
// s_map: VGL_TILE ints of LDS; s_w: VGL_WAVES ints.  All threads of the workgroup must call.
__device__ __forceinline__ void vgl_tile_row_map(int *s_map, int *s_w, const int64_t *starts, int64_t e0,
                                                 int r_first, int r_last)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) s_map[tid + j * VGL_BLOCK] = 0;
    __syncthreads();
    for (int r = r_first + 1 + tid; r <= r_last; r += VGL_BLOCK) {
        const int64_t q = starts[r] - e0;          // > 0 because r_first contains e0
        if (q < VGL_TILE) atomicMax(&s_map[(int)q], r - r_first);
    }
    __syncthreads();
    int m[VGL_EPT];
    int run = 0;
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) { run = max(run, s_map[tid * VGL_EPT + j]); m[j] = run; }
    const int pre = vgl_block_excl_max(run, s_w);
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) s_map[tid * VGL_EPT + j] = max(m[j], pre);
    __syncthreads();
}
#endif  // __HIPCC__
