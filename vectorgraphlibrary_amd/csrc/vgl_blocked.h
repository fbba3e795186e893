// vgl_blocked.h -- "blocked advance": an all-edges gather/accumulate pass whose per-edge random accesses stay inside a CU.
//
// What it replaces: the all-active advance of the reference (multicore/advance_worker.hpp:62-149) with an edge operator of the
// form   y[a] (+)= f(x[g], w)   -- PageRank's pull (pr.hpp:105-124), the Bellman-Ford relax (shortest_paths.hpp:123-133), the
// Shiloach-Vishkin hook (shiloach_vishkin.hpp:37-50).  On MI355X a 4-byte gather x[adj[e]] from a table that does not fit the
// 32 KiB L1 costs one 128-byte L2 -> L1 line whatever the L2 hit rate: ~2.1e11 gathers/s for the whole chip (measured,
// profiles/README.md "gather floor"), i.e. 1.7 TB/s of algorithmic bytes -- the ceiling every all-edges kernel of round 1 sat on.
// LDS serves random 4-byte reads ~20x faster, so the pass is split in two streaming kernels around 128 KiB LDS windows
// (propagation blocking, adapted to a static graph so that only VALUES travel through memory):
//
//   blocks     vertices are cut into blocks of VGL_BLK = 32768 ids (128 KiB of 4-byte values: one LDS window, 1024-thread workgroup);
//              operators with 8-byte accumulators (PageRank's exact fixed-point sums) use 16384-id blocks on the accumulate side
//   segments   the edges with gather-side block gb and accumulate-side block ab form segment (gb, ab); a segment is padded to
//              whole 64-entry CHUNKS (one wavefront-width; pad entries point at dummy accumulators)
//   mid order  chunks sorted by (gb, ab): arrays g_lo (uint16 index of x inside block gb), w_mid (optional f32 edge values)
//   A order    chunks sorted by (ab, gb): arrays a_lo (uint16 index of y inside block ab) and the scratch `vals`
//   mid_to_a   chunk m of the mid order is chunk mid_to_a[m] of the A order  (the values are WRITTEN scattered -- 256-byte chunks, whole
//              segments contiguous -- and read as one stream; the other way round, contiguous writes and scattered reads through an
//              a_to_mid table, measured the same within 3 %: gather 1.41 + accumulate 1.48 ms against 1.46 + 1.22 ms on uniform-25)
//
//   gather kernel      per unit (gb, chunk range): x[block gb] -> LDS; per edge vals[A position] = f(lds[g_lo], w)
//                      reads 2 (+4) B/edge, writes 4 B/edge, all in >= 256-byte contiguous runs
//   accumulate kernel  per unit (ab, chunk range): acc[block ab] in LDS; per edge acc[a_lo] (+)= vals; epilogue per vertex
//                      reads 6 B/edge
// HBM traffic 12 (16 with edge values) B/edge instead of 8 (12) B/edge + one 64-128-byte line per gather.
//
// Within a segment the edges keep CSR order (stable sort), so the layout is a pure function of the graph.
//
// LDS atomics (profiles/microbench/lds_atomic_bench.hip, one 1024-thread workgroup per CU, random addresses in the window):
// ds_add_u32 / ds_min_i32 / ds_add_rtn_u32 6.3-6.4 clocks per wavefront instruction (a plain ds_read_b32: 6.0), ds_add_u64 10.8,
// ds_add_f32 193 -- float atomics are ~30x slower than integer ones, so sums are kept in 64-bit fixed point (which also makes them
// exact and independent of the order of arrival) and minima / maxima work on the integer order of non-negative floats.
#pragma once
#include "vgl_hip_internal.h"

constexpr int VGL_BLK_BITS = 15;
constexpr int VGL_BLK = 1 << VGL_BLK_BITS;       // vertices per block
constexpr int VGL_CHUNK = 64;                    // entries per chunk
constexpr int VGL_BTHREADS = 1024;               // workgroup of the two kernels (16 wavefronts, one workgroup per CU: 128 KiB LDS)
constexpr int VGL_BWAVES = VGL_BTHREADS / 64;
constexpr int VGL_BGROUP = 8;                    // chunks per wavefront step (16 bytes of uint16 indices per lane)
// the windows below are sized for gfx950's 160 KiB of LDS per CU; other targets (gfx942: 64 KiB) cannot hold them
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "vgl_blocked.h: the blocked advance keeps 128 KiB windows in LDS and is written for gfx950 (MI355X) only; build with --offload-arch=gfx950"
#endif
// a_lo of pad entry i of a chunk = (accumulate block size) + i: 64 dummy accumulators behind the window, no branch per entry

struct vgl_blk_unit {                            // one workgroup's share: chunks [chunk0, chunk1) of block `block`
    int32_t block;
    int32_t slab;                                // accumulate units of a block cut into several units: partial-result slab, else -1
    uint32_t chunk0, chunk1;
};
struct vgl_blk_multi {                           // a block whose accumulation is spread over several units
    int32_t block, first_slab, nslabs, pad;
};

// ---- fused tiles (dense segments): both windows in LDS, nothing travels through `vals` ----
// A (gather block, accumulate block) pair that holds many edges -- the hub x hub, hub x anything and anything x hub pairs of a
// degree-sorted RMAT graph -- is processed by ONE kernel: 64 KiB window of x + 64 KiB window of accumulators (blocks of VGL_FBLK = 16384
// ids on both sides), per edge 2 + 2 (+ 4) bytes streamed instead of 12 (16) through the two-pass scheme.  The epilogue of a tile costs a
// sweep over its 16384 accumulators, so only pairs with at least `fuse_min_edges` edges are laid out this way; the rest stays two-pass.
constexpr int VGL_FBLK_BITS = 14;
constexpr int VGL_FBLK = 1 << VGL_FBLK_BITS;
struct vgl_blk_fseg { int32_t ab; uint32_t chunk0, chunk1; };      // one dense pair of the unit's gather block: chunks [chunk0, chunk1)
struct vgl_blk_funit { int32_t gb, seg0, seg1, pad; };             // a workgroup's share: segments [seg0, seg1) of gather block gb

struct vgl_blocked_plan {
    hipStream_t stream = nullptr;                // the stream whose memory pool owns the arrays below
    int32_t g_count = 0, a_count = 0;            // index ranges of the gather / accumulate side
    int32_t nG = 0, nA = 0;
    int a_bits = VGL_BLK_BITS;                   // log2 of the accumulate-side block (15: 4-byte accumulators, 14: 8-byte)
    int64_t edges = 0;                           // edges kept (self loops may be dropped at build time)
    uint32_t nchunks = 0;
    uint16_t *g_lo = nullptr, *a_lo = nullptr;
    float *w_mid = nullptr;
    uint32_t *mid_to_a = nullptr;
    uint32_t *vals = nullptr;                    // scratch: 4 bytes per entry (or 8 per chunk: value_bits = 1)
    vgl_blk_unit *g_units = nullptr, *a_units = nullptr;
    int n_g_units = 0, n_a_units = 0;
    vgl_blk_multi *multi = nullptr;
    int n_multi = 0, n_slabs = 0;
    void *slabs = nullptr;                       // n_slabs * 128 KiB (one window of accumulators each)
    uint8_t *g_dirty = nullptr;                  // nG: gather blocks whose x changed since the last pass (filtered passes)
    // fused tiles (empty unless the plan was built with fuse_min_edges > 0)
    uint16_t *f_g_lo = nullptr, *f_a_lo = nullptr;
    float *f_w = nullptr;
    vgl_blk_fseg *f_segs = nullptr;
    vgl_blk_funit *f_units = nullptr;
    int n_f_segs = 0, n_f_units = 0;
    uint32_t f_nchunks = 0;
    int64_t f_edges = 0;                         // edges laid out as fused tiles (part of `edges`)
    // a direction with 2^32 edges or more is laid out in row-range PIECES of at most 2^31 edges, each a plan of its own over the same
    // index spaces (its rows keep their global numbers); a pass runs the pieces one after the other (min / max-type operators only)
    vgl_blocked_plan *next = nullptr;
    int64_t *piece_rowptr = nullptr;             // the piece's rebased row offsets and tile table (owned; null for a whole-direction plan)
    int32_t *piece_tile_row = nullptr;
    // Edge values (round 5).  The STRUCTURE above depends on the graph alone; a layout that carries edge values also keeps, per slot of w_mid / f_w,
    // the CSR position its value comes from (0xFFFFFFFF: a pad entry) -- the counterpart of the reference's edges_reorder_indexes, from which every
    // weight layout is derived (csr_edges_array.hpp:31-40).  Loading another weights array is then one gather pass (vgl_blocked_plan_load_weights)
    // instead of a second radix sort, and several value arrays can share one structure (vgl_blocked_plan_share).
    uint32_t *w_src_mid = nullptr, *w_src_f = nullptr;
    int64_t w_base = 0;                          // CSR position of the piece's first edge (a piece indexes the direction's weights from there)
    bool shared = false;                         // everything but w_mid / f_w / g_dirty belongs to the plan this one was shared from
    vgl_blocked_plan *shared_from = nullptr;     // (head of a shared chain only) the structure's head
    int sharers = 0;                             // (structure head) plans that share it; a structure destroyed while shared is freed by its last sharer
    bool orphan = false;
};

// Build the plan from one CSR direction.  gather_rows = 0: x is indexed by the adjacency ids (range `ncols`), y by the local rows;
// gather_rows = 1: x by the local rows, y by the adjacency ids.  skip_self: edges whose adjacency id equals row_base + row are left
// out (PageRank, pr.hpp:111).  d_weights (optional): f32 per CSR position, carried to the mid order.  Synchronises; offline cost
// (a 3-pass radix sort of the edges), like the reference's graph import.
// value_bits = 1: what travels is one BIT per edge (the blocked top-down BFS level): `vals` then holds one 64-bit word per chunk.
// fuse_min_edges > 0 (4-byte accumulators and 32-bit values only): block pairs of 16384 x 16384 ids with at least that many edges become
// fused tiles.
int vgl_blocked_plan_build(vgl_hip_ctx *c, const vgl_dir_csr &dir, int32_t nrows, int32_t row_base, int32_t ncols, int gather_rows,
                           int skip_self, const float *d_weights, int a_bits, vgl_blocked_plan **out, int value_bits = 32, int fuse_min_edges = 0);
// keep_edge_index != 0 (implied by d_weights): the layout keeps the CSR position behind every value slot, see vgl_blocked_plan::w_src_mid
int vgl_blocked_plan_build_indexed(vgl_hip_ctx *c, const vgl_dir_csr &dir, int32_t nrows, int32_t row_base, int32_t ncols, int gather_rows,
                                   int skip_self, int a_bits, vgl_blocked_plan **out, int fuse_min_edges);
// (re)fills the value arrays of a layout that keeps its edge index from d_weights (f32 per CSR position of the direction): one gather pass
int vgl_blocked_plan_load_weights(vgl_hip_ctx *c, vgl_blocked_plan *p, const float *d_weights);
// a second set of value arrays over the structure of `structure` (which must outlive the result and keep its edge index)
int vgl_blocked_plan_share(vgl_hip_ctx *c, const vgl_blocked_plan *structure, vgl_blocked_plan **out);
void vgl_blocked_plan_destroy(vgl_blocked_plan *p);
static inline int64_t vgl_blocked_plan_edges(const vgl_blocked_plan *p) { int64_t e = 0; for (; p; p = p->next) e += p->edges; return e; }

#ifdef __HIPCC__
// ---------------------------------------------------------------------------------------------------------------------------
// kernels (templates over the edge operator; instantiated by pr.hip / sssp.hip / cc.hip)
//   OP::load(i)            -> uint32 bits of x[i] (i = index on the gather side, < g_count)
//   OP::edge(xbits, w)     -> uint32 bits of the value that travels to the accumulate side (w = 0 without edge values)
//   OP::acc_t              accumulator type in LDS: uint32_t (blocks of 32768 on the accumulate side) or uint64_t (16384)
//   OP::identity()         -> value the accumulators start from
//   OP::accumulate(p, v)   LDS atomic that folds the travelling uint32 v into *p
//   OP::finish(i, acc)     epilogue of vertex i of the accumulate side when its block was handled by ONE unit
//   OP::MARK               true: the operator keeps a bitmap of improved vertices -- finish_m / partial_m return "improved", mark(v0, mask, shared)
//                          gets one word per wavefront step (shared: other workgroups may write the same word)
//   OP::partial(i, acc)    a block cut into several units: fold this unit's result into memory (min-type operators: a global
//                          atomic), or return false to have the unit write its accumulators to a slab (sum-type operators);
//                          vgl_k_blk_finish_slabs then adds the slabs in unit order and calls finish
// ---------------------------------------------------------------------------------------------------------------------------
template <class OP, bool WEIGHTED>
__global__ __launch_bounds__(VGL_BTHREADS) void vgl_k_blk_gather(const vgl_blk_unit *units, const uint16_t *g_lo, const float *w_mid,
                                                                 const uint32_t *mid_to_a, uint32_t *vals, int32_t g_count,
                                                                 const uint8_t *g_dirty, OP op)
{
    __shared__ uint32_t s_x[VGL_BLK];
    const vgl_blk_unit u = units[blockIdx.x];
    if (g_dirty && !g_dirty[u.block]) return;                       // nothing in this block changed: its values in `vals` still stand
    const int32_t base = u.block << VGL_BLK_BITS;
    const int n = min(VGL_BLK, g_count - base);
    for (int i = threadIdx.x; i < n; i += VGL_BTHREADS) s_x[i] = op.load(base + i);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 3, off = (lane & 7) * 8;                // this lane: entries off..off+7 of chunk m0 + sub
    for (uint32_t m0 = u.chunk0 + wave * VGL_BGROUP; m0 < u.chunk1; m0 += VGL_BWAVES * VGL_BGROUP) {
        const uint32_t m = m0 + sub;
        if (m >= u.chunk1) continue;
        const size_t e = (size_t)m * VGL_CHUNK + off;
        const uint4 gl = *reinterpret_cast<const uint4 *>(g_lo + e);
        const uint32_t a = mid_to_a[m];
        float w[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (WEIGHTED) {
            const float4 w0 = *reinterpret_cast<const float4 *>(w_mid + e), w1 = *reinterpret_cast<const float4 *>(w_mid + e + 4);
            w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w; w[4] = w1.x; w[5] = w1.y; w[6] = w1.z; w[7] = w1.w;
        }
        const uint32_t g[4] = {gl.x, gl.y, gl.z, gl.w};
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            v[2 * k] = op.edge(s_x[g[k] & 0xFFFFu], w[2 * k]);
            v[2 * k + 1] = op.edge(s_x[g[k] >> 16], w[2 * k + 1]);
        }
        uint4 *dst = reinterpret_cast<uint4 *>(vals + (size_t)a * VGL_CHUNK + off);
        dst[0] = make_uint4(v[0], v[1], v[2], v[3]);
        dst[1] = make_uint4(v[4], v[5], v[6], v[7]);
    }
}

template <class OP>
__global__ __launch_bounds__(VGL_BTHREADS) void vgl_k_blk_accumulate(const vgl_blk_unit *units, const uint16_t *a_lo, const uint32_t *vals,
                                                                     int32_t a_count, typename OP::acc_t *slabs, OP op)
{
    typedef typename OP::acc_t acc_t;
    constexpr int ABITS = sizeof(acc_t) == 8 ? VGL_BLK_BITS - 1 : VGL_BLK_BITS, AN = 1 << ABITS;
    __shared__ acc_t s_acc[AN + VGL_CHUNK];
    const vgl_blk_unit u = units[blockIdx.x];
    const int32_t base = u.block << ABITS;
    const int n = min(AN, a_count - base);
    const acc_t ident = op.identity();
    for (int i = threadIdx.x; i < AN + VGL_CHUNK; i += VGL_BTHREADS) s_acc[i] = ident;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 3, off = (lane & 7) * 8;
    for (uint32_t j0 = u.chunk0 + wave * VGL_BGROUP; j0 < u.chunk1; j0 += VGL_BWAVES * VGL_BGROUP) {
        const uint32_t j = j0 + sub;
        if (j >= u.chunk1) continue;
        const size_t e = (size_t)j * VGL_CHUNK + off;
        const uint4 al = *reinterpret_cast<const uint4 *>(a_lo + e);
        const uint4 v0 = *reinterpret_cast<const uint4 *>(vals + e), v1 = *reinterpret_cast<const uint4 *>(vals + e + 4);
        const uint32_t a[4] = {al.x, al.y, al.z, al.w};
        const uint32_t v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            op.accumulate(&s_acc[a[k] & 0xFFFFu], v[2 * k]);
            op.accumulate(&s_acc[a[k] >> 16], v[2 * k + 1]);
        }
    }
    __syncthreads();
    if constexpr (OP::MARK) {
        // operators that keep a bitmap of the vertices they improved (the next frontier): one 64-bit word per wavefront step instead of a
        // store (or an atomic) per vertex -- the lanes of a wavefront handle 64 consecutive vertices of the word-aligned block
        const int nround = (n + 63) & ~63;
        for (int i = threadIdx.x; i < nround; i += VGL_BTHREADS) {
            bool ch = false;
            if (i < n) ch = u.slab < 0 ? op.finish_m(base + i, s_acc[i]) : op.partial_m(base + i, s_acc[i]);
            const unsigned long long m = __ballot(ch);
            if (m && lane == 0) op.mark(base + i, m, u.slab >= 0);
        }
        return;
    }
    if (u.slab < 0) {
        for (int i = threadIdx.x; i < n; i += VGL_BTHREADS) op.finish(base + i, s_acc[i]);
    } else {
        acc_t *slab = slabs + (size_t)u.slab * AN;
        for (int i = threadIdx.x; i < n; i += VGL_BTHREADS)
            if (!op.partial(base + i, s_acc[i])) slab[i] = s_acc[i];
    }
}

// fused tiles: x window of gather block gb stays in LDS for the whole unit; per dense pair (gb, ab) the accumulators of block ab are
// folded in LDS and handed to memory by OP::partial (several units may hold pairs of the same accumulate block: one global atomic per
// IMPROVED vertex, exactly what a multi-unit block of the two-pass scheme does).  Only min / max-type operators (partial() returning true).
template <class OP, bool WEIGHTED>
__global__ __launch_bounds__(VGL_BTHREADS) void vgl_k_blk_fused(const vgl_blk_funit *units, const vgl_blk_fseg *segs, const uint16_t *g_lo, const uint16_t *a_lo,
                                                                const float *w, int32_t g_count, int32_t a_count, OP op)
{
    __shared__ uint32_t s_x[VGL_FBLK];
    __shared__ uint32_t s_acc[VGL_FBLK + VGL_CHUNK];
    const vgl_blk_funit u = units[blockIdx.x];
    const int32_t gbase = u.gb << VGL_FBLK_BITS;
    const int gn = min(VGL_FBLK, g_count - gbase);
    for (int i = threadIdx.x; i < gn; i += VGL_BTHREADS) s_x[i] = op.load(gbase + i);
    const uint32_t ident = op.identity();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 3, off = (lane & 7) * 8;
    for (int32_t sg = u.seg0; sg < u.seg1; sg++) {
        const vgl_blk_fseg seg = segs[sg];
        for (int i = threadIdx.x; i < VGL_FBLK + VGL_CHUNK; i += VGL_BTHREADS) s_acc[i] = ident;
        __syncthreads();                                            // (also orders the x window before its first use)
        for (uint32_t m0 = seg.chunk0 + wave * VGL_BGROUP; m0 < seg.chunk1; m0 += VGL_BWAVES * VGL_BGROUP) {
            const uint32_t m = m0 + sub;
            if (m >= seg.chunk1) continue;
            const size_t e = (size_t)m * VGL_CHUNK + off;
            const uint4 gl = *reinterpret_cast<const uint4 *>(g_lo + e);
            const uint4 al = *reinterpret_cast<const uint4 *>(a_lo + e);
            float wv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (WEIGHTED) {
                const float4 w0 = *reinterpret_cast<const float4 *>(w + e), w1 = *reinterpret_cast<const float4 *>(w + e + 4);
                wv[0] = w0.x; wv[1] = w0.y; wv[2] = w0.z; wv[3] = w0.w; wv[4] = w1.x; wv[5] = w1.y; wv[6] = w1.z; wv[7] = w1.w;
            }
            const uint32_t g[4] = {gl.x, gl.y, gl.z, gl.w}, a[4] = {al.x, al.y, al.z, al.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                op.accumulate(&s_acc[a[k] & 0xFFFFu], op.edge(s_x[g[k] & 0xFFFFu], wv[2 * k]));
                op.accumulate(&s_acc[a[k] >> 16], op.edge(s_x[g[k] >> 16], wv[2 * k + 1]));
            }
        }
        __syncthreads();
        const int32_t abase = seg.ab << VGL_FBLK_BITS;
        const int an = min(VGL_FBLK, a_count - abase);
        if constexpr (OP::MARK) {
            const int anround = (an + 63) & ~63;
            for (int i = threadIdx.x; i < anround; i += VGL_BTHREADS) {
                bool ch = false;
                if (i < an) { const uint32_t acc = s_acc[i]; if (acc != ident) ch = op.partial_m(abase + i, acc); }
                const unsigned long long m = __ballot(ch);
                if (m && lane == 0) op.mark(abase + i, m, true);
            }
        } else {
            for (int i = threadIdx.x; i < an; i += VGL_BTHREADS) {
                const uint32_t acc = s_acc[i];
                if (acc != ident) op.partial(abase + i, acc);
            }
        }
        __syncthreads();
    }
}

// sum-type operators: add the slabs of a multi-unit block in unit order (OP::combine), then the epilogue
template <class OP>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_blk_finish_slabs(const vgl_blk_multi *multi, const typename OP::acc_t *slabs, int32_t a_count, OP op)
{
    typedef typename OP::acc_t acc_t;
    constexpr int ABITS = sizeof(acc_t) == 8 ? VGL_BLK_BITS - 1 : VGL_BLK_BITS, AN = 1 << ABITS;
    const vgl_blk_multi mb = multi[blockIdx.x];
    const int32_t base = mb.block << ABITS;
    const int n = min(AN, a_count - base);
    for (int i = blockIdx.y * VGL_BLOCK + threadIdx.x; i < n; i += gridDim.y * VGL_BLOCK) {
        acc_t acc = op.identity();
        for (int s = 0; s < mb.nslabs; s++) acc = op.combine(acc, slabs[(size_t)(mb.first_slab + s) * AN + i]);
        op.finish(base + i, acc);
    }
}

// one blocked pass: gather kernel, accumulate kernel and (sum-type operators) the slab epilogue, enqueued on the context stream
template <class OP, bool WEIGHTED, bool SLABS>
static inline int vgl_blocked_pass_one(vgl_hip_ctx *c, const vgl_blocked_plan *p, const OP &op, const char *gather_name, const char *accum_name,
                                       bool filtered, const char *fused_name)
{
    if (p->n_g_units > 0) {
        vgl_timed_launch tl(c, gather_name);
        hipLaunchKernelGGL((vgl_k_blk_gather<OP, WEIGHTED>), dim3((unsigned)p->n_g_units), dim3(VGL_BTHREADS), 0, c->stream, (const vgl_blk_unit *)p->g_units,
                           (const uint16_t *)p->g_lo, (const float *)p->w_mid, (const uint32_t *)p->mid_to_a, p->vals, p->g_count,
                           (const uint8_t *)(filtered ? p->g_dirty : nullptr), op);
    }
    if (p->a_bits != (sizeof(typename OP::acc_t) == 8 ? VGL_BLK_BITS - 1 : VGL_BLK_BITS)) VGL_FAIL("blocked pass: the plan's accumulate blocks do not fit the operator's accumulators");
    if (p->n_a_units > 0) {
        vgl_timed_launch tl(c, accum_name);
        hipLaunchKernelGGL((vgl_k_blk_accumulate<OP>), dim3((unsigned)p->n_a_units), dim3(VGL_BTHREADS), 0, c->stream, (const vgl_blk_unit *)p->a_units,
                           (const uint16_t *)p->a_lo, (const uint32_t *)p->vals, p->a_count, (typename OP::acc_t *)p->slabs, op);
    }
    if (p->n_f_units > 0) {
        // after the two-pass part: the fused tiles load their x windows now, so they already see what the accumulate kernel just improved
        // (any order of relaxations reaches the same fixed point; a later read only helps)
        if constexpr (SLABS) VGL_FAIL("blocked pass: fused tiles are for min / max-type operators");
        else {
            vgl_timed_launch tl(c, fused_name ? fused_name : accum_name);
            hipLaunchKernelGGL((vgl_k_blk_fused<OP, WEIGHTED>), dim3((unsigned)p->n_f_units), dim3(VGL_BTHREADS), 0, c->stream, (const vgl_blk_funit *)p->f_units,
                               (const vgl_blk_fseg *)p->f_segs, (const uint16_t *)p->f_g_lo, (const uint16_t *)p->f_a_lo, (const float *)p->f_w, p->g_count, p->a_count, op);
        }
    }
    if constexpr (SLABS) {
        if (p->n_multi > 0)
            hipLaunchKernelGGL((vgl_k_blk_finish_slabs<OP>), dim3((unsigned)p->n_multi, 16), dim3(VGL_BLOCK), 0, c->stream, (const vgl_blk_multi *)p->multi,
                               (const typename OP::acc_t *)p->slabs, p->a_count, op);
    }
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}
template <class OP, bool WEIGHTED, bool SLABS>
static inline int vgl_blocked_pass(vgl_hip_ctx *c, const vgl_blocked_plan *p, const OP &op, const char *gather_name, const char *accum_name,
                                   bool filtered = false, const char *fused_name = nullptr)
{
    if (SLABS && p->next) VGL_FAIL("blocked pass: a plan in several pieces needs a min / max-type operator");
    for (const vgl_blocked_plan *q = p; q; q = q->next) VGL_TRY((vgl_blocked_pass_one<OP, WEIGHTED, SLABS>(c, q, op, gather_name, accum_name, filtered, fused_name)));
    return 0;
}
#endif  // __HIPCC__
