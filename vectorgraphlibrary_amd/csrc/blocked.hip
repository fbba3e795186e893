// blocked.hip -- builds the layout of the blocked advance (vgl_blocked.h) from one CSR direction.  Offline, like the reference's
// graph import (vgl_graph.hpp:57-68): the edge keys (accumulate block, gather block) are sorted with rocPRIM's stable radix sort,
// segments are padded to 64-entry chunks and laid out twice (gather order / accumulate order).
#include "vgl_blocked.h"
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>

namespace {

struct dev_bufs {                                   // frees whatever the build still holds when it leaves, error or not
    hipStream_t st = nullptr;
    std::vector<void *> ptrs;
    template <class T> hipError_t alloc(T **p, size_t n)
    {
        hipError_t e = vgl_pool_alloc(st, (void **)p, sizeof(T) * std::max<size_t>(n, 1));
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
    void release(void *p)
    {
        for (auto &q : ptrs) if (q == p) { vgl_pool_free(st, q); q = nullptr; }
    }
    void free_all() { for (auto &q : ptrs) if (q) { vgl_pool_free(st, q); q = nullptr; } }
    ~dev_bufs() { free_all(); }
};

// edges per (gather block, accumulate block) pair of VGL_FBLK x VGL_FBLK ids: count16[gb16 * nA16 + ab16].  The rows of a 2048-edge tile
// lie in one row block almost always, so the tile counts its column blocks in LDS and adds the non-zero counters to memory once (a
// global atomic per edge would serialise on the hub pairs: millions of increments of one address).
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_blk_hist16(const int64_t *rowptr, const int32_t *adj, const int32_t *tile_row, int64_t E, int32_t row_base,
                                                              int gather_rows, int skip_self, uint32_t nA16, uint32_t ncol16, uint32_t *count16, uint32_t row_off)
{
    __shared__ int s_map[VGL_TILE];
    __shared__ int s_w[VGL_WAVES];
    __shared__ uint32_t s_h[8192];
    const int64_t e0 = (int64_t)blockIdx.x * VGL_TILE;
    const int n = (int)min((int64_t)VGL_TILE, E - e0);
    const int r_first = tile_row[blockIdx.x], r_last = tile_row[blockIdx.x + 1];
    for (uint32_t i = threadIdx.x; i < ncol16; i += VGL_BLOCK) s_h[i] = 0;
    vgl_tile_row_map(s_map, s_w, rowptr, e0, r_first, r_last);       // (starts with a barrier after its own stores: s_h is cleared for everybody)
    const uint32_t row16_tile = ((uint32_t)r_first + row_off) >> VGL_FBLK_BITS;
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) {
        const int i = threadIdx.x + j * VGL_BLOCK;
        if (i < n) {
            const uint32_t r = (uint32_t)(r_first + s_map[i]) + row_off, col = (uint32_t)adj[e0 + i];
            if (skip_self && (uint32_t)row_base + r == col) continue;
            const uint32_t row16 = r >> VGL_FBLK_BITS, col16 = col >> VGL_FBLK_BITS;
            if (row16 == row16_tile) atomicAdd(&s_h[col16], 1u);
            else atomicAdd(&count16[gather_rows ? row16 * nA16 + col16 : col16 * nA16 + row16], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ncol16; i += VGL_BLOCK) {
        const uint32_t h = s_h[i];
        if (h) atomicAdd(&count16[gather_rows ? row16_tile * nA16 + i : i * nA16 + row16_tile], h);
    }
}

// key = ab * nG + gb (sentinel nseg for dropped self loops), value = g_lo | a_lo << 16.  count16 (optional): pairs of 16384-id blocks with
// at least fuse_min edges become fused tiles: key = nseg + 1 + gb16 * nA16 + ab16, value = the ids inside the 16384-id blocks.
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_blk_keys(const int64_t *rowptr, const int32_t *adj, const int32_t *tile_row, int64_t E,
                                                            int32_t row_base, int gather_rows, int skip_self, uint32_t nG, uint32_t nseg,
                                                            int a_bits, uint32_t *keys, uint32_t *packed, const uint32_t *count16, uint32_t fuse_min,
                                                            uint32_t nA16, uint32_t row_off)
{
    __shared__ int s_map[VGL_TILE];
    __shared__ int s_w[VGL_WAVES];
    const int64_t e0 = (int64_t)blockIdx.x * VGL_TILE;
    const int n = (int)min((int64_t)VGL_TILE, E - e0);
    const int r_first = tile_row[blockIdx.x], r_last = tile_row[blockIdx.x + 1];
    vgl_tile_row_map(s_map, s_w, rowptr, e0, r_first, r_last);
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) {
        const int i = threadIdx.x + j * VGL_BLOCK;
        if (i < n) {
            const uint32_t r = (uint32_t)(r_first + s_map[i]) + row_off, col = (uint32_t)adj[e0 + i];        // (row_off: a piece's rows keep their global numbers)
            const uint32_t g = gather_rows ? r : col, a = gather_rows ? col : r;
            const bool drop = skip_self && (uint32_t)row_base + r == col;
            uint32_t key = drop ? nseg : (a >> a_bits) * nG + (g >> VGL_BLK_BITS);
            uint32_t pk = (g & (VGL_BLK - 1)) | ((a & ((1u << a_bits) - 1)) << 16);
            if (count16 && !drop) {
                const uint32_t f = (g >> VGL_FBLK_BITS) * nA16 + (a >> VGL_FBLK_BITS);
                if (count16[f] >= fuse_min) { key = nseg + 1 + f; pk = (g & (VGL_FBLK - 1)) | ((a & (VGL_FBLK - 1)) << 16); }
            }
            keys[e0 + i] = key;
            packed[e0 + i] = pk;
        }
    }
}

// first / one-past-last position of every key present in the sorted key array (absent keys keep 0 / 0)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_blk_runs(int64_t E, const uint32_t *keys, uint32_t *seg_first, uint32_t *seg_end)
{
    for (int64_t i = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; i < E; i += (int64_t)gridDim.x * VGL_BLOCK) {
        const uint32_t k = keys[i];
        if (i == 0 || keys[i - 1] != k) seg_first[k] = (uint32_t)i;
        if (i == E - 1 || keys[i + 1] != k) seg_end[k] = (uint32_t)(i + 1);
    }
}

// chunks per segment in both orders (index nseg = 0: the scans below then leave the totals there)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_blk_nchunks(uint32_t nG, uint32_t nA, const uint32_t *seg_first, const uint32_t *seg_end,
                                                               uint32_t *nch_a, uint32_t *nch_m)
{
    const uint32_t nseg = nG * nA;
    for (uint32_t k = blockIdx.x * VGL_BLOCK + threadIdx.x; k <= nseg; k += gridDim.x * VGL_BLOCK) {
        if (k == nseg) { nch_a[k] = 0; nch_m[k] = 0; continue; }
        const uint32_t n = (seg_end[k] - seg_first[k] + VGL_CHUNK - 1) / VGL_CHUNK;
        const uint32_t ab = k / nG, gb = k % nG;
        nch_a[k] = n;
        nch_m[gb * nA + ab] = n;
    }
}

// one wavefront per A-order chunk: find its segment, place the chunk in both orders
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_blk_fill(uint32_t nchunks, uint32_t nG, uint32_t nA, const uint32_t *a_start, const uint32_t *m_start,
                                                            const uint32_t *seg_first, const uint32_t *seg_end, const uint32_t *packed_sorted,
                                                            const uint32_t *idx_sorted, int a_bits, uint16_t *g_lo, uint16_t *a_lo, uint32_t *w_src_mid, uint32_t *mid_to_a)
{
    const uint32_t nseg = nG * nA;
    const int lane = threadIdx.x & 63;
    for (uint32_t j = blockIdx.x * VGL_WAVES + (threadIdx.x >> 6); j < nchunks; j += gridDim.x * VGL_WAVES) {
        uint32_t lo = 0, hi = nseg;                                 // last k with a_start[k] <= j (a_start[nseg] = nchunks > j)
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (a_start[mid] <= j) lo = mid; else hi = mid;
        }
        // empty segments share their start with the next one: upper_bound - 1 lands on the LAST of equal starts, which is the non-empty one
        const uint32_t k = lo, ab = k / nG, gb = k % nG;
        const uint32_t within = j - a_start[k];
        const uint32_t m = m_start[gb * nA + ab] + within;
        const uint32_t pos = within * VGL_CHUNK + lane, cnt = seg_end[k] - seg_first[k];
        uint16_t gl = 0, al = (uint16_t)((1u << a_bits) + lane);
        uint32_t from = 0xFFFFFFFFu;                               // pad entry: no edge behind it
        if (pos < cnt) {
            const uint32_t pk = packed_sorted[seg_first[k] + pos];
            gl = (uint16_t)(pk & 0xFFFFu);
            al = (uint16_t)(pk >> 16);
            if (idx_sorted) from = idx_sorted[seg_first[k] + pos];
        }
        g_lo[(size_t)m * VGL_CHUNK + lane] = gl;
        a_lo[(size_t)j * VGL_CHUNK + lane] = al;
        if (w_src_mid) w_src_mid[(size_t)m * VGL_CHUNK + lane] = from;
        if (lane == 0) mid_to_a[m] = j;
    }
}

// fused tiles: one wavefront per chunk; seg_chunk0[s] = first chunk of dense pair s (ascending, seg_chunk0[nsegs] = nchunks), seg_key[s] = its sort key
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_blk_fill_fused(uint32_t nchunks, uint32_t nsegs, const uint32_t *seg_chunk0, const uint32_t *seg_key,
                                                                  const uint32_t *seg_first, const uint32_t *seg_end, const uint32_t *packed_sorted,
                                                                  const uint32_t *idx_sorted, uint16_t *g_lo, uint16_t *a_lo, uint32_t *w_src)
{
    const int lane = threadIdx.x & 63;
    for (uint32_t j = blockIdx.x * VGL_WAVES + (threadIdx.x >> 6); j < nchunks; j += gridDim.x * VGL_WAVES) {
        uint32_t lo = 0, hi = nsegs;                                // last s with seg_chunk0[s] <= j (every dense pair has at least one chunk)
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (seg_chunk0[mid] <= j) lo = mid; else hi = mid;
        }
        const uint32_t k = seg_key[lo];
        const uint32_t pos = (j - seg_chunk0[lo]) * VGL_CHUNK + lane, cnt = seg_end[k] - seg_first[k];
        uint16_t gl = 0, al = (uint16_t)(VGL_FBLK + lane);         // pad entries: x[0] folded into a dummy accumulator behind the window
        uint32_t from = 0xFFFFFFFFu;
        if (pos < cnt) {
            const uint32_t pk = packed_sorted[seg_first[k] + pos];
            gl = (uint16_t)(pk & 0xFFFFu);
            al = (uint16_t)(pk >> 16);
            if (idx_sorted) from = idx_sorted[seg_first[k] + pos];
        }
        g_lo[(size_t)j * VGL_CHUNK + lane] = gl;
        a_lo[(size_t)j * VGL_CHUNK + lane] = al;
        if (w_src) w_src[(size_t)j * VGL_CHUNK + lane] = from;
    }
}

// values of a layout from the CSR-order array they come from: out[slot] = weights[w_src[slot]] (0 behind pad entries); 16 bytes per lane and step
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_blk_load_weights(size_t slots, const uint32_t *w_src, const float *weights, float *out)
{
    const size_t quads = slots / 4;                                 // (slots is a multiple of VGL_CHUNK = 64)
    for (size_t q = (size_t)blockIdx.x * VGL_BLOCK + threadIdx.x; q < quads; q += (size_t)gridDim.x * VGL_BLOCK) {
        const uint4 from = reinterpret_cast<const uint4 *>(w_src)[q];
        float4 w;
        w.x = weights[from.x == 0xFFFFFFFFu ? 0 : from.x]; w.y = weights[from.y == 0xFFFFFFFFu ? 0 : from.y];      // (unconditional loads, issued together)
        w.z = weights[from.z == 0xFFFFFFFFu ? 0 : from.z]; w.w = weights[from.w == 0xFFFFFFFFu ? 0 : from.w];
        if (from.x == 0xFFFFFFFFu) w.x = 0.0f;
        if (from.y == 0xFFFFFFFFu) w.y = 0.0f;
        if (from.z == 0xFFFFFFFFu) w.z = 0.0f;
        if (from.w == 0xFFFFFFFFu) w.w = 0.0f;
        reinterpret_cast<float4 *>(out)[q] = w;
    }
}

__global__ void vgl_k_blk_pick(int n, uint32_t stride, const uint32_t *in, uint32_t *out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = in[(size_t)i * stride];
}

// VGL_BLK_BUILD_TRACE=1: wall time of every stage of the build on stderr (each mark waits for the stream)
struct stage_trace {
    bool on;
    hipStream_t st;
    std::chrono::steady_clock::time_point t;
    stage_trace(vgl_hip_ctx *c, hipStream_t s) : on(vgl_env(c, "VGL_BLK_BUILD_TRACE") != nullptr), st(s), t(std::chrono::steady_clock::now()) {}
    void mark(const char *what)
    {
        if (!on) return;
        hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[vgl blocked plan] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

int env_int(vgl_hip_ctx *c, const char *name, int dflt)
{
    const char *s = vgl_env(c, name);
    return (s && *s) ? atoi(s) : dflt;
}

// cut blocks into units of at most `cap` chunks; bounds[b] .. bounds[b+1] = chunk range of block b
void make_units(const std::vector<uint32_t> &bounds, uint32_t cap, bool keep_empty, bool want_slabs, std::vector<vgl_blk_unit> &units,
                std::vector<vgl_blk_multi> &multi, int &n_slabs)
{
    n_slabs = 0;
    for (size_t b = 0; b + 1 < bounds.size(); b++) {
        const uint32_t c0 = bounds[b], c1 = bounds[b + 1], n = c1 - c0;
        if (n == 0 && !keep_empty) continue;
        const uint32_t want = std::max<uint32_t>(1, (n + cap - 1) / cap);
        const uint32_t step = std::max<uint32_t>(1, (n + want - 1) / want);
        const uint32_t parts = std::max<uint32_t>(1, (n + step - 1) / step);        // no empty parts: (parts - 1) * step < n
        if (parts > 1 && want_slabs) multi.push_back(vgl_blk_multi{(int32_t)b, n_slabs, (int32_t)parts, 0});
        for (uint32_t p = 0; p < parts; p++) {
            const uint32_t lo = c0 + std::min(n, p * step), hi = c0 + std::min(n, (p + 1) * step);
            units.push_back(vgl_blk_unit{(int32_t)b, parts > 1 ? (want_slabs ? n_slabs++ : 0) : -1, lo, hi});
        }
    }
    // long units first: the tail of the launch is then made of short ones
    std::stable_sort(units.begin(), units.end(), [](const vgl_blk_unit &x, const vgl_blk_unit &y) { return (int64_t)x.chunk1 - (int64_t)x.chunk0 > (int64_t)y.chunk1 - (int64_t)y.chunk0; });
}

}  // namespace

void vgl_blocked_plan_destroy(vgl_blocked_plan *p)
{
    if (!p) return;
    if (p->next) { vgl_blocked_plan_destroy(p->next); p->next = nullptr; }
    if (p->shared) {                                                // a second set of value arrays over somebody else's structure
        vgl_pool_free(p->stream, p->w_mid); vgl_pool_free(p->stream, p->f_w); vgl_pool_free(p->stream, p->g_dirty);
        vgl_blocked_plan *s = p->shared_from;
        delete p;
        if (s && --s->sharers == 0 && s->orphan) { s->orphan = false; vgl_blocked_plan_destroy(s); }      // its owner is gone already
        return;
    }
    if (p->sharers > 0) { p->orphan = true; return; }               // still in use by plans that share it: the last of them frees it
    if (p->piece_rowptr) hipFree(p->piece_rowptr);
    if (p->piece_tile_row) hipFree(p->piece_tile_row);
    void *ptrs[] = {p->g_lo, p->a_lo, p->w_mid, p->mid_to_a, p->vals, p->g_units, p->a_units, p->multi, p->slabs, p->g_dirty,
                    p->f_g_lo, p->f_a_lo, p->f_w, p->f_segs, p->f_units, p->w_src_mid, p->w_src_f};
    for (void *q : ptrs) vgl_pool_free(p->stream, q);
    delete p;
}

// one plan over the rows of `dir` (all rows of the direction, or a row-range piece of it: row_off = number of rows before the piece,
// nrows_total = rows of the whole direction)
static int vgl_blocked_plan_build_one(vgl_hip_ctx *c, const vgl_dir_csr &dir, int32_t nrows, int32_t row_base, int32_t ncols, int gather_rows,
                                      int skip_self, const float *d_weights, int a_bits, vgl_blocked_plan **out, int value_bits, int fuse_min_edges,
                                      int32_t row_off, int32_t nrows_total, int keep_edge_index, int64_t w_base)
{
    if (d_weights) keep_edge_index = 1;
    if (value_bits != 32 && value_bits != 1) VGL_FAIL("blocked_plan_build: values are 32 bits or 1 bit per edge");
    if (!c || !out) VGL_FAIL("blocked_plan_build: null argument");
    if (a_bits != VGL_BLK_BITS && a_bits != VGL_BLK_BITS - 1) VGL_FAIL("blocked_plan_build: accumulate blocks hold 2^15 (4-byte) or 2^14 (8-byte) accumulators");
    if (dir.edges > 0 && (!dir.rowptr || !dir.adj || !dir.tile_row)) VGL_FAIL("blocked_plan_build: CSR direction is missing");
    if (dir.edges >= (1LL << 32) - VGL_TILE) VGL_FAIL("blocked_plan_build: at most 2^32 edges per plan");
    hipStream_t st = c->stream;
    const int64_t E = dir.edges;
    vgl_blocked_plan *p = new vgl_blocked_plan();
    struct guard { vgl_blocked_plan *p; ~guard() { if (p) vgl_blocked_plan_destroy(p); } } own{p};
    (void)nrows;
    p->g_count = gather_rows ? nrows_total : ncols;
    p->a_count = gather_rows ? ncols : nrows_total;
    p->nG = (int32_t)std::max<int64_t>(1, vgl_ceil_div(p->g_count, VGL_BLK));
    p->a_bits = a_bits;
    p->nA = (int32_t)std::max<int64_t>(1, vgl_ceil_div(p->a_count, 1 << a_bits));
    const uint32_t nG = (uint32_t)p->nG, nA = (uint32_t)p->nA, nseg = nG * nA;
    if ((int64_t)nG * nA >= (1LL << 31)) VGL_FAIL("blocked_plan_build: too many block pairs");
    // fused tiles: pairs of 16384-id blocks; their keys follow the two-pass keys and the sentinel
    const uint32_t nG16 = (uint32_t)std::max<int64_t>(1, vgl_ceil_div(p->g_count, VGL_FBLK)), nA16 = (uint32_t)std::max<int64_t>(1, vgl_ceil_div(p->a_count, VGL_FBLK));
    const uint32_t ncol16 = gather_rows ? nA16 : nG16;
    const bool fuse = fuse_min_edges > 0 && value_bits == 32 && a_bits == VGL_BLK_BITS && dir.edges > 0 && ncol16 <= 8192 &&
                      (int64_t)nG16 * nA16 + nseg + 1 < (1LL << 31);
    const uint32_t nseg16 = fuse ? nG16 * nA16 : 0, nkeys = nseg + 1 + nseg16;

    dev_bufs tmp;
    tmp.st = st;
    p->stream = st;
    stage_trace trace(c, st);
    std::unique_ptr<vgl_timed_launch> timed;
    uint32_t *keys = nullptr, *keys2 = nullptr, *packed = nullptr, *packed2 = nullptr, *seg_first = nullptr, *seg_end = nullptr;
    uint32_t *nch_a = nullptr, *nch_m = nullptr, *a_start = nullptr, *m_start = nullptr, *picked = nullptr;
    uint32_t *idx2 = nullptr;                                       // CSR positions in sorted order (layouts with edge values)
    void *sort_tmp = nullptr;
    uint32_t *count16 = nullptr;
    VGL_HIP_TRY(tmp.alloc(&seg_first, (size_t)nkeys + 1));
    VGL_HIP_TRY(tmp.alloc(&seg_end, (size_t)nkeys + 1));
    VGL_HIP_TRY(tmp.alloc(&nch_a, (size_t)nseg + 1));
    VGL_HIP_TRY(tmp.alloc(&nch_m, (size_t)nseg + 1));
    VGL_HIP_TRY(tmp.alloc(&a_start, (size_t)nseg + 1));
    VGL_HIP_TRY(tmp.alloc(&m_start, (size_t)nseg + 1));
    VGL_HIP_TRY(tmp.alloc(&picked, (size_t)nG + nA + 2));
    VGL_HIP_TRY(hipMemsetAsync(seg_first, 0, sizeof(uint32_t) * ((size_t)nkeys + 1), st));
    VGL_HIP_TRY(hipMemsetAsync(seg_end, 0, sizeof(uint32_t) * ((size_t)nkeys + 1), st));
    if (E > 0) {
        VGL_HIP_TRY(tmp.alloc(&keys, (size_t)E));
        VGL_HIP_TRY(tmp.alloc(&keys2, (size_t)E));
        VGL_HIP_TRY(tmp.alloc(&packed, (size_t)E));
        VGL_HIP_TRY(tmp.alloc(&packed2, (size_t)E));
        trace.mark("allocate keys");
        // (timing on: the stream time from here to the fill kernel is booked under "blk_plan_build" -- what the build costs the GPU,
        // without the allocator, which can stall for seconds right after tens of GB were freed)
        if (fuse) {
            VGL_HIP_TRY(tmp.alloc(&count16, (size_t)nseg16));
            VGL_HIP_TRY(hipMemsetAsync(count16, 0, sizeof(uint32_t) * (size_t)nseg16, st));
        }
        timed.reset(new vgl_timed_launch(c, "blk_plan_build"));
        if (fuse) {
            hipLaunchKernelGGL(vgl_k_blk_hist16, dim3((unsigned)dir.ntiles), dim3(VGL_BLOCK), 0, st, dir.rowptr, dir.adj, (const int32_t *)dir.tile_row, E, row_base,
                               gather_rows, skip_self, nA16, ncol16, count16, (uint32_t)row_off);
            VGL_HIP_TRY(hipGetLastError());
            trace.mark("pair histogram (fused tiles)");
        }
        hipLaunchKernelGGL(vgl_k_blk_keys, dim3((unsigned)dir.ntiles), dim3(VGL_BLOCK), 0, st, dir.rowptr, dir.adj, (const int32_t *)dir.tile_row, E,
                           row_base, gather_rows, skip_self, nG, nseg, a_bits, keys, packed, (const uint32_t *)count16, (uint32_t)std::max(fuse_min_edges, 1), nA16, (uint32_t)row_off);
        VGL_HIP_TRY(hipGetLastError());
        trace.mark("keys kernel");
        int bits = 1;
        while ((1ull << bits) <= (unsigned long long)nkeys) bits++;      // every key, the sentinel nseg and the fused range behind it, must be representable
        size_t need = 0;
        VGL_HIP_TRY(rocprim::radix_sort_pairs(nullptr, need, keys, keys2, packed, packed2, (size_t)E, 0, bits, st));
        VGL_HIP_TRY(tmp.alloc((char **)&sort_tmp, std::max<size_t>(need, 16)));
        VGL_HIP_TRY(rocprim::radix_sort_pairs(sort_tmp, need, keys, keys2, packed, packed2, (size_t)E, 0, bits, st));
        trace.mark("sort (block pair, packed ids)");
        if (keep_edge_index) {                                      // same keys, same stable sort: the CSR positions land in the order of the entries
            VGL_HIP_TRY(hipStreamSynchronize(st));
            tmp.release(packed);
            packed = nullptr;
            VGL_HIP_TRY(tmp.alloc(&idx2, (size_t)E));
            size_t need2 = 0;
            uint32_t *keys3 = nullptr;
            VGL_HIP_TRY(tmp.alloc(&keys3, (size_t)E));
            rocprim::counting_iterator<uint32_t> positions(0);
            VGL_HIP_TRY(rocprim::radix_sort_pairs(nullptr, need2, keys, keys3, positions, idx2, (size_t)E, 0, bits, st));
            if (need2 > need) VGL_FAIL("blocked_plan_build: radix sort scratch grew between two calls of the same size");
            VGL_HIP_TRY(rocprim::radix_sort_pairs(sort_tmp, need2, keys, keys3, positions, idx2, (size_t)E, 0, bits, st));
            VGL_HIP_TRY(hipStreamSynchronize(st));
            tmp.release(keys3);
            trace.mark("sort edge positions");
        }
        hipLaunchKernelGGL(vgl_k_blk_runs, dim3((unsigned)std::min<int64_t>(16384, vgl_ceil_div(E, VGL_BLOCK))), dim3(VGL_BLOCK), 0, st, E,
                           (const uint32_t *)keys2, seg_first, seg_end);
        VGL_HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(vgl_k_blk_nchunks, dim3((unsigned)std::min<int64_t>(4096, vgl_ceil_div((int64_t)nseg + 1, VGL_BLOCK))), dim3(VGL_BLOCK), 0, st, nG, nA,
                       (const uint32_t *)seg_first, (const uint32_t *)seg_end, nch_a, nch_m);
    VGL_HIP_TRY(hipGetLastError());
    {
        size_t need = 0;
        void *scan_tmp = nullptr;
        VGL_HIP_TRY(rocprim::exclusive_scan(nullptr, need, nch_a, a_start, 0u, (size_t)nseg + 1, rocprim::plus<uint32_t>(), st));
        VGL_HIP_TRY(tmp.alloc((char **)&scan_tmp, std::max<size_t>(need, 16)));
        VGL_HIP_TRY(rocprim::exclusive_scan(scan_tmp, need, nch_a, a_start, 0u, (size_t)nseg + 1, rocprim::plus<uint32_t>(), st));
        VGL_HIP_TRY(rocprim::exclusive_scan(scan_tmp, need, nch_m, m_start, 0u, (size_t)nseg + 1, rocprim::plus<uint32_t>(), st));
    }
    trace.mark("runs, chunk counts, scans");
    // chunk ranges of the blocks on both sides -> host
    hipLaunchKernelGGL(vgl_k_blk_pick, dim3(8), dim3(256), 0, st, (int)nA + 1, nG, (const uint32_t *)a_start, picked);
    hipLaunchKernelGGL(vgl_k_blk_pick, dim3(8), dim3(256), 0, st, (int)nG + 1, nA, (const uint32_t *)m_start, picked + nA + 1);
    VGL_HIP_TRY(hipGetLastError());
    std::vector<uint32_t> h((size_t)nA + nG + 2);
    VGL_TRY(vgl_hip_memcpy_d2h(c, h.data(), picked, sizeof(uint32_t) * h.size()));
    std::vector<uint32_t> a_bounds(h.begin(), h.begin() + nA + 1), g_bounds(h.begin() + nA + 1, h.end());
    p->nchunks = a_bounds[nA];
    if (g_bounds[nG] != p->nchunks) VGL_FAIL("blocked_plan_build: the two chunk orders disagree");
    if (E > 0) {
        uint32_t kept_end = 0;                                      // edges kept = end of the last real segment's run = start of the sentinel run
        std::vector<uint32_t> sent(2);
        VGL_TRY(vgl_hip_memcpy_d2h(c, &sent[0], seg_first + nseg, sizeof(uint32_t)));
        VGL_TRY(vgl_hip_memcpy_d2h(c, &sent[1], seg_end + nseg, sizeof(uint32_t)));
        kept_end = (uint32_t)(E - (int64_t)(sent[1] > sent[0] ? sent[1] - sent[0] : 0));     // all but the dropped self loops (their run sits between the two key ranges)
        p->edges = kept_end;
    }
    const size_t slots = (size_t)p->nchunks * VGL_CHUNK;
    VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&p->g_lo, sizeof(uint16_t) * std::max<size_t>(slots, 8)));
    VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&p->a_lo, sizeof(uint16_t) * std::max<size_t>(slots, 8)));
    VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&p->vals, value_bits == 1 ? sizeof(uint64_t) * std::max<size_t>(p->nchunks, 1) : sizeof(uint32_t) * std::max<size_t>(slots, 8)));
    VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&p->mid_to_a, sizeof(uint32_t) * std::max<size_t>(p->nchunks, 1)));
    if (keep_edge_index) {
        VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&p->w_mid, sizeof(float) * std::max<size_t>(slots, 8)));
        VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&p->w_src_mid, sizeof(uint32_t) * std::max<size_t>(slots, 8)));
    }
    p->w_base = w_base;
    VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&p->g_dirty, (size_t)nG));
    VGL_HIP_TRY(hipMemsetAsync(p->g_dirty, 1, (size_t)nG, st));
    trace.mark("allocate plan arrays");
    if (p->nchunks > 0) {
        hipLaunchKernelGGL(vgl_k_blk_fill, dim3((unsigned)std::min<int64_t>(65536, vgl_ceil_div(p->nchunks, VGL_WAVES))), dim3(VGL_BLOCK), 0, st, p->nchunks, nG, nA,
                           (const uint32_t *)a_start, (const uint32_t *)m_start, (const uint32_t *)seg_first, (const uint32_t *)seg_end,
                           (const uint32_t *)packed2, (const uint32_t *)idx2, a_bits, p->g_lo, p->a_lo, p->w_src_mid, p->mid_to_a);
        VGL_HIP_TRY(hipGetLastError());
    }
    trace.mark("fill kernel");
    if (fuse) {
        // dense pairs in key order (gather block major), cut into pieces of at most f_cap chunks (a hub x hub pair holds millions of edges:
        // one workgroup must not be left alone with it), pieces grouped into units per gather block
        std::vector<uint32_t> cnt((size_t)nseg16);
        VGL_TRY(vgl_hip_memcpy_d2h(c, cnt.data(), count16, sizeof(uint32_t) * cnt.size()));
        const uint32_t f_cap = (uint32_t)std::max(64, env_int(c, "VGL_BLK_FUSED_UNIT", 4096));
        std::vector<uint32_t> seg_chunk0, seg_key;
        std::vector<vgl_blk_fseg> fsegs;
        std::vector<vgl_blk_funit> funits;
        uint32_t chunk = 0;
        for (uint32_t gb = 0; gb < nG16; gb++) {
            uint32_t in_unit = 0;
            int32_t unit_seg0 = (int32_t)fsegs.size();
            for (uint32_t ab = 0; ab < nA16; ab++) {
                const uint32_t n_e = cnt[(size_t)gb * nA16 + ab];
                if (n_e < (uint32_t)fuse_min_edges) continue;
                const uint32_t n_ch = (n_e + VGL_CHUNK - 1) / VGL_CHUNK;
                seg_chunk0.push_back(chunk); seg_key.push_back(nseg + 1 + gb * nA16 + ab);
                p->f_edges += n_e;
                for (uint32_t c0 = 0; c0 < n_ch; c0 += f_cap) {
                    const uint32_t c1 = std::min(n_ch, c0 + f_cap);
                    if (in_unit > 0 && in_unit + (c1 - c0) > f_cap) {
                        funits.push_back(vgl_blk_funit{(int32_t)gb, unit_seg0, (int32_t)fsegs.size(), 0});
                        unit_seg0 = (int32_t)fsegs.size(); in_unit = 0;
                    }
                    fsegs.push_back(vgl_blk_fseg{(int32_t)ab, chunk + c0, chunk + c1});
                    in_unit += c1 - c0;
                }
                chunk += n_ch;
            }
            if ((int32_t)fsegs.size() > unit_seg0) funits.push_back(vgl_blk_funit{(int32_t)gb, unit_seg0, (int32_t)fsegs.size(), 0});
        }
        p->f_nchunks = chunk;
        p->n_f_segs = (int)fsegs.size();
        if (chunk > 0) {
            auto unit_chunks = [&](const vgl_blk_funit &u) { uint32_t t = 0; for (int32_t i = u.seg0; i < u.seg1; i++) t += fsegs[(size_t)i].chunk1 - fsegs[(size_t)i].chunk0; return t; };
            std::stable_sort(funits.begin(), funits.end(), [&](const vgl_blk_funit &x, const vgl_blk_funit &y) { return unit_chunks(x) > unit_chunks(y); });
            p->n_f_units = (int)funits.size();
            seg_chunk0.push_back(chunk);
            uint32_t *d_chunk0 = nullptr, *d_key = nullptr;
            VGL_HIP_TRY(tmp.alloc(&d_chunk0, seg_chunk0.size()));
            VGL_HIP_TRY(tmp.alloc(&d_key, seg_key.size()));
            VGL_TRY(vgl_hip_memcpy_h2d(c, d_chunk0, seg_chunk0.data(), sizeof(uint32_t) * seg_chunk0.size()));
            VGL_TRY(vgl_hip_memcpy_h2d(c, d_key, seg_key.data(), sizeof(uint32_t) * seg_key.size()));
            const size_t fslots = (size_t)chunk * VGL_CHUNK;
            VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&p->f_g_lo, sizeof(uint16_t) * fslots));
            VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&p->f_a_lo, sizeof(uint16_t) * fslots));
            if (keep_edge_index) {
                VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&p->f_w, sizeof(float) * fslots));
                VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&p->w_src_f, sizeof(uint32_t) * fslots));
            }
            VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&p->f_segs, sizeof(vgl_blk_fseg) * fsegs.size()));
            VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&p->f_units, sizeof(vgl_blk_funit) * funits.size()));
            VGL_TRY(vgl_hip_memcpy_h2d(c, p->f_segs, fsegs.data(), sizeof(vgl_blk_fseg) * fsegs.size()));
            VGL_TRY(vgl_hip_memcpy_h2d(c, p->f_units, funits.data(), sizeof(vgl_blk_funit) * funits.size()));
            hipLaunchKernelGGL(vgl_k_blk_fill_fused, dim3((unsigned)std::min<int64_t>(65536, vgl_ceil_div(chunk, VGL_WAVES))), dim3(VGL_BLOCK), 0, st, chunk,
                               (uint32_t)seg_key.size(), (const uint32_t *)d_chunk0, (const uint32_t *)d_key, (const uint32_t *)seg_first, (const uint32_t *)seg_end,
                               (const uint32_t *)packed2, (const uint32_t *)idx2, p->f_g_lo, p->f_a_lo, p->w_src_f);
            VGL_HIP_TRY(hipGetLastError());
        }
        trace.mark("fused tiles (tables + fill)");
        if (trace.on)
            fprintf(stderr, "[vgl blocked plan] fused tiles: %lld of %lld edges in %zu pairs (%d pieces, %d units, %u chunks); two-pass chunks %u\n", (long long)p->f_edges,
                    (long long)p->edges, seg_key.size(), p->n_f_segs, p->n_f_units, p->f_nchunks, p->nchunks);
    }
    timed.reset();

    // work units.  Gather units: >= 4 per CU when the graph allows (each reloads its 128 KiB window, so not below ~256 K edges);
    // accumulate units larger (a block cut in several units costs a slab or a round of global atomics per unit)
    // (accumulate blocks up to 1.5x the average stay whole -- on a uniform graph every block is one unit and nothing goes through slabs --
    // but never beyond 16 K chunks = 1 M entries, ~0.25 ms of one CU's share of the HBM stream)
    const uint32_t avg_a = (uint32_t)(p->nchunks / nA);
    const uint32_t g_cap = (uint32_t)std::max(64, env_int(c, "VGL_BLK_GATHER_UNIT", 4096));
    const uint32_t a_cap = (uint32_t)std::max(64, env_int(c, "VGL_BLK_ACCUM_UNIT", (int)std::min<uint32_t>(16384, std::max<uint32_t>(4096, avg_a + avg_a / 2))));
    std::vector<vgl_blk_unit> gu, au;
    std::vector<vgl_blk_multi> multi, none;
    int dummy = 0;
    make_units(g_bounds, g_cap, false, false, gu, none, dummy);
    make_units(a_bounds, a_cap, true, true, au, multi, p->n_slabs);
    p->n_g_units = (int)gu.size(); p->n_a_units = (int)au.size(); p->n_multi = (int)multi.size();
    VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&p->g_units, sizeof(vgl_blk_unit) * std::max<size_t>(gu.size(), 1)));
    VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&p->a_units, sizeof(vgl_blk_unit) * std::max<size_t>(au.size(), 1)));
    VGL_HIP_TRY(vgl_pool_alloc(st, (void **)&p->multi, sizeof(vgl_blk_multi) * std::max<size_t>(multi.size(), 1)));
    if (!gu.empty()) VGL_TRY(vgl_hip_memcpy_h2d(c, p->g_units, gu.data(), sizeof(vgl_blk_unit) * gu.size()));
    if (!au.empty()) VGL_TRY(vgl_hip_memcpy_h2d(c, p->a_units, au.data(), sizeof(vgl_blk_unit) * au.size()));
    if (!multi.empty()) VGL_TRY(vgl_hip_memcpy_h2d(c, p->multi, multi.data(), sizeof(vgl_blk_multi) * multi.size()));
    VGL_HIP_TRY(vgl_pool_alloc(st, &p->slabs, sizeof(uint32_t) * VGL_BLK * (size_t)std::max(p->n_slabs, 1)));
    VGL_HIP_TRY(hipStreamSynchronize(st));
    trace.mark("work units");
    tmp.free_all();
    trace.mark("free temporaries");
    if (d_weights) {
        p->next = nullptr;
        VGL_TRY(vgl_blocked_plan_load_weights(c, p, d_weights - w_base));       // (load_weights takes the direction's array and adds the piece's base itself)
        VGL_HIP_TRY(hipStreamSynchronize(st));
        trace.mark("load weights");
    }
    own.p = nullptr;
    *out = p;
    return 0;
}

__global__ void vgl_k_blk_rebase_rows(int32_t n, const int64_t *rowptr, int64_t base, int64_t *out)
{
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) out[i] = rowptr[i] - base;
}

int vgl_blocked_plan_load_weights(vgl_hip_ctx *c, vgl_blocked_plan *p, const float *d_weights)
{
    if (!c || !p || !d_weights) VGL_FAIL("blocked_plan_load_weights: null argument");
    for (vgl_blocked_plan *q = p; q; q = q->next) {
        if ((q->nchunks > 0 && (!q->w_src_mid || !q->w_mid)) || (q->f_nchunks > 0 && (!q->w_src_f || !q->f_w)))
            VGL_FAIL("blocked_plan_load_weights: the layout was built without its edge index");
        vgl_timed_launch tl(c, "blk_load_weights");
        const size_t slots = (size_t)q->nchunks * VGL_CHUNK, fslots = (size_t)q->f_nchunks * VGL_CHUNK;
        if (slots) hipLaunchKernelGGL(vgl_k_blk_load_weights, dim3((unsigned)std::min<size_t>(16384, (slots / 4 + VGL_BLOCK - 1) / VGL_BLOCK)), dim3(VGL_BLOCK), 0, c->stream,
                                      slots, (const uint32_t *)q->w_src_mid, d_weights + q->w_base, q->w_mid);
        if (fslots) hipLaunchKernelGGL(vgl_k_blk_load_weights, dim3((unsigned)std::min<size_t>(16384, (fslots / 4 + VGL_BLOCK - 1) / VGL_BLOCK)), dim3(VGL_BLOCK), 0, c->stream,
                                       fslots, (const uint32_t *)q->w_src_f, d_weights + q->w_base, q->f_w);
        VGL_HIP_TRY(hipGetLastError());
    }
    return 0;
}

int vgl_blocked_plan_share(vgl_hip_ctx *c, const vgl_blocked_plan *structure, vgl_blocked_plan **out)
{
    if (!c || !structure || !out) VGL_FAIL("blocked_plan_share: null argument");
    vgl_blocked_plan *head = nullptr, *tail = nullptr;
    struct guard { vgl_blocked_plan **h; ~guard() { if (*h) vgl_blocked_plan_destroy(*h); } } own{&head};
    for (const vgl_blocked_plan *s = structure; s; s = s->next) {
        if ((s->nchunks > 0 && !s->w_src_mid) || (s->f_nchunks > 0 && !s->w_src_f)) VGL_FAIL("blocked_plan_share: the layout was built without its edge index");
        vgl_blocked_plan *q = new vgl_blocked_plan(*s);
        q->shared = true; q->next = nullptr; q->w_mid = nullptr; q->f_w = nullptr; q->g_dirty = nullptr;
        q->stream = c->stream;
        if (tail) tail->next = q; else head = q;
        tail = q;
        const size_t slots = (size_t)q->nchunks * VGL_CHUNK, fslots = (size_t)q->f_nchunks * VGL_CHUNK;
        VGL_HIP_TRY(vgl_pool_alloc(c->stream, (void **)&q->w_mid, sizeof(float) * std::max<size_t>(slots, 8)));
        if (fslots) VGL_HIP_TRY(vgl_pool_alloc(c->stream, (void **)&q->f_w, sizeof(float) * fslots));
        VGL_HIP_TRY(vgl_pool_alloc(c->stream, (void **)&q->g_dirty, (size_t)std::max(q->nG, 1)));
        VGL_HIP_TRY(hipMemsetAsync(q->g_dirty, 1, (size_t)std::max(q->nG, 1), c->stream));
    }
    own.h = &tail; tail = nullptr;                                  // (disarm)
    head->shared_from = const_cast<vgl_blocked_plan *>(structure);
    const_cast<vgl_blocked_plan *>(structure)->sharers++;
    *out = head;
    return 0;
}

static int vgl_blocked_plan_build_any(vgl_hip_ctx *c, const vgl_dir_csr &dir, int32_t nrows, int32_t row_base, int32_t ncols, int gather_rows,
                                      int skip_self, const float *d_weights, int a_bits, vgl_blocked_plan **out, int value_bits, int fuse_min_edges, int keep_edge_index);

int vgl_blocked_plan_build(vgl_hip_ctx *c, const vgl_dir_csr &dir, int32_t nrows, int32_t row_base, int32_t ncols, int gather_rows,
                           int skip_self, const float *d_weights, int a_bits, vgl_blocked_plan **out, int value_bits, int fuse_min_edges)
{
    return vgl_blocked_plan_build_any(c, dir, nrows, row_base, ncols, gather_rows, skip_self, d_weights, a_bits, out, value_bits, fuse_min_edges, d_weights ? 1 : 0);
}
int vgl_blocked_plan_build_indexed(vgl_hip_ctx *c, const vgl_dir_csr &dir, int32_t nrows, int32_t row_base, int32_t ncols, int gather_rows,
                                   int skip_self, int a_bits, vgl_blocked_plan **out, int fuse_min_edges)
{
    return vgl_blocked_plan_build_any(c, dir, nrows, row_base, ncols, gather_rows, skip_self, nullptr, a_bits, out, 32, fuse_min_edges, 1);
}

static int vgl_blocked_plan_build_any(vgl_hip_ctx *c, const vgl_dir_csr &dir, int32_t nrows, int32_t row_base, int32_t ncols, int gather_rows,
                                      int skip_self, const float *d_weights, int a_bits, vgl_blocked_plan **out, int value_bits, int fuse_min_edges, int keep_edge_index)
{
    if (!c || !out) VGL_FAIL("blocked_plan_build: null argument");
    // chunk positions are 32-bit: a direction is laid out whole below 2^32 - 2048 edges (VGL_BLK_PIECE_EDGES lowers the bound: tests)
    int64_t limit = (1LL << 32) - VGL_TILE;
    const bool cuttable = value_bits == 32 && a_bits == VGL_BLK_BITS;            // (the variable only lowers the bound of layouts that can be cut)
    if (const char *e = vgl_env(c, "VGL_BLK_PIECE_EDGES")) if (cuttable) limit = std::max<int64_t>(4096, atoll(e));
    if (dir.edges < limit) return vgl_blocked_plan_build_one(c, dir, nrows, row_base, ncols, gather_rows, skip_self, d_weights, a_bits, out, value_bits, fuse_min_edges, 0, nrows, keep_edge_index, 0);
    if (!cuttable) VGL_FAIL("blocked_plan_build: only 4-byte min / max-type layouts can be cut into row-range pieces (2^32 edges or more)");
    // row ranges of at most `piece` edges each (a single row above the bound cannot be cut)
    const int64_t piece = std::min<int64_t>(limit, 1LL << 31);
    const int parts = (int)std::min<int64_t>(4096, vgl_ceil_div(dir.edges, piece) + 1);
    std::vector<int32_t> bounds((size_t)parts + 1);
    VGL_TRY(vgl_hip_partition_rows(c, nrows, dir.rowptr, parts, bounds.data()));
    std::vector<int64_t> starts((size_t)parts + 1);
    for (int k = 0; k <= parts; k++) VGL_TRY(vgl_hip_memcpy_d2h(c, &starts[(size_t)k], dir.rowptr + bounds[(size_t)k], sizeof(int64_t)));
    vgl_blocked_plan *head = nullptr, *tail = nullptr;
    struct guard { vgl_blocked_plan **h; ~guard() { if (*h) vgl_blocked_plan_destroy(*h); } } own{&head};
    for (int k = 0; k < parts; k++) {
        const int32_t lo = bounds[(size_t)k], hi = bounds[(size_t)k + 1];
        const int64_t e0 = starts[(size_t)k], e1 = starts[(size_t)k + 1];
        if (hi <= lo || e1 <= e0) continue;
        if (e1 - e0 >= (1LL << 32) - VGL_TILE) VGL_FAIL("blocked_plan_build: a row range of one piece holds 2^32 edges or more");
        vgl_dir_csr view;
        int64_t *rp = nullptr;
        VGL_HIP_TRY(hipMalloc((void **)&rp, sizeof(int64_t) * ((size_t)(hi - lo) + 1)));
        hipLaunchKernelGGL(vgl_k_blk_rebase_rows, dim3((unsigned)std::min<int64_t>(4096, vgl_ceil_div((int64_t)(hi - lo) + 1, 256))), dim3(256), 0, c->stream, hi - lo,
                           dir.rowptr + lo, e0, rp);
        view.rowptr = rp; view.adj = dir.adj + e0; view.edges = e1 - e0;
        int rc = vgl_build_tile_rows(c, view, hi - lo);
        vgl_blocked_plan *q = nullptr;
        if (!rc) rc = vgl_blocked_plan_build_one(c, view, hi - lo, row_base, ncols, gather_rows, skip_self, d_weights ? d_weights + e0 : nullptr, a_bits, &q, value_bits,
                                                 fuse_min_edges, lo, nrows, keep_edge_index, e0);
        if (rc) { hipFree(rp); if (view.tile_row) hipFree(view.tile_row); return rc; }
        q->piece_rowptr = rp; q->piece_tile_row = view.tile_row;
        if (tail) tail->next = q; else head = q;
        tail = q;
    }
    if (!head) return vgl_blocked_plan_build_one(c, dir, nrows, row_base, ncols, gather_rows, skip_self, d_weights, a_bits, out, value_bits, fuse_min_edges, 0, nrows, keep_edge_index, 0);
    *out = head;
    head = nullptr;
    return 0;
}
