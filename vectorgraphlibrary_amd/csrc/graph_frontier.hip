// graph_frontier.hip -- graph handle (borrowed CSR + derived tile tables), frontier object,
// generate_new_frontier entry points and the reduce primitive.
#include "vgl_hip_internal.h"
#include <atomic>
#include "vgl_gnf.h"
#include "vgl_blocked.h"

// tile_row[t] = local row containing edge t*VGL_TILE: row r covers tiles [ceil(start/T), ceil(end/T))
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_tile_rows(int32_t nrows, const int64_t *rowptr, int32_t *tile_row, int64_t ntiles)
{
    for (int32_t r = blockIdx.x * VGL_BLOCK + threadIdx.x; r < nrows; r += gridDim.x * VGL_BLOCK) {
        const int64_t t0 = (rowptr[r] + VGL_TILE - 1) / VGL_TILE;
        const int64_t t1 = (rowptr[r + 1] + VGL_TILE - 1) / VGL_TILE;
        for (int64_t t = t0; t < t1; t++) tile_row[t] = r;
        // sentinel for the last tile = the row that owns the LAST edge (not nrows-1: trailing empty rows -- 45 % of a
        // degree-sorted RMAT graph -- would otherwise all be walked by the last tile's row map)
        if (rowptr[r] < rowptr[r + 1] && rowptr[r + 1] == rowptr[nrows]) tile_row[ntiles] = r;
    }
}

// bit (row_base + r) = row r is non-empty; whole words are written (row_base is a multiple of 64)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_nonempty_rows(int32_t nrows, int32_t row_base, const int64_t *rowptr, uint64_t *bits)
{
    const int32_t nround = (nrows + 63) & ~63;
    for (int32_t r = blockIdx.x * VGL_BLOCK + threadIdx.x; r < nround; r += gridDim.x * VGL_BLOCK) {
        const bool on = r < nrows && rowptr[r + 1] > rowptr[r];
        const unsigned long long m = __ballot(on);
        if ((threadIdx.x & 63) == 0) bits[(row_base + r) >> 6] = m;
    }
}

// rank[g] = set bits of the words before group g (exclusive prefix over the owned words); rank[ngroups] = total.  One workgroup.
__global__ __launch_bounds__(1024) void vgl_k_nz_rank(int64_t ngroups, const uint64_t *bits, int32_t *rank)
{
    __shared__ int s_w[16];
    const int64_t per = (ngroups + 1023) / 1024;
    const int64_t lo = min(ngroups, (int64_t)threadIdx.x * per), hi = min(ngroups, lo + per);
    int sum = 0;
    for (int64_t g = lo; g < hi; g++) sum += __popcll(bits[g]);
    const int inc = vgl_wave_incl_add(sum);
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = inc;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < 16; w++) { if (w < (int)(threadIdx.x >> 6)) base += s_w[w]; total += s_w[w]; }
    int run = base + inc - sum;
    for (int64_t g = lo; g < hi; g++) { rank[g] = run; run += __popcll(bits[g]); }
    if (threadIdx.x == 0) rank[ngroups] = total;
}

// vt_min[t] = smallest degree among the rows of vertex tile t (2048 rows)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_tile_min_degree(int32_t nrows, const int64_t *rowptr, int32_t *vt_min)
{
    __shared__ int s_min[VGL_WAVES];
    const int32_t r0 = blockIdx.x * VGL_TILE;
    int64_t m = INT32_MAX;
    for (int32_t r = r0 + threadIdx.x; r < min(nrows, r0 + VGL_TILE); r += VGL_BLOCK) m = min(m, rowptr[r + 1] - rowptr[r]);
    int mi = (int)m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mi = min(mi, __shfl_xor(mi, o));
    if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = mi;
    __syncthreads();
    if (threadIdx.x == 0) { for (int w = 1; w < VGL_WAVES; w++) mi = min(mi, s_min[w]); vt_min[blockIdx.x] = mi; }
}

// The eight smallest distinct ids seen so far, ascending (INT32_MAX = free slot); `overflow` = a distinct id did not fit.
struct vgl_small8 {
    int32_t a[8];
    bool overflow;
    __device__ void init()
    {
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = INT32_MAX;
        overflow = false;
    }
    __device__ void insert(int32_t x)
    {
        if (x >= a[7]) { overflow |= x > a[7]; return; }             // (the common case in a long row; ids are < INT32_MAX)
        bool dup = false;
#pragma unroll
        for (int i = 0; i < 7; i++) dup |= a[i] == x;
        if (dup) return;
        overflow |= a[7] != INT32_MAX;
        a[7] = x;
#pragma unroll
        for (int i = 7; i > 0; i--)
            if (a[i] < a[i - 1]) { const int32_t t = a[i]; a[i] = a[i - 1]; a[i - 1] = t; }
    }
    __device__ void pop()
    {
#pragma unroll
        for (int i = 0; i < 7; i++) a[i] = a[i + 1];
        a[7] = INT32_MAX;
    }
};

// head0[r] / head1[r] = the eight SMALLEST distinct ids of row r, ascending, -1 padded: two aligned 16-byte records per row, so the
// bottom-up probe reads its candidates with coalesced loads instead of row offsets + a dependent adjacency load.  Smallest ids first:
// with the degree renumbering (ids ascending = degrees descending) these are the row's best-connected in-neighbours -- the ones most
// likely to be in a large frontier: on the first bottom-up level of an RMAT traversal half as many candidates go on to the second round
// of probes and a quarter as many rows are deferred (tests/studies/bu_head_order.py).  long_bits[v] = row v holds ids that are not in
// its head; the deferred pass then scans the WHOLE row: the head is a selection, not a prefix.  One
// wavefront per 64 rows: rows of more than 64 entries are scanned by the whole wavefront (per-lane selections merged by eight wave
// minima), the others by their own lane.
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_row_heads(int32_t nrows, int32_t row_base, const int64_t *rowptr, const int32_t *adj, int4 *head0,
                                                             int4 *head1, uint64_t *long_bits, const int32_t *nz_rank)
{
    const int lane = threadIdx.x & 63;
    const int32_t ngroups = (nrows + 63) >> 6;
    for (int32_t grp = blockIdx.x * VGL_WAVES + (threadIdx.x >> 6); grp < ngroups; grp += gridDim.x * VGL_WAVES) {
        const int32_t r = (grp << 6) + lane;
        int64_t b = 0, n = 0;
        if (r < nrows) { b = rowptr[r]; n = rowptr[r + 1] - b; }
        // record of this row: the rows with incoming edges are numbered consecutively (rows without have no record)
        const unsigned long long nzw = __ballot(n > 0);
        const int32_t rec = nz_rank[grp] + (int32_t)__popcll(nzw & ((1ULL << lane) - 1ULL));
        bool has_more = false;
        unsigned long long big = __ballot(n > 64);
        while (big) {                                               // long rows: all lanes on one row
            const int j = __ffsll((long long)big) - 1;
            big &= big - 1;
            const int64_t bj = __shfl(b, j), nj = __shfl(n, j);
            vgl_small8 s;
            s.init();
            for (int64_t i = lane; i < nj; i += 64) s.insert(adj[bj + i]);
            int32_t out[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                int32_t m = s.a[0];
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) m = min(m, __shfl_xor(m, d));
                out[k] = m == INT32_MAX ? -1 : m;
                if (s.a[0] == m && m != INT32_MAX) s.pop();
            }
            const bool more = __any(s.overflow || s.a[0] != INT32_MAX);
            if (lane == j) {
                head0[rec] = make_int4(out[0], out[1], out[2], out[3]);
                head1[rec] = make_int4(out[4], out[5], out[6], out[7]);
                has_more = more;
            }
        }
        if (r < nrows && n > 0 && n <= 64) {
            vgl_small8 s;
            s.init();
            for (int64_t i = 0; i < n; i++) s.insert(adj[b + i]);
            int32_t out[8];
#pragma unroll
            for (int k = 0; k < 8; k++) out[k] = s.a[k] == INT32_MAX ? -1 : s.a[k];
            head0[rec] = make_int4(out[0], out[1], out[2], out[3]);
            head1[rec] = make_int4(out[4], out[5], out[6], out[7]);
            has_more = s.overflow;
        }
        const unsigned long long m = __ballot(has_more);
        if (lane == 0) long_bits[(row_base + (grp << 6)) >> 6] = m;
    }
}

__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_iota_flags(int32_t n, int32_t *ids, int32_t *flags, int32_t flag)
{
    for (int32_t i = blockIdx.x * VGL_BLOCK + threadIdx.x; i < n; i += gridDim.x * VGL_BLOCK) {
        if (ids) ids[i] = i;
        flags[i] = flag;
    }
}
__global__ void vgl_k_add_vertex(int32_t v, int32_t *ids, int32_t *flags) { ids[0] = v; flags[v] = 1; }

// deterministic two-stage sums: per-workgroup partials (fixed shape) then one workgroup adds them in order
template <class T, class ACC, int MODE>   // MODE 0: all, 1: flags, 2: ids
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_reduce_partial(int32_t n, const T *values, const int32_t *flags,
                                                                  const int32_t *ids, double *partials)
{
    __shared__ ACC s[VGL_WAVES];
    ACC acc = 0;
    for (int32_t i = blockIdx.x * VGL_BLOCK + threadIdx.x; i < n; i += gridDim.x * VGL_BLOCK) {
        if (MODE == 0) acc += (ACC)values[i];
        else if (MODE == 1) { if (flags[i]) acc += (ACC)values[i]; }
        else acc += (ACC)values[ids[i]];
    }
    acc = vgl_block_reduce_add(acc, s);
    if (threadIdx.x == 0) partials[blockIdx.x] = (double)acc;
}
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_neq_partial(int32_t n, const uint32_t *a, const uint32_t *b, double *partials)
{
    __shared__ int64_t s[VGL_WAVES];
    int64_t acc = 0;
    for (int32_t i = blockIdx.x * VGL_BLOCK + threadIdx.x; i < n; i += gridDim.x * VGL_BLOCK) acc += (a[i] != b[i]);
    acc = vgl_block_reduce_add(acc, s);
    if (threadIdx.x == 0) partials[blockIdx.x] = (double)acc;
}
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_reduce_final(int nparts, const double *partials, double *out)
{
    __shared__ double s[VGL_WAVES];
    double acc = 0;
    for (int i = threadIdx.x; i < nparts; i += VGL_BLOCK) acc += partials[i];
    acc = vgl_block_reduce_add(acc, s);
    if (threadIdx.x == 0) *out = acc;
}

// ---- advance plan of a sparse frontier: exclusive scan of the degrees of ids[0..F) (three passes, like the GNF) ----
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_plan_sums(int32_t F, const int32_t *ids, int32_t row_base, const int64_t *rowptr,
                                                             int64_t *blk_sum)
{
    __shared__ int64_t s64[VGL_WAVES];
    int64_t deg = 0;
    const int32_t p0 = blockIdx.x * VGL_TILE + threadIdx.x * VGL_EPT;
    for (int j = 0; j < VGL_EPT; j++)
        if (p0 + j < F) { const int32_t r = ids[p0 + j] - row_base; deg += rowptr[r + 1] - rowptr[r]; }
    deg = vgl_block_reduce_add(deg, s64);
    if (threadIdx.x == 0) blk_sum[blockIdx.x] = deg;
}
__global__ __launch_bounds__(VGL_SCAN_THREADS) void vgl_k_plan_scan(int64_t nblk, const int64_t *blk_sum, int64_t *blk_off, int64_t *counters,
                                                                    int64_t *offs, int32_t F)
{
    __shared__ int64_t s_d[VGL_SCAN_THREADS / 64];
    const int64_t per = (nblk + VGL_SCAN_THREADS - 1) / VGL_SCAN_THREADS;
    const int64_t lo = min(nblk, (int64_t)threadIdx.x * per), hi = min(nblk, lo + per);
    int64_t d = 0;
    for (int64_t t = lo; t < hi; t++) d += blk_sum[t];
    const int64_t di = vgl_wave_incl_add(d);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) s_d[w] = di;
    __syncthreads();
    int64_t db = 0, dtot = 0;
    for (int i = 0; i < VGL_SCAN_THREADS / 64; i++) { if (i < w) db += s_d[i]; dtot += s_d[i]; }
    int64_t dpre = db + di - d;
    for (int64_t t = lo; t < hi; t++) { blk_off[t] = dpre; dpre += blk_sum[t]; }
    if (threadIdx.x == 0) { counters[C_NEIGH] = dtot; offs[F] = dtot; }
}
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_plan_write(int32_t F, const int32_t *ids, int32_t row_base, const int64_t *rowptr,
                                                              const int64_t *blk_off, int64_t *offs)
{
    __shared__ int64_t s64[VGL_WAVES];
    int64_t degs[VGL_EPT], deg = 0;
    const int32_t p0 = blockIdx.x * VGL_TILE + threadIdx.x * VGL_EPT;
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) {
        degs[j] = 0;
        if (p0 + j < F) { const int32_t r = ids[p0 + j] - row_base; degs[j] = rowptr[r + 1] - rowptr[r]; deg += degs[j]; }
    }
    int64_t tot;
    int64_t e = blk_off[blockIdx.x] + vgl_block_excl_add(deg, s64, &tot);
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++)
        if (p0 + j < F) { offs[p0 + j] = e; e += degs[j]; }
}
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_plan_tile_first(int32_t F, const int64_t *offs, int32_t *tile_first)
{
    for (int32_t p = blockIdx.x * VGL_BLOCK + threadIdx.x; p < F; p += gridDim.x * VGL_BLOCK) {
        const int64_t t0 = (offs[p] + VGL_TILE - 1) / VGL_TILE;
        const int64_t t1 = (offs[p + 1] + VGL_TILE - 1) / VGL_TILE;
        for (int64_t t = t0; t < t1; t++) tile_first[t] = p;
        if (offs[p] < offs[p + 1] && offs[p + 1] == offs[F]) tile_first[(offs[F] + VGL_TILE - 1) / VGL_TILE] = p;   // owner of the last edge
    }
}

int vgl_build_tile_rows(vgl_hip_ctx *c, vgl_dir_csr &d, int32_t nrows)
{
    d.ntiles = vgl_ceil_div(d.edges, VGL_TILE);
    VGL_HIP_TRY(hipMalloc((void **)&d.tile_row, sizeof(int32_t) * ((size_t)d.ntiles + 2)));
    int grid = (int)std::min<int64_t>(8192, std::max<int64_t>(1, vgl_ceil_div(nrows, VGL_BLOCK)));
    hipLaunchKernelGGL(vgl_k_tile_rows, dim3(grid), dim3(VGL_BLOCK), 0, c->stream, nrows, d.rowptr, d.tile_row, d.ntiles);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

template <class T>
static int vgl_alloc(T **p, size_t n) { VGL_HIP_TRY(hipMalloc((void **)p, sizeof(T) * (n ? n : 1))); return 0; }

static int vgl_reduce_common(vgl_hip_ctx *c, int nblocks, double *result)
{
    hipLaunchKernelGGL(vgl_k_reduce_final, dim3(1), dim3(VGL_BLOCK), 0, c->stream, nblocks, c->d_partials, c->d_partials + nblocks);
    VGL_HIP_TRY(hipGetLastError());
    VGL_HIP_TRY(hipMemcpyAsync(result, c->d_partials + nblocks, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    VGL_HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" {

int vgl_hip_graph_create(vgl_hip_ctx *c, int32_t V, int32_t row_begin, int32_t row_end,
                         const int64_t *d_out_rowptr, const int32_t *d_out_adj, int64_t out_edges,
                         const int64_t *d_in_rowptr, const int32_t *d_in_adj, int64_t in_edges, vgl_hip_graph **out)
{
    if (!c || !out) VGL_FAIL("graph_create: null argument");
    if (V <= 0) VGL_FAIL("graph_create: vertices_count must be positive");
    if (row_begin < 0 || row_end > V || row_begin >= row_end) VGL_FAIL("graph_create: bad owned row range");
    if (row_begin % 64 != 0) VGL_FAIL("graph_create: row_begin must be a multiple of 64 (use vgl_hip_partition_rows)");
    if (!d_out_rowptr || (!d_out_adj && out_edges > 0)) VGL_FAIL("graph_create: outgoing CSR is required");
    if (out_edges < 0 || in_edges < 0) VGL_FAIL("graph_create: negative edge count");
    VGL_HIP_TRY(hipSetDevice(c->device));
    vgl_hip_graph *g = new vgl_hip_graph();
    static std::atomic<uint64_t> next_uid{1};
    g->uid = next_uid.fetch_add(1);
    // a failure below (an allocation, a launch) must not leave the half-built handle and its device arrays behind
    struct rollback { vgl_hip_ctx *c; vgl_hip_graph *g; ~rollback() { if (g) vgl_hip_graph_destroy(c, g); } } undo{c, g};
    g->V = V; g->row_begin = row_begin; g->row_end = row_end; g->nrows = row_end - row_begin;
    g->out.rowptr = d_out_rowptr; g->out.adj = d_out_adj; g->out.edges = out_edges;
    VGL_TRY(vgl_build_tile_rows(c, g->out, g->nrows));
    if (d_in_rowptr) {
        g->in.rowptr = d_in_rowptr; g->in.adj = d_in_adj; g->in.edges = in_edges;
        VGL_TRY(vgl_build_tile_rows(c, g->in, g->nrows));
    }
    const size_t words = (size_t)vgl_ceil_div(V, 64) + 1;
    VGL_TRY(vgl_alloc(&g->bm_visited, words));
    VGL_TRY(vgl_alloc(&g->bm_front, words));
    VGL_TRY(vgl_alloc(&g->bm_next, words));
    VGL_TRY(vgl_alloc(&g->bm_in_nz, words));
    VGL_TRY(vgl_alloc(&g->ids, (size_t)g->nrows));
    VGL_TRY(vgl_alloc(&g->offs, (size_t)g->nrows + 1));
    g->nvtiles = vgl_ceil_div(g->nrows, VGL_TILE);
    VGL_TRY(vgl_alloc(&g->vt_cnt, (size_t)g->nvtiles));
    VGL_TRY(vgl_alloc(&g->vt_cnt_off, (size_t)g->nvtiles));
    VGL_TRY(vgl_alloc(&g->vt_deg, (size_t)g->nvtiles));
    VGL_TRY(vgl_alloc(&g->vt_deg_off, (size_t)g->nvtiles));
    VGL_TRY(vgl_alloc(&g->vt_min_deg, (size_t)g->nvtiles));
    hipLaunchKernelGGL(vgl_k_tile_min_degree, dim3((unsigned)g->nvtiles), dim3(VGL_BLOCK), 0, c->stream, g->nrows, g->out.rowptr, g->vt_min_deg);
    VGL_TRY(vgl_alloc(&g->tile_first, (size_t)g->out.ntiles + 2));
    VGL_TRY(vgl_alloc(&g->heavy, (size_t)g->nrows + 4096 * VGL_BLOCK));
    VGL_TRY(vgl_alloc(&g->heavy_cnt, (size_t)4096));
    VGL_TRY(vgl_alloc(&g->heavy_off, (size_t)4097));
    VGL_TRY(vgl_alloc(&g->bu_partials, (size_t)4096 * 4));
    VGL_TRY(vgl_alloc(&g->tickets, (size_t)4 * VGL_TICKET_WORDS));
    VGL_HIP_TRY(hipMemsetAsync(g->tickets, 0, 4 * VGL_TICKET_WORDS * sizeof(uint32_t), c->stream));
    VGL_TRY(vgl_alloc(&g->epoch, (size_t)V));
    VGL_TRY(vgl_alloc(&g->fscratch, (size_t)V));
    VGL_TRY(vgl_alloc(&g->fscratch2, (size_t)V));
    VGL_TRY(vgl_alloc(&g->fscratch3, (size_t)V));
    VGL_TRY(vgl_alloc(&g->iscratch, (size_t)V));
    VGL_HIP_TRY(hipMemsetAsync(g->bm_visited, 0, words * 8, c->stream));
    VGL_HIP_TRY(hipMemsetAsync(g->bm_front, 0, words * 8, c->stream));
    VGL_HIP_TRY(hipMemsetAsync(g->bm_next, 0, words * 8, c->stream));
    VGL_HIP_TRY(hipMemsetAsync(g->bm_in_nz, 0, words * 8, c->stream));
    if (g->in.rowptr) {
        int grid = (int)std::min<int64_t>(8192, std::max<int64_t>(1, vgl_ceil_div(g->nrows, VGL_BLOCK)));
        hipLaunchKernelGGL(vgl_k_nonempty_rows, dim3(grid), dim3(VGL_BLOCK), 0, c->stream, g->nrows, g->row_begin, g->in.rowptr, g->bm_in_nz);
        const int64_t ngroups = vgl_ceil_div(g->nrows, 64);
        VGL_TRY(vgl_alloc(&g->in_nz_rank, (size_t)ngroups + 1));
        hipLaunchKernelGGL(vgl_k_nz_rank, dim3(1), dim3(1024), 0, c->stream, ngroups, (const uint64_t *)(g->bm_in_nz + (row_begin >> 6)), g->in_nz_rank);
        VGL_HIP_TRY(hipGetLastError());
        VGL_TRY(vgl_hip_memcpy_d2h(c, &g->in_nz_rows, g->in_nz_rank + ngroups, sizeof(int32_t)));
        VGL_TRY(vgl_alloc(&g->in_head, (size_t)std::max(g->in_nz_rows, 1) * 8));
        VGL_TRY(vgl_alloc(&g->bm_in_long, words));
        VGL_HIP_TRY(hipMemsetAsync(g->bm_in_long, 0, words * 8, c->stream));
        hipLaunchKernelGGL(vgl_k_row_heads, dim3(grid), dim3(VGL_BLOCK), 0, c->stream, g->nrows, g->row_begin, g->in.rowptr, g->in.adj,
                           reinterpret_cast<int4 *>(g->in_head), reinterpret_cast<int4 *>(g->in_head) + g->in_nz_rows, g->bm_in_long, (const int32_t *)g->in_nz_rank);
        VGL_HIP_TRY(hipGetLastError());
    }
    VGL_HIP_TRY(hipStreamSynchronize(c->stream));
    undo.g = nullptr;
    *out = g;
    return 0;
}

int vgl_hip_graph_destroy(vgl_hip_ctx *c, vgl_hip_graph *g)
{
    if (!g) return 0;
    if (c) hipStreamSynchronize(c->stream);
    if (g->transposed) { vgl_hip_graph_destroy(c, g->transposed); g->transposed = nullptr; }
    if (g->blk_pr) { vgl_blocked_plan_destroy(g->blk_pr); g->blk_pr = nullptr; }
    if (g->blk_cc) { vgl_blocked_plan_destroy(g->blk_cc); g->blk_cc = nullptr; }
    if (g->blk_bfs) { vgl_blocked_plan_destroy(g->blk_bfs); g->blk_bfs = nullptr; }
    if (g->blk_path) { vgl_blocked_plan_destroy(g->blk_path); g->blk_path = nullptr; }
    void *ptrs[] = {g->out.tile_row, g->in.tile_row, g->bm_visited, g->bm_front, g->bm_next, g->bm_in_nz, g->in_head, g->in_nz_rank, g->bm_in_long, g->ids, g->offs, g->vt_cnt,
                    g->vt_cnt_off, g->vt_deg, g->vt_deg_off, g->tile_first, g->heavy, g->heavy_cnt, g->heavy_off, g->bu_partials, g->tickets, g->epoch, g->fscratch, g->fscratch2,
                    g->fscratch3, g->iscratch, g->ds_tile_active, g->ds_partials, g->out.hub_rows, g->in.hub_rows, g->out.giant_rows, g->in.giant_rows, g->out.pull_blk_row,
                    g->in.pull_blk_row, g->vt_min_deg, g->gnf_bits, g->out.hub_chunks, g->in.hub_chunks, g->out.hub_chunk_sums, g->in.hub_chunk_sums, g->pr_indeg};
    for (void *p : ptrs) if (p) hipFree(p);
    delete g;
    return 0;
}

int vgl_hip_frontier_create(vgl_hip_ctx *c, vgl_hip_graph *g, vgl_hip_frontier **out)
{
    if (!c || !g || !out) VGL_FAIL("frontier_create: null argument");
    vgl_hip_frontier *f = new vgl_hip_frontier();
    f->g = g;
    VGL_TRY(vgl_alloc(&f->flags, (size_t)g->V));
    VGL_TRY(vgl_alloc(&f->ids, (size_t)g->V));
    *out = f;
    return vgl_hip_frontier_set_all_active(c, f);   // VGL frontiers start all-active (base_frontier.h ctor)
}
// a frontier handle over arrays the CALLER owns (a backend bound to another library's frontier container, whose flags / ids arrays host code of
// that library reads and writes: FrontierCSR / FrontierVectorCSR, base_frontier.h:5-62).  Nothing is initialised: vgl_hip_frontier_set_state says
// what the arrays hold before every use.
int vgl_hip_frontier_create_on(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t *d_flags, int32_t *d_ids, vgl_hip_frontier **out)
{
    if (!c || !g || !d_flags || !d_ids || !out) VGL_FAIL("frontier_create_on: null argument");
    vgl_hip_frontier *f = new vgl_hip_frontier();
    f->g = g; f->flags = d_flags; f->ids = d_ids; f->borrowed = true;
    f->size = 0; f->neighbours = 0; f->sparsity = VGL_HIP_FRONTIER_SPARSE; f->plan_dir = -1;
    *out = f;
    return 0;
}
// the caller changed the frontier behind the handle (host-side add_vertex / clear / set_all_active of the owning container, or a traversal
// direction switch to the other direction's graph handle): take its description as given and forget the advance plan
int vgl_hip_frontier_set_state(vgl_hip_ctx *c, vgl_hip_frontier *f, vgl_hip_graph *g, int32_t size, int64_t neighbours, int sparsity)
{
    if (!c || !f || !g) VGL_FAIL("frontier_set_state: null argument");
    if (g->V != f->g->V) VGL_FAIL("frontier_set_state: the graph handle has another vertex count");
    if (size < 0 || size > g->V || sparsity < VGL_HIP_FRONTIER_DENSE || sparsity > VGL_HIP_FRONTIER_ALL_ACTIVE) VGL_FAIL("frontier_set_state: bad description");
    f->g = g; f->size = size; f->neighbours = neighbours; f->sparsity = sparsity; f->plan_dir = -1;
    return 0;
}
int vgl_hip_frontier_destroy(vgl_hip_ctx *c, vgl_hip_frontier *f)
{
    if (!f) return 0;
    if (c) hipStreamSynchronize(c->stream);
    if (!f->borrowed) { hipFree(f->flags); hipFree(f->ids); }
    if (f->offs) { hipFree(f->offs); hipFree(f->tile_first); hipFree(f->blk_sum); hipFree(f->blk_off); }
    delete f;
    return 0;
}
int vgl_hip_frontier_set_all_active(vgl_hip_ctx *c, vgl_hip_frontier *f)
{
    if (!c || !f) VGL_FAIL("null argument");
    const int32_t V = f->g->V;
    hipLaunchKernelGGL(vgl_k_iota_flags, dim3((unsigned)std::min<int64_t>(4096, vgl_ceil_div(V, VGL_BLOCK))), dim3(VGL_BLOCK), 0,
                       c->stream, V, f->ids, f->flags, 1);
    VGL_HIP_TRY(hipGetLastError());
    f->size = V; f->neighbours = f->g->out.edges; f->sparsity = VGL_HIP_FRONTIER_ALL_ACTIVE; f->plan_dir = -1;
    return 0;
}
int vgl_hip_frontier_clear(vgl_hip_ctx *c, vgl_hip_frontier *f)
{
    if (!c || !f) VGL_FAIL("null argument");
    VGL_HIP_TRY(hipMemsetAsync(f->flags, 0, sizeof(int32_t) * (size_t)f->g->V, c->stream));
    f->size = 0; f->neighbours = 0; f->sparsity = VGL_HIP_FRONTIER_SPARSE; f->plan_dir = -1;
    return 0;
}
int vgl_hip_frontier_add_vertex(vgl_hip_ctx *c, vgl_hip_frontier *f, int32_t v)
{
    if (!c || !f) VGL_FAIL("null argument");
    if (f->size > 0) VGL_FAIL("VGL error! can not add vertex to non-empty frontier");   // modification.hpp:33-36
    if (v < 0 || v >= f->g->V) VGL_FAIL("frontier_add_vertex: vertex id out of range");
    hipLaunchKernelGGL(vgl_k_add_vertex, dim3(1), dim3(1), 0, c->stream, v, f->ids, f->flags);
    VGL_HIP_TRY(hipGetLastError());
    f->size = 1; f->sparsity = VGL_HIP_FRONTIER_SPARSE; f->plan_dir = -1;
    int64_t rp[2] = {0, 0};
    if (v >= f->g->row_begin && v < f->g->row_end)
        VGL_TRY(vgl_hip_memcpy_d2h(c, rp, f->g->out.rowptr + (v - f->g->row_begin), sizeof(rp)));
    f->neighbours = rp[1] - rp[0];
    return 0;
}
int vgl_hip_frontier_info(vgl_hip_ctx *c, vgl_hip_frontier *f, int32_t *size, int64_t *neighbours, int *sparsity)
{
    if (!c || !f) VGL_FAIL("null argument");
    if (size) *size = f->size;
    if (neighbours) *neighbours = f->neighbours;
    if (sparsity) *sparsity = f->sparsity;
    return 0;
}
int vgl_hip_graph_tile_rows(vgl_hip_graph *g, int direction, const int32_t **d_tile_row, int64_t *ntiles)
{
    if (!g || !d_tile_row || !ntiles) VGL_FAIL("graph_tile_rows: null argument");
    const vgl_dir_csr &d = direction ? g->in : g->out;
    if (!d.rowptr) VGL_FAIL("graph_tile_rows: this direction of the graph is not stored");
    *d_tile_row = d.tile_row; *ntiles = d.ntiles;
    return 0;
}

}  // extern "C"

// plan buffers of a frontier (edge offsets of its ids, tile table, per-block sums), allocated on first use
static int vgl_frontier_reserve(vgl_hip_graph *g, vgl_hip_frontier *f)
{
    if (f->offs) return 0;
    const int64_t emax = std::max(g->out.edges, g->in.edges);
    VGL_TRY(vgl_alloc(&f->offs, (size_t)g->V + 1));
    VGL_TRY(vgl_alloc(&f->tile_first, (size_t)vgl_ceil_div(emax, VGL_TILE) + 2));
    VGL_TRY(vgl_alloc(&f->blk_sum, (size_t)vgl_ceil_div(g->V, VGL_TILE) + 1));
    VGL_TRY(vgl_alloc(&f->blk_off, (size_t)vgl_ceil_div(g->V, VGL_TILE) + 1));
    return 0;
}

// VGL_GNF_INT_FLAGS=1: the count pass of the operator classes writes the int32 flags of every result, as the reference does (read once per process)
static bool vgl_gnf_int_flags()
{
    static const bool on = [] { const char *e = getenv("VGL_GNF_INT_FLAGS"); return e && e[0] == '1'; }();
    return on;
}

extern "C" {

int vgl_hip_gnf_begin(vgl_hip_ctx *c, vgl_hip_graph *g, vgl_hip_frontier *f, int want_plan, vgl_hip_gnf_buffers *out)
{
    if (!c || !g || !f || !out) VGL_FAIL("gnf_begin: null argument");
    if (g->row_begin != 0 || g->row_end != g->V) VGL_FAIL("generate_new_frontier: graph handle must own all rows");
    if (want_plan) VGL_TRY(vgl_frontier_reserve(g, f));
    out->nrows = g->nrows; out->row_begin = g->row_begin; out->nvtiles = g->nvtiles; out->out_rowptr = g->out.rowptr;
    out->vt_cnt = g->vt_cnt; out->vt_cnt_off = g->vt_cnt_off; out->vt_deg = g->vt_deg; out->vt_deg_off = g->vt_deg_off;
    out->ticket = g->nvtiles <= 16384 ? g->tickets + 0 * VGL_TICKET_WORDS : nullptr;      // beyond 2^25 vertices vgl_hip_gnf_complete runs the scan pass
    out->counters = c->d_counters; out->host_counters = (volatile int64_t *)c->h_counters;
    if (!g->gnf_bits) {
        VGL_TRY(vgl_alloc(&g->gnf_bits, (size_t)vgl_ceil_div(g->V, 8) + 8));
        VGL_HIP_TRY(hipMemsetAsync(g->gnf_bits, 0, (size_t)vgl_ceil_div(g->V, 8) + 8, c->stream));
    }
    out->front_bytes = g->gnf_bits;
    out->flags = vgl_gnf_int_flags() ? f->flags : nullptr;     // (NULL: the count pass leaves the bitmap only, vgl_hip_gnf_complete does the rest)
    out->plan_offs = want_plan ? f->offs : nullptr;
    out->seq = vgl_next_seq(c);
    return 0;
}

// second half of a frontier generation whose count pass ran in the caller's translation unit (the C++ operator class evaluates the
// user's predicate in vgl_k_gnf_count itself: flags, per-tile counts, totals in the context's pinned counters): size / neighbours /
// sparsity, and for a SPARSE result the ascending-id compaction -- with the exclusive out-edge offsets of the ids when want_plan, so
// that the advance that follows needs no offset pass of its own.
int vgl_hip_gnf_complete(vgl_hip_ctx *c, vgl_hip_graph *g, vgl_hip_frontier *f, double dense_threshold, int want_plan, int64_t seq)
{
    if (!c || !g || !f) VGL_FAIL("gnf_complete: null argument");
    if (g->row_begin != 0 || g->row_end != g->V) VGL_FAIL("generate_new_frontier: graph handle must own all rows");
    VGL_HIP_TRY(hipGetLastError());
    if (g->nvtiles > 16384)                                  // the count launch left per-tile counts only (see vgl_gnf_run)
        hipLaunchKernelGGL(vgl_k_gnf_scan, dim3(1), dim3(VGL_SCAN_THREADS), 0, c->stream, g->nvtiles, g->vt_cnt, g->vt_deg, g->vt_cnt_off, g->vt_deg_off, c->d_counters,
                           want_plan ? f->offs : (int64_t *)nullptr, (volatile int64_t *)c->h_counters, seq);
    VGL_TRY(vgl_wait_counters(c, seq));
    f->size = (int32_t)c->h_counters[C_FRONT];
    f->neighbours = c->h_counters[C_NEIGH];
    f->plan_dir = -1;
    const bool all = f->size == g->V, dense = !all && dense_threshold > 0.0 && (double)f->size / g->V > dense_threshold;
    if (all || dense) {
        f->sparsity = all ? VGL_HIP_FRONTIER_ALL_ACTIVE : VGL_HIP_FRONTIER_DENSE;
        if (!vgl_gnf_int_flags()) {                          // the flags of a frontier that is walked by its flags: from the bitmap of the count pass
            if (!g->gnf_bits) VGL_FAIL("gnf_complete: vgl_hip_gnf_begin has not run on this graph handle");
            vgl_timed_launch tl(c, "gnf");
            hipLaunchKernelGGL(vgl_k_bits_to_flags, dim3((unsigned)std::min<int64_t>(4096, std::max<int64_t>(1, vgl_ceil_div(vgl_ceil_div(g->nrows, 8), VGL_BLOCK)))), dim3(VGL_BLOCK), 0,
                               c->stream, g->nrows, g->row_begin, (const uint8_t *)g->gnf_bits, f->flags);
            VGL_HIP_TRY(hipGetLastError());
        }
        return 0;
    }
    f->sparsity = VGL_HIP_FRONTIER_SPARSE;
    if (want_plan) VGL_TRY(vgl_frontier_reserve(g, f));
    {
        vgl_timed_launch tl(c, "gnf");
        if (!g->gnf_bits) VGL_FAIL("gnf_complete: vgl_hip_gnf_begin has not run on this graph handle");
        const vgl_pred_bits pred{g->gnf_bits};               // (the bitmap is there in either mode: 2 MiB to read instead of 64 MiB of flags)
        hipLaunchKernelGGL(vgl_k_gnf_write<vgl_pred_bits>, dim3((unsigned)g->nvtiles), dim3(VGL_BLOCK), 0, c->stream, pred, g->nrows, g->row_begin,
                           g->out.rowptr, g->vt_cnt_off, g->vt_deg_off, f->ids, want_plan ? f->offs : (int64_t *)nullptr,
                           want_plan ? f->tile_first : (int32_t *)nullptr, (int64_t)f->neighbours);
    }
    VGL_HIP_TRY(hipGetLastError());
    if (want_plan) { f->plan_dir = 0; f->plan_edges = f->neighbours; }      // offsets AND tile table of the outgoing direction are in place
    return 0;
}

int vgl_hip_frontier_advance_plan(vgl_hip_ctx *c, vgl_hip_graph *g, vgl_hip_frontier *f, int direction, const int64_t **d_offs,
                                  const int32_t **d_tile_first, int64_t *edges)
{
    if (!c || !g || !f || !d_offs || !d_tile_first || !edges) VGL_FAIL("frontier_advance_plan: null argument");
    const vgl_dir_csr &d = direction ? g->in : g->out;
    if (!d.rowptr) VGL_FAIL("frontier_advance_plan: this direction of the graph is not stored");
    if (f->sparsity != VGL_HIP_FRONTIER_SPARSE) VGL_FAIL("frontier_advance_plan: only sparse frontiers need a plan");
    const int32_t F = f->size;
    const int64_t nblk = vgl_ceil_div(std::max<int64_t>(F, 1), VGL_TILE);
    VGL_TRY(vgl_frontier_reserve(g, f));
    *d_offs = f->offs; *d_tile_first = f->tile_first; *edges = 0;
    if (F == 0) return 0;
    const int grid = (int)std::min<int64_t>(4096, vgl_ceil_div(F, VGL_BLOCK));
    if (f->plan_dir == direction) {              // the frontier generation (or an earlier call) left offsets and tile table of these ids behind
        *edges = f->plan_edges;
        return 0;
    }
    hipLaunchKernelGGL(vgl_k_plan_sums, dim3((unsigned)nblk), dim3(VGL_BLOCK), 0, c->stream, F, f->ids, g->row_begin, d.rowptr, f->blk_sum);
    hipLaunchKernelGGL(vgl_k_plan_scan, dim3(1), dim3(VGL_SCAN_THREADS), 0, c->stream, nblk, f->blk_sum, f->blk_off, c->d_counters, f->offs, F);
    hipLaunchKernelGGL(vgl_k_plan_write, dim3((unsigned)nblk), dim3(VGL_BLOCK), 0, c->stream, F, f->ids, g->row_begin, d.rowptr, f->blk_off, f->offs);
    hipLaunchKernelGGL(vgl_k_plan_tile_first, dim3(grid), dim3(VGL_BLOCK), 0, c->stream, F, f->offs, f->tile_first);
    VGL_HIP_TRY(hipGetLastError());
    VGL_TRY(vgl_read_counters(c, false));
    *edges = c->h_counters[C_NEIGH];
    f->plan_dir = direction; f->plan_edges = *edges;
    return 0;
}

int vgl_hip_reduce_sum_f64_buffer(vgl_hip_ctx *c, int64_t n, const double *d_values, double *result)
{
    if (!c || !d_values || !result) VGL_FAIL("reduce_sum_f64_buffer: null argument");
    if (n <= 0) { *result = 0; return 0; }
    if (n > 0x7FFFFFFFLL) VGL_FAIL("reduce_sum_f64_buffer: n too large");
    const int nb = (int)std::min<int64_t>(1024, vgl_ceil_div(n, VGL_BLOCK));
    VGL_TRY(vgl_ensure_partials(c, (size_t)nb + 1));
    hipLaunchKernelGGL((vgl_k_reduce_partial<double, double, 0>), dim3(nb), dim3(VGL_BLOCK), 0, c->stream, (int32_t)n, d_values,
                       (const int32_t *)nullptr, (const int32_t *)nullptr, c->d_partials);
    return vgl_reduce_common(c, nb, result);
}

int32_t *vgl_hip_frontier_ids(vgl_hip_frontier *f) { return f ? f->ids : nullptr; }
int32_t *vgl_hip_frontier_flags(vgl_hip_frontier *f) { return f ? f->flags : nullptr; }

}  // extern "C"

template <class Pred>
static int vgl_gnf_frontier(vgl_hip_ctx *c, vgl_hip_graph *g, Pred pred, double dense_threshold, vgl_hip_frontier *f,
                            int32_t *flags_out)
{
    if (g->row_begin != 0 || g->row_end != g->V) VGL_FAIL("generate_new_frontier: graph handle must own all rows");
    // pass 1+2: flags, counts, totals
    VGL_TRY(vgl_gnf_run(c, g, pred, f->ids, nullptr, nullptr, nullptr, flags_out, false, true));
    f->size = (int32_t)c->h_counters[C_FRONT];
    f->plan_dir = -1;
    f->neighbours = c->h_counters[C_NEIGH];
    if (f->size == g->V) { f->sparsity = VGL_HIP_FRONTIER_ALL_ACTIVE; }
    else if (dense_threshold > 0.0 && (double)f->size / g->V > dense_threshold) { f->sparsity = VGL_HIP_FRONTIER_DENSE; }
    else {
        f->sparsity = VGL_HIP_FRONTIER_SPARSE;
        vgl_timed_launch tl(c, "gnf");
        hipLaunchKernelGGL(vgl_k_gnf_write<Pred>, dim3((unsigned)g->nvtiles), dim3(VGL_BLOCK), 0, c->stream, pred, g->nrows,
                           g->row_begin, g->out.rowptr, g->vt_cnt_off, g->vt_deg_off, f->ids, (int64_t *)nullptr, (int32_t *)nullptr, (int64_t)0);
    }
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" {

int vgl_hip_gnf_from_flags(vgl_hip_ctx *c, vgl_hip_graph *g, const int32_t *d_flags, double dense_threshold, vgl_hip_frontier *f)
{
    if (!c || !g || !f || !d_flags) VGL_FAIL("gnf_from_flags: null argument");
    vgl_pred_nonzero_i32 pred{d_flags};
    return vgl_gnf_frontier(c, g, pred, dense_threshold, f, f->flags);
}
int vgl_hip_gnf_equal_i32(vgl_hip_ctx *c, vgl_hip_graph *g, const int32_t *d_values, int32_t value, double dense_threshold,
                          vgl_hip_frontier *f)
{
    if (!c || !g || !f || !d_values) VGL_FAIL("gnf_equal_i32: null argument");
    vgl_pred_equal_i32 pred{d_values, value};
    return vgl_gnf_frontier(c, g, pred, dense_threshold, f, f->flags);
}

int vgl_hip_reduce_sum_i32(vgl_hip_ctx *c, vgl_hip_frontier *f, const int32_t *d_values, int64_t *result)
{
    if (!c || !f || !d_values || !result) VGL_FAIL("reduce: null argument");
    const int32_t n = (f->sparsity == VGL_HIP_FRONTIER_SPARSE) ? f->size : f->g->V;
    if (n == 0) { *result = 0; return 0; }
    const int nb = (int)std::min<int64_t>(1024, vgl_ceil_div(n, VGL_BLOCK));
    VGL_TRY(vgl_ensure_partials(c, (size_t)nb + 1));
    if (f->sparsity == VGL_HIP_FRONTIER_ALL_ACTIVE)
        hipLaunchKernelGGL((vgl_k_reduce_partial<int32_t, int64_t, 0>), dim3(nb), dim3(VGL_BLOCK), 0, c->stream, n, d_values, f->flags, f->ids, c->d_partials);
    else if (f->sparsity == VGL_HIP_FRONTIER_DENSE)
        hipLaunchKernelGGL((vgl_k_reduce_partial<int32_t, int64_t, 1>), dim3(nb), dim3(VGL_BLOCK), 0, c->stream, n, d_values, f->flags, f->ids, c->d_partials);
    else
        hipLaunchKernelGGL((vgl_k_reduce_partial<int32_t, int64_t, 2>), dim3(nb), dim3(VGL_BLOCK), 0, c->stream, n, d_values, f->flags, f->ids, c->d_partials);
    double r = 0;
    VGL_TRY(vgl_reduce_common(c, nb, &r));
    *result = (int64_t)r;     // exact: every partial is an integer below 2^53
    return 0;
}
int vgl_hip_reduce_sum_f32(vgl_hip_ctx *c, vgl_hip_frontier *f, const float *d_values, double *result)
{
    if (!c || !f || !d_values || !result) VGL_FAIL("reduce: null argument");
    const int32_t n = (f->sparsity == VGL_HIP_FRONTIER_SPARSE) ? f->size : f->g->V;
    if (n == 0) { *result = 0; return 0; }
    const int nb = (int)std::min<int64_t>(1024, vgl_ceil_div(n, VGL_BLOCK));
    VGL_TRY(vgl_ensure_partials(c, (size_t)nb + 1));
    if (f->sparsity == VGL_HIP_FRONTIER_ALL_ACTIVE)
        hipLaunchKernelGGL((vgl_k_reduce_partial<float, double, 0>), dim3(nb), dim3(VGL_BLOCK), 0, c->stream, n, d_values, f->flags, f->ids, c->d_partials);
    else if (f->sparsity == VGL_HIP_FRONTIER_DENSE)
        hipLaunchKernelGGL((vgl_k_reduce_partial<float, double, 1>), dim3(nb), dim3(VGL_BLOCK), 0, c->stream, n, d_values, f->flags, f->ids, c->d_partials);
    else
        hipLaunchKernelGGL((vgl_k_reduce_partial<float, double, 2>), dim3(nb), dim3(VGL_BLOCK), 0, c->stream, n, d_values, f->flags, f->ids, c->d_partials);
    return vgl_reduce_common(c, nb, result);
}
int vgl_hip_count_not_equal_u32(vgl_hip_ctx *c, int32_t n, const void *d_a, const void *d_b, int64_t *result)
{
    if (!c || !d_a || !d_b || !result) VGL_FAIL("count_not_equal: null argument");
    if (n <= 0) { *result = 0; return 0; }
    const int nb = (int)std::min<int64_t>(1024, vgl_ceil_div(n, VGL_BLOCK));
    VGL_TRY(vgl_ensure_partials(c, (size_t)nb + 1));
    hipLaunchKernelGGL(vgl_k_neq_partial, dim3(nb), dim3(VGL_BLOCK), 0, c->stream, n, (const uint32_t *)d_a, (const uint32_t *)d_b, c->d_partials);
    double r = 0;
    VGL_TRY(vgl_reduce_common(c, nb, &r));
    *result = (int64_t)r;
    return 0;
}

}  // extern "C"
