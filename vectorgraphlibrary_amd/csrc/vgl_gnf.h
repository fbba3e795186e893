// vgl_gnf.h -- generate_new_frontier kernels (multicore/generate_new_frontier.hpp:113-164 + copy_if.hpp:128-191),
// shared by the generic frontier API and the fused BFS.  Two passes over the owned vertices:
//   count : per 2048-vertex tile, number of active vertices and sum of their degrees (+ optional bitmaps / flags)
//   scan  : one workgroup turns the per-tile counts into offsets and the totals (size, neighbours)
//   write : ascending-id compaction of vertex ids and (optionally) exclusive edge offsets of the new frontier
// Algorithmic traffic per call: 4 B/vertex predicate read per pass (+16 B row offsets for ACTIVE vertices only).
#pragma once
#include "vgl_hip_internal.h"

#ifdef __HIPCC__
// A predicate functor provides:  __device__ uint32_t bits8(int32_t v0, int nvalid, uint32_t *aux) const
// returning bit j set iff global vertex v0+j is active (only j < nvalid are evaluated).  *aux receives a second bit set
// derived from the same loads (the fused BFS uses it for "visited": values[v] != -1), so the count pass reads 4 B/vertex.

struct vgl_pred_equal_i32 {                 // values[v] == value   (BFS on_next_level, bfs.hpp:40-45)
    const int32_t *values; int32_t value;
    __device__ uint32_t bits8(int32_t v0, int nvalid, uint32_t *aux) const
    {
        uint32_t b = 0, n = 0;
        if (nvalid == 8) {                  // v0 is a multiple of 8 => 32-byte aligned, two 16-byte loads
            const int4 a = *reinterpret_cast<const int4 *>(values + v0);
            const int4 c = *reinterpret_cast<const int4 *>(values + v0 + 4);
            b = (a.x == value) | ((a.y == value) << 1) | ((a.z == value) << 2) | ((a.w == value) << 3) |
                ((c.x == value) << 4) | ((c.y == value) << 5) | ((c.z == value) << 6) | ((c.w == value) << 7);
            n = (a.x != -1) | ((a.y != -1) << 1) | ((a.z != -1) << 2) | ((a.w != -1) << 3) |
                ((c.x != -1) << 4) | ((c.y != -1) << 5) | ((c.z != -1) << 6) | ((c.w != -1) << 7);
        } else {
            for (int j = 0; j < nvalid; j++) {
                const int32_t x = values[v0 + j];
                b |= (uint32_t)(x == value) << j;
                n |= (uint32_t)(x != -1) << j;
            }
        }
        *aux = n;
        return b;
    }
};
struct vgl_pred_nonzero_i32 {               // flags[v] != 0
    const int32_t *flags;
    __device__ uint32_t bits8(int32_t v0, int nvalid, uint32_t *aux) const
    {
        uint32_t b = 0;
        *aux = 0;
        if (nvalid == 8) {
            const int4 a = *reinterpret_cast<const int4 *>(flags + v0);
            const int4 c = *reinterpret_cast<const int4 *>(flags + v0 + 4);
            b = (a.x != 0) | ((a.y != 0) << 1) | ((a.z != 0) << 2) | ((a.w != 0) << 3) |
                ((c.x != 0) << 4) | ((c.y != 0) << 5) | ((c.z != 0) << 6) | ((c.w != 0) << 7);
        } else {
            for (int j = 0; j < nvalid; j++) b |= (uint32_t)(flags[v0 + j] != 0) << j;
        }
        return b;
    }
};

// count pass.  Workgroup = tile of 2048 owned vertices, thread = 8 consecutive vertices.
// front_bytes (optional): bitmap of active vertices (byte v>>3, bit v&7; little-endian uint64 words).
// visited_bytes (optional): bitmap of the predicate's auxiliary bits (fused BFS: values[v] != -1).
// flags_out (optional): int32 0/1 per vertex (VGL frontier flags).
template <class Pred>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_gnf_count(Pred pred, int32_t nrows, int32_t row_base, const int64_t *rowptr,
                                                             int32_t *vt_cnt, int64_t *vt_deg, uint8_t *front_bytes,
                                                             uint8_t *visited_bytes, int32_t *flags_out)
{
    __shared__ int64_t s64[VGL_WAVES];
    __shared__ int s32[VGL_WAVES];
    const int32_t r0 = blockIdx.x * VGL_TILE + threadIdx.x * VGL_EPT;   // local row
    int cnt = 0;
    int64_t deg = 0;
    if (r0 < nrows) {
        const int nvalid = min(VGL_EPT, nrows - r0);
        const int32_t v0 = row_base + r0;
        uint32_t aux;
        const uint32_t bits = pred.bits8(v0, nvalid, &aux);
        cnt = __popc(bits);
        if (bits) {
            for (int j = 0; j < nvalid; j++)
                if ((bits >> j) & 1) deg += rowptr[r0 + j + 1] - rowptr[r0 + j];
        }
        if (front_bytes) front_bytes[v0 >> 3] = (uint8_t)bits;
        if (visited_bytes) visited_bytes[v0 >> 3] = (uint8_t)aux;
        if (flags_out)
            for (int j = 0; j < nvalid; j++) flags_out[v0 + j] = (bits >> j) & 1;
    }
    const int tc = vgl_block_reduce_add(cnt, s32);
    const int64_t td = vgl_block_reduce_add(deg, s64);
    if (threadIdx.x == 0) { vt_cnt[blockIdx.x] = tc; vt_deg[blockIdx.x] = td; }
}

// scan pass: single workgroup of 1024 threads; exclusive offsets per tile; totals to counters[C_FRONT], counters[C_NEIGH];
// also terminates the edge-offset array: offs[size] = neighbours.
constexpr int VGL_SCAN_THREADS = 1024;
static __global__ __launch_bounds__(VGL_SCAN_THREADS) void vgl_k_gnf_scan(int64_t ntiles, const int32_t *vt_cnt, const int64_t *vt_deg,
                                                                   int32_t *vt_cnt_off, int64_t *vt_deg_off, int64_t *counters,
                                                                   int64_t *offs)
{
    __shared__ int64_t s_c[VGL_SCAN_THREADS / 64], s_d[VGL_SCAN_THREADS / 64];
    const int64_t per = (ntiles + VGL_SCAN_THREADS - 1) / VGL_SCAN_THREADS;
    const int64_t lo = min(ntiles, (int64_t)threadIdx.x * per), hi = min(ntiles, lo + per);
    int64_t c = 0, d = 0;
    for (int64_t t = lo; t < hi; t++) { c += vt_cnt[t]; d += vt_deg[t]; }
    const int64_t ci = vgl_wave_incl_add(c), di = vgl_wave_incl_add(d);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) { s_c[w] = ci; s_d[w] = di; }
    __syncthreads();
    int64_t cb = 0, db = 0, ctot = 0, dtot = 0;
#pragma unroll
    for (int i = 0; i < VGL_SCAN_THREADS / 64; i++) {
        if (i < w) { cb += s_c[i]; db += s_d[i]; }
        ctot += s_c[i]; dtot += s_d[i];
    }
    int64_t cpre = cb + ci - c, dpre = db + di - d;
    for (int64_t t = lo; t < hi; t++) {
        vt_cnt_off[t] = (int32_t)cpre; vt_deg_off[t] = dpre;
        cpre += vt_cnt[t]; dpre += vt_deg[t];
    }
    if (threadIdx.x == 0) {
        counters[C_FRONT] = ctot; counters[C_NEIGH] = dtot;
        if (offs) offs[ctot] = dtot;
    }
}

// write pass: ids[pos] = v ascending; offs[pos] = exclusive sum of degrees (optional)
template <class Pred>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_gnf_write(Pred pred, int32_t nrows, int32_t row_base, const int64_t *rowptr,
                                                             const int32_t *vt_cnt_off, const int64_t *vt_deg_off,
                                                             int32_t *ids, int64_t *offs)
{
    __shared__ int64_t s64[VGL_WAVES];
    __shared__ int s32[VGL_WAVES];
    const int32_t r0 = blockIdx.x * VGL_TILE + threadIdx.x * VGL_EPT;
    uint32_t bits = 0;
    int nvalid = 0;
    int64_t degs[VGL_EPT];
    int64_t deg = 0;
    if (r0 < nrows) {
        nvalid = min(VGL_EPT, nrows - r0);
        uint32_t aux;
        bits = pred.bits8(row_base + r0, nvalid, &aux);
        if (bits && offs) {
#pragma unroll
            for (int j = 0; j < VGL_EPT; j++) {
                degs[j] = 0;
                if (j < nvalid && ((bits >> j) & 1)) { degs[j] = rowptr[r0 + j + 1] - rowptr[r0 + j]; deg += degs[j]; }
            }
        }
    }
    int ctot; int64_t dtot;
    int pos = vt_cnt_off[blockIdx.x] + vgl_block_excl_add((int)__popc(bits), s32, &ctot);
    int64_t eoff = 0;
    if (offs) eoff = vt_deg_off[blockIdx.x] + vgl_block_excl_add(deg, s64, &dtot);
    if (bits) {
#pragma unroll
        for (int j = 0; j < VGL_EPT; j++) {
            if ((bits >> j) & 1) {
                ids[pos] = row_base + r0 + j;
                if (offs) { offs[pos] = eoff; eoff += degs[j]; }
                pos++;
            }
        }
    }
}

// host-side driver of the three passes; results (size, neighbours) land in ctx->h_counters[C_FRONT/C_NEIGH]
// when read_back is true (synchronises).
template <class Pred>
static int vgl_gnf_run(vgl_hip_ctx *c, vgl_hip_graph *g, Pred pred, int32_t *ids, int64_t *offs,
                       uint8_t *front_bytes, uint8_t *visited_bytes, int32_t *flags_out,
                       bool write_ids, bool read_back)
{
    const int64_t nt = g->nvtiles;
    {
        vgl_timed_launch tl(c, "gnf");
        hipLaunchKernelGGL(vgl_k_gnf_count<Pred>, dim3((unsigned)nt), dim3(VGL_BLOCK), 0, c->stream, pred, g->nrows, g->row_begin,
                           g->out.rowptr, g->vt_cnt, g->vt_deg, front_bytes, visited_bytes, flags_out);
    }
    hipLaunchKernelGGL(vgl_k_gnf_scan, dim3(1), dim3(VGL_SCAN_THREADS), 0, c->stream, nt, g->vt_cnt, g->vt_deg, g->vt_cnt_off,
                       g->vt_deg_off, c->d_counters, offs);
    if (write_ids) {
        vgl_timed_launch tl(c, "gnf");
        hipLaunchKernelGGL(vgl_k_gnf_write<Pred>, dim3((unsigned)nt), dim3(VGL_BLOCK), 0, c->stream, pred, g->nrows, g->row_begin,
                           g->out.rowptr, g->vt_cnt_off, g->vt_deg_off, ids, offs);
    }
    VGL_HIP_TRY(hipGetLastError());
    if (read_back) VGL_TRY(vgl_read_counters(c, false));
    return 0;
}
#endif
