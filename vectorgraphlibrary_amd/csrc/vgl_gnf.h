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

struct vgl_pred_bits {                      // bit v of a bitmap (byte v >> 3, bit v & 7) left by an earlier count pass; v0 is a multiple of 8
    const uint8_t *bytes;
    __device__ uint32_t bits8(int32_t v0, int nvalid, uint32_t *aux) const
    {
        *aux = 0;
        return (uint32_t)bytes[v0 >> 3] & ((1u << nvalid) - 1u);
    }
};
// int32 flags (0 / 1) of a DENSE or ALL_ACTIVE frontier from the bitmap of its count pass: one byte -> eight flags per thread
static __global__ __launch_bounds__(VGL_BLOCK) void vgl_k_bits_to_flags(int32_t nrows, int32_t row_base, const uint8_t *bytes, int32_t *flags)
{
    const int64_t nbytes = ((int64_t)nrows + 7) >> 3;
    for (int64_t i = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; i < nbytes; i += (int64_t)gridDim.x * VGL_BLOCK) {
        const int32_t v0 = row_base + (int32_t)(i << 3);
        const uint32_t bits = bytes[v0 >> 3];
        if (v0 + 8 <= row_base + nrows && (v0 & 7) == 0) {
            *reinterpret_cast<int4 *>(flags + v0) = make_int4(bits & 1, (bits >> 1) & 1, (bits >> 2) & 1, (bits >> 3) & 1);
            *reinterpret_cast<int4 *>(flags + v0 + 4) = make_int4((bits >> 4) & 1, (bits >> 5) & 1, (bits >> 6) & 1, (bits >> 7) & 1);
        } else
            for (int j = 0; v0 + j < row_base + nrows && j < 8; j++) flags[v0 + j] = (bits >> j) & 1;
    }
}

// count pass.  Workgroup = tile of 2048 owned vertices, thread = 8 consecutive vertices.
// front_bytes (optional): bitmap of active vertices (byte v>>3, bit v&7; little-endian uint64 words).
// visited_bytes (optional): bitmap of the predicate's auxiliary bits (fused BFS: values[v] != -1).
// flags_out (optional): int32 0/1 per vertex (VGL frontier flags).
// ticket (optional): the last workgroup to finish also does the scan pass (exclusive offsets per tile, totals, offs[size], hand-over to the
// host when `host` is given) -- a scan kernel of its own, however small, waits for this kernel's caches to be written back and
// invalidated before it starts: 18 us per frontier generation on RMAT-24, in every super-step of every frontier-driven algorithm.
// the VGL_EPT + 1 row offsets of a thread's eight consecutive rows in ONE round of loads (72 contiguous bytes): fetched one by one under
// `if (bit j)` each was a branch awaited before the next -- up to eight dependent round trips for the same one or two cache lines.
// nvalid < VGL_EPT (the last rows of the graph): entries past nvalid repeat the last offset (bits8 reports no such row).
typedef long long vgl_ll2_u __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void vgl_load_row_offsets(const int64_t *p, int nvalid, int64_t (&rp)[VGL_EPT + 1])
{
    static_assert(VGL_EPT == 8, "eight rows per thread");
    if (nvalid == VGL_EPT) {
        const vgl_ll2_u a = *reinterpret_cast<const vgl_ll2_u *>(p), b = *reinterpret_cast<const vgl_ll2_u *>(p + 2),
                        c = *reinterpret_cast<const vgl_ll2_u *>(p + 4), d = *reinterpret_cast<const vgl_ll2_u *>(p + 6);
        const int64_t e = p[8];
        rp[0] = a.x; rp[1] = a.y; rp[2] = b.x; rp[3] = b.y; rp[4] = c.x; rp[5] = c.y; rp[6] = d.x; rp[7] = d.y; rp[8] = e;
    } else {
#pragma unroll
        for (int j = 0; j <= VGL_EPT; j++) rp[j] = p[min(j, nvalid)];
    }
}

template <class Pred>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_gnf_count(Pred pred, int32_t nrows, int32_t row_base, const int64_t *rowptr,
                                                             int32_t *vt_cnt, int64_t *vt_deg, uint8_t *front_bytes,
                                                             uint8_t *visited_bytes, int32_t *flags_out, uint32_t *ticket,
                                                             int32_t *vt_cnt_off, int64_t *vt_deg_off, int64_t *counters, int64_t *offs,
                                                             volatile int64_t *host, int64_t seq)
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
            int64_t rp[VGL_EPT + 1];
            vgl_load_row_offsets(rowptr + r0, nvalid, rp);
#pragma unroll
            for (int j = 0; j < VGL_EPT; j++)
                if ((bits >> j) & 1) deg += rp[j + 1] - rp[j];
        }
        if (front_bytes) front_bytes[v0 >> 3] = (uint8_t)bits;
        if (visited_bytes) visited_bytes[v0 >> 3] = (uint8_t)aux;
        if (flags_out) {
            if (nvalid == 8 && (v0 & 7) == 0) {         // 32-byte aligned: two 16-byte stores
                *reinterpret_cast<int4 *>(flags_out + v0) = make_int4(bits & 1, (bits >> 1) & 1, (bits >> 2) & 1, (bits >> 3) & 1);
                *reinterpret_cast<int4 *>(flags_out + v0 + 4) = make_int4((bits >> 4) & 1, (bits >> 5) & 1, (bits >> 6) & 1, (bits >> 7) & 1);
            } else
                for (int j = 0; j < nvalid; j++) flags_out[v0 + j] = (bits >> j) & 1;
        }
    }
    const int tc = vgl_block_reduce_add(cnt, s32);
    const int64_t td = vgl_block_reduce_add(deg, s64);
    if (!ticket) {
        if (threadIdx.x == 0) { vt_cnt[blockIdx.x] = tc; vt_deg[blockIdx.x] = td; }
        return;
    }
    uint32_t dep = 0;
    if (threadIdx.x == 0) dep = vgl_put_agent(vt_cnt + blockIdx.x, tc) ^ vgl_put_agent(vt_deg + blockIdx.x, td);
    if (!vgl_last_block(ticket, dep)) return;
    // last workgroup: a contiguous run of tiles per thread, eight loads in flight at a time
    const int nt = (int)gridDim.x;
    const int per = (nt + VGL_BLOCK - 1) / VGL_BLOCK;
    const int lo = min(nt, (int)threadIdx.x * per), hi = min(nt, lo + per);
    int c = 0;
    int64_t d = 0;
    for (int t0 = lo; t0 < hi; t0 += 8) {
        int cv[8];
        int64_t dv[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int t = min(t0 + j, hi - 1);
            cv[j] = vgl_load_agent(vt_cnt + t);
            dv[j] = vgl_load_agent(vt_deg + t);
        }
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (t0 + j < hi) { c += cv[j]; d += dv[j]; }
    }
    int ctot;
    int64_t dtot;
    int cpre = vgl_block_excl_add(c, s32, &ctot);
    int64_t dpre = vgl_block_excl_add(d, s64, &dtot);
    for (int t0 = lo; t0 < hi; t0 += 8) {
        int cv[8];
        int64_t dv[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int t = min(t0 + j, hi - 1);
            cv[j] = vgl_load_agent(vt_cnt + t);
            dv[j] = vgl_load_agent(vt_deg + t);
        }
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (t0 + j < hi) { vt_cnt_off[t0 + j] = cpre; vt_deg_off[t0 + j] = dpre; cpre += cv[j]; dpre += dv[j]; }
    }
    if (threadIdx.x == 0) {
        if (offs) offs[ctot] = dtot;
        counters[C_FRONT] = ctot; counters[C_NEIGH] = dtot;
    }
    if (host) {                                             // all slots, as vgl_k_publish would (callers read more than these two)
        if (threadIdx.x < C_NSLOTS)
            host[threadIdx.x] = threadIdx.x == C_FRONT ? (int64_t)ctot : threadIdx.x == C_NEIGH ? dtot : counters[threadIdx.x];
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) { host[C_NSLOTS] = seq; __threadfence_system(); }
    }
}

// scan pass: single workgroup of 1024 threads; exclusive offsets per tile; totals to counters[C_FRONT], counters[C_NEIGH];
// also terminates the edge-offset array: offs[size] = neighbours.  A thread takes eight consecutive tiles per round and loads them before
// it adds anything (one entry per loop iteration was a chain of 2 x ntiles / 1024 dependent L2 round trips: 19 us for the 8192 tiles of
// RMAT-24, in every super-step of every frontier-driven algorithm); rounds of 8192 tiles carry their totals forward.  host (optional):
// all counter slots and then the sequence number go to the pinned mirror from here, instead of from a vgl_k_publish launch of their own.
constexpr int VGL_SCAN_THREADS = 1024;
constexpr int VGL_SCAN_PER = 8;
static __global__ __launch_bounds__(VGL_SCAN_THREADS) void vgl_k_gnf_scan(int64_t ntiles, const int32_t *vt_cnt, const int64_t *vt_deg,
                                                                   int32_t *vt_cnt_off, int64_t *vt_deg_off, int64_t *counters,
                                                                   int64_t *offs, volatile int64_t *host, int64_t seq)
{
    __shared__ int64_t s_c[VGL_SCAN_THREADS / 64], s_d[VGL_SCAN_THREADS / 64];
    const int w = threadIdx.x >> 6;
    int64_t ccarry = 0, dcarry = 0;
    for (int64_t r0 = 0; r0 < ntiles; r0 += (int64_t)VGL_SCAN_THREADS * VGL_SCAN_PER) {
        const int64_t lo = r0 + (int64_t)threadIdx.x * VGL_SCAN_PER;
        int32_t cv[VGL_SCAN_PER];
        int64_t dv[VGL_SCAN_PER];
#pragma unroll
        for (int j = 0; j < VGL_SCAN_PER; j++) {            // (the tables have room for ntiles entries only: no vector loads across the end)
            const bool in = lo + j < ntiles;
            cv[j] = in ? vt_cnt[lo + j] : 0;
            dv[j] = in ? vt_deg[lo + j] : 0;
        }
        int64_t c = 0, d = 0;
#pragma unroll
        for (int j = 0; j < VGL_SCAN_PER; j++) { c += cv[j]; d += dv[j]; }
        const int64_t ci = vgl_wave_incl_add(c), di = vgl_wave_incl_add(d);
        __syncthreads();                                    // (s_c / s_d of the round before have been read)
        if ((threadIdx.x & 63) == 63) { s_c[w] = ci; s_d[w] = di; }
        __syncthreads();
        int64_t cb = 0, db = 0, ctot = 0, dtot = 0;
#pragma unroll
        for (int i = 0; i < VGL_SCAN_THREADS / 64; i++) {
            if (i < w) { cb += s_c[i]; db += s_d[i]; }
            ctot += s_c[i]; dtot += s_d[i];
        }
        int64_t cpre = ccarry + cb + ci - c, dpre = dcarry + db + di - d;
#pragma unroll
        for (int j = 0; j < VGL_SCAN_PER; j++) {
            if (lo + j < ntiles) { vt_cnt_off[lo + j] = (int32_t)cpre; vt_deg_off[lo + j] = dpre; }
            cpre += cv[j]; dpre += dv[j];
        }
        ccarry += ctot; dcarry += dtot;
    }
    if (threadIdx.x == 0) {
        counters[C_FRONT] = ccarry; counters[C_NEIGH] = dcarry;
        if (offs) offs[ccarry] = dcarry;
    }
    if (host) {                                             // what vgl_k_publish does (context.hip)
        __syncthreads();
        if (threadIdx.x < C_NSLOTS) host[threadIdx.x] = threadIdx.x == C_FRONT ? ccarry : threadIdx.x == C_NEIGH ? dcarry : counters[threadIdx.x];
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) { host[C_NSLOTS] = seq; __threadfence_system(); }
    }
}

// write pass: ids[pos] = v ascending; offs[pos] = exclusive sum of degrees (optional).  tile_first (optional, with offs; M = the degree sum of
// the whole frontier): the frontier position that owns edge t * VGL_TILE of the frontier's edge space, and in entry [#tiles] the owner of the last
// edge -- every frontier vertex knows its own edge range, so the advance that follows needs no table pass of its own (vgl_k_plan_tile_first).
template <class Pred>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_gnf_write(Pred pred, int32_t nrows, int32_t row_base, const int64_t *rowptr,
                                                             const int32_t *vt_cnt_off, const int64_t *vt_deg_off,
                                                             int32_t *ids, int64_t *offs, int32_t *tile_first, int64_t M)
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
            int64_t rp[VGL_EPT + 1];
            vgl_load_row_offsets(rowptr + r0, nvalid, rp);
#pragma unroll
            for (int j = 0; j < VGL_EPT; j++) {
                degs[j] = 0;
                if ((bits >> j) & 1) { degs[j] = rp[j + 1] - rp[j]; deg += degs[j]; }
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
                if (offs) {
                    offs[pos] = eoff;
                    const int64_t eend = eoff + degs[j];
                    if (tile_first) {
                        for (int64_t t = (eoff + VGL_TILE - 1) / VGL_TILE; t < (eend + VGL_TILE - 1) / VGL_TILE; t++) tile_first[t] = pos;
                        if (eoff < eend && eend == M) tile_first[(M + VGL_TILE - 1) / VGL_TILE] = pos;
                    }
                    eoff = eend;
                }
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
    // up to 16384 tiles (32 M owned vertices) the count kernel's last workgroup scans (64 tiles per thread at most); beyond that the
    // 1024-thread scan kernel does.  Either hands F and M to the host itself when they are wanted (read_back) -- the host then has them
    // while the write pass is still running.
    const bool fused_scan = nt <= 16384;
    const int64_t seq = read_back ? vgl_next_seq(c) : 0;
    volatile int64_t *host = read_back ? (volatile int64_t *)c->h_counters : (volatile int64_t *)nullptr;
    {
        vgl_timed_launch tl(c, "gnf");
        hipLaunchKernelGGL(vgl_k_gnf_count<Pred>, dim3((unsigned)nt), dim3(VGL_BLOCK), 0, c->stream, pred, g->nrows, g->row_begin,
                           g->out.rowptr, g->vt_cnt, g->vt_deg, front_bytes, visited_bytes, flags_out,
                           fused_scan ? g->tickets + 0 * VGL_TICKET_WORDS : (uint32_t *)nullptr, g->vt_cnt_off, g->vt_deg_off, c->d_counters, offs,
                           host, seq);
    }
    if (!fused_scan)
        hipLaunchKernelGGL(vgl_k_gnf_scan, dim3(1), dim3(VGL_SCAN_THREADS), 0, c->stream, nt, g->vt_cnt, g->vt_deg, g->vt_cnt_off,
                           g->vt_deg_off, c->d_counters, offs, host, seq);
    if (write_ids) {
        vgl_timed_launch tl(c, "gnf");
        hipLaunchKernelGGL(vgl_k_gnf_write<Pred>, dim3((unsigned)nt), dim3(VGL_BLOCK), 0, c->stream, pred, g->nrows, g->row_begin,
                           g->out.rowptr, g->vt_cnt_off, g->vt_deg_off, ids, offs, (int32_t *)nullptr, (int64_t)0);
    }
    VGL_HIP_TRY(hipGetLastError());
    if (read_back) VGL_TRY(vgl_wait_counters(c, seq));
    return 0;
}
#endif
