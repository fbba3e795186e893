// vgl_hip_kernels.hpp -- the templated HIP kernels of the operator API that call USER operators (device lambdas): advance over static edge
// tiles / over a sparse frontier's edge space / over sequential rows, per-vertex operators, reductions, the frontier predicate adaptor, the
// merge kernels of exchange_vertices_array, the degree-class splitters of the six-functor advance.  They depend on plain pointers only --
// not on any graph / frontier / array class -- so the same file serves this repository's own minimal classes (vgl_hip.hpp) and the
// backend class bound to the REFERENCE's containers (integration/vgl_compute_api/hip/graph_abstractions_hip.h: CSRGraph,
// VectorCSRGraph, FrontierCSR, FrontierVectorCSR through friend access, vgl_compute_api/template/graph_abstractions_template.h:5-107).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>
#include "../../include/vgl_hip.h"
#include "../csrc/vgl_hip_internal.h"
#include "../csrc/vgl_gnf.h"

// VGL_SRC_ID_ADD (architecture_independent_api.h:48): "+= into the source vertex's slot" from an edge operator.  The edges of a row
// sit in consecutive lanes, so when a wavefront walks a hub every lane adds to the SAME address and plain atomics serialise (~12 ns
// each: 0.7 ms per PageRank iteration for one 60 K-edge hub).  When all active lanes agree on the address the wavefront sums its
// values in lane order (v_readlane, a uniform loop) and issues one atomic; otherwise every lane adds on its own.
template <class T> __device__ __forceinline__ T vgl_lane_value(T v, int l);
template <> __device__ __forceinline__ float vgl_lane_value<float>(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
template <> __device__ __forceinline__ int vgl_lane_value<int>(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
template <> __device__ __forceinline__ double vgl_lane_value<double>(double v, int l)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <class T, class U>
__device__ __forceinline__ void vgl_src_id_add(T &slot, U value)
{
    T *addr = &slot;
    const T val = (T)value;
    const unsigned long long active = __ballot(1);
    const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)active) - 1);
    const unsigned long long mine = (unsigned long long)(uintptr_t)addr;
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mine, leader), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mine >> 32), leader);
    const bool same = mine == (((unsigned long long)hi << 32) | lo);
    if (__ballot(same) == active && __popcll(active) >= 8) {
        T sum = (T)0;
        unsigned long long m = active;
        while (m) {                                            // wave-uniform loop; only active lanes are read
            const int l = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
            m &= m - 1;
            sum += vgl_lane_value<T>(val, l);
        }
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(addr, sum);
    } else atomicAdd(addr, val);
}

// ------------------------------------------------------------------------------------------------------------------
// templated kernels calling user operators
// ------------------------------------------------------------------------------------------------------------------
// advance over ALL edges of a direction (ALL_ACTIVE) or over the rows flagged in a DENSE frontier: static edge tiles
template <bool DENSE, class EdgeOp>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_advance_static(const long long *rowptr, const int *adj, const int32_t *tile_row,
                                                                  long long E, long long process_shift, const int *flags, int row_lo, int row_hi,
                                                                  EdgeOp edge_op)
{
    __shared__ int s_map[VGL_TILE];
    __shared__ int s_w[VGL_WAVES];
    const int64_t e0 = (int64_t)blockIdx.x * VGL_TILE;
    const int n = (int)min((int64_t)VGL_TILE, (int64_t)E - e0);
    const int r_first = tile_row[blockIdx.x];
    const int r_last = tile_row[blockIdx.x + 1];
    if (r_last < row_lo || r_first >= row_hi) return;       // (several ranks: tiles outside this rank's vertex range)
    vgl_tile_row_map(s_map, s_w, (const int64_t *)rowptr, e0, r_first, r_last);
#pragma unroll
    for (int j = 0; j < VGL_EPT; j++) {
        const int i = threadIdx.x + j * VGL_BLOCK;
        if (i < n) {
            const int src = r_first + s_map[i];
            if (src >= row_lo && src < row_hi && (!DENSE || flags[src] > 0)) {
                const long long e = e0 + i;
                edge_op(src, adj[e], (int)(e - rowptr[src]), process_shift + e, (int)(threadIdx.x & 63));
            }
        }
    }
}
// advance over a SPARSE frontier: tiles of the frontier's own edge list (plan = offs + tile_first).  Per frontier position of the tile
// the source id, the offset of its first edge inside the tile and (row start - frontier edge offset) are staged in LDS, so that an edge
// costs LDS lookups plus its adjacency load instead of three dependent global loads (ids[p], offs[p], rowptr[src]); tiles spanning more
// than VGL_ADV_STAGE positions (thousands of empty or one-edge rows) read them from memory.
constexpr int VGL_ADV_STAGE = 1024;
#ifndef VGL_ADV_THREADS_VALUE
#define VGL_ADV_THREADS_VALUE 512
#endif
constexpr int VGL_ADV_THREADS = VGL_ADV_THREADS_VALUE;   // threads per 2048-edge tile: four edges each -- the per-edge chains (adjacency, then whatever the
                                               // user's operator gathers and stores) of a thread cannot overlap (its stores may alias its next
                                               // loads), so more, shorter chains per tile keep more requests in flight than 256 threads x 8
// vgl_tile_row_map (csrc/vgl_hip_internal.h) for a workgroup of THREADS threads
template <int THREADS>
__device__ __forceinline__ void vgl_tile_row_map_wide(int *s_map, int *s_w, const int64_t *starts, int64_t e0, int r_first, int r_last)
{
    constexpr int EPT = VGL_TILE / THREADS, WAVES = THREADS / 64;
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < EPT; j++) s_map[tid + j * THREADS] = 0;
    __syncthreads();
    for (int r = r_first + 1 + tid; r <= r_last; r += THREADS) {
        const int64_t q = starts[r] - e0;          // > 0 because r_first contains e0
        if (q < VGL_TILE) atomicMax(&s_map[(int)q], r - r_first);
    }
    __syncthreads();
    int m[EPT];
    int run = 0;
#pragma unroll
    for (int j = 0; j < EPT; j++) { run = max(run, s_map[tid * EPT + j]); m[j] = run; }
    const int inc = vgl_wave_incl_max(run);
    if ((tid & 63) == 63) s_w[tid >> 6] = inc;
    __syncthreads();
    int pre = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) if (w < (tid >> 6)) pre = max(pre, s_w[w]);
    int up = __shfl_up(inc, 1);
    if ((tid & 63) == 0) up = 0;
    pre = max(pre, up);
#pragma unroll
    for (int j = 0; j < EPT; j++) s_map[tid * EPT + j] = max(m[j], pre);
    __syncthreads();
}
template <class EdgeOp>
__global__ __launch_bounds__(VGL_ADV_THREADS) void vgl_k_advance_sparse(const int *ids, const int64_t *offs, const int32_t *tile_first, int F,
                                                                        long long M, const long long *rowptr, const int *adj,
                                                                        long long process_shift, int row_lo, int row_hi, EdgeOp edge_op)
{
    constexpr int EPT = VGL_TILE / VGL_ADV_THREADS;
    __shared__ int s_map[VGL_TILE];
    __shared__ int s_w[VGL_ADV_THREADS / 64];
    __shared__ int s_src[VGL_ADV_STAGE];
    __shared__ int s_first[VGL_ADV_STAGE];
    __shared__ long long s_base[VGL_ADV_STAGE];
    const int64_t e0 = (int64_t)blockIdx.x * VGL_TILE;
    const int n = (int)min((int64_t)VGL_TILE, (int64_t)M - e0);
    const int p_first = tile_first[blockIdx.x];
    const int p_last = tile_first[blockIdx.x + 1];      // last tile: owner of the last edge (written by the plan)
    const int np = p_last - p_first + 1;
    const bool staged = np <= VGL_ADV_STAGE;
    if (staged)
        for (int k = threadIdx.x; k < np; k += VGL_ADV_THREADS) {
            const int p = p_first + k;
            const int src = ids[p];
            const int64_t o = offs[p];
            s_src[k] = src; s_first[k] = (int)(o - e0); s_base[k] = rowptr[src] - o;
        }
    vgl_tile_row_map_wide<VGL_ADV_THREADS>(s_map, s_w, offs, e0, p_first, p_last);      // ends with a barrier: the staged arrays are visible too
    int srcs[EPT], locals[EPT], dsts[EPT];
    long long es[EPT];
#pragma unroll
    for (int j = 0; j < EPT; j++) {                     // every adjacency load of the thread is issued before the first operator call
        const int i = threadIdx.x + j * VGL_ADV_THREADS;
        srcs[j] = -1;
        if (i < n) {
            const int k = s_map[i];
            if (staged) { srcs[j] = s_src[k]; locals[j] = i - s_first[k]; es[j] = s_base[k] + e0 + i; }
            else { const int p = p_first + k; srcs[j] = ids[p]; locals[j] = (int)(e0 + i - offs[p]); es[j] = rowptr[srcs[j]] + locals[j]; }
            if (srcs[j] < row_lo || srcs[j] >= row_hi) srcs[j] = -1;
            else dsts[j] = adj[es[j]];
        }
    }
#pragma unroll
    for (int j = 0; j < EPT; j++)
        if (srcs[j] >= 0) edge_op(srcs[j], dsts[j], locals[j], process_shift + es[j], (int)(threadIdx.x & 63));
}
// advance with SEQUENTIAL ROWS (GraphAbstractionsHIP::enable_sequential_rows): one lane walks the edges of an active vertex in adjacency order
// between its pre and post operators -- the execution shape of the reference's vector-core kernels (multicore/advance_worker.hpp:62-149: one
// thread per vertex), whose algorithms accumulate into per-vertex state without atomics and therefore sum in adjacency order.  An operator that
// only stores to src-indexed data needs no atomics here, and its sums are the reference's chains bit for bit (apps/algorithms/pr.hpp).  No load
// balancing: a hub row is one lane's loop -- meant for graphs without hubs or for results that must not depend on the schedule.
template <int MODE, class EdgeOp, class PreOp, class PostOp>     // 0 all-active, 1 dense (flags), 2 sparse (ids)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_advance_rows(int n, const long long *rowptr, const int *adj, const int *flags, const int *ids, long long process_shift,
                                                                int row_lo, int row_hi, EdgeOp edge_op, PreOp pre_op, PostOp post_op)
{
    if (MODE == 2) {                                    // listed vertices: rows anywhere, adjacency read by the lane itself
        for (int i = blockIdx.x * VGL_BLOCK + threadIdx.x; i < n; i += gridDim.x * VGL_BLOCK) {
            const int src = ids[i];
            if (src < row_lo || src >= row_hi) continue;
            const long long lo = rowptr[src], hi = rowptr[src + 1];
            const int lane = (int)(threadIdx.x & 63);
            pre_op(src, (int)(hi - lo), lane);
            for (long long e = lo; e < hi; e++) edge_op(src, adj[e], (int)(e - lo), process_shift + e, lane);
            post_op(src, (int)(hi - lo), lane);
        }
        return;
    }
    // all vertices / flagged vertices: a workgroup owns 256 consecutive rows, i.e. one contiguous run of the adjacency, and streams it through
    // LDS in chunks (coalesced loads); every lane then takes its row's entries of the chunk from LDS, in order.  (A lane reading its own row from
    // memory touches a line of its own per step: the PageRank pull fetched 13.8 x its algorithmic bytes that way, profiles/r04_operator_roofline.json.)
    constexpr int CHUNK = 4096;
    __shared__ int s_adj[CHUNK];
    const int lane = (int)(threadIdx.x & 63);
    for (int r0 = blockIdx.x * VGL_BLOCK; r0 < n; r0 += gridDim.x * VGL_BLOCK) {
        const int src = r0 + (int)threadIdx.x;
        const bool mine = src < n && src >= row_lo && src < row_hi && (MODE != 1 || flags[src] > 0);
        long long lo = 0, hi = 0;
        if (src < n) { lo = rowptr[src]; hi = rowptr[src + 1]; }
        const long long e_first = rowptr[r0], e_last = rowptr[min(r0 + VGL_BLOCK, n)];
        if (mine) pre_op(src, (int)(hi - lo), lane);
        for (long long c0 = e_first; c0 < e_last; c0 += CHUNK) {
            const int m = (int)min((long long)CHUNK, e_last - c0);
            __syncthreads();                            // (the chunk before has been consumed)
            for (int k = threadIdx.x; k < m; k += VGL_BLOCK) s_adj[k] = adj[c0 + k];
            __syncthreads();
            if (mine) {
                const long long a = max(lo, c0), b = min(hi, c0 + m);
                for (long long e = a; e < b; e++) edge_op(src, s_adj[(int)(e - c0)], (int)(e - lo), process_shift + e, lane);
            }
        }
        if (mine) post_op(src, (int)(hi - lo), lane);
    }
}
// per-vertex operator over all vertices / flagged vertices / listed vertices (compute_worker, multicore/compute.hpp:6-58)
template <int MODE, class Op>     // 0 all-active, 1 dense (flags), 2 sparse (ids)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_vertex_op(int n, const long long *rowptr, const int *flags, const int *ids, int row_lo, int row_hi, Op op)
{
    for (int i = blockIdx.x * VGL_BLOCK + threadIdx.x; i < n; i += gridDim.x * VGL_BLOCK) {
        if (MODE == 1 && flags[i] <= 0) continue;
        const int src = (MODE == 2) ? ids[i] : i;
        if (src < row_lo || src >= row_hi) continue;
        op(src, (int)(rowptr[src + 1] - rowptr[src]), (int)(threadIdx.x & 63));
    }
}
// reduce_worker (multicore/reduce.hpp:6-152) in ONE pass: every workgroup folds the reduce_op values of its vertices (f64: exact for int /
// float operands) in a fixed tree and leaves one partial; MAX = max(0, values), the reference's definition (reduce.hpp:80).
template <int MODE, bool IS_MAX, class Op>     // 0 all-active, 1 dense (flags), 2 sparse (ids)
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_reduce_partials(int n, const long long *rowptr, const int *flags, const int *ids, Op op, double *partials)
{
    __shared__ double s[VGL_BLOCK / 64];
    double acc = 0.0;
    for (int i = blockIdx.x * VGL_BLOCK + threadIdx.x; i < n; i += gridDim.x * VGL_BLOCK) {
        if (MODE == 1 && flags[i] <= 0) continue;
        const int src = (MODE == 2) ? ids[i] : i;
        const double v = (double)op(src, (int)(rowptr[src + 1] - rowptr[src]), (int)(threadIdx.x & 63));
        if (IS_MAX) acc = v > acc ? v : acc; else acc += v;
    }
    for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(acc, o); if (IS_MAX) acc = t > acc ? t : acc; else acc += t; }
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < VGL_BLOCK / 64; w++) { if (IS_MAX) acc = s[w] > acc ? s[w] : acc; else acc += s[w]; }
        partials[blockIdx.x] = acc;
    }
}
// REDUCE_MAX: the maximum of the partials, folded on the device by one workgroup
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_max_fold(int n, const double *partials, double *out)
{
    __shared__ double s[VGL_BLOCK / 64];
    double m = 0.0;
    for (int i = threadIdx.x; i < n; i += VGL_BLOCK) m = partials[i] > m ? partials[i] : m;
    for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) { for (int w = 1; w < VGL_BLOCK / 64; w++) m = s[w] > m ? s[w] : m; out[0] = m; }
}
// generate_new_frontier: the user's filter condition as a predicate of the frontier-generation kernels (csrc/vgl_gnf.h)
template <class Cond>
struct vgl_pred_user {
    Cond cond; const long long *rowptr;
    __device__ uint32_t bits8(int32_t v0, int nvalid, uint32_t *aux) const
    {
        uint32_t b = 0;
        *aux = 0;
        if (nvalid == 8) {
            // a full thread: the eight evaluations UNROLLED, so that whatever the condition loads (levels[v], distances[v] ...) is requested for
            // all eight vertices before the first result is needed.  As a loop with a run-time trip count each evaluation waited for its own loads:
            // eight dependent round trips per thread, 57 us for the 64 MiB of a BFS level's `levels` on RMAT-24 against 13 us for the library's own
            // scan of the same array (profiles/r05_binding_bfs_vcsr_rmat24_kernel_stats.csv, vgl_k_bfs_scan_bound).
            int hit[8];
#pragma unroll
            for (int j = 0; j < 8; j++) hit[j] = cond(v0 + j, (int)(rowptr[v0 + j + 1] - rowptr[v0 + j]));
#pragma unroll
            for (int j = 0; j < 8; j++) b |= (uint32_t)(hit[j] > 0) << j;
            return b;
        }
        for (int j = 0; j < nvalid; j++) b |= (uint32_t)(cond(v0 + j, (int)(rowptr[v0 + j + 1] - rowptr[v0 + j])) > 0) << j;
        return b;
    }
};
// exchange_vertices_array helpers (common/mpi_exchange.hpp:78-150): the merge of the other ranks' copies / changed entries with the user's operator
template <class T, class MergeOp>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_merge_copies(int n, int parts, int self, const T *all, T *data, MergeOp merge_op)
{
    for (int i = blockIdx.x * VGL_BLOCK + threadIdx.x; i < n; i += gridDim.x * VGL_BLOCK) {
        T acc = data[i];
        for (int p = 0; p < parts; p++)
            if (p != self) acc = merge_op(all[(size_t)p * n + i], acc);      // _new_data[i] = _merge_op(received_data[i], _new_data[i])  (mpi_exchange.hpp:146-149)
        data[i] = acc;
    }
}
// one rank's (index, value bits) list: its indexes are distinct, so plain stores are safe (the lists of different ranks are applied one after the other)
template <class T, class MergeOp>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_merge_pairs(const int *list, long long max_pairs, int n, T *data, MergeOp merge_op)
{
    static_assert(sizeof(T) == 4, "pair lists carry 4-byte values");
    const long long count = list[0] < max_pairs ? list[0] : max_pairs;
    for (long long k = (long long)blockIdx.x * VGL_BLOCK + threadIdx.x; k < count; k += (long long)gridDim.x * VGL_BLOCK) {
        const int idx = list[1 + 2 * k];
        if (idx < 0 || idx >= n) continue;
        T v;
        const int bits = list[2 + 2 * k];
        memcpy(&v, &bits, 4);
        data[idx] = merge_op(v, data[idx]);
    }
}
__global__ void vgl_k_list_heads(const int *lists, long long stride, int parts, int *out)
{
    if ((int)threadIdx.x < parts) out[threadIdx.x] = lists[(long long)threadIdx.x * stride];
}

// VECTOR_CSR_GRAPH: the reference's advance (multicore/advance_worker.hpp:204-319) hands rows of at least VECTOR_CORE_THRESHOLD_VALUE
// entries to (edge_op, pre, post) and the shorter rows -- its collective range -- to the collective functor set.  The row's own
// degree in the traversed direction decides here (one numbering serves both directions, so the classes are not id ranges).
template <class A, class B>
struct vgl_split_edge_op {
    A big; B small; const long long *rowptr; int threshold;
    __device__ __forceinline__ void operator()(int src, int dst, int local, long long global, int lane) const
    {
        if ((int)(rowptr[src + 1] - rowptr[src]) >= threshold) big(src, dst, local, global, lane);
        else small(src, dst, local, global, lane);
    }
};
template <class A, class B>
struct vgl_split_vertex_op {
    A big; B small; int threshold;
    __device__ __forceinline__ void operator()(int src, int connections, int lane) const
    {
        if (connections >= threshold) big(src, connections, lane);
        else small(src, connections, lane);
    }
};

