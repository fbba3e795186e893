// comm.hip -- the communicator behind the C ABI: RCCL collectives over xGMI on the context's stream, and a host-staged transport
// through POSIX shared memory for ranks that share a GPU (tests, rehearsals).  What it replaces in the reference: the MPI state of
// vgl_runtime/helpers/library_data/library_data.h and the point-to-point / Allgatherv calls of vgl_compute_api/common/mpi_exchange.hpp
// (exchange_data_cycle_mode :78-150, exchange_data_recently_changed_and_all :156-187, in_group_exchange :222-247).
// Every rank is one process with one GPU (or a share of one, HOSTED); all collectives are issued in the same order on every rank.
#include "vgl_comm.h"
#include <signal.h>
#include <errno.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#define VGL_NCCL_TRY(expr)                                                                         \
    do {                                                                                           \
        ncclResult_t _r = (expr);                                                                  \
        if (_r != ncclSuccess) return vgl_set_error(__FILE__, __LINE__, ncclGetErrorString(_r));   \
    } while (0)

// ------------------------------------------------------------------------------------------------------------------
// HOSTED transport: header + one staging slot per rank in a shared-memory object
// ------------------------------------------------------------------------------------------------------------------
struct vgl_hosted_header {
    std::atomic<uint32_t> magic;
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> generation;
    uint32_t world;
    uint64_t slot_bytes;
    std::atomic<uint32_t> turn;          // VGL_HOSTED_SERIALIZE=1: ranks that have finished the phase after the last barrier
    std::atomic<uint32_t> aborted;       // a rank gave up (vgl_hip_comm_abort, or its own timeout): everybody waiting at a barrier fails at once
    int32_t creator_pid;                 // rank 0's process: a segment whose creator is gone is a leftover of a crashed run, not this run's
    char pad[256 - 36];
};
static_assert(sizeof(vgl_hosted_header) == 256, "hosted header is one 256-byte block");
constexpr uint32_t VGL_HOSTED_MAGIC = 0x56474C48u;      // "VGLH"
static double vgl_hosted_timeout() { static const double t = getenv("VGL_HOSTED_TIMEOUT") ? atof(getenv("VGL_HOSTED_TIMEOUT")) : 180.0; return t; }
#define VGL_HOSTED_TIMEOUT_S (vgl_hosted_timeout())

static inline char *vgl_hosted_slot(vgl_hip_comm *m, int p) { return reinterpret_cast<char *>(m->shm) + sizeof(vgl_hosted_header) + (size_t)p * m->slot_bytes; }

// VGL_HOSTED_SERIALIZE=1 (rehearsals of more ranks than GPUs): between two barriers the ranks work ONE AT A TIME, in rank order, so that the
// kernel times a rank measures with HIP events are those of its own work, not of P processes or threads sharing the card.
int vgl_hosted_barrier(vgl_hip_comm *m)
{
    vgl_hosted_header *h = m->shm;
    static const bool serialize = getenv("VGL_HOSTED_SERIALIZE") && getenv("VGL_HOSTED_SERIALIZE")[0] == '1';
    const auto t0 = std::chrono::steady_clock::now();
    auto timed_out = [&](long spin) {
        if ((spin & 0xFFFF) != 0xFFFF) return false;
        usleep(50);
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() <= VGL_HOSTED_TIMEOUT_S) return false;
        h->aborted.store(1, std::memory_order_release);      // the others need not wait out their own timeouts
        return true;
    };
    if (h->aborted.load(std::memory_order_acquire)) VGL_FAIL("hosted transport: another rank gave up (see its error)");
    const uint32_t gen = h->generation.load(std::memory_order_acquire);
    if (serialize) h->turn.fetch_add(1, std::memory_order_acq_rel);         // my phase is over: the next rank may start its own
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) == (uint32_t)m->world - 1) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->turn.store(0, std::memory_order_relaxed);
        h->generation.fetch_add(1, std::memory_order_release);
    } else {
        for (long spin = 0; h->generation.load(std::memory_order_acquire) == gen; spin++) {
            if ((spin & 0x3FF) == 0x3FF && h->aborted.load(std::memory_order_acquire)) VGL_FAIL("hosted transport: another rank gave up (see its error)");
            if (timed_out(spin)) VGL_FAIL("hosted transport: a rank did not reach the barrier (timeout)");
            __builtin_ia32_pause();
        }
    }
    if (serialize)
        for (long spin = 0; h->turn.load(std::memory_order_acquire) != (uint32_t)m->rank; spin++) {
            if (timed_out(spin)) VGL_FAIL("hosted transport: the rank before this one did not finish its phase (timeout)");
            __builtin_ia32_pause();
        }
    return 0;
}

static int vgl_hosted_sync(vgl_hip_comm *m)
{
    VGL_HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    return 0;
}

// d_recv[p * bytes + ...] = rank p's d_send, in pieces of at most one slot
static int vgl_hosted_allgather(vgl_hip_comm *m, const void *d_send, void *d_recv, int64_t bytes)
{
    const int P = m->world;
    for (int64_t off = 0; off < bytes; off += (int64_t)m->slot_bytes) {
        const size_t n = (size_t)std::min<int64_t>((int64_t)m->slot_bytes, bytes - off);
        VGL_HIP_TRY(hipMemcpyAsync(vgl_hosted_slot(m, m->rank), (const char *)d_send + off, n, hipMemcpyDeviceToHost, m->ctx->stream));
        VGL_TRY(vgl_hosted_sync(m));
        VGL_TRY(vgl_hosted_barrier(m));
        for (int p = 0; p < P; p++)
            VGL_HIP_TRY(hipMemcpyAsync((char *)d_recv + (int64_t)p * bytes + off, vgl_hosted_slot(m, p), n, hipMemcpyHostToDevice, m->ctx->stream));
        VGL_TRY(vgl_hosted_sync(m));
        VGL_TRY(vgl_hosted_barrier(m));
    }
    return 0;
}

static int vgl_hosted_alltoall(vgl_hip_comm *m, const void *d_send, void *d_recv, int64_t bpr)
{
    const int P = m->world;
    const int64_t piece = (int64_t)(m->slot_bytes / (size_t)P) & ~(int64_t)7;
    if (piece <= 0) VGL_FAIL("hosted transport: slot too small for an all-to-all");
    for (int64_t off = 0; off < bpr; off += piece) {
        const size_t n = (size_t)std::min<int64_t>(piece, bpr - off);
        for (int q = 0; q < P; q++)
            VGL_HIP_TRY(hipMemcpyAsync(vgl_hosted_slot(m, m->rank) + (int64_t)q * piece, (const char *)d_send + (int64_t)q * bpr + off, n, hipMemcpyDeviceToHost,
                                       m->ctx->stream));
        VGL_TRY(vgl_hosted_sync(m));
        VGL_TRY(vgl_hosted_barrier(m));
        for (int p = 0; p < P; p++)
            VGL_HIP_TRY(hipMemcpyAsync((char *)d_recv + (int64_t)p * bpr + off, vgl_hosted_slot(m, p) + (int64_t)m->rank * piece, n, hipMemcpyHostToDevice,
                                       m->ctx->stream));
        VGL_TRY(vgl_hosted_sync(m));
        VGL_TRY(vgl_hosted_barrier(m));
    }
    return 0;
}

static int vgl_hosted_allgatherv_inplace(vgl_hip_comm *m, void *d_buf, const int64_t *bb)
{
    const int P = m->world;
    int64_t longest = 0;
    for (int p = 0; p < P; p++) longest = std::max(longest, bb[p + 1] - bb[p]);
    for (int64_t off = 0; off < longest; off += (int64_t)m->slot_bytes) {
        const int64_t mine = bb[m->rank + 1] - bb[m->rank];
        if (off < mine)
            VGL_HIP_TRY(hipMemcpyAsync(vgl_hosted_slot(m, m->rank), (const char *)d_buf + bb[m->rank] + off, (size_t)std::min<int64_t>((int64_t)m->slot_bytes, mine - off),
                                       hipMemcpyDeviceToHost, m->ctx->stream));
        VGL_TRY(vgl_hosted_sync(m));
        VGL_TRY(vgl_hosted_barrier(m));
        for (int p = 0; p < P; p++) {
            const int64_t theirs = bb[p + 1] - bb[p];
            if (p == m->rank || off >= theirs) continue;
            VGL_HIP_TRY(hipMemcpyAsync((char *)d_buf + bb[p] + off, vgl_hosted_slot(m, p), (size_t)std::min<int64_t>((int64_t)m->slot_bytes, theirs - off),
                                       hipMemcpyHostToDevice, m->ctx->stream));
        }
        VGL_TRY(vgl_hosted_sync(m));
        VGL_TRY(vgl_hosted_barrier(m));
    }
    return 0;
}

// out[i] = fold over p of in[p * n + i] in rank order (the same order on every rank: bit-identical sums everywhere)
template <class T, int OP>
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_fold_parts(int64_t n, int parts, const T *in, T *out)
{
    for (int64_t i = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * VGL_BLOCK) {
        T acc = in[i];
        for (int p = 1; p < parts; p++) {
            const T v = in[(int64_t)p * n + i];
            if (OP == VGL_OP_SUM) acc = acc + v;
            else if (OP == VGL_OP_MIN) acc = v < acc ? v : acc;
            else if (OP == VGL_OP_MAX) acc = v > acc ? v : acc;
            else acc = (T)((uint64_t)acc | (uint64_t)v);
        }
        out[i] = acc;
    }
}
template <class T>
static int vgl_fold_launch(vgl_hip_ctx *c, int64_t n, int parts, const void *in, void *out, int op)
{
    const unsigned nb = (unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, vgl_ceil_div(n, VGL_BLOCK)));
    switch (op) {
    case VGL_OP_SUM: hipLaunchKernelGGL((vgl_k_fold_parts<T, VGL_OP_SUM>), dim3(nb), dim3(VGL_BLOCK), 0, c->stream, n, parts, (const T *)in, (T *)out); break;
    case VGL_OP_MIN: hipLaunchKernelGGL((vgl_k_fold_parts<T, VGL_OP_MIN>), dim3(nb), dim3(VGL_BLOCK), 0, c->stream, n, parts, (const T *)in, (T *)out); break;
    case VGL_OP_MAX: hipLaunchKernelGGL((vgl_k_fold_parts<T, VGL_OP_MAX>), dim3(nb), dim3(VGL_BLOCK), 0, c->stream, n, parts, (const T *)in, (T *)out); break;
    case VGL_OP_OR:
        if (!std::is_same<T, uint64_t>::value) VGL_FAIL("all-reduce: only 64-bit words are merged with OR");
        hipLaunchKernelGGL((vgl_k_fold_parts<uint64_t, VGL_OP_OR>), dim3(nb), dim3(VGL_BLOCK), 0, c->stream, n, parts, (const uint64_t *)in, (uint64_t *)out);
        break;
    default: VGL_FAIL("all-reduce: unknown operator");
    }
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

static size_t vgl_dt_bytes(int dtype) { return (dtype == VGL_DT_I32 || dtype == VGL_DT_F32) ? 4 : 8; }

int vgl_fold(vgl_hip_ctx *c, int64_t n, int parts, const void *in, void *out, int dtype, int op)
{
    switch (dtype) {
    case VGL_DT_I32: return vgl_fold_launch<int32_t>(c, n, parts, in, out, op);
    case VGL_DT_F32: return vgl_fold_launch<float>(c, n, parts, in, out, op);
    case VGL_DT_F64: return vgl_fold_launch<double>(c, n, parts, in, out, op);
    case VGL_DT_I64: return vgl_fold_launch<int64_t>(c, n, parts, in, out, op);
    case VGL_DT_U64: return vgl_fold_launch<uint64_t>(c, n, parts, in, out, op);
    }
    VGL_FAIL("all-reduce: unknown element type");
}

// ------------------------------------------------------------------------------------------------------------------
// transport-independent collectives
// ------------------------------------------------------------------------------------------------------------------
int vgl_comm_scratch(vgl_hip_comm *m, int slot, size_t bytes, void **out)
{
    if (slot < 0 || slot >= VGL_COMM_SCRATCH_SLOTS) VGL_FAIL("comm scratch: bad slot");
    if (m->scratch_cap[slot] < bytes) {
        if (m->scratch[slot]) { VGL_HIP_TRY(hipStreamSynchronize(m->ctx->stream)); VGL_HIP_TRY(hipFree(m->scratch[slot])); m->scratch[slot] = nullptr; }
        const size_t cap = std::max<size_t>(bytes + bytes / 4, 4096);
        VGL_HIP_TRY(hipMalloc(&m->scratch[slot], cap));
        m->scratch_cap[slot] = cap;
    }
    *out = m->scratch[slot];
    return 0;
}

void vgl_comm_group_begin(vgl_hip_comm *m)
{
    if (!vgl_comm_active(m) || m->grouped) return;
    if (m->transport == VGL_HIP_COMM_RCCL) { ncclGroupStart(); m->grouped = true; }
    else if (m->transport == VGL_HIP_COMM_PEER) m->grouped = true;          // the exchanges queue up and share one arrival flag
}
int vgl_comm_group_end(vgl_hip_comm *m)
{
    if (m && m->grouped) {
        m->grouped = false;
        if (m->transport == VGL_HIP_COMM_PEER) return vgl_peer_group_end(m);
        VGL_NCCL_TRY(ncclGroupEnd());
    }
    return 0;
}

int vgl_comm_allreduce(vgl_hip_comm *m, void *d_buf, int64_t count, int dtype, int op)
{
    if (!vgl_comm_active(m) || count <= 0) return 0;
    m->stats.collectives++;
    m->stats.bytes_received += count * (int64_t)vgl_dt_bytes(dtype);
    if (m->transport == VGL_HIP_COMM_RCCL) {
        static const ncclDataType_t dt[] = {ncclInt32, ncclFloat32, ncclFloat64, ncclInt64, ncclUint64};
        static const ncclRedOp_t ro[] = {ncclSum, ncclMin, ncclMax};
        if (op == VGL_OP_OR) VGL_FAIL("all-reduce: OR goes through vgl_hip_exchange_bitmap_or");
        VGL_NCCL_TRY(ncclAllReduce(d_buf, d_buf, (size_t)count, dt[dtype], ro[op], m->nccl, m->ctx->stream));
        return 0;
    }
    if (m->transport == VGL_HIP_COMM_PEER) return vgl_peer_allreduce(m, d_buf, count, dtype, op);
    void *all = nullptr;
    const size_t bytes = (size_t)count * vgl_dt_bytes(dtype);
    VGL_TRY(vgl_comm_scratch(m, 5, bytes * (size_t)m->world, &all));
    VGL_TRY(vgl_hosted_allgather(m, d_buf, all, (int64_t)bytes));
    return vgl_fold(m->ctx, count, m->world, all, d_buf, dtype, op);
}

int vgl_comm_allgather(vgl_hip_comm *m, const void *d_send, void *d_recv, int64_t bytes)
{
    if (!vgl_comm_active(m)) {
        if (d_send != d_recv && bytes > 0) VGL_HIP_TRY(hipMemcpyAsync(d_recv, d_send, (size_t)bytes, hipMemcpyDeviceToDevice, m ? m->ctx->stream : nullptr));
        return 0;
    }
    m->stats.collectives++;
    m->stats.bytes_received += bytes * m->world;
    if (m->transport == VGL_HIP_COMM_RCCL) {
        VGL_NCCL_TRY(ncclAllGather(d_send, d_recv, (size_t)bytes, ncclChar, m->nccl, m->ctx->stream));
        return 0;
    }
    if (m->transport == VGL_HIP_COMM_PEER) return vgl_peer_allgather(m, d_send, d_recv, bytes);
    return vgl_hosted_allgather(m, d_send, d_recv, bytes);
}

int vgl_comm_alltoall(vgl_hip_comm *m, const void *d_send, void *d_recv, int64_t bpr)
{
    if (!vgl_comm_active(m)) {
        if (d_send != d_recv && bpr > 0) VGL_HIP_TRY(hipMemcpyAsync(d_recv, d_send, (size_t)bpr, hipMemcpyDeviceToDevice, m ? m->ctx->stream : nullptr));
        return 0;
    }
    m->stats.collectives++;
    m->stats.bytes_received += bpr * m->world;
    if (m->transport == VGL_HIP_COMM_RCCL) {
        VGL_NCCL_TRY(ncclAllToAll(d_send, d_recv, (size_t)bpr, ncclChar, m->nccl, m->ctx->stream));
        return 0;
    }
    if (m->transport == VGL_HIP_COMM_PEER) return vgl_peer_alltoall(m, d_send, d_recv, bpr);
    return vgl_hosted_alltoall(m, d_send, d_recv, bpr);
}

int vgl_comm_allgatherv_inplace(vgl_hip_comm *m, void *d_buf, const int64_t *bb)
{
    if (!vgl_comm_active(m)) return 0;
    m->stats.collectives++;
    m->stats.bytes_received += bb[m->world] - bb[0];
    if (m->transport == VGL_HIP_COMM_RCCL) {
        bool equal = true;
        for (int p = 1; p < m->world; p++) equal = equal && (bb[p + 1] - bb[p]) == (bb[1] - bb[0]);
        if (equal) {                                 // in place: rank r's part already sits at its position
            if (bb[1] - bb[0] > 0)
                VGL_NCCL_TRY(ncclAllGather((const char *)d_buf + bb[m->rank], (char *)d_buf + bb[0], (size_t)(bb[1] - bb[0]), ncclChar, m->nccl, m->ctx->stream));
            return 0;
        }
        // MPI_Allgatherv in place = one broadcast per owner, fused into a single launch by the group
        const bool outer = m->grouped;
        if (!outer) VGL_NCCL_TRY(ncclGroupStart());
        ncclResult_t first = ncclSuccess;                // (a failing broadcast must not leave the local group open)
        for (int p = 0; p < m->world && first == ncclSuccess; p++) {
            const int64_t n = bb[p + 1] - bb[p];
            if (n > 0) first = ncclBroadcast((const char *)d_buf + bb[p], (char *)d_buf + bb[p], (size_t)n, ncclChar, p, m->nccl, m->ctx->stream);
        }
        if (!outer) { const ncclResult_t e = ncclGroupEnd(); if (first == ncclSuccess) first = e; }
        VGL_NCCL_TRY(first);
        return 0;
    }
    if (m->transport == VGL_HIP_COMM_PEER) return vgl_peer_allgatherv_inplace(m, d_buf, bb);
    return vgl_hosted_allgatherv_inplace(m, d_buf, bb);
}

__global__ void vgl_k_comm_publish(const int64_t *src, int n, volatile int64_t *host, int64_t seq, const unsigned long long *peer_error)
{
    for (int i = threadIdx.x; i < n; i += blockDim.x) host[i] = src[i];
    if (threadIdx.x == 0) host[VGL_COMM_SMALL + 1] = peer_error ? (int64_t)__hip_atomic_load(peer_error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { host[VGL_COMM_SMALL] = seq; __threadfence_system(); }
}

// n device words -> host.  Works without a communicator too (ctx's stream, the context's own pinned mirror is not touched).
int vgl_comm_read_small(vgl_hip_comm *m, const int64_t *d_vals, int n, int64_t *h_out)
{
    if (n < 0 || n > VGL_COMM_SMALL) VGL_FAIL("comm read_small: too many values");
    const int64_t seq = ++m->small_seq;
    hipLaunchKernelGGL(vgl_k_comm_publish, dim3(1), dim3(64), 0, m->ctx->stream, d_vals, n, (volatile int64_t *)m->h_small, seq,
                       m->transport == VGL_HIP_COMM_PEER ? vgl_peer_error_word(m) : (const unsigned long long *)nullptr);
    VGL_HIP_TRY(hipGetLastError());
    volatile int64_t *flag = (volatile int64_t *)m->h_small + VGL_COMM_SMALL;
    for (long spin = 0; *flag != seq; spin++) {
        if (spin > 4000000) {
            VGL_HIP_TRY(hipStreamSynchronize(m->ctx->stream));
            if (*flag != seq) VGL_FAIL("comm read_small: the hand-over did not arrive");
            break;
        }
        __builtin_ia32_pause();
    }
    for (int i = 0; i < n; i++) h_out[i] = m->h_small[i];
    if (m->h_small[VGL_COMM_SMALL + 1] != 0) VGL_FAIL("peer transport: a rank did not arrive at an exchange (timeout): the results of this run are void");
    return 0;
}

// after the last exchange of a driver that reads no counters of its own (PageRank, HITS): a flag wait that ran out under the PEER transport only
// sets the window's error word -- fetch it, so that the driver fails on this rank like the drivers that poll their counters every super-step
int vgl_comm_check(vgl_hip_comm *m)
{
    if (!m || m->transport != VGL_HIP_COMM_PEER) return 0;
    int64_t none = 0;
    return vgl_comm_read_small(m, m->d_small, 0, &none);
}

int vgl_comm_allreduce_host_i64(vgl_hip_comm *m, int64_t *vals, int n, int op)
{
    if (!vgl_comm_active(m) || n <= 0) return 0;
    if (n > VGL_COMM_SMALL) VGL_FAIL("comm allreduce_host: too many values");
    VGL_HIP_TRY(hipMemcpyAsync(m->d_small, vals, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, m->ctx->stream));
    VGL_HIP_TRY(hipStreamSynchronize(m->ctx->stream));          // (vals may be a stack array)
    VGL_TRY(vgl_comm_allreduce(m, m->d_small, n, VGL_DT_I64, op));
    return vgl_comm_read_small(m, m->d_small, n, vals);
}

int vgl_comm_row_bounds(vgl_hip_comm *m, const vgl_hip_graph *g, const int64_t **bounds)
{
    auto it = m->bounds.find(g->uid);
    if (it == m->bounds.end()) {
        const int P = m->world;
        std::vector<int64_t> mine = {g->row_begin, g->row_end, g->in.rowptr ? (int64_t)g->in_nz_rows : 0}, all((size_t)3 * P);
        int64_t *d_mine = m->d_small, *d_all = m->d_small + 3;
        if (3 + 3 * P > VGL_COMM_SMALL) VGL_FAIL("comm row_bounds: too many ranks");
        VGL_HIP_TRY(hipMemcpyAsync(d_mine, mine.data(), sizeof(int64_t) * 3, hipMemcpyHostToDevice, m->ctx->stream));
        VGL_HIP_TRY(hipStreamSynchronize(m->ctx->stream));
        VGL_TRY(vgl_comm_allgather(m, d_mine, d_all, sizeof(int64_t) * 3));
        VGL_TRY(vgl_comm_read_small(m, d_all, 3 * P, all.data()));
        std::vector<int64_t> b((size_t)P + 2, 0);
        for (int p = 0; p < P; p++) {
            if (all[(size_t)3 * p + 1] < all[(size_t)3 * p]) VGL_FAIL("sharded run: a rank reports an empty-inverted row range");
            if (p > 0 && all[(size_t)3 * p] != all[(size_t)3 * p - 2]) VGL_FAIL("sharded run: the ranks' row ranges must tile [0, V) in rank order");
            b[(size_t)p] = all[(size_t)3 * p];
            b[(size_t)P + 1] += all[(size_t)3 * p + 2];                  // rows with incoming edges over all ranks (0: no incoming lists loaded)
        }
        b[(size_t)P] = all[(size_t)3 * P - 2];
        if (b[0] != 0 || b[(size_t)P] != g->V) VGL_FAIL("sharded run: the ranks' row ranges must tile [0, V) in rank order");
        if (m->bounds.size() > 64) m->bounds.clear();
        it = m->bounds.emplace(g->uid, std::move(b)).first;
    }
    *bounds = it->second.data();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------------
static int vgl_comm_common_init(vgl_hip_comm *m)
{
    VGL_HIP_TRY(hipMalloc((void **)&m->d_small, sizeof(int64_t) * VGL_COMM_SMALL));
    VGL_HIP_TRY(hipMemsetAsync(m->d_small, 0, sizeof(int64_t) * VGL_COMM_SMALL, m->ctx->stream));
    VGL_HIP_TRY(hipHostMalloc((void **)&m->h_small, sizeof(int64_t) * (VGL_COMM_SMALL + 8), hipHostMallocDefault));
    memset(m->h_small, 0, sizeof(int64_t) * (VGL_COMM_SMALL + 8));
    const char *f = getenv("VGL_SHARD_FORCE_COLLECTIVES");
    m->force = f && f[0] == '1';
    return 0;
}

extern "C" {

int vgl_hip_comm_unique_id(void *id_out)
{
    if (!id_out) VGL_FAIL("comm_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == VGL_HIP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    VGL_NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int vgl_hip_comm_create(vgl_hip_ctx *c, int rank, int world, const void *unique_id, vgl_hip_comm **out)
{
    if (!c || !out) VGL_FAIL("comm_create: null argument");
    if (world < 1 || rank < 0 || rank >= world) VGL_FAIL("comm_create: rank / world out of range");
    if (!unique_id) VGL_FAIL("comm_create: the unique id of rank 0 is required (vgl_hip_comm_unique_id)");
    VGL_HIP_TRY(hipSetDevice(c->device));
    vgl_hip_comm *m = new vgl_hip_comm();
    m->ctx = c; m->rank = rank; m->world = world; m->transport = VGL_HIP_COMM_RCCL;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&m->nccl, world, id, rank);
    if (r != ncclSuccess) { delete m; return vgl_set_error(__FILE__, __LINE__, ncclGetErrorString(r)); }
    if (vgl_comm_common_init(m)) { ncclCommDestroy(m->nccl); delete m; return 1; }
    *out = m;
    return 0;
}

// attaches (rank 0: creates) the shared-memory object `name` of `payload` bytes behind the 256-byte header; `check` must agree on all ranks
static int vgl_shm_attach(vgl_hip_ctx *c, int rank, int world, const char *name, size_t payload, uint64_t check, int transport, vgl_hip_comm **out)
{
    const size_t total = sizeof(vgl_hosted_header) + payload;
    int fd = -1;
    const auto t0 = std::chrono::steady_clock::now();
    auto waited = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    if (rank == 0) {
        shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) VGL_FAIL("comm_create: shm_open failed on rank 0");
        if (ftruncate(fd, (off_t)total) != 0) { close(fd); shm_unlink(name); VGL_FAIL("comm_create: ftruncate failed (is /dev/shm large enough?)"); }
    } else {
        for (;;) {
            fd = shm_open(name, O_RDWR, 0600);
            if (fd >= 0) {
                struct stat st;
                if (fstat(fd, &st) == 0 && (size_t)st.st_size == total) {
                    // an object of this name and size may be what a crashed run left behind (rank 0 unlinks the name only once everybody is
                    // attached): it is this run's if its creator is alive -- or has not written its pid yet
                    void *q = mmap(nullptr, sizeof(vgl_hosted_header), PROT_READ, MAP_SHARED, fd, 0);
                    bool stale = false;
                    if (q != MAP_FAILED) {
                        const vgl_hosted_header *hh = reinterpret_cast<const vgl_hosted_header *>(q);
                        const int32_t pid = hh->magic.load(std::memory_order_acquire) == VGL_HOSTED_MAGIC ? hh->creator_pid : 0;
                        stale = pid > 0 && kill((pid_t)pid, 0) != 0 && errno == ESRCH;
                        munmap(q, sizeof(vgl_hosted_header));
                    }
                    if (!stale) break;
                }
                close(fd); fd = -1;
            }
            if (waited() > VGL_HOSTED_TIMEOUT_S) VGL_FAIL("comm_create: rank 0's segment did not appear (timeout)");
            usleep(1000);
        }
    }
    void *p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) VGL_FAIL("comm_create: mmap failed");
    vgl_hip_comm *m = new vgl_hip_comm();
    m->ctx = c; m->rank = rank; m->world = world; m->transport = transport;
    m->shm = reinterpret_cast<vgl_hosted_header *>(p); m->shm_bytes = total; m->shm_name = name;
    if (rank == 0) {
        m->shm->arrived.store(0); m->shm->generation.store(0); m->shm->turn.store(0); m->shm->aborted.store(0);
        m->shm->creator_pid = (int32_t)getpid();
        m->shm->world = (uint32_t)world; m->shm->slot_bytes = check;
        memset(m->shm->pad, 0, sizeof(m->shm->pad));
        m->shm->magic.store(VGL_HOSTED_MAGIC, std::memory_order_release);
    } else {
        while (m->shm->magic.load(std::memory_order_acquire) != VGL_HOSTED_MAGIC) {
            if (waited() > VGL_HOSTED_TIMEOUT_S) { munmap(p, total); delete m; VGL_FAIL("comm_create: rank 0 did not initialise the segment (timeout)"); }
            usleep(200);
        }
        if (m->shm->world != (uint32_t)world || m->shm->slot_bytes != check) { munmap(p, total); delete m; VGL_FAIL("comm_create: the ranks disagree on world / buffer size"); }
    }
    if (vgl_comm_common_init(m) || vgl_hosted_barrier(m)) { munmap(p, total); delete m; return 1; }
    if (rank == 0) shm_unlink(name);            // everybody is attached: the object lives on until the last rank unmaps it
    *out = m;
    return 0;
}

int vgl_hip_comm_create_hosted(vgl_hip_ctx *c, int rank, int world, const char *name, size_t slot_bytes, vgl_hip_comm **out)
{
    if (!c || !out || !name || name[0] != '/') VGL_FAIL("comm_create_hosted: null argument or a name that does not start with '/'");
    if (world < 1 || world > 64 || rank < 0 || rank >= world) VGL_FAIL("comm_create_hosted: rank / world out of range");
    slot_bytes = (std::max<size_t>(slot_bytes, 4096) + 255) & ~(size_t)255;
    VGL_TRY(vgl_shm_attach(c, rank, world, name, slot_bytes * (size_t)world, slot_bytes, VGL_HIP_COMM_HOSTED, out));
    (*out)->slot_bytes = slot_bytes;
    return 0;
}

int vgl_hip_comm_set_timeout_ms(vgl_hip_comm *m, double ms)
{
    if (!m) VGL_FAIL("comm_set_timeout_ms: null communicator");
    if (m->transport == VGL_HIP_COMM_PEER) vgl_peer_set_timeout_ms(m, ms);
    return 0;
}

int vgl_hip_comm_create_peer(vgl_hip_ctx *c, int rank, int world, const char *name, size_t window_bytes, vgl_hip_comm **out)
{
    if (!c || !out || !name || name[0] != '/') VGL_FAIL("comm_create_peer: null argument or a name that does not start with '/'");
    if (world < 1 || world > 64 || rank < 0 || rank >= world) VGL_FAIL("comm_create_peer: rank / world out of range");
    VGL_HIP_TRY(hipSetDevice(c->device));
    vgl_hip_comm *m = nullptr;
    VGL_TRY(vgl_shm_attach(c, rank, world, name, (size_t)128 * (size_t)world, (uint64_t)window_bytes, VGL_HIP_COMM_PEER, &m));
    if (vgl_peer_setup(m, window_bytes)) {
        // the set-up failed on this rank (an allocation, an IPC handle) or on all ranks alike (the agreed vote): say so in the shared header before
        // leaving, so that no peer waits out the hosted timeout in a barrier -- of its own set-up or of its teardown -- for a rank that is gone
        const std::string err = vgl_hip_last_error();
        (void)vgl_hip_comm_abort(m);
        vgl_hip_comm_destroy(m);
        return vgl_set_error(__FILE__, __LINE__, err.c_str());
    }
    *out = m;
    return 0;
}

// a rank that cannot go on (an error anywhere between two exchanges) tells the others, who would otherwise wait at the next barrier until
// their timeout: hosted / peer transports only (RCCL has its own abort); safe to call more than once
int vgl_hip_comm_abort(vgl_hip_comm *m)
{
    if (!m) return 0;
    if (m->shm) m->shm->aborted.store(1, std::memory_order_release);
    return 0;
}
int vgl_hip_comm_destroy(vgl_hip_comm *m)
{
    if (!m) return 0;
    hipSetDevice(m->ctx->device);
    hipStreamSynchronize(m->ctx->stream);
    if (m->nccl) ncclCommDestroy(m->nccl);
    if (m->peer) vgl_peer_teardown(m);
    if (m->shm) {
        // serialised rehearsal: leaving counts as the end of this rank's last phase (the next rank is waiting for its turn)
        if (getenv("VGL_HOSTED_SERIALIZE") && getenv("VGL_HOSTED_SERIALIZE")[0] == '1') m->shm->turn.fetch_add(1, std::memory_order_acq_rel);
        munmap(m->shm, m->shm_bytes);
    }
    for (int i = 0; i < VGL_COMM_SCRATCH_SLOTS; i++) if (m->scratch[i]) hipFree(m->scratch[i]);
    if (m->d_small) hipFree(m->d_small);
    if (m->h_small) hipHostFree(m->h_small);
    delete m;
    return 0;
}

int vgl_hip_comm_info(vgl_hip_comm *m, int *rank, int *world, int *transport)
{
    if (!m) VGL_FAIL("comm_info: null communicator");
    if (rank) *rank = m->rank;
    if (world) *world = m->world;
    if (transport) *transport = m->transport;
    return 0;
}

int vgl_hip_comm_barrier(vgl_hip_comm *m)
{
    if (!m) VGL_FAIL("comm_barrier: null communicator");
    int64_t one = 1;
    VGL_TRY(vgl_comm_allreduce_host_i64(m, &one, 1, VGL_OP_SUM));
    VGL_HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    if (vgl_comm_active(m) && one != m->world) VGL_FAIL("comm_barrier: ranks out of step");
    return 0;
}

int vgl_hip_comm_stats(vgl_hip_comm *m, vgl_hip_exchange_stats *out)
{
    if (!m || !out) VGL_FAIL("comm_stats: null argument");
    *out = m->stats;
    return 0;
}

#define VGL_ALLREDUCE_ENTRY(name, ctype, dt, op)                                                  \
    int vgl_hip_exchange_allreduce_##name(vgl_hip_comm *m, ctype *d_values, int64_t n)              \
    {                                                                                             \
        if (!m || !d_values) VGL_FAIL("exchange_allreduce_" #name ": null argument");              \
        if (n < 0) VGL_FAIL("exchange_allreduce_" #name ": negative count");                       \
        return vgl_comm_allreduce(m, d_values, n, dt, op);                                        \
    }
VGL_ALLREDUCE_ENTRY(min_i32, int32_t, VGL_DT_I32, VGL_OP_MIN)
VGL_ALLREDUCE_ENTRY(min_f32, float, VGL_DT_F32, VGL_OP_MIN)
VGL_ALLREDUCE_ENTRY(max_f32, float, VGL_DT_F32, VGL_OP_MAX)
VGL_ALLREDUCE_ENTRY(sum_i32, int32_t, VGL_DT_I32, VGL_OP_SUM)
VGL_ALLREDUCE_ENTRY(sum_i64, int64_t, VGL_DT_I64, VGL_OP_SUM)
VGL_ALLREDUCE_ENTRY(sum_f32, float, VGL_DT_F32, VGL_OP_SUM)
VGL_ALLREDUCE_ENTRY(sum_f64, double, VGL_DT_F64, VGL_OP_SUM)

int vgl_hip_exchange_allgather(vgl_hip_comm *m, const void *d_send, void *d_recv, int64_t bytes_per_rank)
{
    if (!m || !d_send || !d_recv) VGL_FAIL("exchange_allgather: null argument");
    if (bytes_per_rank < 0) VGL_FAIL("exchange_allgather: negative size");
    return vgl_comm_allgather(m, d_send, d_recv, bytes_per_rank);
}

int vgl_hip_exchange_allgather_slices(vgl_hip_comm *m, void *d_array, const int64_t *bounds_host, int elem_bytes)
{
    if (!m || !d_array || !bounds_host) VGL_FAIL("exchange_allgather_slices: null argument");
    if (elem_bytes < 1) VGL_FAIL("exchange_allgather_slices: element size");
    std::vector<int64_t> bb((size_t)m->world + 1);
    for (int p = 0; p <= m->world; p++) {
        if (p > 0 && bounds_host[p] < bounds_host[p - 1]) VGL_FAIL("exchange_allgather_slices: bounds must not decrease");
        bb[(size_t)p] = bounds_host[p] * elem_bytes;
    }
    return vgl_comm_allgatherv_inplace(m, d_array, bb.data());
}

int vgl_hip_exchange_bitmap_or(vgl_hip_comm *m, uint64_t *d_bits, int64_t words)
{
    if (!m || !d_bits) VGL_FAIL("exchange_bitmap_or: null argument");
    if (words < 0) VGL_FAIL("exchange_bitmap_or: negative size");
    const int P = m->world;
    if (!vgl_comm_active(m) || words == 0) return 0;
    if (words % P == 0) {
        // slice r of every rank's bitmap -> rank r, OR, all-gather of the merged slices
        const int64_t sw = words / P;
        void *in = nullptr, *mine = nullptr;
        VGL_TRY(vgl_comm_scratch(m, 3, sizeof(uint64_t) * (size_t)words, &in));
        VGL_TRY(vgl_comm_scratch(m, 4, sizeof(uint64_t) * (size_t)sw, &mine));
        VGL_TRY(vgl_comm_alltoall(m, d_bits, in, sw * 8));
        VGL_TRY(vgl_fold(m->ctx, sw, P, in, mine, VGL_DT_U64, VGL_OP_OR));
        return vgl_comm_allgather(m, mine, d_bits, sw * 8);
    }
    void *all = nullptr;
    VGL_TRY(vgl_comm_scratch(m, 3, sizeof(uint64_t) * (size_t)words * (size_t)P, &all));
    VGL_TRY(vgl_comm_allgather(m, d_bits, all, words * 8));
    return vgl_fold(m->ctx, words, P, all, d_bits, VGL_DT_U64, VGL_OP_OR);
}

}  // extern "C"
