// peer.hip -- the PEER transport of the communicator (round 4): every rank owns a WINDOW in device memory that the other ranks map
// (hipIpcGetMemHandle / hipIpcOpenMemHandle; ranks that are threads of one process hand the pointer over as it is) and write into with
// plain stores from a kernel on their own stream -- over xGMI when the ranks are GPUs of one node, through the same HBM when they share a
// card (tests, rehearsals).  No collective library and no host in the data path: what a rank enqueues per exchange is
//     put   : its contribution copied into EVERY peer's window (one kernel reads the source once and stores it `world` times), a
//             system-scope fence, then this rank's slot of every peer's arrival flags set to the exchange's sequence number;
//     wait  : one workgroup polls its own flags (device memory, system-scope loads) until every rank's contribution of this sequence
//             number has arrived;
//     take  : the payload copied (or folded: all-reduce) out of the window by an ordinary kernel;
//     ack   : every peer's consumption flags set, so that a rank may overwrite this half of the window two exchanges later.
// The window has two halves used alternately (sequence number parity): a rank that runs ahead writes exchange k + 1 into the other half
// while a slow peer still reads exchange k, and blocks (inside its put kernel, bounded) only before exchange k + 2.  The exchanges of a
// group (vgl_comm_group_begin / end: the sharded BFS sends a level's owned frontier slices and its four counters together) share ONE
// sequence number: several puts, one signal, one wait -- the counters ride on the payload's flag, as the reference's MPI code sends sizes
// and data in one message pair (vgl_compute_api/common/mpi_exchange.hpp:110-150); its changed-entries exchange (:156-187) and the slice
// all-gather (:222-271) are the callers above this file (exchange.hpp, sharded.hip, bfs_sharded.hip), unchanged.
// Every spin is bounded (VGL_PEER_TIMEOUT_MS, default 20 s): a rank that never arrives turns into an error on all the others, not a hang.
#include "vgl_comm.h"
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

constexpr size_t VGL_PEER_FLAG_BYTES = 4096;            // head of every window: flags, then the two payload halves
constexpr int VGL_PEER_MAX = 64;
struct vgl_peer_flags {
    unsigned long long data_seq[VGL_PEER_MAX];          // [p]: last exchange whose payload from rank p has landed in this window
    unsigned long long ack_seq[VGL_PEER_MAX];           // [p]: last exchange rank p has finished reading out of ITS window
    unsigned long long error;                           // a bounded spin of this rank ran out
};
static_assert(sizeof(vgl_peer_flags) <= VGL_PEER_FLAG_BYTES, "flags fit their block");
struct vgl_peer_tab { char *win[VGL_PEER_MAX]; };      // every rank's window as mapped in this process

struct vgl_peer_record {                                // one per rank in the shared control segment (host memory)
    int32_t pid, device;
    uint64_t raw_ptr;
    hipIpcMemHandle_t handle;
    char pad[128 - 16 - sizeof(hipIpcMemHandle_t)];
};
static_assert(sizeof(vgl_peer_record) == 128, "peer record is 128 bytes");

enum { PEER_ALLGATHER, PEER_ALLTOALL, PEER_ALLGATHERV, PEER_ALLREDUCE };
struct vgl_peer_op {
    int kind;
    const void *src; void *dst;
    int64_t bytes;                                      // per rank (all-gather / all-to-all / all-reduce)
    int64_t src_stride, dst_stride;                     // distance of the per-peer parts in src (all-to-all) / of the per-rank parts in dst: `bytes` unless the op is a piece
    std::vector<int64_t> bb;                            // all-gatherv: world + 1 byte bounds
    int dtype, op;                                      // all-reduce
};
struct vgl_peer_state {
    vgl_peer_tab tab;
    char *window = nullptr;                             // this rank's (owned)
    size_t half_bytes = 0;
    unsigned long long seq = 0;
    uint32_t *ticket = nullptr;
    std::vector<vgl_peer_op> pending;                   // ops of an open group
    std::vector<bool> opened;                           // windows mapped through IPC (to be closed)
    long long timeout_ticks = 0;
};

// ---------------------------------------------------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long vgl_ld_sys(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void vgl_st_sys(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// waits until *p >= want; a spin that runs out sets *err and returns (the exchange's result is then garbage, the host reports the error)
__device__ __forceinline__ void vgl_peer_spin(const unsigned long long *p, unsigned long long want, unsigned long long *err, long long timeout_ticks)
{
    const long long t0 = wall_clock64();                // constant 100 MHz
    while (vgl_ld_sys(p) < want) {
        __builtin_amdgcn_s_sleep(16);
        if (wall_clock64() - t0 > timeout_ticks) { vgl_st_sys(err, 1ULL); return; }
    }
}

// n bytes from src (+ q * src_stride for peer q: all-to-all) into every peer's window at dst_off.  signal: this is the last put of the
// exchange -- the last workgroup to finish sets the arrival flags.
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_peer_put(vgl_peer_tab tab, int rank, int world, const char *src, int64_t src_stride, int64_t n,
                                                           int64_t dst_off, unsigned long long seq, int signal, uint32_t *ticket, long long timeout_ticks)
{
    vgl_peer_flags *mine = reinterpret_cast<vgl_peer_flags *>(tab.win[rank]);
    // flow control: every peer has read what this half of its window held two exchanges ago
    if ((int)threadIdx.x < world && seq > 2) vgl_peer_spin(&mine->ack_seq[threadIdx.x], seq - 2, &mine->error, timeout_ticks);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    const int64_t tid = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x, nthreads = (int64_t)gridDim.x * VGL_BLOCK;
    for (int q = 0; q < world; q++) {
        const char *s = src + (int64_t)q * src_stride;
        char *d = tab.win[q] + dst_off;
        if ((((uintptr_t)s | (uintptr_t)d | (uintptr_t)n) & 15) == 0) {
            const uint4 *s4 = reinterpret_cast<const uint4 *>(s);
            uint4 *d4 = reinterpret_cast<uint4 *>(d);
            for (int64_t i = tid; i < n / 16; i += nthreads) d4[i] = s4[i];
        } else if ((((uintptr_t)s | (uintptr_t)d | (uintptr_t)n) & 3) == 0) {
            const uint32_t *s1 = reinterpret_cast<const uint32_t *>(s);
            uint32_t *d1 = reinterpret_cast<uint32_t *>(d);
            for (int64_t i = tid; i < n / 4; i += nthreads) d1[i] = s1[i];
        } else
            for (int64_t i = tid; i < n; i += nthreads) d[i] = s[i];
    }
    __threadfence_system();                             // this thread's stores are visible to every agent before the ticket moves
    if (!signal) return;
    __syncthreads();                                    // ... and every thread of the workgroup has got here
    if (!vgl_last_block(ticket, 0u)) return;
    __threadfence_system();
    if ((int)threadIdx.x < world) vgl_st_sys(&reinterpret_cast<vgl_peer_flags *>(tab.win[threadIdx.x])->data_seq[rank], seq);
}

__global__ __launch_bounds__(64) void vgl_k_peer_wait(vgl_peer_tab tab, int rank, int world, unsigned long long seq, long long timeout_ticks)
{
    vgl_peer_flags *mine = reinterpret_cast<vgl_peer_flags *>(tab.win[rank]);
    if ((int)threadIdx.x < world) vgl_peer_spin(&mine->data_seq[threadIdx.x], seq, &mine->error, timeout_ticks);
    __threadfence_system();
}

__global__ __launch_bounds__(64) void vgl_k_peer_ack(vgl_peer_tab tab, int rank, int world, unsigned long long seq)
{
    if ((int)threadIdx.x < world) vgl_st_sys(&reinterpret_cast<vgl_peer_flags *>(tab.win[threadIdx.x])->ack_seq[rank], seq);
}

// up to 64 (window offset -> destination, bytes) segments in one launch
struct vgl_peer_segs { int n; const char *src[VGL_PEER_MAX]; char *dst[VGL_PEER_MAX]; int64_t bytes[VGL_PEER_MAX]; };
__global__ __launch_bounds__(VGL_BLOCK) void vgl_k_peer_take(vgl_peer_segs sg)
{
    const int64_t tid = (int64_t)blockIdx.x * VGL_BLOCK + threadIdx.x, nthreads = (int64_t)gridDim.x * VGL_BLOCK;
    for (int k = 0; k < sg.n; k++) {
        const char *s = sg.src[k];
        char *d = sg.dst[k];
        const int64_t n = sg.bytes[k];
        if ((((uintptr_t)s | (uintptr_t)d | (uintptr_t)n) & 15) == 0) {
            for (int64_t i = tid; i < n / 16; i += nthreads) reinterpret_cast<uint4 *>(d)[i] = reinterpret_cast<const uint4 *>(s)[i];
        } else if ((((uintptr_t)s | (uintptr_t)d | (uintptr_t)n) & 3) == 0) {
            for (int64_t i = tid; i < n / 4; i += nthreads) reinterpret_cast<uint32_t *>(d)[i] = reinterpret_cast<const uint32_t *>(s)[i];
        } else
            for (int64_t i = tid; i < n; i += nthreads) d[i] = s[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------
static inline vgl_peer_state *peer_of(vgl_hip_comm *m) { return reinterpret_cast<vgl_peer_state *>(m->peer); }
static inline int64_t round16(int64_t x) { return (x + 15) & ~(int64_t)15; }
static inline unsigned put_grid(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(128, vgl_ceil_div(n, (int64_t)VGL_BLOCK * 64))); }

static int64_t peer_need(const vgl_hip_comm *m, const vgl_peer_op &o)
{
    if (o.kind == PEER_ALLGATHERV) return round16(o.bb[(size_t)m->world] - o.bb[0]);
    return round16(o.bytes) * m->world;
}

// one exchange (= one sequence number) carrying ops[first .. last): every op's window space has been checked to fit a half
static int peer_exchange(vgl_hip_comm *m, const vgl_peer_op *ops, int nops)
{
    vgl_peer_state *s = peer_of(m);
    vgl_hip_ctx *c = m->ctx;
    const int P = m->world, r = m->rank;
    const unsigned long long seq = ++s->seq;
    m->stats.exchanges++;
    const int64_t half0 = (int64_t)VGL_PEER_FLAG_BYTES + (int64_t)(seq & 1ULL) * (int64_t)s->half_bytes;
    std::vector<int64_t> off((size_t)nops);
    int64_t run = 0;
    for (int k = 0; k < nops; k++) { off[(size_t)k] = half0 + run; run += peer_need(m, ops[k]); }
    if (run > (int64_t)s->half_bytes) VGL_FAIL("peer transport: internal error (exchange larger than a window half)");
    // a put is launched even for an empty contribution: the LAST one carries the signal
    for (int k = 0; k < nops; k++) {
        const vgl_peer_op &o = ops[k];
        const int signal = k == nops - 1;
        const char *src = (const char *)o.src;
        int64_t stride = 0, n = o.bytes, dst = off[(size_t)k] + (int64_t)r * round16(o.bytes);
        if (o.kind == PEER_ALLTOALL) stride = o.src_stride;
        if (o.kind == PEER_ALLGATHERV) { src += o.bb[(size_t)r]; n = o.bb[(size_t)r + 1] - o.bb[(size_t)r]; dst = off[(size_t)k] + (o.bb[(size_t)r] - o.bb[0]); }
        hipLaunchKernelGGL(vgl_k_peer_put, dim3(put_grid(n)), dim3(VGL_BLOCK), 0, c->stream, s->tab, r, P, src, stride, n, dst, seq, signal, s->ticket, s->timeout_ticks);
    }
    hipLaunchKernelGGL(vgl_k_peer_wait, dim3(1), dim3(64), 0, c->stream, s->tab, r, P, seq, s->timeout_ticks);
    VGL_HIP_TRY(hipGetLastError());
    for (int k = 0; k < nops; k++) {
        const vgl_peer_op &o = ops[k];
        const char *w = s->window + off[(size_t)k];
        if (o.kind == PEER_ALLREDUCE) {
            // parts lie round16(bytes) apart; the fold wants them `count` elements apart: equal whenever bytes is a multiple of 16, else staged compactly
            const int64_t esz = (o.dtype == VGL_DT_I32 || o.dtype == VGL_DT_F32) ? 4 : 8;
            if (round16(o.bytes) == o.bytes) VGL_TRY(vgl_fold(c, o.bytes / esz, P, w, o.dst, o.dtype, o.op));
            else {
                void *all = nullptr;
                VGL_TRY(vgl_comm_scratch(m, 5, (size_t)o.bytes * (size_t)P, &all));
                vgl_peer_segs sg; sg.n = P;
                for (int p = 0; p < P; p++) { sg.src[p] = w + (int64_t)p * round16(o.bytes); sg.dst[p] = (char *)all + (int64_t)p * o.bytes; sg.bytes[p] = o.bytes; }
                hipLaunchKernelGGL(vgl_k_peer_take, dim3(put_grid(o.bytes * P)), dim3(VGL_BLOCK), 0, c->stream, sg);
                VGL_TRY(vgl_fold(c, o.bytes / esz, P, all, o.dst, o.dtype, o.op));
            }
            continue;
        }
        vgl_peer_segs sg; sg.n = 0;
        int64_t total = 0;
        for (int p = 0; p < P; p++) {
            int64_t n = o.bytes;
            const char *from = w + (int64_t)p * round16(o.bytes);
            char *to = (char *)o.dst + (int64_t)p * o.dst_stride;
            if (o.kind == PEER_ALLGATHERV) {
                if (p == r) continue;                               // in place: this rank's part is where it belongs
                n = o.bb[(size_t)p + 1] - o.bb[(size_t)p];
                from = w + (o.bb[(size_t)p] - o.bb[0]);
                to = (char *)o.dst + o.bb[(size_t)p];
            }
            if (n <= 0) continue;
            sg.src[sg.n] = from; sg.dst[sg.n] = to; sg.bytes[sg.n] = n; sg.n++;
            total += n;
        }
        if (sg.n > 0) hipLaunchKernelGGL(vgl_k_peer_take, dim3(put_grid(total)), dim3(VGL_BLOCK), 0, c->stream, sg);
    }
    hipLaunchKernelGGL(vgl_k_peer_ack, dim3(1), dim3(64), 0, c->stream, s->tab, r, P, seq);
    VGL_HIP_TRY(hipGetLastError());
    return 0;
}

// an op larger than a window half goes in pieces, each an exchange of its own
static int peer_run_alone(vgl_hip_comm *m, const vgl_peer_op &o)
{
    vgl_peer_state *s = peer_of(m);
    const int P = m->world;
    if (peer_need(m, o) <= (int64_t)s->half_bytes) return peer_exchange(m, &o, 1);
    if (o.kind == PEER_ALLGATHERV) {
        // pieces of the byte range [bb[0], bb[P]): every rank sends the part of its slice that falls into the piece
        const int64_t piece = (int64_t)s->half_bytes & ~(int64_t)15;
        for (int64_t lo = o.bb[0]; lo < o.bb[(size_t)P]; lo += piece) {
            const int64_t hi = std::min(o.bb[(size_t)P], lo + piece);
            vgl_peer_op q = o;
            for (int p = 0; p <= P; p++) q.bb[(size_t)p] = std::min(std::max(o.bb[(size_t)p], lo), hi);
            VGL_TRY(peer_exchange(m, &q, 1));
        }
        return 0;
    }
    const int64_t piece = ((int64_t)s->half_bytes / P) & ~(int64_t)15;
    if (piece < 16) VGL_FAIL("peer transport: window too small for this world size");
    for (int64_t lo = 0; lo < o.bytes; lo += piece) {   // the strides stay those of the whole op: a piece lands where it belongs
        vgl_peer_op q = o;
        q.bytes = std::min(piece, o.bytes - lo);
        q.src = (const char *)o.src + lo;
        q.dst = (char *)o.dst + lo;
        VGL_TRY(peer_exchange(m, &q, 1));
    }
    return 0;
}

static int peer_submit(vgl_hip_comm *m, vgl_peer_op &&o)
{
    vgl_peer_state *s = peer_of(m);
    if (m->grouped) { s->pending.emplace_back(std::move(o)); return 0; }
    return peer_run_alone(m, o);
}

int vgl_peer_group_end(vgl_hip_comm *m)
{
    vgl_peer_state *s = peer_of(m);
    std::vector<vgl_peer_op> ops;
    ops.swap(s->pending);
    // as many consecutive ops as fit a half share an exchange; an op that does not fit by itself runs alone (in pieces)
    size_t first = 0;
    while (first < ops.size()) {
        int64_t run = 0;
        size_t last = first;
        while (last < ops.size() && run + peer_need(m, ops[last]) <= (int64_t)s->half_bytes) { run += peer_need(m, ops[last]); last++; }
        if (last == first) { VGL_TRY(peer_run_alone(m, ops[first])); first++; continue; }
        VGL_TRY(peer_exchange(m, ops.data() + first, (int)(last - first)));
        first = last;
    }
    return 0;
}

int vgl_peer_allgather(vgl_hip_comm *m, const void *d_send, void *d_recv, int64_t bytes)
{
    vgl_peer_op o; o.kind = PEER_ALLGATHER; o.src = d_send; o.dst = d_recv; o.bytes = o.src_stride = o.dst_stride = bytes; o.dtype = o.op = 0;
    return peer_submit(m, std::move(o));
}
int vgl_peer_alltoall(vgl_hip_comm *m, const void *d_send, void *d_recv, int64_t bpr)
{
    vgl_peer_op o; o.kind = PEER_ALLTOALL; o.src = d_send; o.dst = d_recv; o.bytes = o.src_stride = o.dst_stride = bpr; o.dtype = o.op = 0;
    return peer_submit(m, std::move(o));
}
int vgl_peer_allgatherv_inplace(vgl_hip_comm *m, void *d_buf, const int64_t *bb)
{
    vgl_peer_op o; o.kind = PEER_ALLGATHERV; o.src = d_buf; o.dst = d_buf; o.bytes = o.src_stride = o.dst_stride = 0; o.dtype = o.op = 0;
    o.bb.assign(bb, bb + m->world + 1);
    return peer_submit(m, std::move(o));
}
int vgl_peer_allreduce(vgl_hip_comm *m, void *d_buf, int64_t count, int dtype, int op)
{
    vgl_peer_op o; o.kind = PEER_ALLREDUCE; o.src = d_buf; o.dst = d_buf; o.dtype = dtype; o.op = op;
    o.bytes = o.src_stride = o.dst_stride = count * ((dtype == VGL_DT_I32 || dtype == VGL_DT_F32) ? 4 : 8);
    return peer_submit(m, std::move(o));
}

// the error word of this rank's window (a spin ran out), read with the small hand-over of vgl_comm_read_small
const unsigned long long *vgl_peer_error_word(vgl_hip_comm *m)
{
    vgl_peer_state *s = peer_of(m);
    return s ? &reinterpret_cast<vgl_peer_flags *>(s->window)->error : nullptr;
}

void vgl_peer_set_timeout_ms(vgl_hip_comm *m, double ms)
{
    if (vgl_peer_state *s = peer_of(m)) s->timeout_ticks = (long long)((ms > 0.0 ? ms : 20000.0) * 1e5);        // wall_clock64: 100 MHz
}

int vgl_peer_setup(vgl_hip_comm *m, size_t window_bytes)
{
    // m->shm (control segment: hosted header + one vgl_peer_record per rank) is attached and m->rank / m->world are set
    vgl_peer_state *s = new vgl_peer_state();
    m->peer = s;
    const int P = m->world, r = m->rank;
    s->half_bytes = (std::max<size_t>(window_bytes, 65536) + 255) & ~(size_t)255;
    double ms = 20000.0;
    if (const char *e = getenv("VGL_PEER_TIMEOUT_MS")) ms = atof(e);
    s->timeout_ticks = (long long)(ms * 1e5);           // wall_clock64: 100 MHz
    const size_t total = VGL_PEER_FLAG_BYTES + 2 * s->half_bytes;
    // fine-grained device memory: stores arriving over xGMI and the polling loads of the owner must not be served from a stale L2 line
    // (coarse-grained hipMalloc memory is only coherent at kernel boundaries); plain hipMalloc when the allocator refuses the flag
    bool fine_grained = true;
    if (hipExtMallocWithFlags((void **)&s->window, total, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        fine_grained = false;
        VGL_HIP_TRY(hipMalloc((void **)&s->window, total));
    }
    VGL_HIP_TRY(hipMemset(s->window, 0, VGL_PEER_FLAG_BYTES));
    VGL_HIP_TRY(hipMalloc((void **)&s->ticket, sizeof(uint32_t) * VGL_TICKET_WORDS));
    VGL_HIP_TRY(hipMemset(s->ticket, 0, sizeof(uint32_t) * VGL_TICKET_WORDS));
    VGL_HIP_TRY(hipDeviceSynchronize());
    vgl_peer_record *rec = reinterpret_cast<vgl_peer_record *>(reinterpret_cast<char *>(m->shm) + 256);
    memset(&rec[r], 0, sizeof(vgl_peer_record));
    rec[r].pid = (int32_t)getpid(); rec[r].device = m->ctx->device; rec[r].raw_ptr = (uint64_t)(uintptr_t)s->window;
    hipError_t e = hipIpcGetMemHandle(&rec[r].handle, s->window);
    const bool have_handle = e == hipSuccess;
    if (!have_handle) (void)hipGetLastError();
    rec[r].pad[0] = have_handle ? 1 : 0;
    rec[r].pad[1] = fine_grained ? 1 : 0;
    {   // which physical GPU the window lives on (device indices are per process): a hash of the PCI bus id
        char bus[64] = {0};
        uint32_t h = 2166136261u;
        if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), m->ctx->device) != hipSuccess) (void)hipGetLastError();
        for (const char *q = bus; *q; q++) h = (h ^ (uint8_t)*q) * 16777619u;
        memcpy(&rec[r].pad[4], &h, sizeof(h));
    }
    VGL_TRY(vgl_hosted_barrier(m));                     // every record is written
    s->opened.assign((size_t)P, false);
    int failed = 0;
    for (int p = 0; p < P; p++) {
        if (p == r) { s->tab.win[p] = s->window; continue; }
        if (rec[p].pid == (int32_t)getpid()) { s->tab.win[p] = (char *)(uintptr_t)rec[p].raw_ptr; continue; }      // a thread of this process
        void *ptr = nullptr;
        if (!rec[p].pad[0] || hipIpcOpenMemHandle(&ptr, rec[p].handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); failed = 1; continue; }
        s->tab.win[p] = (char *)ptr;
        s->opened[(size_t)p] = true;
    }
    for (int p = P; p < VGL_PEER_MAX; p++) s->tab.win[p] = nullptr;
    // a coarse-grained window (the allocator refused the fine-grained flag) is coherent at kernel boundaries only: good enough while every rank sits
    // on ONE GPU (tests, rehearsals), not across GPUs, where polled flags and payload could be served from stale L2 lines
    {
        uint32_t mine = 0;
        memcpy(&mine, &rec[r].pad[4], sizeof(mine));
        for (int p = 0; p < P; p++) {
            uint32_t theirs = 0;
            memcpy(&theirs, &rec[p].pad[4], sizeof(theirs));
            if (theirs != mine && (!rec[p].pad[1] || !rec[r].pad[1])) failed = 1;
        }
    }
    // all ranks agree on success: one failure anywhere voids the transport everywhere (the caller falls back to RCCL)
    std::atomic<uint32_t> *bad = reinterpret_cast<std::atomic<uint32_t> *>(&reinterpret_cast<char *>(m->shm)[192]);
    if (failed) bad->fetch_add(1);
    VGL_TRY(vgl_hosted_barrier(m));
    if (bad->load() != 0) VGL_FAIL("peer transport: a rank's window cannot be mapped by another rank (hipIpcGetMemHandle / hipIpcOpenMemHandle failed)");
    return 0;
}

void vgl_peer_teardown(vgl_hip_comm *m)
{
    vgl_peer_state *s = peer_of(m);
    if (!s) return;
    hipStreamSynchronize(m->ctx->stream);
    if (m->shm) (void)vgl_hosted_barrier(m);            // nobody writes into a window that is about to go
    for (size_t p = 0; p < s->opened.size(); p++) if (s->opened[p]) hipIpcCloseMemHandle(s->tab.win[p]);
    if (m->shm) (void)vgl_hosted_barrier(m);            // ... and nobody frees a window that is still mapped
    if (s->window) hipFree(s->window);
    if (s->ticket) hipFree(s->ticket);
    delete s;
    m->peer = nullptr;
}
