// vgl_comm.h -- internals of the multi-GPU communicator (comm.hip) shared with the super-step loops (sharded.hip, bfs_sharded.hip).
#pragma once
#include "vgl_hip_internal.h"
#include <rccl/rccl.h>
#include <map>
#include <string>
#include <vector>

enum { VGL_DT_I32 = 0, VGL_DT_F32 = 1, VGL_DT_F64 = 2, VGL_DT_I64 = 3, VGL_DT_U64 = 4 };
enum { VGL_OP_SUM = 0, VGL_OP_MIN = 1, VGL_OP_MAX = 2, VGL_OP_OR = 3 };

struct vgl_hosted_header;      // layout of the shared-memory segment (comm.hip)

constexpr int VGL_COMM_SCRATCH_SLOTS = 6;
constexpr int VGL_COMM_SMALL = 256;          // int64 entries of the pinned hand-over buffer (+ sequence number)

struct vgl_hip_comm {
    vgl_hip_ctx *ctx = nullptr;
    int rank = 0, world = 1, transport = VGL_HIP_COMM_RCCL;
    ncclComm_t nccl = nullptr;
    bool force = false;                       // VGL_SHARD_FORCE_COLLECTIVES=1: a world of one still issues every collective (one-GPU tests of the RCCL path)
    bool grouped = false;                     // inside vgl_comm_group_begin / end (RCCL: one fused launch)
    // HOSTED transport
    vgl_hosted_header *shm = nullptr;
    size_t shm_bytes = 0, slot_bytes = 0;
    std::string shm_name;
    uint32_t barrier_gen = 0;
    void *peer = nullptr;                     // PEER transport state (peer.hip)
    // device scratch owned by the communicator, grown on demand (never shrinks)
    void *scratch[VGL_COMM_SCRATCH_SLOTS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t scratch_cap[VGL_COMM_SCRATCH_SLOTS] = {0, 0, 0, 0, 0, 0};
    int64_t *d_small = nullptr;               // VGL_COMM_SMALL device words
    int64_t *h_small = nullptr;               // pinned mirror + sequence number at [VGL_COMM_SMALL]
    int64_t small_seq = 0;
    std::map<uint64_t, std::vector<int64_t>> bounds;                 // row bounds of every rank per graph handle uid (gathered once)
    vgl_hip_exchange_stats stats = {0, 0, 0, 0, 0, 0};
};

static inline int vgl_comm_world(const vgl_hip_comm *m) { return m ? m->world : 1; }
static inline int vgl_comm_rank(const vgl_hip_comm *m) { return m ? m->rank : 0; }
static inline bool vgl_comm_active(const vgl_hip_comm *m) { return m && (m->world > 1 || m->force); }

// scratch slot `slot` with at least `bytes` (contents are lost when it grows)
int vgl_comm_scratch(vgl_hip_comm *m, int slot, size_t bytes, void **out);
// collectives on the context's stream (asynchronous for RCCL; HOSTED drains the stream and meets the other ranks)
int vgl_comm_allreduce(vgl_hip_comm *m, void *d_buf, int64_t count, int dtype, int op);
int vgl_comm_allgather(vgl_hip_comm *m, const void *d_send, void *d_recv, int64_t bytes_per_rank);
int vgl_comm_alltoall(vgl_hip_comm *m, const void *d_send, void *d_recv, int64_t bytes_per_rank);
int vgl_comm_allgatherv_inplace(vgl_hip_comm *m, void *d_buf, const int64_t *byte_bounds);   // world + 1 byte offsets
void vgl_comm_group_begin(vgl_hip_comm *m);
int vgl_comm_group_end(vgl_hip_comm *m);
// n <= VGL_COMM_SMALL device words -> host (pinned hand-over + poll; a few microseconds after the stream reaches it)
int vgl_comm_read_small(vgl_hip_comm *m, const int64_t *d_vals, int n, int64_t *h_out);
// all-reduce of a few HOST values (setup paths; synchronises)
int vgl_comm_allreduce_host_i64(vgl_hip_comm *m, int64_t *vals, int n, int op);
// [row_begin, row_end) of every rank for this graph handle: world + 1 bounds when the ranges tile [0, V) in rank order, else an error;
// bounds[world + 1] = the number of rows with incoming edges over all ranks (0 when a rank has no incoming lists)
int vgl_comm_row_bounds(vgl_hip_comm *m, const vgl_hip_graph *g, const int64_t **bounds);

// host-side barrier of the ranks attached to m->shm (HOSTED data path; set-up and tear-down of PEER)
int vgl_hosted_barrier(vgl_hip_comm *m);
// PEER transport (peer.hip): windows in device memory mapped by every rank, arrival / consumption flags in the windows
int vgl_peer_setup(vgl_hip_comm *m, size_t window_bytes);
void vgl_peer_teardown(vgl_hip_comm *m);
int vgl_peer_allreduce(vgl_hip_comm *m, void *d_buf, int64_t count, int dtype, int op);
int vgl_peer_allgather(vgl_hip_comm *m, const void *d_send, void *d_recv, int64_t bytes);
int vgl_peer_alltoall(vgl_hip_comm *m, const void *d_send, void *d_recv, int64_t bpr);
int vgl_peer_allgatherv_inplace(vgl_hip_comm *m, void *d_buf, const int64_t *bb);
int vgl_peer_group_end(vgl_hip_comm *m);
const unsigned long long *vgl_peer_error_word(vgl_hip_comm *m);
int vgl_comm_check(vgl_hip_comm *m);          // PEER: fails when a flag wait of an earlier exchange ran out (reads the window's error word)
void vgl_peer_set_timeout_ms(vgl_hip_comm *m, double ms);
int vgl_fold(vgl_hip_ctx *c, int64_t n, int parts, const void *in, void *out, int dtype, int op);      // out[i] = fold over p of in[p * n + i] (comm.hip)

// enqueue-only forms of the owned-row super-steps (no host read; defined next to their kernels)
int vgl_sssp_relax_enqueue(vgl_hip_ctx *c, vgl_hip_graph *g, const float *d_weights, float *d_dist, bool widest);
int vgl_cc_hook_launch(vgl_hip_ctx *c, vgl_hip_graph *g, int32_t *comp);
int vgl_pr_iteration(vgl_hip_ctx *c, vgl_hip_graph *g, const int32_t *indeg, const float *rdeg, float *ranks, float *contrib,
                     float *ranks_out, int mode);
int vgl_pr_longest_row(vgl_hip_ctx *c, vgl_hip_graph *g, int64_t *out);
int vgl_pr_env_mode(vgl_hip_ctx *c, int mode, int *out);
int vgl_hits_init(vgl_hip_ctx *c, int32_t V, double *d_auth, double *d_hub);
int vgl_hits_pull_owned(vgl_hip_ctx *c, vgl_hip_graph *g, bool incoming, const double *x, double *out, double *d_sumsq);
int vgl_hits_scale_owned(vgl_hip_ctx *c, vgl_hip_graph *g, double *x, const double *d_sumsq);
