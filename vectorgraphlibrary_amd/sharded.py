"""Thin ctypes callers of the multi-GPU part of the C ABI (include/vgl_hip.h: vgl_hip_comm_*, vgl_hip_exchange_*,
vgl_hip_*_run_sharded).  The communicator, every collective (RCCL over xGMI) and the super-step loops live in
libvgl_hip.so; nothing here touches torch.distributed on the data path -- a launcher (bench.py, tests) may use it, or a
file, to hand rank 0's 128-byte RCCL id to the other ranks.

Reference: GraphAbstractions::exchange_vertices_array (vgl_compute_api/common/graph_abstractions.h:157-168) and its MPI
implementation (common/mpi_exchange.hpp:110-271); call sites algorithms/sssp/shortest_paths.hpp:136-141, algorithms/pr/pr.hpp:58,127.
"""
import ctypes as C

import torch

from . import lib as _l
from .api import BFS_DIRECTION_OPT, PR_AUTO, _ptr, _stats

TRANSPORT_RCCL, TRANSPORT_HOSTED, TRANSPORT_PEER = 0, 1, 2
ID_BYTES = 128


class Comm:
    """vgl_hip_comm: one per rank and context."""

    def __init__(self, ctx, handle, rank, world):
        self.ctx, self.h, self.rank, self.world = ctx, handle, rank, world

    @staticmethod
    def unique_id():
        """rank 0: the RCCL id (bytes) to be handed to every rank"""
        buf = C.create_string_buffer(ID_BYTES)
        _l.check(_l.load().vgl_hip_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def rccl(cls, ctx, rank, world, unique_id):
        h = C.c_void_p()
        _l.check(ctx.L.vgl_hip_comm_create(ctx.h, int(rank), int(world), C.c_char_p(bytes(unique_id)), C.byref(h)))
        return cls(ctx, h, rank, world)

    @classmethod
    def hosted(cls, ctx, rank, world, name, slot_bytes=1 << 22):
        """host-staged transport through a shared-memory object: ranks = processes of this host, possibly sharing one GPU"""
        h = C.c_void_p()
        _l.check(ctx.L.vgl_hip_comm_create_hosted(ctx.h, int(rank), int(world), name.encode(), int(slot_bytes), C.byref(h)))
        return cls(ctx, h, rank, world)

    @classmethod
    def peer(cls, ctx, rank, world, name, window_bytes=32 << 20):
        """direct peer-to-peer transport: every rank's window in device memory is mapped by the others (hipIpc) and written from kernels;
        raises VglHipError on every rank alike when the windows cannot be mapped (fall back to rccl())"""
        h = C.c_void_p()
        _l.check(ctx.L.vgl_hip_comm_create_peer(ctx.h, int(rank), int(world), name.encode(), int(window_bytes), C.byref(h)))
        return cls(ctx, h, rank, world)

    @classmethod
    def from_torch_group(cls, ctx, group=None):
        """RCCL communicator for the ranks of an initialised torch.distributed group (the group only carries the id)"""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls.rccl(ctx, rank, world, box[0])

    def barrier(self):
        _l.check(self.ctx.L.vgl_hip_comm_barrier(self.h))

    def set_timeout_ms(self, ms):
        """bound of every in-kernel flag wait of the PEER transport from now on (<= 0: the default, 20 s); no-op for the other transports"""
        _l.check(self.ctx.L.vgl_hip_comm_set_timeout_ms(self.h, C.c_double(float(ms))))

    def stats(self):
        st = _l.ExchangeStats()
        _l.check(self.ctx.L.vgl_hip_comm_stats(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def abort(self):
        """this rank gives up: ranks of a hosted / peer communicator waiting at a barrier fail at once instead of after their timeout"""
        if self.h:
            self.ctx.L.vgl_hip_comm_abort(self.h)

    def close(self):
        if self.h:
            self.ctx.L.vgl_hip_comm_destroy(self.h)
            self.h = None

    # ---- exchanges on device tensors (in place, asynchronous on the context's stream) ----
    def allreduce(self, t, op):
        name = {("min", torch.int32): "min_i32", ("min", torch.float32): "min_f32", ("max", torch.float32): "max_f32",
                ("sum", torch.int32): "sum_i32", ("sum", torch.int64): "sum_i64", ("sum", torch.float32): "sum_f32",
                ("sum", torch.float64): "sum_f64"}[(op, t.dtype)]
        _l.check(getattr(self.ctx.L, "vgl_hip_exchange_allreduce_" + name)(self.h, _ptr(t), t.numel()))
        return t

    def allgather(self, send, recv):
        _l.check(self.ctx.L.vgl_hip_exchange_allgather(self.h, _ptr(send), _ptr(recv), send.numel() * send.element_size()))
        return recv

    def allgather_slices(self, array, bounds):
        b = (C.c_int64 * (self.world + 1))(*[int(x) for x in bounds])
        _l.check(self.ctx.L.vgl_hip_exchange_allgather_slices(self.h, _ptr(array), b, array.element_size()))
        return array

    def bitmap_or(self, bits):
        _l.check(self.ctx.L.vgl_hip_exchange_bitmap_or(self.h, _ptr(bits), bits.numel()))
        return bits

    def exchange_changed(self, before, values, take_min=True):
        ch = C.c_int()
        _l.check(self.ctx.L.vgl_hip_exchange_changed_u32(self.h, values.numel(), _ptr(before), _ptr(values), int(bool(take_min)), C.byref(ch)))
        return bool(ch.value)


def _h(comm):
    return comm.h if comm is not None else None


def bfs_run_sharded(graph, comm, source, mode=BFS_DIRECTION_OPT, global_edges=0, gather_levels=True, levels=None, want_stats=True):
    """vgl_hip_bfs_run_sharded; returns (levels, stats).  levels are complete on every rank when gather_levels, else on the owned rows."""
    ctx = graph.ctx
    if levels is None:
        levels = torch.empty(graph.V, dtype=torch.int32, device=ctx.device)
    st = _l.BfsStats()
    _l.check(ctx.L.vgl_hip_bfs_run_sharded(ctx.h, _h(comm), graph.h, int(source), int(mode), int(global_edges), int(bool(gather_levels)),
                                           _ptr(levels), C.byref(st) if want_stats else None))
    return levels, (_stats(st) if want_stats else None)


def sssp_run_sharded(graph, comm, weights, source, dist=None):
    ctx = graph.ctx
    if dist is None:
        dist = torch.empty(graph.V, dtype=torch.float32, device=ctx.device)
    st = _l.SsspStats()
    _l.check(ctx.L.vgl_hip_sssp_run_sharded(ctx.h, _h(comm), graph.h, _ptr(weights), int(source), _ptr(dist), C.byref(st)))
    return dist, _stats(st)


def sswp_run_sharded(graph, comm, capacities, source, widths=None):
    ctx = graph.ctx
    if widths is None:
        widths = torch.empty(graph.V, dtype=torch.float32, device=ctx.device)
    st = _l.SsspStats()
    _l.check(ctx.L.vgl_hip_sswp_run_sharded(ctx.h, _h(comm), graph.h, _ptr(capacities), int(source), _ptr(widths), C.byref(st)))
    return widths, _stats(st)


def cc_run_sharded(graph, comm, comp=None):
    ctx = graph.ctx
    if comp is None:
        comp = torch.empty(graph.V, dtype=torch.int32, device=ctx.device)
    st = _l.CcStats()
    _l.check(ctx.L.vgl_hip_cc_run_sharded(ctx.h, _h(comm), graph.h, _ptr(comp), C.byref(st)))
    return comp, _stats(st)


def pr_run_sharded(graph, comm, iterations, mode=PR_AUTO, ranks=None):
    ctx = graph.ctx
    if ranks is None:
        ranks = torch.empty(graph.V, dtype=torch.float32, device=ctx.device)
    st = _l.PrStats()
    _l.check(ctx.L.vgl_hip_pr_run_sharded(ctx.h, _h(comm), graph.h, int(iterations), int(mode), _ptr(ranks), C.byref(st)))
    return ranks, _stats(st)


def hits_run_sharded(graph, comm, steps, auth=None, hub=None):
    """vgl_hip_hits_run_sharded: replicated f64 authority / hub vectors after `steps` steps"""
    ctx = graph.ctx
    auth = torch.empty(graph.V, dtype=torch.float64, device=ctx.device) if auth is None else auth
    hub = torch.empty(graph.V, dtype=torch.float64, device=ctx.device) if hub is None else hub
    _l.check(ctx.L.vgl_hip_hits_run_sharded(ctx.h, _h(comm), graph.h, int(steps), _ptr(auth), _ptr(hub)))
    return auth, hub
