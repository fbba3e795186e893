"""The Python PROTOCOL MODEL of the edge-cut multi-GPU path (test infrastructure; moved out of the package in round 4: the product path is the
C++ super-step loops of libvgl_hip.so behind vectorgraphlibrary_amd/sharded.py).

Super-step drivers written against torch.distributed (bfs_sharded, ChangedExchange, sssp / sswp / cc / page_rank_sharded) with a pluggable `ops`
object -- the model the C++ loops were written from.  tests/test_distributed_cpu.py exercises it with gloo, world size 2, and a numpy double for
the kernels (no GPU code runs there); tests/test_distributed_gpu.py runs it once through a one-rank RCCL group with HipShardOps (the per-shard
kernels through the C ABI).  Vertex arrays replicated, one exchange per super-step (common/mpi_exchange.hpp:110-150,222-271 in the reference).

Exchange payloads of the model:
  BFS  : bitmap of the vertices discovered in this super-step (V/8 bytes per rank, all-gather + OR) instead of the reference's whole-array exchange
  SSSP : the (index, value) pairs of the distances each rank's step lowered, all-gathered and merged with min (EXCHANGE_RECENTLY_CHANGED,
         mpi_exchange.hpp:110-150); allreduce(min) of the whole f32 array (EXCHANGE_ALL with min_op, shortest_paths.hpp:136-141) only while more
         than V/(2P) entries change per rank
  CC   : the same with the int32 labels
  PR   : all-gather of the owned slices of the new ranks (EXCHANGE_PRIVATE_DATA, pr.hpp:127, mpi_exchange.hpp:222-271)

The shard builder and the certificates are re-exported from the package so that a test can keep one alias for both."""
import ctypes as C
import os

import torch
import torch.distributed as dist

from vectorgraphlibrary_amd import lib as _l
from vectorgraphlibrary_amd.api import _ptr
from vectorgraphlibrary_amd.distributed import _exchanging, _world, bfs_levels_certificate, build_generated_shard, pagerank_step_residual  # noqa: F401


class HipShardOps:
    """per-shard super-step kernels through the C ABI (graph = api.Graph restricted to the owned rows)."""

    def __init__(self, graph, weights=None):
        self.g, self.ctx, self.L = graph, graph.ctx, graph.ctx.L
        self.V = graph.V
        self.device = graph.ctx.device
        self.weights = weights

    def new_i32(self):
        return torch.empty(self.V, dtype=torch.int32, device=self.device)

    def new_f32(self):
        return torch.empty(self.V, dtype=torch.float32, device=self.device)

    def new_words(self, parts):
        return torch.empty(parts * ((self.V + 63) // 64), dtype=torch.int64, device=self.device)

    def scalar(self, values):
        return torch.tensor(values, dtype=torch.int64, device=self.device)

    def bfs_init(self, levels, source):
        _l.check(self.L.vgl_hip_bfs_init(self.ctx.h, self.V, int(source), _ptr(levels)))

    def bfs_step(self, levels, level, visited=None):
        f, m = C.c_int64(), C.c_int64()
        _l.check(self.L.vgl_hip_bfs_step_top_down(self.ctx.h, self.g.h, _ptr(levels), int(level), _ptr(visited), C.byref(f), C.byref(m)))
        return f.value, m.value

    def row_range(self):
        return self.g.row_begin, self.g.row_end

    def bfs_step_bits(self, levels, level, visited, front, mine):
        """top-down step from the replicated frontier bitmap; `mine` receives the bitmap of this shard's discoveries"""
        f, m = C.c_int64(), C.c_int64()
        _l.check(self.L.vgl_hip_bfs_step_top_down_bits(self.ctx.h, self.g.h, _ptr(levels), int(level), _ptr(visited), _ptr(front),
                                                       _ptr(mine), C.byref(f), C.byref(m)))
        return f.value, m.value

    def bfs_step_bu(self, levels, level, visited, front, mine, want_counts=False):
        """bottom-up step over the owned rows; want_counts: wait for and return (found, adjacency entries probed) of THIS shard"""
        if not want_counts:
            _l.check(self.L.vgl_hip_bfs_step_bottom_up(self.ctx.h, self.g.h, _ptr(levels), int(level), _ptr(visited), _ptr(front), _ptr(mine),
                                                       None, None))
            return None
        f, m = C.c_int64(), C.c_int64()
        _l.check(self.L.vgl_hip_bfs_step_bottom_up(self.ctx.h, self.g.h, _ptr(levels), int(level), _ptr(visited), _ptr(front), _ptr(mine),
                                                   C.byref(f), C.byref(m)))
        return f.value, m.value

    def or_parts(self, parts, bits_in, bits_out):
        _l.check(self.L.vgl_hip_bitmap_or_parts(self.ctx.h, bits_out.numel(), int(parts), _ptr(bits_in), _ptr(bits_out)))

    def levels_to_bitmap(self, levels, level, bits):
        _l.check(self.L.vgl_hip_levels_to_bitmap(self.ctx.h, self.V, _ptr(levels), int(level), _ptr(bits)))

    def apply_bitmaps(self, parts, bits_all, levels, level, visited=None, front=None, degrees=None):
        n, d = C.c_int64(), C.c_int64()
        _l.check(self.L.vgl_hip_bfs_apply_bitmaps(self.ctx.h, self.V, parts, _ptr(bits_all), _ptr(levels), int(level), _ptr(visited),
                                                  _ptr(front), _ptr(degrees), C.byref(n), C.byref(d)))
        return n.value, d.value

    def new_id_lists(self, parts, cap):
        return torch.empty(parts * (1 + cap), dtype=torch.int32, device=self.device)

    def bits_to_ids(self, bits, cap, out):
        """out[0] = number of set bits of `bits` (V bits), out[1:1+cap] = ids of the first cap of them (unordered); asynchronous"""
        _l.check(self.L.vgl_hip_bitmap_to_ids(self.ctx.h, (self.V + 63) // 64, _ptr(bits), int(cap), _ptr(out)))

    def list_counts(self, lists, parts, cap):
        return lists.view(parts, 1 + cap)[:, 0].tolist()             # (one small device -> host read)

    def apply_ids(self, parts, cap, lists, levels, level, visited, front, degrees=None):
        n, d = C.c_int64(), C.c_int64()
        _l.check(self.L.vgl_hip_bfs_apply_ids(self.ctx.h, self.V, int(parts), int(cap), _ptr(lists), _ptr(levels), int(level), _ptr(visited),
                                              _ptr(front), _ptr(degrees), C.byref(n), C.byref(d)))
        return n.value, d.value

    def apply_bitmaps_owned(self, parts, bits_all, levels, level, visited, front, degrees=None):
        """apply_bitmaps with the per-vertex part (levels, counts) restricted to the owned rows; returns the OWNED (newly, degree sum)"""
        n, d = C.c_int64(), C.c_int64()
        _l.check(self.L.vgl_hip_bfs_apply_bitmaps_owned(self.ctx.h, self.V, parts, _ptr(bits_all), _ptr(levels), int(level), _ptr(visited), _ptr(front),
                                                        _ptr(degrees), self.g.row_begin, self.g.row_end, C.byref(n), C.byref(d)))
        return n.value, d.value

    def new_pair_lists(self, parts, cap):
        return torch.empty(parts * (1 + 2 * cap), dtype=torch.int32, device=self.device)

    def diff_to_pairs(self, before, after, cap, out):
        """out[0] = number of entries where after != before (may exceed cap), then (index, value bits) pairs; asynchronous"""
        _l.check(self.L.vgl_hip_diff_to_pairs_u32(self.ctx.h, self.V, _ptr(before), _ptr(after), int(cap), _ptr(out)))

    def apply_pairs(self, parts, stride, skip_part, lists, take_min, values):
        _l.check(self.L.vgl_hip_apply_pairs_u32(self.ctx.h, int(parts), int(stride), int(skip_part), _ptr(lists), int(bool(take_min)), self.V, _ptr(values), None))

    def sssp_init(self, d, source):
        _l.check(self.L.vgl_hip_sssp_init(self.ctx.h, self.V, int(source), _ptr(d)))

    def sssp_relax(self, d):
        ch = C.c_int()
        _l.check(self.L.vgl_hip_sssp_relax_owned(self.ctx.h, self.g.h, _ptr(self.weights), _ptr(d), C.byref(ch)))
        return ch.value

    def sswp_init(self, wd, source):
        _l.check(self.L.vgl_hip_sswp_init(self.ctx.h, self.V, int(source), _ptr(wd)))

    def sswp_relax(self, wd):
        ch = C.c_int()
        _l.check(self.L.vgl_hip_sswp_relax_owned(self.ctx.h, self.g.h, _ptr(self.weights), _ptr(wd), C.byref(ch)))
        return ch.value

    def cc_init(self, comp):
        _l.check(self.L.vgl_hip_cc_init(self.ctx.h, self.V, _ptr(comp)))

    def cc_hook(self, comp):
        ch = C.c_int()
        _l.check(self.L.vgl_hip_cc_hook_owned(self.ctx.h, self.g.h, _ptr(comp), C.byref(ch)))
        return ch.value

    def cc_jump(self, comp):
        _l.check(self.L.vgl_hip_cc_jump(self.ctx.h, self.V, _ptr(comp)))

    def indeg_add(self, indeg):
        _l.check(self.L.vgl_hip_indegree_noloops_add(self.ctx.h, self.g.h, _ptr(indeg)))

    def pr_setup(self, indeg, ranks, rdeg):
        _l.check(self.L.vgl_hip_pr_setup(self.ctx.h, self.V, _ptr(indeg), _ptr(ranks), _ptr(rdeg)))

    def pr_iteration(self, indeg, rdeg, ranks, contrib):
        _l.check(self.L.vgl_hip_pr_iteration_owned(self.ctx.h, self.g.h, _ptr(indeg), _ptr(rdeg), _ptr(ranks), _ptr(contrib)))

    def sync(self):
        self.ctx.sync()


def _allreduce(t, op, group):
    if _exchanging(_world(group)[0]):
        dist.all_reduce(t, op=op, group=group)


ALPHA, BETA = 15, 18          # change_state.hpp:5-6


def bfs_sharded(ops, source, group=None, degrees=None, edges=None, equal_ranges=False, two_phase=None, stats=None, sparse_cap=None,
                owned_levels=False):
    """BFS over edge-cut shards; returns the replicated levels array and the number of levels.
    degrees (int32[V] out-degrees of ALL vertices, replicated) + edges (global E) enable direction optimisation: every rank
    evaluates the same switch rule (gpu_change_state, change_state.hpp:100-141) on replicated counters, bottom-up steps scan
    the owned rows' incoming edges.  Without them the traversal is top-down only.
    Exchange per level: all-gather of V/8-byte discovery bitmaps.  equal_ranges=True (every rank owns V/P rows, V/P a multiple
    of 64; build_generated_shard(placement="dealt")) lets the bottom-up levels -- which only discover owned vertices -- gather
    the owned V/(8P)-byte slices instead, P times less traffic; the caller guarantees the flag is the same on all ranks.
    two_phase (needs equal_ranges; default: on for P >= 4): top-down levels, whose discoveries lie anywhere, exchange in two steps
    -- all-to-all of the V/(8P)-byte slices (every rank receives the P versions of ITS slice and ORs them), then all-gather of the
    merged slices -- 2*V/8 bytes per rank instead of P*V/8.
    sparse_cap (default 4096, env VGL_SHARD_SPARSE_CAP; 0 = off): a top-down level whose frontier has at most this many vertices
    first tries to exchange its discoveries as id lists -- every rank all-gathers 4 * (1 + cap) bytes instead of V/8 (16 MiB at
    scale 27), and the merge touches the listed vertices instead of P bitmaps.  When some rank found more than cap vertices (the
    counts travel with the lists, so every rank sees the same thing) the level falls back to the bitmap exchange.  The first and the
    last levels of a traversal are of this kind.
    owned_levels=True: every rank keeps `levels` for the rows it owns only (the returned array is complete on the owned range; other
    entries are unspecified) -- the merge of a level then touches V/P vertices per rank instead of V, and the frontier size / degree
    sum are all-reduced (two scalars).  False (default): replicated levels, as the tests compare them.
    stats (dict, optional): accumulates THIS shard's work -- bu_steps / bu_edges / bu_found, td_steps / td_edges / td_frontier,
    levels -- for the roofline accounting of bench.py; bottom-up steps then wait for their counters (one more host read per level)."""
    P, rank = _world(group)
    V = ops.V
    levels = ops.new_i32()
    ops.bfs_init(levels, source)
    mine = ops.new_words(1)
    exchanging = _exchanging(P)
    everyone = ops.new_words(P) if exchanging else mine
    merged = None
    if equal_ranges and exchanging:
        lo, hi = ops.row_range()
        if V % (64 * P) or lo != rank * (V // P) or hi != lo + V // P:
            raise ValueError("bfs_sharded: equal_ranges needs rank r to own rows [r*V/P, (r+1)*V/P) with V/P a multiple of 64")
        merged = ops.new_words(1)
    if two_phase is None:
        two_phase = merged is not None and P >= 4
    if two_phase and merged is None:
        raise ValueError("bfs_sharded: two_phase needs equal_ranges and more than one rank")
    if two_phase:
        slices_in, my_slice = ops.new_words(1), ops.new_words(1)[:(hi - lo) // 64]      # P received slices / their OR
    if sparse_cap is None:
        sparse_cap = int(os.environ.get("VGL_SHARD_SPARSE_CAP", "4096"))
    if not (exchanging and hasattr(ops, "bits_to_ids")):
        sparse_cap = 0
    if sparse_cap:
        my_list, all_lists = ops.new_id_lists(1, sparse_cap), ops.new_id_lists(P, sparse_cap)
    visited, front = ops.new_words(1), ops.new_words(1)
    ops.levels_to_bitmap(levels, 1, front)
    visited.copy_(front)
    direction_opt = degrees is not None and edges is not None
    F, M = 1, (int(degrees[source]) if direction_opt else 0)
    prevF, visited_total, bottom_up = 0, 0, False
    factor = max(1, (edges // V) // 2) if direction_opt else 1
    level, nlevels = 1, 0
    while True:
        visited_total += F
        if direction_opt:
            if not bottom_up:
                if F > prevF and M >= ((V - visited_total) * factor + V) // ALPHA:
                    bottom_up = True
            elif F <= prevF and F < ((V - visited_total) * factor + V) // (factor * BETA):
                bottom_up = False
        prevF = F
        parts, bits = P, everyone
        if bottom_up:
            counts = ops.bfs_step_bu(levels, level, visited, front, mine, *((True,) if stats is not None else ()))   # owned unvisited vertices look for a parent
            if stats is not None:
                stats["bu_steps"] = stats.get("bu_steps", 0) + 1
                stats["bu_found"] = stats.get("bu_found", 0) + counts[0]
                stats["bu_edges"] = stats.get("bu_edges", 0) + counts[1]
        else:
            fm = ops.bfs_step_bits(levels, level, visited, front, mine)   # owned frontier vertices expand; mine = discoveries
            if stats is not None and fm is not None:
                stats["td_steps"] = stats.get("td_steps", 0) + 1
                stats["td_frontier"] = stats.get("td_frontier", 0) + fm[0]
                stats["td_edges"] = stats.get("td_edges", 0) + fm[1]
        nlevels += 1
        if stats is not None:
            stats["levels"] = stats.get("levels", 0) + 1
        merged_sparse = False
        if exchanging and sparse_cap and not bottom_up and F <= sparse_cap:
            ops.bits_to_ids(mine, sparse_cap, my_list)
            ops.sync()
            dist.all_gather_into_tensor(all_lists, my_list, group=group)
            if max(ops.list_counts(all_lists, P, sparse_cap)) <= sparse_cap:        # the same P counts on every rank
                F, M = ops.apply_ids(P, sparse_cap, all_lists, levels, level + 1, visited, front, degrees if direction_opt else None)
                merged_sparse = True
                if stats is not None:
                    stats["sparse_levels"] = stats.get("sparse_levels", 0) + 1
        if merged_sparse:
            if F == 0:
                break
            level += 1
            continue
        if exchanging:
            ops.sync()
            if bottom_up and merged is not None:
                dist.all_gather_into_tensor(merged, mine[lo // 64:hi // 64], group=group)
                parts, bits = 1, merged
            elif two_phase:
                dist.all_to_all_single(slices_in, mine, group=group)       # slice r of every rank's bitmap -> rank r
                ops.or_parts(P, slices_in, my_slice)
                ops.sync()
                dist.all_gather_into_tensor(merged, my_slice, group=group)
                parts, bits = 1, merged
            else:
                dist.all_gather_into_tensor(everyone, mine, group=group)
        if owned_levels and exchanging and hasattr(ops, "apply_bitmaps_owned"):
            f_own, m_own = ops.apply_bitmaps_owned(parts, bits, levels, level + 1, visited, front, degrees if direction_opt else None)
            fm = ops.scalar([f_own, m_own])
            dist.all_reduce(fm, op=dist.ReduceOp.SUM, group=group)
            F, M = (int(x) for x in fm.tolist())
        else:
            F, M = ops.apply_bitmaps(parts, bits, levels, level + 1, visited, front, degrees if direction_opt else None)
        if F == 0:
            break
        level += 1
    return levels, nlevels


class ChangedExchange:
    """Merge of a replicated 4-byte vertex array after a super-step in which every rank changed some entries of ITS copy
    (EXCHANGE_RECENTLY_CHANGED, common/mpi_exchange.hpp:110-150).  Per step: snapshot() before the local work, then merge():
      1. the entries that differ from the snapshot are compacted into (index, value) pairs on the device,
      2. the P counts are all-gathered (4 bytes each) -- every rank now knows how much everybody changed, which also answers
         "did anything change anywhere" without a separate flag reduction,
      3. while no rank changed more than V / (2 P) entries the pair lists (padded to the next power of two of the largest count)
         are all-gathered and merged with the operator (min / max); otherwise the whole array is all-reduced, which then moves
         fewer bytes than the lists would.
    stats (dict): "list_steps", "dense_steps", "pair_bytes" (bytes this rank received as lists)."""

    def __init__(self, ops, take_min, group=None, stats=None, dense_only=False):
        self.ops, self.take_min, self.group, self.stats = ops, take_min, group, stats if stats is not None else {}
        self.P, self.rank = _world(group)
        self.active = _exchanging(self.P)
        self.lists_ok = self.active and hasattr(ops, "diff_to_pairs") and not dense_only
        if self.lists_ok:
            self.cap = max(64, ops.V // (2 * max(self.P, 1)))
            self.before = ops.new_i32()
            self.mine = ops.new_pair_lists(1, self.cap)
            self.all = ops.new_pair_lists(self.P, self.cap)
            self.counts = ops.new_pair_lists(self.P, 0)              # P int32

    def snapshot(self, values):
        if self.lists_ok:
            self.before.copy_(values.view(torch.int32))

    def merge(self, values, changed_locally):
        """returns True when some rank changed something in this step"""
        if not self.active:
            return bool(changed_locally)
        ops, st = self.ops, self.stats
        if self.lists_ok:
            ops.diff_to_pairs(self.before, values, self.cap, self.mine)
            ops.sync()
            dist.all_gather_into_tensor(self.counts, self.mine[:1], group=self.group)
            counts = self.counts.tolist()
            most = max(counts)
            if most == 0:
                return False
            if most <= self.cap:
                n = 1 << (most - 1).bit_length()
                n = min(n, self.cap)
                stride = 1 + 2 * n
                dist.all_gather_into_tensor(self.all[:self.P * stride], self.mine[:stride].contiguous(), group=self.group)
                ops.apply_pairs(self.P, stride, self.rank, self.all, self.take_min, values)
                st["list_steps"] = st.get("list_steps", 0) + 1
                st["pair_bytes"] = st.get("pair_bytes", 0) + 4 * stride * self.P
                return True
        else:
            ops.sync()
        dist.all_reduce(values, op=dist.ReduceOp.MIN if self.take_min else dist.ReduceOp.MAX, group=self.group)
        st["dense_steps"] = st.get("dense_steps", 0) + 1
        if self.lists_ok:
            return True
        flag = ops.scalar([int(changed_locally)])
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        return bool(int(flag.item()))


def sssp_sharded(ops, source, group=None, stats=None, dense_only=False):
    """Bellman-Ford over edge-cut shards: every rank relaxes the out-edges of its rows into its copy of the distances, the copies are
    merged with min (ChangedExchange).  dense_only=True keeps the reference's EXCHANGE_ALL (whole-array all-reduce) every step."""
    d = ops.new_f32()
    ops.sssp_init(d, source)
    ex = ChangedExchange(ops, True, group, stats, dense_only)
    iters = 0
    while True:
        ex.snapshot(d)
        changed = ops.sssp_relax(d)
        iters += 1
        if not ex.merge(d, changed):
            break
    return d, iters


def sswp_sharded(ops, source, group=None, stats=None, dense_only=False):
    """single-source widest paths over edge-cut shards: every rank relaxes its owned rows, the widths are merged with max
    (the exchange of SSWP::vgl_dijkstra under MPI would be EXCHANGE_ALL with a max op, like shortest_paths.hpp:136-141 with min)"""
    wd = ops.new_f32()
    ops.sswp_init(wd, source)
    ex = ChangedExchange(ops, False, group, stats, dense_only)
    iters = 0
    while True:
        ex.snapshot(wd)
        changed = ops.sswp_relax(wd)
        iters += 1
        if not ex.merge(wd, changed):
            break
    return wd, iters


def cc_sharded(ops, group=None, stats=None, dense_only=False):
    """Shiloach-Vishkin over edge-cut shards (shiloach_vishkin.hpp:7-88): hook over the owned rows, labels merged with min, pointer
    jumping on the merged (replicated) labels -- every rank jumps the same array, so no exchange follows the jump."""
    comp = ops.new_i32()
    ops.cc_init(comp)
    ex = ChangedExchange(ops, True, group, stats, dense_only)
    passes = 0
    while True:
        ex.snapshot(comp)
        changed = ops.cc_hook(comp)
        passes += 1
        if not ex.merge(comp, changed):
            break
        ops.cc_jump(comp)
    return comp, passes


def page_rank_sharded(ops, iterations, row_begin, row_end, group=None, stats=None):
    """PageRank over edge-cut shards: every rank pulls the new ranks of the rows it owns from the replicated old ranks; the owned
    slices are all-gathered (EXCHANGE_PRIVATE_DATA, pr.hpp:127).  Ranks own different numbers of rows in general (edge-balanced
    cut), so the slices travel padded to the longest one: V/P * 4 bytes per rank instead of the V * 4 of a zero-padded sum."""
    P, rank = _world(group)
    indeg = ops.new_i32()
    indeg.zero_()
    ops.indeg_add(indeg)
    exchanging = _exchanging(P)
    if exchanging:
        ops.sync()
        dist.all_reduce(indeg, op=dist.ReduceOp.SUM, group=group)
        bounds = ops.scalar([row_begin, row_end])
        every = ops.scalar([0] * (2 * P))
        dist.all_gather_into_tensor(every, bounds, group=group)
        every = every.view(P, 2).tolist()
        longest = max(hi - lo for lo, hi in every)
        send = ops.new_f32()[:longest]
        recv = ops.new_f32()[:0].new_empty(P * longest)
    ranks, rdeg, contrib = ops.new_f32(), ops.new_f32(), ops.new_f32()
    ops.pr_setup(indeg, ranks, rdeg)
    for _ in range(iterations):
        ops.pr_iteration(indeg, rdeg, ranks, contrib)     # writes the owned rows of `ranks`
        if exchanging:
            ops.sync()
            send[:row_end - row_begin] = ranks[row_begin:row_end]
            dist.all_gather_into_tensor(recv, send, group=group)
            for p, (lo, hi) in enumerate(every):
                if p != rank and hi > lo:
                    ranks[lo:hi] = recv[p * longest:p * longest + (hi - lo)]
            if stats is not None:
                stats["gathered_bytes"] = stats.get("gathered_bytes", 0) + 4 * longest * P
    ops.sync()
    return ranks
