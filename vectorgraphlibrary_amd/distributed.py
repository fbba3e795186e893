"""Streaming shard builder + the earlier Python protocol model of the edge-cut multi-GPU path.

PRODUCT PATH (round 3): the communicator, every collective and the super-step loops live in libvgl_hip.so -- include/vgl_hip.h
"multi-GPU behind the boundary", csrc/{comm,exchange,sharded,bfs_sharded}.hip -- and are reached through vectorgraphlibrary_amd/sharded.py
(ctypes) or GraphAbstractionsHIP::exchange_vertices_array (C++); bench.py --gpus N runs those.

What stays here:
  * build_generated_shard: a rank's edge-cut shard of a synthetic graph too large to materialise on one GPU, built by streaming the
    counter-based generator (no build-time communication); used by bench.py and the tests for every sharded run.
  * the super-step drivers written against torch.distributed (bfs_sharded, ChangedExchange, sssp / sswp / cc / page_rank_sharded) with
    a pluggable `ops` object: the PROTOCOL MODEL the C++ loops were written from.  tests/test_distributed_cpu.py exercises it with gloo,
    world size 2, and a numpy double for the kernels (no GPU code runs there); tests/test_distributed_gpu.py runs it once through a
    one-rank RCCL group.  One process per GPU, vertex arrays replicated, every rank owns the out-edges (and in-edges) of a contiguous
    vertex range with ~E/P edges (VectorCSRGraph::get_mpi_thresholds, vect_csr/get_api.hpp:66-94), one exchange per super-step
    (common/mpi_exchange.hpp:110-150,222-271 in the reference).

Exchange payloads of the model:
  BFS  : bitmap of the vertices discovered in this super-step (V/8 bytes per rank, all-gather + OR) instead of the
         reference's whole-array exchange
  SSSP : the (index, value) pairs of the distances each rank's step lowered, all-gathered and merged with min
         (EXCHANGE_RECENTLY_CHANGED, mpi_exchange.hpp:110-150); allreduce(min) of the whole f32 array (EXCHANGE_ALL with min_op,
         shortest_paths.hpp:136-141) only while more than V/(2P) entries change per rank
  CC   : the same with the int32 labels
  PR   : all-gather of the owned slices of the new ranks (EXCHANGE_PRIVATE_DATA, pr.hpp:127, mpi_exchange.hpp:222-271)
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import lib as _l
from .api import _ptr


def _exchanging(P, group=None):
    """True when the super-step exchanges run.  VGL_SHARD_FORCE_COLLECTIVES=1 runs them in a one-rank process group too: every
    collective call of the N > 1 path then executes through RCCL on a one-GPU box (tests/test_distributed_gpu.py)"""
    return P > 1 or (os.environ.get("VGL_SHARD_FORCE_COLLECTIVES") == "1" and dist.is_available() and dist.is_initialized())


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


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


def build_generated_shard(ctx, scale, edge_factor, seed, rank, world, kind="rmat", renumber="total", chunk_edges=1 << 27,
                          placement="ranges", piece_edges=1 << 30, symmetric=False, with_incoming=True):
    """This rank's edge-cut shard of a synthetic graph that is too large to materialise on one GPU (RMAT-27 over 8 GPUs,
    BASELINE.json configs[2]).  Every rank streams the whole counter-based edge list in chunks twice -- once for the degree
    histograms (renumbering + partition bounds, identical on all ranks), once to keep the edges whose source (outgoing
    direction) or destination (incoming direction) it owns -- so no rank ever holds more than chunk + E/P edges and there is
    no build-time communication.  The result equals Graph.from_coo(whole graph, renumber).shard(lo, hi) up to the order of
    the incoming adjacency lists (generation order here, outgoing-CSR order there; vgl_graph.hpp:57-68).
    placement="ranges": contiguous edge-balanced row ranges of the (renumbered) graph (get_mpi_thresholds,
    vect_csr/get_api.hpp:66-94).  placement="dealt": the 64-vertex blocks of the (renumbered) order are dealt round-robin to the
    ranks and the stored numbering is made rank-major, so that every rank owns exactly V/P consecutive stored ids with the
    same mix of hubs and leaves -- rows AND edges are balanced (bottom-up work follows rows, not edges) and bottom-up levels can
    exchange owned bitmap slices (bfs_sharded(equal_ranges=True)).  shard.fwd/bwd map original <-> stored ids either way.
    symmetric=True: every generated edge is stored in both directions (the undirected input of the cc app, apps/cc/cc.cpp:24-36:
    2 * E stored edges).  with_incoming=False: only the outgoing CSR is built (push-only algorithms: CC, PageRank's pull over the
    outgoing lists).
    Returns (shard, out_degrees[V] replicated, bounds)."""
    from .api import Graph
    V, E = 1 << scale, (1 << scale) * edge_factor

    def chunk(e0, n):
        if kind == "rmat":
            s, d = ctx.gen_rmat(scale, edge_factor, seed, first_edge=e0, count=n)
        else:
            s, d = ctx.gen_uniform(scale, edge_factor, seed, first_edge=e0, count=n)
        if symmetric:
            s, d = torch.cat([s, d]), torch.cat([d, s])
        return s, d

    chunks = [(e0, min(chunk_edges, E - e0)) for e0 in range(0, E, chunk_edges)]
    outdeg = torch.zeros(V, dtype=torch.int32, device=ctx.device)
    key = torch.zeros(V, dtype=torch.int32, device=ctx.device) if renumber in ("in", "total") else None
    for e0, n in chunks:
        s, d = chunk(e0, n)
        ctx.degree_hist_add(s, d, "out", outdeg)
        if key is not None:
            ctx.degree_hist_add(s, d, renumber, key)
        del s, d
    fwd = bwd = None
    if renumber:
        fwd, bwd = ctx.degree_order_from_degrees(outdeg if key is None else key)
        outdeg = ctx.permute(bwd, outdeg)                       # out-degrees in the sorted numbering
    del key
    if placement == "dealt":
        if V % (64 * world):
            raise ValueError("placement='dealt' needs V to be a multiple of 64 * world")
        pos = torch.arange(V, device=ctx.device, dtype=torch.int64)
        blk = pos >> 6
        deal = ((blk % world) * (V // world) + (blk // world) * 64 + (pos & 63)).to(torch.int32)    # position -> stored id
        undeal = torch.empty_like(deal)
        undeal[deal.long()] = pos.to(torch.int32)
        del pos, blk
        fwd = deal if fwd is None else ctx.relabel(deal, fwd)
        bwd = undeal if bwd is None else ctx.permute(undeal, bwd)      # bwd'[stored] = bwd[position of stored]
        outdeg = ctx.permute(undeal, outdeg)
        del deal, undeal
        bounds = [p * (V // world) for p in range(world + 1)]
    elif placement == "ranges":
        rowptr = torch.zeros(V + 1, dtype=torch.int64, device=ctx.device)
        torch.cumsum(outdeg, 0, dtype=torch.int64, out=rowptr[1:])
        bounds = ctx.partition_rows(rowptr, world)
        del rowptr
    else:
        raise ValueError("placement must be 'ranges' or 'dealt'")
    lo, hi = bounds[rank], bounds[rank + 1]
    # The stable COO -> CSR sort takes at most 2^31 - 16 edges per call.  A shard with more (RMAT-27 on one or two GPUs) is built in
    # row-range pieces of at most `piece_edges` edges each: the generator is streamed once per piece and the piece CSRs are
    # concatenated (rows are independent, so the result is the same CSR).  In-degrees for the incoming direction are counted here.
    indeg_owned = torch.zeros(V, dtype=torch.int32, device=ctx.device)
    for e0, n in (chunks if with_incoming else []):
        s, d = chunk(e0, n)
        if fwd is not None:
            s, d = ctx.relabel(fwd, s), ctx.relabel(fwd, d)
        ctx.degree_hist_add(s, d, "in", indeg_owned)
        del s, d

    def pieces(deg):
        """row ranges [a, b) inside [lo, hi) whose rows hold at most piece_edges edges (a single row above the bound is its own piece)"""
        pre = torch.zeros(hi - lo + 1, dtype=torch.int64, device=ctx.device)
        torch.cumsum(deg[lo:hi], 0, dtype=torch.int64, out=pre[1:])
        total = int(pre[-1])
        if total <= piece_edges:
            return [(lo, hi)]
        cuts, a = [], 0
        while a < hi - lo:
            b = int(torch.searchsorted(pre, pre[a] + piece_edges, right=True)) - 1
            b = max(b, a + 1)
            cuts.append((lo + a, lo + min(b, hi - lo)))
            a = min(b, hi - lo)
        return cuts

    def build(direction, ranges):
        rps, adjs, base = [], [], 0
        for a, b in ranges:
            ks, kd = [], []
            for e0, n in chunks:
                s, d = chunk(e0, n)
                if fwd is not None:
                    s, d = ctx.relabel(fwd, s), ctx.relabel(fwd, d)
                if direction == "in":
                    s, d = d, s                                  # transposed: rows of the incoming direction are destinations
                m = (s >= a) & (s < b)
                ks.append(s[m]); kd.append(d[m])
                del s, d, m
            cs = torch.cat(ks) if ks else torch.empty(0, dtype=torch.int32, device=ctx.device)
            cd = torch.cat(kd) if kd else torch.empty(0, dtype=torch.int32, device=ctx.device)
            del ks, kd
            rp, adj, _ = ctx.coo_to_csr(V, cs, cd, a, b)
            del cs, cd
            rps.append(rp[:-1] + base)
            adjs.append(adj)
            base += int(rp[-1])
            del rp
        rowptr = torch.cat(rps + [torch.tensor([base], dtype=torch.int64, device=ctx.device)])
        adj = adjs[0].clone() if len(adjs) == 1 else torch.cat(adjs)
        return rowptr, adj

    out_pieces, in_pieces = pieces(outdeg), pieces(indeg_owned)
    del indeg_owned
    if not with_incoming:
        orp, oadj = build("out", out_pieces)
        irp = iadj = None
    elif len(out_pieces) == 1 and len(in_pieces) == 1:           # the usual case: both directions in ONE pass over the generator
        keep = {"os": [], "od": [], "is": [], "id": []}
        for e0, n in chunks:
            s, d = chunk(e0, n)
            if fwd is not None:
                s, d = ctx.relabel(fwd, s), ctx.relabel(fwd, d)
            m = (s >= lo) & (s < hi)
            keep["os"].append(s[m]); keep["od"].append(d[m])
            m = (d >= lo) & (d < hi)
            keep["is"].append(d[m]); keep["id"].append(s[m])     # transposed: rows of the incoming direction are destinations
            del s, d, m
        cat = {k: (torch.cat(v) if v else torch.empty(0, dtype=torch.int32, device=ctx.device)) for k, v in keep.items()}
        del keep
        orp, oadj, _ = ctx.coo_to_csr(V, cat["os"], cat["od"], lo, hi)
        del cat["os"], cat["od"]
        irp, iadj, _ = ctx.coo_to_csr(V, cat["is"], cat["id"], lo, hi)
        del cat
        oadj, iadj = oadj.clone(), iadj.clone()
    else:
        orp, oadj = build("out", out_pieces)
        irp, iadj = build("in", in_pieces)
    g = Graph(ctx, V, orp, oadj, irp, iadj, lo, hi)
    g.fwd, g.bwd = fwd, bwd
    return g, outdeg, bounds


def bfs_levels_certificate(levels, shard, source, chunk_rows=1 << 22):
    """A size-independent proof that `levels` (int32[V], 1 at the source, -1 for unreached, all vertices) are breadth-first levels, checked on
    ONE shard's rows (every rank checks its own; the conjunction over the ranks is the proof):
      edges   : for every owned out-edge u -> v with u reached, v is reached and levels[v] <= levels[u] + 1   (no level is too large,
                and everything reachable is reached);
      parents : every owned reached vertex other than the source has an in-neighbour exactly one level below   (no level is too small,
                and everything reached is reachable) -- needs the shard's incoming lists.
    Returns (edges_ok, parents_ok); parents_ok is None without incoming lists.  Test / bench infrastructure: plain torch ops on the GPU."""
    import torch
    lv = levels.long()
    lo, hi = int(shard.row_begin), int(shard.row_end)
    dev = lv.device

    def row_index(rowptr, r0, r1):
        deg = rowptr[r0 + 1:r1 + 1] - rowptr[r0:r1]
        return torch.repeat_interleave(torch.arange(lo + r0, lo + r1, device=dev), deg), int(rowptr[r0]), int(rowptr[r1])

    edges_ok = int(lv[source]) == 1
    nrows = hi - lo
    for r0 in range(0, nrows, chunk_rows):
        r1 = min(nrows, r0 + chunk_rows)
        rows, e0, e1 = row_index(shard.out_rowptr, r0, r1)
        ls, ld = lv[rows], lv[shard.out_adj[e0:e1].long()]
        reached = ls > 0
        edges_ok = edges_ok and bool(((ld[reached] > 0) & (ld[reached] <= ls[reached] + 1)).all())
        del rows, ls, ld, reached
    parents_ok = None
    if shard.in_rowptr is not None and shard.in_adj is not None:
        parents_ok = True
        for r0 in range(0, nrows, chunk_rows):
            r1 = min(nrows, r0 + chunk_rows)
            rows, e0, e1 = row_index(shard.in_rowptr, r0, r1)
            has = torch.zeros(r1 - r0, dtype=torch.int32, device=dev)
            good = (lv[shard.in_adj[e0:e1].long()] == lv[rows] - 1) & (lv[rows] > 1)
            has.index_add_(0, rows - (lo + r0), good.to(torch.int32))
            mine = lv[lo + r0:lo + r1]
            need = mine > 1                                            # reached, not the source
            parents_ok = parents_ok and bool((has[need] > 0).all()) and bool(((mine == 1).sum() == (1 if lo + r0 <= source < lo + r1 else 0)))
            del rows, has, good, mine, need
    return edges_ok, parents_ok


def pagerank_step_residual(shard, old, new, allreduce_sum=None, chunk_rows=1 << 22):
    """One PageRank iteration of the reference recipe (pr.hpp:31-136; the CPU checker restates it expression for
    expression) recomputed in f64 with plain torch on ONE shard's rows: max over the owned rows of |new - F(old)| / F(old), where
    F(old)[u] = k + d * (sum over u's stored neighbours v != u of old[v] * rdeg[v] + dangling).  `old` / `new` are the rank vectors after
    i and i + 1 iterations (all V entries); allreduce_sum(tensor) adds the in-degree counts of the other ranks' rows (None: one rank).
    Test / bench infrastructure -- an independent check of a full-size iteration that needs neither the whole graph nor the CPU oracle."""
    import torch
    V = int(old.numel())
    lo, hi = int(shard.row_begin), int(shard.row_end)
    dev = old.device
    d32 = torch.tensor(0.85, dtype=torch.float32)
    k = float(((1.0 - d32.double()) / torch.tensor(float(V), dtype=torch.float32).double()).float())
    d = float(d32)
    indeg = torch.zeros(V, dtype=torch.int64, device=dev)
    nrows = hi - lo
    for r0 in range(0, nrows, chunk_rows):
        r1 = min(nrows, r0 + chunk_rows)
        e0, e1 = int(shard.out_rowptr[r0]), int(shard.out_rowptr[r1])
        rows = torch.repeat_interleave(torch.arange(lo + r0, lo + r1, device=dev), shard.out_rowptr[r0 + 1:r1 + 1] - shard.out_rowptr[r0:r1])
        nb = shard.out_adj[e0:e1].long()
        indeg += torch.bincount(nb[nb != rows], minlength=V)
        del rows, nb
    if allreduce_sum is not None:
        allreduce_sum(indeg)
    rdeg = torch.where(indeg > 0, (1.0 / indeg.double()).float(), torch.zeros(V, dtype=torch.float32, device=dev)).double()
    o = old.double()
    dangling = float(((old / V).double()[indeg == 0]).sum().float())
    contrib = o * rdeg
    worst = 0.0
    for r0 in range(0, nrows, chunk_rows):
        r1 = min(nrows, r0 + chunk_rows)
        e0, e1 = int(shard.out_rowptr[r0]), int(shard.out_rowptr[r1])
        rows = torch.repeat_interleave(torch.arange(r0, r1, device=dev), shard.out_rowptr[r0 + 1:r1 + 1] - shard.out_rowptr[r0:r1])
        nb = shard.out_adj[e0:e1].long()
        vals = torch.where(nb != rows + lo, contrib[nb], torch.zeros((), dtype=torch.float64, device=dev))
        acc = torch.zeros(r1 - r0, dtype=torch.float64, device=dev).index_add_(0, rows - r0, vals)
        expect = k + d * (acc + dangling)
        worst = max(worst, float(((new[lo + r0:lo + r1].double() - expect).abs() / expect).max()))
        del rows, nb, vals, acc, expect
    return worst


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
