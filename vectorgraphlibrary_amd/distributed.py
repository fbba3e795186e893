"""Streaming shard builder and the size-independent checkers of the edge-cut multi-GPU path.

PRODUCT PATH: the communicator, every collective and the super-step loops live in libvgl_hip.so -- include/vgl_hip.h "multi-GPU behind the
boundary", csrc/{comm,peer,exchange,sharded,bfs_sharded}.hip -- and are reached through vectorgraphlibrary_amd/sharded.py (ctypes) or
GraphAbstractionsHIP::exchange_vertices_array (C++); bench.py --gpus N runs those.

What is here:
  * build_generated_shard: a rank's edge-cut shard of a synthetic graph too large to materialise on one GPU, built by streaming the
    counter-based generator (no build-time communication); used by bench.py and the tests for every sharded run.  One process per GPU, every
    rank owns the out-edges (and in-edges) of a contiguous vertex range with ~E/P edges (VectorCSRGraph::get_mpi_thresholds,
    vect_csr/get_api.hpp:66-94) or of blocks dealt round-robin.
  * bfs_levels_certificate, pagerank_step_residual: checks of a sharded result that need no reference run (bench.py --gpus N, tests).
(The Python protocol model the C++ loops were written from -- super-step drivers against torch.distributed with a pluggable kernel object --
is test infrastructure now: tests/protocol_model.py.)
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

    def row_chunks(rowptr, max_edges=1 << 27):
        # at most chunk_rows rows AND about max_edges entries per piece: the first rows of a degree-sorted RMAT-27 hold billions of entries, and
        # torch's masked indexing fails beyond 2^31 elements (it asked for 6.7e7 GiB in the first round-5 run of the bench)
        r0 = 0
        while r0 < nrows:
            cut = int(torch.searchsorted(rowptr[r0:nrows + 1], int(rowptr[r0]) + max_edges, right=True)) - 1 + r0
            r1 = max(r0 + 1, min(cut, r0 + chunk_rows, nrows))
            yield r0, r1
            r0 = r1

    edges_ok = int(lv[source]) == 1
    nrows = hi - lo
    for r0, r1 in row_chunks(shard.out_rowptr):
        rows, e0, e1 = row_index(shard.out_rowptr, r0, r1)
        ls, ld = lv[rows], lv[shard.out_adj[e0:e1].long()]
        reached = ls > 0
        edges_ok = edges_ok and bool(((ld[reached] > 0) & (ld[reached] <= ls[reached] + 1)).all())
        del rows, ls, ld, reached
    parents_ok = None
    if shard.in_rowptr is not None and shard.in_adj is not None:
        parents_ok = True
        for r0, r1 in row_chunks(shard.in_rowptr):
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
