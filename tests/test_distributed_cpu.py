"""CPU test (gloo, world_size 2) of the multi-GPU super-step drivers of the Python protocol model (tests/protocol_model.py; the shipped loops are C++): the
edge-cut partition, the per-step exchange (bitmap all-gather / min all-reduce / owned-slice sum) and the termination
logic.  The per-shard kernels are replaced by a numpy test double defined HERE (no GPU code can run in this container);
the HIP kernels behind the same `ops` interface are covered by tests/test_gpu_parity.py::test_sharded_super_steps_single_process."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FLT_MAX = np.float32(3.4028234663852886e38)


class NumpyShardOps:
    """numpy stand-in for HipShardOps over the rows [lo,hi) of a host CSR (test double, not product code)."""

    def __init__(self, V, rowptr, adj, w, lo, hi):
        self.V, self.rowptr, self.adj, self.w, self.lo, self.hi = V, rowptr, adj, w, lo, hi
        self.src = np.repeat(np.arange(V, dtype=np.int64), np.diff(rowptr))
        e0, e1 = rowptr[lo], rowptr[hi]
        self.es, self.ed = self.src[e0:e1], adj[e0:e1].astype(np.int64)
        self.ew = w[e0:e1] if w is not None else None
        own = (adj >= lo) & (adj < hi)                    # incoming edges of the owned vertices (bottom-up steps)
        self.in_src, self.in_dst = self.src[own], adj[own].astype(np.int64)

    def new_i32(self): return torch.empty(self.V, dtype=torch.int32)
    def new_f32(self): return torch.empty(self.V, dtype=torch.float32)
    def new_words(self, parts): return torch.empty(parts * ((self.V + 63) // 64), dtype=torch.int64)
    def scalar(self, v): return torch.tensor(v, dtype=torch.int64)
    def sync(self): pass

    def bfs_init(self, levels, source):
        levels.fill_(-1); levels[source] = 1

    def bfs_step(self, levels, level, visited=None):
        lv = levels.numpy()
        m = lv[self.es] == level
        tgt = self.ed[m]
        tgt = tgt[lv[tgt] == -1]
        lv[tgt] = level + 1
        return int(((lv[self.lo:self.hi]) == level).sum()), int(m.sum())

    def row_range(self):
        return self.lo, self.hi

    def bfs_step_bits(self, levels, level, visited, front, mine):
        lv = levels.numpy()
        fr, vis = self._unpack(front), self._unpack(visited)
        m = fr[self.es]
        tgt = self.ed[m]
        tgt = tgt[~vis[tgt]]
        lv[tgt] = level + 1
        found = np.zeros(self.V, bool)
        found[tgt] = True
        self._pack(found, mine)
        return int(fr[self.lo:self.hi].sum()), int(m.sum())

    def new_id_lists(self, parts, cap): return torch.zeros(parts * (1 + cap), dtype=torch.int32)

    def bits_to_ids(self, bits, cap, out):
        ids = np.nonzero(self._unpack(bits))[0].astype(np.int32)
        o = out.numpy()
        o[0] = len(ids)
        o[1:1 + min(cap, len(ids))] = ids[::-1][:cap]                # any order, any subset beyond cap: take the LAST ones to prove it

    def list_counts(self, lists, parts, cap): return lists.view(parts, 1 + cap)[:, 0].tolist()

    def apply_ids(self, parts, cap, lists, levels, level, visited, front, degrees=None):
        l = lists.numpy().reshape(parts, 1 + cap)
        ids = np.unique(np.concatenate([l[p, 1:1 + l[p, 0]] for p in range(parts)])) if parts else np.zeros(0, np.int32)
        vis = self._unpack(visited)
        new = ids[~vis[ids]]
        levels.numpy()[new] = level
        vis[new] = True
        self._pack(vis, visited)
        fr = np.zeros(self.V, bool)
        fr[new] = True
        self._pack(fr, front)
        return int(len(new)), (int(degrees.numpy()[new].sum()) if degrees is not None else 0)

    def or_parts(self, parts, bits_in, bits_out):
        bits_out.numpy()[:] = np.bitwise_or.reduce(bits_in.numpy().reshape(parts, -1), axis=0)

    def levels_to_bitmap(self, levels, level, bits):
        b = np.zeros(((self.V + 63) // 64) * 64, np.uint8)
        b[:self.V] = levels.numpy() == level
        bits.numpy().view(np.uint8)[:] = np.packbits(b, bitorder="little")

    def apply_bitmaps(self, parts, bits_all, levels, level, visited=None, front=None, degrees=None):
        words = (self.V + 63) // 64
        w = bits_all.numpy().reshape(parts, words)
        merged = np.bitwise_or.reduce(w, axis=0)
        on = np.unpackbits(merged.view(np.uint8), bitorder="little")[:self.V].astype(bool)
        lv = levels.numpy()
        lv[on & (lv == -1)] = level
        new = on & (lv == level)
        if front is not None:
            self._pack(new, front)
        if visited is not None:
            visited.numpy()[:] |= self._packed(new)
        return int(new.sum()), (int(degrees.numpy()[new].sum()) if degrees is not None else 0)

    def apply_bitmaps_owned(self, parts, bits_all, levels, level, visited, front, degrees=None):
        words = (self.V + 63) // 64
        merged = np.bitwise_or.reduce(bits_all.numpy().reshape(parts, words), axis=0)
        on = np.unpackbits(merged.view(np.uint8), bitorder="little")[:self.V].astype(bool)
        vis = self._unpack(visited)
        new = on & ~vis
        self._pack(vis | new, visited)
        self._pack(new, front)
        own = np.zeros(self.V, bool); own[self.lo:self.hi] = True
        mine = new & own
        levels.numpy()[mine] = level
        return int(mine.sum()), (int(degrees.numpy()[mine].sum()) if degrees is not None else 0)

    def _packed(self, mask):
        b = np.zeros(((self.V + 63) // 64) * 64, np.uint8)
        b[:self.V] = mask
        return np.packbits(b, bitorder="little").view(np.int64)

    def _pack(self, mask, out):
        out.numpy()[:] = self._packed(mask)

    def _unpack(self, bits):
        return np.unpackbits(bits.numpy().view(np.uint8), bitorder="little")[:self.V].astype(bool)

    def bfs_step_bu(self, levels, level, visited, front, mine, want_counts=False):
        """owned unvisited vertices with an in-neighbour in the frontier get level+1 (uses the transposed owned edges)"""
        lv = levels.numpy()
        fr, vis = self._unpack(front), self._unpack(visited)
        found = np.zeros(self.V, bool)
        m = fr[self.in_src] & ~vis[self.in_dst]
        found[self.in_dst[m]] = True
        lv[found] = level + 1
        self._pack(found, mine)
        return (int(found.sum()), int(m.sum())) if want_counts else None

    def new_pair_lists(self, parts, cap): return torch.zeros(parts * (1 + 2 * cap), dtype=torch.int32)

    def diff_to_pairs(self, before, after, cap, out):
        b, a = before.numpy().view(np.int32), after.numpy().view(np.int32)
        idx = np.nonzero(a != b)[0][::-1]                            # any order
        o = out.numpy()
        o[0] = len(idx)
        k = min(cap, len(idx))
        o[1:1 + 2 * k:2] = idx[:k]
        o[2:2 + 2 * k:2] = a[idx[:k]]

    def apply_pairs(self, parts, stride, skip_part, lists, take_min, values):
        l = lists.numpy()
        v = values.numpy().view(np.int32)
        for p in range(parts):
            if p == skip_part:
                continue
            n = l[p * stride]
            idx, val = l[p * stride + 1:p * stride + 1 + 2 * n:2], l[p * stride + 2:p * stride + 2 + 2 * n:2]
            (np.minimum if take_min else np.maximum).at(v, idx, val)

    def sssp_init(self, d, source):
        d.fill_(float(FLT_MAX)); d[source] = 0

    def sssp_relax(self, d):
        dv = d.numpy()
        live = dv[self.es] < FLT_MAX
        cand = (dv[self.es][live] + self.ew[live]).astype(np.float32)
        before = dv.copy()
        np.minimum.at(dv, self.ed[live], cand)
        return int((before != dv).any())

    def sswp_init(self, wd, source):
        wd.fill_(0.0); wd[source] = float(FLT_MAX)

    def sswp_relax(self, wd):
        wv = wd.numpy()
        live = wv[self.es] > 0
        cand = np.minimum(wv[self.es][live], self.ew[live]).astype(np.float32)
        before = wv.copy()
        np.maximum.at(wv, self.ed[live], cand)
        return int((before != wv).any())

    def cc_init(self, comp): comp.copy_(torch.arange(self.V, dtype=torch.int32))

    def cc_hook(self, comp):
        c = comp.numpy()
        before = c.copy()
        np.minimum.at(c, self.ed, c[self.es])
        return int((before != c).any())

    def cc_jump(self, comp):
        c = comp.numpy()
        while True:
            n = c[c]
            if (n == c).all():
                break
            c[:] = n

    def indeg_add(self, indeg):
        m = self.es != self.ed
        np.add.at(indeg.numpy(), self.ed[m], 1)

    def pr_setup(self, indeg, ranks, rdeg):
        ranks.fill_(float(np.float32(1.0 / self.V)))
        dg = indeg.numpy()
        rdeg.numpy()[:] = np.where(dg == 0, np.float32(0), (1.0 / np.maximum(dg, 1)).astype(np.float32))

    def pr_iteration(self, indeg, rdeg, ranks, contrib):
        V = self.V
        old = ranks.numpy().copy()
        c = (old * rdeg.numpy()).astype(np.float32)
        dang = np.float32((old[indeg.numpy() == 0] / np.float32(V)).astype(np.float64).sum())
        d, k = np.float32(0.85), np.float32((1.0 - float(np.float32(0.85))) / float(np.float32(V)))
        out = ranks.numpy()
        for u in range(self.lo, self.hi):                 # f32 adjacency-order sums, like the kernel
            acc = np.float32(0)
            for p in range(self.rowptr[u], self.rowptr[u + 1]):
                v = self.adj[p]
                if v != u:
                    acc = np.float32(acc + c[v])
            out[u] = np.float32(k + np.float32(d * np.float32(acc + dang)))


def _worker(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    import protocol_model as vd
    scale, ef, seed = 9, 8, 13
    V = 1 << scale
    src, dst = O.gen_rmat(scale, ef, seed)
    rowptr, adj, perm = O.coo_to_csr(V, src, dst)
    w = O.gen_weights(len(src), seed)[perm]
    # edge-balanced cut rounded to multiples of 64 (same rule as vgl_hip_partition_rows)
    bounds = [0]
    for p in range(1, world):
        b = int(np.searchsorted(rowptr, len(adj) * p // world))
        bounds.append(min(V, max(bounds[-1], (b + 32) // 64 * 64)))
    bounds.append(V)
    ops = NumpyShardOps(V, rowptr, adj, w, bounds[rank], bounds[rank + 1])
    source = O.pick_source(rowptr, seed)
    levels, _ = vd.bfs_sharded(ops, source)
    degrees = torch.from_numpy(np.diff(rowptr).astype(np.int32))
    levels_do, _ = vd.bfs_sharded(ops, source, degrees=degrees, edges=len(adj))
    assert (levels_do.numpy() == levels.numpy()).all(), "direction-optimising sharded BFS != top-down sharded BFS"
    # equal row ranges: bottom-up levels exchange owned slices only
    eq = NumpyShardOps(V, rowptr, adj, w, rank * (V // world), (rank + 1) * (V // world))
    levels_eq, _ = vd.bfs_sharded(eq, source, degrees=degrees, edges=len(adj), equal_ranges=True)
    assert (levels_eq.numpy() == levels.numpy()).all(), "sliced exchange (equal ranges) != full-bitmap exchange"
    levels_2p, _ = vd.bfs_sharded(eq, source, degrees=degrees, edges=len(adj), equal_ranges=True, two_phase=True)
    assert (levels_2p.numpy() == levels.numpy()).all(), "two-phase top-down exchange != full-bitmap exchange"
    levels_2p_td, _ = vd.bfs_sharded(eq, source, equal_ranges=True, two_phase=True)          # top-down only: every level two-phase
    assert (levels_2p_td.numpy() == levels.numpy()).all()
    # levels kept per owner: complete on the owned range, frontier size / degree sum all-reduced
    for kw in (dict(), dict(equal_ranges=True), dict(equal_ranges=True, two_phase=True)):
        o = eq if kw else ops
        for cap in (0, 4096):
            lv, _ = vd.bfs_sharded(o, source, degrees=degrees, edges=len(adj), owned_levels=True, sparse_cap=cap, **kw)
            lo_, hi_ = o.row_range()
            assert (lv.numpy()[lo_:hi_] == levels.numpy()[lo_:hi_]).all(), f"owned levels {kw} sparse {cap}"
    # sparse exchange of small levels: off, with a bound every level fits (V <= 4096 here), and with a bound only the tiny ones fit
    for cap in (0, 4096, 8):
        st = {}
        lv, _ = vd.bfs_sharded(ops, source, degrees=degrees, edges=len(adj), sparse_cap=cap, stats=st)
        assert (lv.numpy() == levels.numpy()).all(), f"sparse exchange (cap {cap}) != bitmap exchange"
        assert (st.get("sparse_levels", 0) > 0) == (cap > 0), (cap, st)
        lv, _ = vd.bfs_sharded(eq, source, equal_ranges=True, two_phase=True, sparse_cap=cap)
        assert (lv.numpy() == levels.numpy()).all()
    st_s, st_c = {}, {}
    d, _ = vd.sssp_sharded(ops, source, stats=st_s)
    assert st_s.get("list_steps", 0) > 0 and st_s.get("dense_steps", 0) > 0, st_s           # both forms of the exchange ran (V/(2P) pairs is the switch)
    d_dense, _ = vd.sssp_sharded(ops, source, dense_only=True)
    assert (d_dense.numpy().view(np.int32) == d.numpy().view(np.int32)).all(), "changed-only exchange != whole-array exchange"
    wd, _ = vd.sswp_sharded(ops, source)
    assert (wd.numpy().view(np.int32) == O.sswp_bellman_ford(rowptr, adj, w, source)[0].view(np.int32)).all(), "sharded SSWP != oracle"
    assert (vd.sswp_sharded(ops, source, dense_only=True)[0].numpy().view(np.int32) == wd.numpy().view(np.int32)).all()
    comp, _ = vd.cc_sharded(ops, stats=st_c)
    assert st_c.get("list_steps", 0) + st_c.get("dense_steps", 0) > 0
    assert (vd.cc_sharded(ops, dense_only=True)[0].numpy() == comp.numpy()).all()
    st_p = {}
    ranks = vd.page_rank_sharded(ops, 3, bounds[rank], bounds[rank + 1], stats=st_p)
    assert st_p["gathered_bytes"] == 3 * 4 * world * max(bounds[r + 1] - bounds[r] for r in range(world))
    ok = [(levels.numpy() == O.bfs_top_down(rowptr, adj, source)[0]).all(),
          (d.numpy().view(np.int32) == O.sssp_bellman_ford(rowptr, adj, w, source)[0].view(np.int32)).all(),
          (comp.numpy() == O.cc_sv(rowptr, adj)[0]).all(),
          (ranks.numpy().view(np.int32) == O.pagerank(rowptr, adj, 3, 1).view(np.int32)).all()]
    results[rank] = [bool(x) for x in ok]
    dist.destroy_process_group()


def test_sharded_drivers_gloo_world2():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, port, results), nprocs=world, join=True)
    assert len(results) == world
    for r in range(world):
        assert results[r] == [True, True, True, True], f"rank {r}: bfs/sssp/cc/pr = {results[r]}"


def test_single_process_driver_without_process_group(oracle):
    O = oracle
    import protocol_model as vd
    V = 256
    src, dst = O.gen_uniform(8, 4, 3)
    rowptr, adj, perm = O.coo_to_csr(V, src, dst)
    ops = NumpyShardOps(V, rowptr, adj, O.gen_weights(len(src), 3)[perm], 0, V)
    source = O.pick_source(rowptr, 3)
    assert (vd.bfs_sharded(ops, source)[0].numpy() == O.bfs_top_down(rowptr, adj, source)[0]).all()
    assert (vd.cc_sharded(ops)[0].numpy() == O.cc_sv(rowptr, adj)[0]).all()
