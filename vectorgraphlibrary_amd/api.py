"""Host-side harness over the C ABI (include/vgl_hip.h): graph construction, frontier objects and the
fused BFS / SSSP / PageRank / CC entry points, with torch tensors used only as device-memory handles
(pointers and sizes cross the boundary; no torch types do).

Names follow the reference: VGL_Graph -> Graph (outgoing + incoming CSR, vgl_graph.h:7-79),
VGL_Frontier -> Frontier (base_frontier.h:5-62), algorithms/{bfs,sssp,pr,cc} -> bfs(), sssp(), page_rank(),
connected_components().  The C++ drop-in class for arbitrary user lambdas is
vectorgraphlibrary_amd/hip/vgl_hip.hpp.
"""
import ctypes as C

import torch

from . import lib as _l

RMAT_ABCD = (57, 19, 19, 5)            # vgl_runtime.hpp:36
BFS_TOP_DOWN, BFS_DIRECTION_OPT = 0, 1
SSSP_ALL_ACTIVE, SSSP_ACTIVE_TILES, SSSP_DELTA_STEPPING, SSSP_PULL, SSSP_DIRECTION_OPT = 0, 1, 2, 3, 4
DENSE, SPARSE, ALL_ACTIVE = 0, 1, 2    # framework_types.h:156-160
PR_EXACT_ORDER, PR_BLOCKED, PR_AUTO = 0, 1, 2


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class Context:
    """One per GPU / process.  Work is ordered on torch's current stream of `device` so that it composes
    with torch.distributed (RCCL) collectives issued from the same process."""

    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise _l.VglHipError("no HIP device visible: vectorgraphlibrary_amd has no CPU fallback")
        self.L = _l.load()
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        h = C.c_void_p()
        _l.check(self.L.vgl_hip_ctx_create(device, C.c_void_p(stream), C.byref(h)))
        self.h = h

    def sync(self):
        _l.check(self.L.vgl_hip_ctx_sync(self.h))

    def close(self):
        if self.h:
            self.L.vgl_hip_ctx_destroy(self.h)
            self.h = None

    def empty(self, n, dtype):
        return torch.empty(int(n), dtype=dtype, device=self.device)

    # ---- timing hooks (bench.py roofline) ----
    def timing(self, enable=True, only=None, stride=1):
        """bracket kernel launches with HIP events (only: one kernel name; the rest then run without the event records; stride: only every
        stride-th of the bracketed launches)"""
        _l.check(self.L.vgl_hip_timing_enable(self.h, int(enable)))
        _l.check(self.L.vgl_hip_timing_only(self.h, only.encode() if only else None))
        _l.check(self.L.vgl_hip_timing_stride(self.h, int(stride)))
        _l.check(self.L.vgl_hip_timing_reset(self.h))

    def timing_get(self, name):
        n, ms = C.c_int64(), C.c_double()
        _l.check(self.L.vgl_hip_timing_get(self.h, name.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    # ---- synthetic inputs ----
    def gen_rmat(self, scale, edge_factor, seed, relabel=True, first_edge=0, count=None):
        E = (1 << scale) * edge_factor if count is None else count
        src, dst = self.empty(E, torch.int32), self.empty(E, torch.int32)
        a, b, c, d = RMAT_ABCD
        _l.check(self.L.vgl_hip_gen_rmat(self.h, scale, first_edge, E, seed, a, b, c, d, int(relabel), _ptr(src), _ptr(dst)))
        return src, dst

    def gen_uniform(self, scale, edge_factor, seed, first_edge=0, count=None):
        E = (1 << scale) * edge_factor if count is None else count
        src, dst = self.empty(E, torch.int32), self.empty(E, torch.int32)
        _l.check(self.L.vgl_hip_gen_uniform(self.h, scale, first_edge, E, seed, _ptr(src), _ptr(dst)))
        return src, dst

    def gen_weights(self, E, seed, first_edge=0):
        w = self.empty(E, torch.float32)
        _l.check(self.L.vgl_hip_gen_weights(self.h, first_edge, E, seed, _ptr(w)))
        return w

    def coo_to_csr(self, V, src, dst, row_begin=0, row_end=None, want_perm=False):
        """stable COO -> CSR of the edges whose source lies in [row_begin,row_end)."""
        row_end = V if row_end is None else row_end
        E = src.numel()
        rowptr = self.empty(row_end - row_begin + 1, torch.int64)
        adj = self.empty(max(E, 1), torch.int32)
        perm = self.empty(max(E, 1), torch.int64) if want_perm else None
        kept = C.c_int64()
        _l.check(self.L.vgl_hip_coo_to_csr(self.h, V, E, _ptr(src), _ptr(dst), row_begin, row_end, _ptr(rowptr), _ptr(adj),
                                           _ptr(perm), C.byref(kept)))
        k = kept.value
        return rowptr, adj[:k], (perm[:k] if want_perm else None)

    def gather_u32(self, perm, values):
        out = torch.empty(perm.numel(), dtype=values.dtype, device=self.device)
        _l.check(self.L.vgl_hip_gather_u32(self.h, perm.numel(), _ptr(perm), _ptr(values), _ptr(out)))
        return out

    def degree_order(self, V, src, dst, kind="total"):
        """VectCSR-style renumbering: fwd[orig] = sorted id, bwd[sorted] = orig (degree descending, id ascending)."""
        fwd, bwd = self.empty(V, torch.int32), self.empty(V, torch.int32)
        k = {"out": 0, "in": 1, "total": 2}[kind]
        _l.check(self.L.vgl_hip_degree_order(self.h, V, src.numel(), _ptr(src), _ptr(dst), k, _ptr(fwd), _ptr(bwd)))
        return fwd, bwd

    def degree_hist_add(self, src, dst, kind, degree):
        """degree[v] += degree of v in the chunk (src,dst); degree is a zero-initialised 4-byte vertex array"""
        k = {"out": 0, "in": 1, "total": 2}[kind]
        _l.check(self.L.vgl_hip_degree_hist_add(self.h, src.numel(), _ptr(src), _ptr(dst), k, _ptr(degree)))

    def degree_order_from_degrees(self, degree):
        V = degree.numel()
        fwd, bwd = self.empty(V, torch.int32), self.empty(V, torch.int32)
        _l.check(self.L.vgl_hip_degree_order_from_degrees(self.h, V, _ptr(degree), _ptr(fwd), _ptr(bwd)))
        return fwd, bwd

    def relabel(self, mapping, ids):
        out = torch.empty_like(ids)
        _l.check(self.L.vgl_hip_relabel_i32(self.h, ids.numel(), _ptr(mapping), _ptr(ids), _ptr(out)))
        return out

    def permute(self, idx, values):
        """out[i] = values[idx[i]] for a 4-byte vertex array"""
        out = torch.empty_like(values)
        _l.check(self.L.vgl_hip_permute_u32(self.h, values.numel(), _ptr(idx), _ptr(values), _ptr(out)))
        return out

    def partition_rows(self, rowptr, parts):
        V = rowptr.numel() - 1
        bounds = (C.c_int32 * (parts + 1))()
        _l.check(self.L.vgl_hip_partition_rows(self.h, V, _ptr(rowptr), parts, bounds))
        return list(bounds)


class Graph:
    """Device CSR in both directions (or outgoing only), optionally restricted to the owned rows [row_begin,row_end)."""

    def __init__(self, ctx, V, out_rowptr, out_adj, in_rowptr=None, in_adj=None, row_begin=0, row_end=None):
        self.ctx, self.V = ctx, int(V)
        self.row_begin, self.row_end = int(row_begin), int(V if row_end is None else row_end)
        self.out_rowptr, self.out_adj, self.in_rowptr, self.in_adj = out_rowptr, out_adj, in_rowptr, in_adj
        self.E = int(out_adj.numel())
        self.fwd = self.bwd = None            # set by from_coo(renumber=...): original <-> sorted vertex ids
        self.perm = None
        h = C.c_void_p()
        _l.check(ctx.L.vgl_hip_graph_create(ctx.h, self.V, self.row_begin, self.row_end, _ptr(out_rowptr), _ptr(out_adj), self.E,
                                            _ptr(in_rowptr), _ptr(in_adj), int(in_adj.numel()) if in_adj is not None else 0,
                                            C.byref(h)))
        self.h = h

    @classmethod
    def from_coo(cls, ctx, V, src, dst, with_incoming=True, want_perm=False, renumber=None):
        """VGL_Graph::import (vgl_graph.hpp:57-68): outgoing CSR from (src,dst), incoming CSR from the OUT-CSR-ordered
        transposed list (the container is sorted in place by the outgoing import before it is transposed).
        renumber in {None, "out", "in", "total"}: VectCSR-style degree renumbering of the vertices before the build
        (vect_csr/import.hpp:61-99); vertex arrays of such a graph live in the sorted numbering (see to_original)."""
        fwd = bwd = None
        if renumber:
            fwd, bwd = ctx.degree_order(V, src, dst, renumber)
            src, dst = ctx.relabel(fwd, src), ctx.relabel(fwd, dst)
        rowptr, adj, perm = ctx.coo_to_csr(V, src, dst, want_perm=want_perm)
        in_rowptr = in_adj = None
        if with_incoming:
            deg = rowptr[1:] - rowptr[:-1]
            csr_src = torch.repeat_interleave(torch.arange(V, device=ctx.device, dtype=torch.int32), deg)
            in_rowptr, in_adj, _ = ctx.coo_to_csr(V, adj, csr_src)
            del csr_src
        g = cls(ctx, V, rowptr, adj, in_rowptr, in_adj)
        g.perm, g.fwd, g.bwd = perm, fwd, bwd
        return g

    def vertex_id(self, original_id):
        """original vertex id -> id in this graph's numbering (VGL_Graph::reorder(v, ORIGINAL, SCATTER))"""
        return int(self.fwd[original_id]) if self.fwd is not None else int(original_id)

    def to_original(self, values):
        """vertex array in this graph's numbering -> ORIGINAL numbering (VerticesArray::reorder(ORIGINAL))"""
        return self.ctx.permute(self.fwd, values) if self.fwd is not None else values

    def shard(self, row_begin, row_end):
        """edge-cut shard owning rows [row_begin,row_end) of both directions (own, aligned copies of the slices)."""
        def cut(rowptr, adj):
            if rowptr is None:
                return None, None
            lo, hi = int(rowptr[row_begin]), int(rowptr[row_end])
            return (rowptr[row_begin:row_end + 1] - lo).contiguous(), adj[lo:hi].clone()
        orp, oadj = cut(self.out_rowptr, self.out_adj)
        irp, iadj = cut(self.in_rowptr, self.in_adj)
        s = Graph(self.ctx, self.V, orp, oadj, irp, iadj, row_begin, row_end)
        s.fwd, s.bwd = self.fwd, self.bwd
        return s

    def out_edge_range(self, row_begin, row_end):
        return int(self.out_rowptr[row_begin]), int(self.out_rowptr[row_end])

    def prepare_page_rank(self, mode=PR_AUTO):
        """build now what the first page_rank() call would build (blocked layout or hub schedule); returns the resolved mode"""
        m = C.c_int()
        _l.check(self.ctx.L.vgl_hip_pr_prepare(self.ctx.h, self.h, int(mode), C.byref(m)))
        return m.value

    def prepare_sssp(self):
        """build the blocked STRUCTURE of the path algorithms now (once per graph; vgl_hip_sssp_prepare): afterwards a pull plan for any weights
        array is one gather pass and SSSP_ALL_ACTIVE runs as blocked passes"""
        _l.check(self.ctx.L.vgl_hip_sssp_prepare(self.ctx.h, self.h))

    def prepare_cc(self):
        _l.check(self.ctx.L.vgl_hip_cc_prepare(self.ctx.h, self.h))

    def prepare_blocked_bfs(self):
        """one-time layout for the blocked top-down BFS levels (vgl_hip_bfs_prepare_blocked); bfs() results do not change"""
        _l.check(self.ctx.L.vgl_hip_bfs_prepare_blocked(self.ctx.h, self.h))

    def close(self):
        if self.h:
            self.ctx.L.vgl_hip_graph_destroy(self.ctx.h, self.h)
            self.h = None


class Frontier:
    def __init__(self, graph):
        self.g, self.ctx = graph, graph.ctx
        h = C.c_void_p()
        _l.check(self.ctx.L.vgl_hip_frontier_create(self.ctx.h, graph.h, C.byref(h)))
        self.h = h

    def set_all_active(self):
        _l.check(self.ctx.L.vgl_hip_frontier_set_all_active(self.ctx.h, self.h))

    def clear(self):
        _l.check(self.ctx.L.vgl_hip_frontier_clear(self.ctx.h, self.h))

    def add_vertex(self, v):
        _l.check(self.ctx.L.vgl_hip_frontier_add_vertex(self.ctx.h, self.h, int(v)))

    def info(self):
        s, n, t = C.c_int32(), C.c_int64(), C.c_int()
        _l.check(self.ctx.L.vgl_hip_frontier_info(self.ctx.h, self.h, C.byref(s), C.byref(n), C.byref(t)))
        return s.value, n.value, t.value

    def size(self):
        return self.info()[0]

    def _view(self, ptr, n):
        if n == 0:
            return torch.empty(0, dtype=torch.int32)
        buf = torch.empty(n, dtype=torch.int32)
        _l.check(self.ctx.L.vgl_hip_memcpy_d2h(self.ctx.h, C.c_void_p(buf.data_ptr()), C.c_void_p(ptr), n * 4))
        return buf

    def ids(self):
        return self._view(self.ctx.L.vgl_hip_frontier_ids(self.h), self.size())

    def flags(self):
        return self._view(self.ctx.L.vgl_hip_frontier_flags(self.h), self.g.V)

    def generate_from_flags(self, flags, dense_threshold=0.0):
        _l.check(self.ctx.L.vgl_hip_gnf_from_flags(self.ctx.h, self.g.h, _ptr(flags), float(dense_threshold), self.h))

    def generate_equal(self, values, value, dense_threshold=0.0):
        _l.check(self.ctx.L.vgl_hip_gnf_equal_i32(self.ctx.h, self.g.h, _ptr(values), int(value), float(dense_threshold), self.h))

    def reduce_sum(self, values):
        if values.dtype == torch.int32:
            r = C.c_int64()
            _l.check(self.ctx.L.vgl_hip_reduce_sum_i32(self.ctx.h, self.h, _ptr(values), C.byref(r)))
        else:
            r = C.c_double()
            _l.check(self.ctx.L.vgl_hip_reduce_sum_f32(self.ctx.h, self.h, _ptr(values), C.byref(r)))
        return r.value

    def close(self):
        if self.h:
            self.ctx.L.vgl_hip_frontier_destroy(self.ctx.h, self.h)
            self.h = None


def _stats(s):
    return {k: getattr(s, k) for k, _ in s._fields_}


# The entry points below take the source in ORIGINAL vertex ids and, by default, return the result in ORIGINAL
# numbering.  raw=True skips both conversions (source and result in the graph's own numbering): that is the timed region
# of the reference, which reorders before tm.start() and after tm.end() (bfs.hpp:62-85, verify_results.h:33-92).

def bfs(graph, source, mode=BFS_DIRECTION_OPT, levels=None, raw=False):
    ctx = graph.ctx
    levels = ctx.empty(graph.V, torch.int32) if levels is None else levels
    st = _l.BfsStats()
    s = int(source) if raw else graph.vertex_id(source)
    _l.check(ctx.L.vgl_hip_bfs_run(ctx.h, graph.h, s, mode, _ptr(levels), C.byref(st)))
    return (levels if raw else graph.to_original(levels)), _stats(st)


def bfs_batch(graph, sources, mode=BFS_DIRECTION_OPT, levels=None):
    """vgl_hip_bfs_run_batch: the traversals from `sources` (the graph's own numbering) one after the other behind ONE call of the C ABI -- the rounds
    loop of the reference's bfs app; returns (levels of the last source, list of per-traversal stats)."""
    ctx = graph.ctx
    levels = ctx.empty(graph.V, torch.int32) if levels is None else levels
    n = len(sources)
    src = (C.c_int32 * n)(*[int(s) for s in sources])
    st = (_l.BfsStats * n)()
    _l.check(ctx.L.vgl_hip_bfs_run_batch(ctx.h, graph.h, src, n, mode, _ptr(levels), st))
    return levels, [_stats(st[i]) for i in range(n)]


class SsspPlan:
    """light/heavy partitioned copy of (adjacency, weights) for the bucketed SSSP schedule; reusable across sources."""

    def __init__(self, graph, weights, delta):
        self.g, self.ctx, self.delta = graph, graph.ctx, float(delta)
        h = C.c_void_p()
        _l.check(self.ctx.L.vgl_hip_sssp_plan_create(self.ctx.h, graph.h, _ptr(weights), self.delta, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            self.ctx.L.vgl_hip_sssp_plan_destroy(self.ctx.h, self.h)
            self.h = None


class SsspPullPlan:
    """blocked copy of (outgoing adjacency, edge values) for the pull steps of SSSP / SSWP (vgl_blocked.h); reusable across sources."""

    def __init__(self, graph, weights):
        self.g, self.ctx, self.weights = graph, graph.ctx, weights
        h = C.c_void_p()
        _l.check(self.ctx.L.vgl_hip_sssp_pull_plan_create(self.ctx.h, graph.h, _ptr(weights), C.byref(h)))
        self.h = h

    def info(self):
        e, f, b, m = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        _l.check(self.ctx.L.vgl_hip_sssp_pull_plan_info(self.h, C.byref(e), C.byref(f), C.byref(b), C.byref(m)))
        return {"edges": e.value, "fused_edges": f.value, "streamed_bytes_per_pass": b.value, "plan_bytes": m.value}

    def close(self):
        if self.h:
            self.ctx.L.vgl_hip_sssp_pull_plan_destroy(self.ctx.h, self.h)
            self.h = None


def sssp(graph, weights, source, mode=SSSP_ACTIVE_TILES, dist=None, raw=False, delta=16.0, plan=None):
    """mode: SSSP_ALL_ACTIVE / SSSP_ACTIVE_TILES (push), SSSP_DELTA_STEPPING, SSSP_PULL, SSSP_DIRECTION_OPT (push <-> pull).
    plan: an SsspPlan (bucketed schedule) or an SsspPullPlan (with mode SSSP_PULL or SSSP_DIRECTION_OPT)."""
    ctx = graph.ctx
    dist = ctx.empty(graph.V, torch.float32) if dist is None else dist
    st = _l.SsspStats()
    s = int(source) if raw else graph.vertex_id(source)
    if isinstance(plan, SsspPullPlan):
        m = mode if mode in (SSSP_PULL, SSSP_DIRECTION_OPT) else SSSP_DIRECTION_OPT
        _l.check(ctx.L.vgl_hip_sssp_run_pull(ctx.h, graph.h, _ptr(weights), plan.h, s, int(m), _ptr(dist), C.byref(st)))
    elif plan is not None:
        _l.check(ctx.L.vgl_hip_sssp_run_plan(ctx.h, graph.h, plan.h, s, _ptr(dist), C.byref(st)))
    elif mode == SSSP_DELTA_STEPPING:
        _l.check(ctx.L.vgl_hip_sssp_run_delta(ctx.h, graph.h, _ptr(weights), s, float(delta), _ptr(dist), C.byref(st)))
    else:
        _l.check(ctx.L.vgl_hip_sssp_run(ctx.h, graph.h, _ptr(weights), s, mode, _ptr(dist), C.byref(st)))
    return (dist if raw else graph.to_original(dist)), _stats(st)


def sswp(graph, capacities, source, mode=SSSP_ACTIVE_TILES, widths=None, raw=False, plan=None):
    """single-source widest paths (SSWP::vgl_dijkstra): widths[source] = FLT_MAX, unreachable vertices 0; capacities in the
    order of the outgoing CSR.  source / result in ORIGINAL numbering unless raw=True."""
    ctx = graph.ctx
    widths = ctx.empty(graph.V, torch.float32) if widths is None else widths
    st = _l.SsspStats()
    src = int(source) if raw else graph.vertex_id(source)
    if isinstance(plan, SsspPullPlan):
        m = mode if mode in (SSSP_PULL, SSSP_DIRECTION_OPT) else SSSP_DIRECTION_OPT
        _l.check(ctx.L.vgl_hip_sswp_run_pull(ctx.h, graph.h, _ptr(capacities), plan.h, src, int(m), _ptr(widths), C.byref(st)))
    else:
        _l.check(ctx.L.vgl_hip_sswp_run(ctx.h, graph.h, _ptr(capacities), src, int(mode), _ptr(widths), C.byref(st)))
    return (widths if raw else graph.to_original(widths)), _stats(st)


def page_rank(graph, iterations, indeg_noloops=None, ranks=None, raw=False, mode=PR_AUTO):
    """indeg_noloops (optional) is indexed in the graph's own numbering.  mode: PR_EXACT_ORDER (adjacency-order f32 sums, bit-identical
    to seq_page_rank), PR_BLOCKED (LDS-window gather / sum, within a few ulp) or PR_AUTO (blocked from 2^25 edges)."""
    ctx = graph.ctx
    ranks = ctx.empty(graph.V, torch.float32) if ranks is None else ranks
    st = _l.PrStats()
    _l.check(ctx.L.vgl_hip_pr_run_mode(ctx.h, graph.h, _ptr(indeg_noloops), int(iterations), int(mode), _ptr(ranks), C.byref(st)))
    return (ranks if raw else graph.to_original(ranks)), _stats(st)


def sum_over_edges(graph, values, bound=1.0):
    """sums[src] = sum of values[dst] over the edges src -> dst with dst != src (graph's own numbering; the declared operator VGL_SUM_OVER_EDGES):
    exact fixed-point sums rounded to f32 once.  values: non-negative f32, every per-vertex sum at most `bound`."""
    ctx = graph.ctx
    sums = ctx.empty(graph.V, torch.float32)
    _l.check(ctx.L.vgl_hip_sum_over_edges_f32(ctx.h, graph.h, _ptr(values), C.c_float(float(bound)), _ptr(sums)))
    ctx.sync()
    return sums


def strongly_connected_components(graph, raw=False):
    """labels = smallest ORIGINAL vertex id of each strongly connected component (raw=True: smallest id in the graph's numbering)"""
    ctx = graph.ctx
    comp = ctx.empty(graph.V, torch.int32)
    st = _l.SccStats()
    _l.check(ctx.L.vgl_hip_scc_run(ctx.h, graph.h, _ptr(comp), C.byref(st)))
    if raw or graph.fwd is None:
        return comp, _stats(st)
    out, scratch = ctx.empty(graph.V, torch.int32), ctx.empty(graph.V, torch.int32)
    _l.check(ctx.L.vgl_hip_cc_labels_to_original(ctx.h, graph.V, _ptr(comp), _ptr(graph.fwd), _ptr(graph.bwd), _ptr(scratch), _ptr(out)))
    return out, _stats(st)


def hits(graph, steps, raw=False):
    """HITS authorities and hubs (f64) after `steps` steps; needs the incoming CSR.  ORIGINAL numbering unless raw=True."""
    ctx = graph.ctx
    auth, hub = ctx.empty(graph.V, torch.float64), ctx.empty(graph.V, torch.float64)
    _l.check(ctx.L.vgl_hip_hits_run(ctx.h, graph.h, int(steps), _ptr(auth), _ptr(hub)))
    if raw or graph.fwd is None:
        return auth, hub
    idx = graph.fwd.long()
    return auth[idx], hub[idx]


def connected_components(graph, comp=None, raw=False, symmetric=False):
    """labels = smallest ORIGINAL vertex id that reaches each vertex (raw=True: smallest id in the graph's numbering).
    On a DIRECTED graph the hook / jump fixed point ("smallest id that reaches v") depends on the vertex numbering -- as in the
    reference (exact under identical numbering) -- so with a renumbered graph the conversion back to original ids is
    only meaningful for symmetric inputs, where the labels describe the components.
    symmetric=True: the caller vouches that every edge is stored in both directions; the same labels then come from a min-id
    union-find (vgl_hip_cc_run_symmetric) instead of repeated sweeps over all edges."""
    ctx = graph.ctx
    comp = ctx.empty(graph.V, torch.int32) if comp is None else comp
    st = _l.CcStats()
    run = ctx.L.vgl_hip_cc_run_symmetric if symmetric else ctx.L.vgl_hip_cc_run
    _l.check(run(ctx.h, graph.h, _ptr(comp), C.byref(st)))
    if raw or graph.fwd is None:
        return comp, _stats(st)
    out, scratch = ctx.empty(graph.V, torch.int32), ctx.empty(graph.V, torch.int32)
    _l.check(ctx.L.vgl_hip_cc_labels_to_original(ctx.h, graph.V, _ptr(comp), _ptr(graph.fwd), _ptr(graph.bwd), _ptr(scratch), _ptr(out)))
    return out, _stats(st)


def count_not_equal(ctx, a, b):
    r = C.c_int64()
    _l.check(ctx.L.vgl_hip_count_not_equal_u32(ctx.h, a.numel(), _ptr(a), _ptr(b), C.byref(r)))
    return r.value
