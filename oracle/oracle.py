"""ctypes/numpy binding of the CPU oracle (oracle/vgl_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this module.  The product package
(vectorgraphlibrary_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libvgl_oracle.so")


def build(force=False):
    """Compile the C restatement (gcc) if the shared object is missing or stale."""
    src = os.path.join(_HERE, "vgl_oracle.c")
    hdr = os.path.join(_HERE, "vgl_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "_build/libvgl_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


class BfsStats(C.Structure):
    _fields_ = [("levels", C.c_int32), ("edges_examined", C.c_int64),
                ("frontier_total", C.c_int64), ("discovered", C.c_int64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        p = C.c_void_p
        i32, i64, u64 = C.c_int32, C.c_int64, C.c_uint64
        L.vgo_splitmix64.restype = u64
        L.vgo_splitmix64.argtypes = [u64]
        L.vgo_relabel.restype = C.c_uint32
        L.vgo_relabel.argtypes = [C.c_uint32, C.c_int, u64]
        L.vgo_gen_rmat.argtypes = [C.c_int, i64, i64, u64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, p, p]
        L.vgo_gen_uniform.argtypes = [C.c_int, i64, i64, u64, p, p]
        L.vgo_gen_weights.argtypes = [i64, i64, u64, p]
        L.vgo_coo_to_csr.argtypes = [i32, i64, p, p, p, p, p]
        L.vgo_degree_renumber.argtypes = [i32, p, p, p]
        L.vgo_bfs_top_down.argtypes = [i32, p, p, i32, p, C.POINTER(BfsStats), C.c_int]
        L.vgo_bfs_seq.argtypes = [i32, p, p, i32, p]
        L.vgo_sssp_bellman_ford.restype = i32
        L.vgo_sssp_bellman_ford.argtypes = [i32, p, p, p, i32, p, C.c_int]
        L.vgo_sssp_dijkstra.argtypes = [i32, p, p, p, i32, p]
        L.vgo_sswp_bellman_ford.restype = i32
        L.vgo_sswp_bellman_ford.argtypes = [i32, p, p, p, i32, p, C.c_int]
        L.vgo_sswp_seq.argtypes = [i32, p, p, p, i32, p]
        L.vgo_hits.argtypes = [i32, p, p, p, p, i32, p, p]
        L.vgo_scc_tarjan.argtypes = [i32, p, p, p]
        L.vgo_indegree_noloops.argtypes = [i32, i64, p, p, p]
        L.vgo_pagerank.argtypes = [i32, p, p, p, C.c_int, C.c_int, p, C.c_int]
        L.vgo_cc_sv.restype = i32
        L.vgo_cc_sv.argtypes = [i32, p, p, p, C.c_int]
        L.vgo_cc_seq_bfs.argtypes = [i32, p, p, p]
        L.vgo_fnv1a64.restype = u64
        L.vgo_fnv1a64.argtypes = [p, i64]
        L.vgo_max_threads.restype = C.c_int
        L.vgo_set_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


RMAT_ABCD = (57, 19, 19, 5)   # vgl_runtime.hpp:36 / graph_generation.hpp:94


def gen_rmat(scale, edge_factor, seed, relabel=True, first_edge=0, count=None):
    V = 1 << scale
    E = V * edge_factor if count is None else count
    src = np.empty(E, np.int32)
    dst = np.empty(E, np.int32)
    a, b, c, d = RMAT_ABCD
    lib().vgo_gen_rmat(scale, first_edge, E, seed, a, b, c, d, int(relabel), _p(src), _p(dst))
    return src, dst


def gen_uniform(scale, edge_factor, seed, first_edge=0, count=None):
    V = 1 << scale
    E = V * edge_factor if count is None else count
    src = np.empty(E, np.int32)
    dst = np.empty(E, np.int32)
    lib().vgo_gen_uniform(scale, first_edge, E, seed, _p(src), _p(dst))
    return src, dst


def gen_weights(E, seed, first_edge=0):
    w = np.empty(E, np.float32)
    lib().vgo_gen_weights(first_edge, E, seed, _p(w))
    return w


def symmetrize(src, dst):
    """undirected input for CC: every edge followed (as a block) by its reverse (graph_generation.hpp:41-49)"""
    return np.concatenate([src, dst]), np.concatenate([dst, src])


def coo_to_csr(V, src, dst, want_perm=True):
    E = len(src)
    src = np.ascontiguousarray(src, np.int32)
    dst = np.ascontiguousarray(dst, np.int32)
    rowptr = np.empty(V + 1, np.int64)
    adj = np.empty(E, np.int32)
    perm = np.empty(E, np.int64) if want_perm else None
    lib().vgo_coo_to_csr(V, E, _p(src), _p(dst), _p(rowptr), _p(adj), _p(perm) if want_perm else None)
    return rowptr, adj, perm


def degree_renumber(rowptr):
    V = len(rowptr) - 1
    fwd = np.empty(V, np.int32)
    bwd = np.empty(V, np.int32)
    lib().vgo_degree_renumber(V, _p(rowptr), _p(fwd), _p(bwd))
    return fwd, bwd


def bfs_top_down(rowptr, adj, source, parallel=False):
    V = len(rowptr) - 1
    levels = np.empty(V, np.int32)
    st = BfsStats()
    lib().vgo_bfs_top_down(V, _p(rowptr), _p(adj), source, _p(levels), C.byref(st), int(parallel))
    return levels, dict(levels=st.levels, edges_examined=st.edges_examined,
                        frontier_total=st.frontier_total, discovered=st.discovered)


def bfs_seq(rowptr, adj, source):
    V = len(rowptr) - 1
    levels = np.empty(V, np.int32)
    lib().vgo_bfs_seq(V, _p(rowptr), _p(adj), source, _p(levels))
    return levels


def sssp_bellman_ford(rowptr, adj, w, source, parallel=False):
    V = len(rowptr) - 1
    dist = np.empty(V, np.float32)
    iters = lib().vgo_sssp_bellman_ford(V, _p(rowptr), _p(adj), _p(w), source, _p(dist), int(parallel))
    return dist, iters


def transpose_csr(rowptr, adj):
    """incoming CSR the way VGL_Graph::import builds it (vgl_graph.hpp:57-68): the OUT-CSR-ordered edge list, transposed, stable"""
    V = len(rowptr) - 1
    csr_src = np.repeat(np.arange(V, dtype=np.int32), np.diff(rowptr))
    in_rowptr, in_adj, _ = coo_to_csr(V, adj, csr_src, want_perm=False)
    return in_rowptr, in_adj


def scc_tarjan(rowptr, adj):
    """canonical SCC labels: comp[v] = smallest vertex id of v's strongly connected component"""
    V = len(rowptr) - 1
    comp = np.empty(V, np.int32)
    lib().vgo_scc_tarjan(V, _p(rowptr), _p(adj), _p(comp))
    return comp


def canonical_labels(labels):
    """any labelling of a partition -> the smallest member id of each part (what scc_tarjan / cc_sv return)"""
    labels = np.asarray(labels)
    _, inv = np.unique(labels, return_inverse=True)
    mins = np.full(inv.max() + 1, len(labels), np.int64)
    np.minimum.at(mins, inv, np.arange(len(labels)))
    return mins[inv].astype(np.int32)


def hits(rowptr, adj, steps):
    V = len(rowptr) - 1
    in_rowptr, in_adj = transpose_csr(rowptr, adj)
    auth, hub = np.empty(V, np.float64), np.empty(V, np.float64)
    lib().vgo_hits(V, _p(rowptr), _p(adj), _p(in_rowptr), _p(in_adj), int(steps), _p(auth), _p(hub))
    return auth, hub


def sswp_bellman_ford(rowptr, adj, cap, source, parallel=False):
    V = len(rowptr) - 1
    width = np.empty(V, np.float32)
    iters = lib().vgo_sswp_bellman_ford(V, _p(rowptr), _p(adj), _p(cap), source, _p(width), int(parallel))
    return width, iters


def sswp_seq(rowptr, adj, cap, source):
    V = len(rowptr) - 1
    width = np.empty(V, np.float32)
    lib().vgo_sswp_seq(V, _p(rowptr), _p(adj), _p(cap), source, _p(width))
    return width


def sssp_dijkstra(rowptr, adj, w, source):
    V = len(rowptr) - 1
    dist = np.empty(V, np.float32)
    lib().vgo_sssp_dijkstra(V, _p(rowptr), _p(adj), _p(w), source, _p(dist))
    return dist


def indegree_noloops(rowptr, adj):
    V = len(rowptr) - 1
    indeg = np.empty(V, np.int32)
    lib().vgo_indegree_noloops(V, len(adj), _p(rowptr), _p(adj), _p(indeg))
    return indeg


def pagerank(rowptr, adj, iterations, dangling_mode=1, parallel=False, indeg=None):
    V = len(rowptr) - 1
    if indeg is None:
        indeg = indegree_noloops(rowptr, adj)
    ranks = np.empty(V, np.float32)
    lib().vgo_pagerank(V, _p(rowptr), _p(adj), _p(indeg), iterations, dangling_mode, _p(ranks), int(parallel))
    return ranks


def cc_sv(rowptr, adj, parallel=False):
    V = len(rowptr) - 1
    comp = np.empty(V, np.int32)
    passes = lib().vgo_cc_sv(V, _p(rowptr), _p(adj), _p(comp), int(parallel))
    return comp, passes


def cc_seq_bfs(rowptr, adj):
    V = len(rowptr) - 1
    comp = np.empty(V, np.int32)
    lib().vgo_cc_seq_bfs(V, _p(rowptr), _p(adj), _p(comp))
    return comp


def fnv1a64(a):
    a = np.ascontiguousarray(a)
    return int(lib().vgo_fnv1a64(_p(a), a.nbytes))


def max_threads():
    return int(lib().vgo_max_threads())


def host_cpus():
    """CPUs this process may really use: the affinity mask, cut down by a cgroup CPU quota where one is set (a GPU box shows every
    core of the host but grants a share of them; an OpenMP team of all visible cores is then heavily over-subscribed)."""
    if os.environ.get("VGL_HOST_CPUS"):             # counted before an OpenMP runtime with OMP_PROC_BIND narrowed this thread's own mask
        return max(1, int(os.environ["VGL_HOST_CPUS"]))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
        except (OSError, ValueError, IndexError):
            pass
    return n


def host_description():
    """model name / sockets / cores of the host the CPU baseline ran on (for the bench line)"""
    model, sockets, cores = "unknown", set(), set()
    try:
        phys = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
                sockets.add(phys)
            elif line.startswith("core id"):
                cores.add((phys, line.split(":", 1)[1].strip()))
    except OSError:
        pass
    return {"model": model, "sockets": len(sockets) or None, "physical_cores": len(cores) or None,
            "logical_cpus_visible": os.cpu_count(), "cpus_granted": host_cpus()}


def set_threads(n=None):
    """size of the OpenMP team of the parallel=True runs; default = host_cpus()"""
    n = host_cpus() if n is None else int(n)
    lib().vgo_set_threads(n)
    return n


def pick_source(rowptr, seed, k=0):
    """deterministic stand-in for VGL_Graph::select_random_nz_vertex: first vertex at or after a
    hashed position that has at least one outgoing edge."""
    V = len(rowptr) - 1
    v = int(lib().vgo_splitmix64((seed + 0x9999 + k) & 0xFFFFFFFFFFFFFFFF) % V)
    for _ in range(V):
        if rowptr[v + 1] > rowptr[v]:
            return v
        v = (v + 1) % V
    raise ValueError("graph has no edges")


def same_partition(a, b):
    """verify_results.h:198-254 equal_components: label bijection between two labelings"""
    fwd, bwd = {}, {}
    for x, y in zip(a.tolist(), b.tolist()):
        if fwd.setdefault(x, y) != y or bwd.setdefault(y, x) != x:
            return False
    return True


def write_el_container(path, V, src, dst):
    """EdgesContainer binary format (edges_container.h:58-77): int V; long long E; int type=4; src[E]; dst[E]"""
    with open(path, "wb") as f:
        f.write(np.int32(V).tobytes())
        f.write(np.int64(len(src)).tobytes())
        f.write(np.int32(4).tobytes())
        f.write(np.ascontiguousarray(src, np.int32).tobytes())
        f.write(np.ascontiguousarray(dst, np.int32).tobytes())


def _rw_draw(seed, step, walk):
    """counter-based draw of the random-walk app (apps/algorithms/rw.hpp RandomWalk::draw): splitmix64 finaliser, uint64 wrap-around"""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (np.asarray(step, np.uint64) + np.uint64(1)) \
            + np.uint64(0xD1B54A32D192ED03) * (np.asarray(walk, np.uint64) + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def random_walk(rowptr, adj, seed, percent, length):
    """algorithms/rw/random_walk.hpp:60-85 (seq_random_walk's loop) with the app's reproducible draws in place of rand():
    walk vertices v (draw(seed, 2^64-1, v) % 100 < percent) start at themselves; per step every live walk moves to out-neighbour
    number draw(seed, step, v) % degree of its current vertex, or to DEAD_END (-1) where the degree is 0.  Returns int32[V]."""
    V = len(rowptr) - 1
    ids = np.arange(V, dtype=np.uint64)
    walkers = np.nonzero((_rw_draw(seed, np.uint64(0xFFFFFFFFFFFFFFFF), ids) % np.uint64(100)).astype(np.int64) < percent)[0]
    cur = walkers.astype(np.int64)
    for step in range(length):
        live = cur >= 0
        c = cur[live]
        deg = rowptr[c + 1] - rowptr[c]
        r = (_rw_draw(seed, np.uint64(step), walkers[live].astype(np.uint64)) % np.maximum(deg, 1).astype(np.uint64)).astype(np.int64)
        nxt = np.where(deg > 0, adj[np.minimum(rowptr[c] + r, len(adj) - 1)], -1)
        cur[live] = nxt
    out = np.full(V, -1, np.int32)
    out[walkers] = cur
    return out


def max_flow_ford_fulkerson(rowptr, adj, source, sink, capacity=100):
    """algorithms/mf/mf.hpp:67-128 + seq_mf.hpp:51-97 with the reproducible parent rule of apps/algorithms/mf.hpp: every stored edge
    holds a residual value (initially `capacity`); level-synchronous search over positive edges, parent = smallest id among the
    previous level's vertices reaching a vertex; bottleneck from the FIRST u->v match, update of ALL parallel u->v (minus) and ALL
    stored v->u (plus).  Returns (flow value, number of augmentations).  On a symmetric graph the value is the maximum flow
    (tests compare it with scipy.sparse.csgraph.maximum_flow)."""
    V = len(rowptr) - 1
    rowptr = np.asarray(rowptr, np.int64)
    adj = np.asarray(adj, np.int64)
    res = np.full(len(adj), capacity, np.int64)
    deg = np.diff(rowptr)
    total, rounds = 0, 0
    while source != sink:
        level = np.full(V, -1, np.int64)
        parent = np.full(V, np.iinfo(np.int64).max, np.int64)
        level[source] = 1
        front = np.array([source], np.int64)
        cur = 1
        while front.size:
            cnt = deg[front]
            srcs = np.repeat(front, cnt)
            pos = np.repeat(rowptr[front] - np.concatenate(([0], np.cumsum(cnt)[:-1])), cnt) + np.arange(cnt.sum())
            dsts = adj[pos]
            ok = (res[pos] > 0) & (level[dsts] == -1)
            np.minimum.at(parent, dsts[ok], srcs[ok])
            front = np.unique(dsts[ok])
            level[front] = cur + 1
            cur += 1
        if level[sink] == -1:
            break
        path = []
        v = sink
        while v != source:
            u = int(parent[v])
            path.append((u, v))
            v = u
        flow = min(int(res[rowptr[u] + np.nonzero(adj[rowptr[u]:rowptr[u + 1]] == v)[0][0]]) for u, v in path)
        if flow <= 0:
            break
        for u, v in path:
            res[rowptr[u] + np.nonzero(adj[rowptr[u]:rowptr[u + 1]] == v)[0]] -= flow
            res[rowptr[v] + np.nonzero(adj[rowptr[v]:rowptr[v + 1]] == u)[0]] += flow
        total += flow
        rounds += 1
    return total, rounds
