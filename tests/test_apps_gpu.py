"""GPU tests of the C++ drop-in layer: the example applications (apps/*.cpp) run the reference-style algorithm code
(apps/algorithms/*.hpp, user device lambdas through GraphAbstractionsHIP: scatter / compute / reduce /
generate_new_frontier) and are checked against the CPU oracle on the same seeded inputs, plus their own -check mode
(`error count: 0`, the line the reference's verification harness greps, scripts/verification_api.py:23-44)."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "apps", "bin")


def run_app(app, args, tmp_path, env=None):
    dump = str(tmp_path / (app + ".bin"))
    cmd = [os.path.join(BIN, app + "_hip")] + [str(a) for a in args] + ["-dump", dump]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout, dump


def graph(O, kind, scale, ef, seed, symmetric=False):
    src, dst = (O.gen_rmat if kind == "rmat" else O.gen_uniform)(scale, ef, seed)
    if symmetric:
        src, dst = O.symmetrize(src, dst)
    rowptr, adj, perm = O.coo_to_csr(1 << scale, src, dst)
    return src, dst, rowptr, adj, perm


CASES = [("rmat", 12, 16, 3), ("ru", 11, 8, 5)]


@pytest.mark.parametrize("kind,scale,ef,seed", CASES)
@pytest.mark.parametrize("mode", [[], ["-fused", "-td"], ["-fused", "-do"], ["-fused", "-td", "-blocked"]])
def test_bfs_app(kind, scale, ef, seed, mode, tmp_path, oracle, ctx):
    O = oracle
    src, dst, rowptr, adj, perm = graph(O, kind, scale, ef, seed)
    source = O.pick_source(rowptr, seed)
    out, dump = run_app("bfs", ["-s", scale, "-e", ef, "-type", kind, "-seed", seed, "-source", source, "-check"] + mode, tmp_path)
    assert "error count: 0" in out and re.search(r"AVG_PERF: [0-9.e+]+ MTEPS", out)
    assert (np.fromfile(dump, np.int32) == O.bfs_top_down(rowptr, adj, source)[0]).all()


@pytest.mark.parametrize("kind,scale,ef,seed", CASES)
@pytest.mark.parametrize("mode", [[], ["-fused"], ["-pull"], ["-push", "-all-active"], ["-fused", "-pull"], ["-fused", "-do"], ["-pull", "-format", "vcsr"],
                                  ["-declared"], ["-declared", "-format", "vcsr"]],
                         ids=["push", "fused", "pull", "push_all_active", "fused_pull", "fused_do", "pull_vcsr", "declared", "declared_vcsr"])
def test_sssp_app(kind, scale, ef, seed, mode, tmp_path, oracle, ctx):
    O = oracle
    src, dst, rowptr, adj, perm = graph(O, kind, scale, ef, seed)
    source = O.pick_source(rowptr, seed)
    out, dump = run_app("sssp", ["-s", scale, "-e", ef, "-type", kind, "-seed", seed, "-source", source, "-check"] + mode, tmp_path)
    assert "error count: 0" in out
    ref, _ = O.sssp_bellman_ford(rowptr, adj, O.gen_weights(len(src), seed)[perm], source)
    assert (np.fromfile(dump, np.float32).view(np.int32) == ref.view(np.int32)).all()


@pytest.mark.parametrize("mode", [[], ["-fused"]], ids=["operator_api", "fused"])
@pytest.mark.parametrize("kind,scale,ef,seed", CASES)
def test_sswp_app(kind, scale, ef, seed, mode, tmp_path, oracle, ctx):
    """widest paths (f1 widening): the lambda version on the generic operator path and the fused kernel, both bit-exact"""
    O = oracle
    src, dst, rowptr, adj, perm = graph(O, kind, scale, ef, seed)
    source = O.pick_source(rowptr, seed)
    out, dump = run_app("sswp", ["-s", scale, "-e", ef, "-type", kind, "-seed", seed, "-source", source, "-check"] + mode, tmp_path)
    assert "error count: 0" in out
    ref, _ = O.sswp_bellman_ford(rowptr, adj, O.gen_weights(len(src), seed)[perm], source)
    assert (np.fromfile(dump, np.float32).view(np.int32) == ref.view(np.int32)).all()


@pytest.mark.parametrize("mode", [[], ["-fused"]], ids=["operator_api", "fused"])
@pytest.mark.parametrize("kind,scale,ef,seed", CASES)
def test_hits_app(kind, scale, ef, seed, mode, tmp_path, oracle, ctx):
    """HITS (f1 widening, f64): gather/scatter with vertex pre-ops + reduce<double> on the operator path, and the fused kernel"""
    O = oracle
    src, dst, rowptr, adj, perm = graph(O, kind, scale, ef, seed)
    out, dump = run_app("hits", ["-s", scale, "-e", ef, "-type", kind, "-seed", seed, "-it", 4, "-check"] + mode, tmp_path)
    assert "error count: 0" in out
    auth, hub = O.hits(rowptr, adj, 4)
    got = np.fromfile(dump, np.float64)
    V = len(rowptr) - 1
    tol = 1e-12 if mode else 1e-9          # the lambda version accumulates with f64 atomics (order-dependent last bits)
    assert np.max(np.abs(got[:V] - auth) / np.maximum(np.abs(auth), 1e-300)) <= tol
    assert np.max(np.abs(got[V:] - hub) / np.maximum(np.abs(hub), 1e-300)) <= tol


@pytest.mark.parametrize("mode", [[], ["-fused"]], ids=["operator_api", "fused"])
@pytest.mark.parametrize("kind,scale,ef,seed", CASES + [("ru", 12, 1, 8)])
def test_scc_app(kind, scale, ef, seed, mode, tmp_path, oracle, ctx):
    """SCC app (f1 widening): canonical labels equal the oracle's Tarjan partition; the app's own -check (partition equality) agrees.
    operator_api = trim + colour forward-backward through scatter / compute / generate_new_frontier (apps/algorithms/scc.hpp), fused = vgl_hip_scc_run"""
    O = oracle
    src, dst, rowptr, adj, perm = graph(O, kind, scale, ef, seed)
    out, dump = run_app("scc", ["-s", scale, "-e", ef, "-type", kind, "-seed", seed, "-check"] + mode, tmp_path)
    assert "error count: 0" in out
    assert (np.fromfile(dump, np.int32) == O.scc_tarjan(rowptr, adj)).all()


@pytest.mark.parametrize("kind,scale,ef,seed", CASES)
def test_pr_app(kind, scale, ef, seed, tmp_path, oracle, ctx):
    O = oracle
    src, dst, rowptr, adj, perm = graph(O, kind, scale, ef, seed)
    ref = O.pagerank(rowptr, adj, 5, 1)
    out, dump = run_app("pr", ["-s", scale, "-e", ef, "-type", kind, "-seed", seed, "-it", 5, "-check", "-fused"], tmp_path)
    assert "error count: 0" in out
    assert (np.fromfile(dump, np.float32).view(np.int32) == ref.view(np.int32)).all()
    # operator-API version accumulates with float atomics (like the reference's GPU variant): order-dependent in the last bits
    out, dump = run_app("pr", ["-s", scale, "-e", ef, "-type", kind, "-seed", seed, "-it", 5, "-check"], tmp_path)
    assert "error count: 0" in out
    got = np.fromfile(dump, np.float32)
    assert np.max(np.abs(got - ref) / ref) < 2e-5
    # ... and with -deterministic (round 4): one lane per vertex adds the products in adjacency order without atomics
    # (GraphAbstractionsHIP::enable_sequential_rows, the execution shape of the reference's multicore kernels): within north_star's 1e-6 of
    # the multicore recipe -- the same f32 chain; only the dangling term is folded in another (fixed) f64 order
    out, dump = run_app("pr", ["-s", scale, "-e", ef, "-type", kind, "-seed", seed, "-it", 5, "-check", "-deterministic"], tmp_path)
    assert "error count: 0" in out
    got = np.fromfile(dump, np.float32)
    assert np.max(np.abs(got - ref) / ref) <= 1e-6
    out2, dump2 = run_app("pr", ["-s", scale, "-e", ef, "-type", kind, "-seed", seed, "-it", 5, "-deterministic"], tmp_path)
    assert (np.fromfile(dump2, np.float32).view(np.int32) == got.view(np.int32)).all()          # run to run: bit-identical
    # -pull: the same chain summed in a register inside compute() (adjacency through a view captured by the lambda): the same bits
    out3, dump3 = run_app("pr", ["-s", scale, "-e", ef, "-type", kind, "-seed", seed, "-it", 5, "-check", "-pull"], tmp_path)
    assert "error count: 0" in out3
    assert (np.fromfile(dump3, np.float32).view(np.int32) == got.view(np.int32)).all()
    # -declared (round 5): the pull as the declared operator VGL_SUM_OVER_EDGES -- the class runs the library's blocked pass (exact sums): within 1e-6
    # of the chain on uniform graphs (short rows), within the chain's own rounding on RMAT hubs; the same bits from run to run
    out4, dump4 = run_app("pr", ["-s", scale, "-e", ef, "-type", kind, "-seed", seed, "-it", 5, "-check", "-declared"], tmp_path)
    assert "error count: 0" in out4
    got4 = np.fromfile(dump4, np.float32)
    assert np.max(np.abs(got4 - ref) / ref) <= (1e-6 if kind == "ru" else 2e-5)
    out5, dump5 = run_app("pr", ["-s", scale, "-e", ef, "-type", kind, "-seed", seed, "-it", 5, "-declared"], tmp_path)
    assert (np.fromfile(dump5, np.float32).view(np.int32) == got4.view(np.int32)).all()


@pytest.mark.parametrize("kind,scale,ef,seed", CASES)
@pytest.mark.parametrize("mode", [[], ["-fused"], ["-declared"], ["-declared", "-blocked-hook"]], ids=["operator_api", "fused", "declared", "declared_blocked"])
def test_cc_app(kind, scale, ef, seed, mode, tmp_path, oracle, ctx):
    """-declared: the hook as a DECLARED operator (VGL_MIN_LABEL_OVER_EDGES, an extension of the operator API): the backend runs it as a library
    pass -- the atomic kernel on a graph this small, the blocked LDS-window pass when forced (VGL_CC_BLOCKED=1) -- same labels"""
    O = oracle
    src, dst, rowptr, adj, perm = graph(O, kind, scale, ef, seed, symmetric=True)
    env = None
    if "-blocked-hook" in mode:
        mode = [m for m in mode if m != "-blocked-hook"]
        env = dict(os.environ, VGL_CC_BLOCKED="1")
    out, dump = run_app("cc", ["-s", scale, "-e", ef, "-type", kind, "-seed", seed, "-check"] + mode, tmp_path, env=env)
    assert "error count: 0" in out
    assert (np.fromfile(dump, np.int32) == O.cc_sv(rowptr, adj)[0]).all()


def test_operator_api_error_convention(tmp_path, ctx):
    """errors surface as thrown C strings caught in main, like the reference (apps/bfs/bfs.cpp:53-61)."""
    out = subprocess.run([os.path.join(BIN, "bfs_hip"), "-bogus"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "unknown command line option" in out.stdout


def test_import_el_container(tmp_path, oracle, ctx):
    """-import reads the reference's EdgesContainer binary format (edges_container.h:58-99), here written by the oracle helper
    with an arbitrary (non power-of-two) vertex count."""
    O = oracle
    rng = np.random.default_rng(7)
    V, E = 3001, 20000
    src, dst = rng.integers(0, V, E).astype(np.int32), rng.integers(0, V, E).astype(np.int32)
    path = str(tmp_path / "g.el_container")
    O.write_el_container(path, V, src, dst)
    rowptr, adj, _ = O.coo_to_csr(V, src, dst)
    source = O.pick_source(rowptr, 7)
    ref = O.bfs_top_down(rowptr, adj, source)[0]
    for mode in ([], ["-fused", "-do"]):
        out, dump = run_app("bfs", ["-import", path, "-source", source, "-check"] + mode, tmp_path)
        assert "error count: 0" in out
        assert (np.fromfile(dump, np.int32) == ref).all()


@pytest.mark.parametrize("mode", [[], ["-fused"]], ids=["operator_api", "fused"])
@pytest.mark.parametrize("app", ["bfs", "sssp", "sswp", "pr", "cc", "scc", "hits"])
def test_apps_vector_csr_format(app, mode, tmp_path, oracle, ctx):
    """-format vcsr (VECTOR_CSR_GRAPH: vertices renumbered by degree, the reference's default format): sources given and results
    dumped in ORIGINAL numbering must equal what the plain CSR run / the oracle give; the apps' -check runs in the stored numbering"""
    O = oracle
    kind, scale, ef, seed = "rmat", 12, 16, 3
    src, dst, rowptr, adj, perm = graph(O, kind, scale, ef, seed, symmetric=(app == "cc"))
    w = O.gen_weights(len(perm), seed)[perm] if app != "cc" else None
    source = O.pick_source(rowptr, seed)
    args = ["-s", scale, "-e", ef, "-type", kind, "-seed", seed, "-format", "vcsr", "-check"] + mode
    if app in ("bfs", "sssp", "sswp"):
        args += ["-source", source]
    if app in ("pr", "hits"):
        args += ["-it", 4]
    out, dump = run_app(app, args, tmp_path)
    assert "error count: 0" in out
    if app == "bfs":
        assert (np.fromfile(dump, np.int32) == O.bfs_top_down(rowptr, adj, source)[0]).all()
    elif app == "sssp":
        assert (np.fromfile(dump, np.float32).view(np.int32) == O.sssp_bellman_ford(rowptr, adj, w, source)[0].view(np.int32)).all()
    elif app == "sswp":
        assert (np.fromfile(dump, np.float32).view(np.int32) == O.sswp_bellman_ford(rowptr, adj, w, source)[0].view(np.int32)).all()
    elif app == "pr":           # the renumbered rows sum their neighbours in another order: f32 rounding, not bit-exact
        ref = O.pagerank(rowptr, adj, 4, 1)
        assert np.max(np.abs(np.fromfile(dump, np.float32) - ref) / ref) < 2e-5
    elif app == "cc":
        assert (np.fromfile(dump, np.int32) == O.cc_sv(rowptr, adj)[0]).all()
    elif app == "scc":
        assert (np.fromfile(dump, np.int32) == O.scc_tarjan(rowptr, adj)).all()
    elif app == "hits":
        auth, hub = O.hits(rowptr, adj, 4)
        got = np.fromfile(dump, np.float64)
        V = len(rowptr) - 1
        assert np.max(np.abs(got[:V] - auth) / np.maximum(np.abs(auth), 1e-300)) < 1e-9
        assert np.max(np.abs(got[V:] - hub) / np.maximum(np.abs(hub), 1e-300)) < 1e-9


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
@pytest.mark.parametrize("kind,scale,ef,seed", CASES + [("rmat", 14, 16, 9)])
def test_coloring_app(kind, scale, ef, seed, fmt, tmp_path, oracle, ctx):
    """greedy speculative colouring (f1 widening) entirely on the generic operator path -- sparse frontiers, vertex post-ops, 64-bit
    vertex arrays, reduce<int>, generate_new_frontier.  The result is order-dependent (in the reference too): the test checks that
    it is a PROPER colouring of the symmetrised graph with no more colours than the largest degree + 1."""
    O = oracle
    src, dst, rowptr, adj, perm = graph(O, kind, scale, ef, seed, symmetric=True)
    out, dump = run_app("coloring", ["-s", scale, "-e", ef, "-type", kind, "-seed", seed, "-format", fmt, "-check"], tmp_path)
    assert "error count: 0" in out
    colors = np.fromfile(dump, np.int32)                       # ORIGINAL numbering
    V = len(rowptr) - 1
    assert colors.shape == (V,) and (colors >= 0).all()
    u = np.repeat(np.arange(V), np.diff(rowptr))
    proper = (colors[u] != colors[adj]) | (u == adj)
    assert proper.all(), f"{int((~proper).sum())} edges join vertices of one colour"
    assert colors.max() <= int(np.diff(rowptr).max())


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
@pytest.mark.parametrize("kind,scale,ef,seed", [("rmat", 12, 16, 3), ("ru", 11, 1, 5), ("ru", 11, 2, 5), ("ru", 10, 2, 9)])
def test_tc_app(kind, scale, ef, seed, fmt, tmp_path, oracle, ctx):
    """transitive-closure queries (algorithms/tc/tc.hpp): Purdom's algorithm (SCC + condensation, REDUCE_MAX, copy_if_indexes, an
    EdgesArray written in scatter) against one oracle BFS per distinct source; the app's own -check runs BFS per source on the device"""
    O = oracle
    src, dst, rowptr, adj, perm = graph(O, kind, scale, ef, seed)
    out, dump = run_app("tc", ["-s", scale, "-e", ef, "-type", kind, "-seed", seed, "-it", 96, "-check", "-format", fmt], tmp_path)
    assert "error count: 0" in out and "condensation:" in out
    t = np.fromfile(dump, np.int32).reshape(-1, 3)
    assert len(t) == 96 and ((t[:, :2] >= 0) & (t[:, :2] < (1 << scale))).all()
    for s in np.unique(t[:, 0]):
        levels = O.bfs_top_down(rowptr, adj, int(s))[0]
        rows = t[t[:, 0] == s]
        assert np.array_equal(rows[:, 2] != 0, levels[rows[:, 1]] != -1), int(s)
    assert 0 < t[:, 2].sum() < len(t) or ef != 2              # uniform graphs of out-degree 2 give both answers


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
@pytest.mark.parametrize("kind,scale,ef,seed,percent,length", [("rmat", 12, 8, 3, 10, 7), ("ru", 11, 4, 5, 100, 12), ("ru", 10, 1, 9, 50, 3)])
def test_rw_app(kind, scale, ef, seed, percent, length, fmt, tmp_path, oracle, ctx):
    """random walks (algorithms/rw/random_walk.hpp) on compute() over a sparse frontier: identical to the CPU restatement under
    both storage formats (the draws are keyed on ORIGINAL ids), every hop is an edge, walks stop at vertices without out-edges"""
    O = oracle
    src, dst, rowptr, adj, perm = graph(O, kind, scale, ef, seed, symmetric=True)
    out, dump = run_app("rw", ["-s", scale, "-e", ef, "-type", kind, "-seed", seed, "-wv", percent, "-it", length, "-check", "-format", fmt], tmp_path)
    assert "error count: 0" in out
    got = np.fromfile(dump, np.int32)
    ref = O.random_walk(rowptr, adj, seed, percent, length)
    assert np.array_equal(got, ref)
    walkers = np.nonzero(O.random_walk(rowptr, adj, seed, percent, 0) >= 0)[0]      # length 0: every walk vertex holds itself
    assert abs(len(walkers) / (1 << scale) - percent / 100) < 0.05 or percent == 100
    one = O.random_walk(rowptr, adj, seed, percent, 1)             # a single hop lands on a neighbour
    w = np.nonzero(one >= 0)[0]
    assert all(one[v] in adj[rowptr[v]:rowptr[v + 1]] for v in w[:200])


def scipy_max_flow(V, rowptr, adj, source, sink, capacity=100):
    import scipy.sparse as sp
    from scipy.sparse.csgraph import maximum_flow
    rows = np.repeat(np.arange(V), np.diff(rowptr))
    keep = rows != adj
    m = sp.csr_matrix((np.ones(keep.sum(), np.int32), (rows[keep], adj[keep])), shape=(V, V))
    m.data[:] = capacity                                   # parallel edges act as ONE edge (they are always updated together)
    return int(maximum_flow(m.astype(np.int32), source, sink).flow_value)


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
@pytest.mark.parametrize("kind,scale,ef,seed", [("rmat", 10, 8, 3), ("ru", 10, 3, 5), ("ru", 9, 1, 9)])
def test_mf_app_symmetric(kind, scale, ef, seed, fmt, tmp_path, oracle, ctx):
    """maximum flow (algorithms/mf/mf.hpp) on symmetric inputs, where the reference's residual scheme is the textbook one: the app,
    the CPU restatement and scipy's independent maximum_flow agree, for both storage formats"""
    O = oracle
    V = 1 << scale
    src, dst, rowptr, adj, perm = graph(O, kind, scale, ef, seed, symmetric=True)
    nz = np.nonzero(np.diff(rowptr))[0]
    pairs = [(int(nz[(7 * k + 1) % len(nz)]), int(nz[(13 * k + 5) % len(nz)])) for k in range(3)]
    hubs = np.argsort(-np.diff(rowptr), kind="stable")[:2]
    pairs.append((int(hubs[0]), int(hubs[1])))                # many augmentations, residual edges actually used
    for s, t in pairs:
        out, dump = run_app("mf", ["-s", scale, "-e", ef, "-type", kind, "-seed", seed, "-undirected", "-source", s, "-sink", t, "-check", "-format", fmt], tmp_path)
        assert "error count: 0" in out
        got = np.fromfile(dump, np.int32)
        assert tuple(got[:2]) == (s, t)
        ref, rounds = O.max_flow_ford_fulkerson(rowptr, adj, s, t)
        assert got[2] == ref == scipy_max_flow(V, rowptr, adj, s, t), (s, t, rounds)


@pytest.mark.parametrize("kind,scale,ef,seed", [("rmat", 10, 8, 3), ("ru", 10, 3, 5)])
def test_mf_app_directed(kind, scale, ef, seed, tmp_path, oracle, ctx):
    """directed inputs: reverse residuals exist only where the graph stores the reverse edge, so the value depends on the paths taken;
    with the smallest-id parent rule the device run reproduces the CPU restatement exactly (csr numbering) and never exceeds the
    true maximum flow"""
    O = oracle
    V = 1 << scale
    src, dst, rowptr, adj, perm = graph(O, kind, scale, ef, seed)
    nz = np.nonzero(np.diff(rowptr))[0]
    for k in range(3):
        s, t = int(nz[(5 * k + 2) % len(nz)]), int(nz[(11 * k + 7) % len(nz)])
        out, dump = run_app("mf", ["-s", scale, "-e", ef, "-type", kind, "-seed", seed, "-source", s, "-sink", t, "-check"], tmp_path)
        assert "error count: 0" in out
        got = int(np.fromfile(dump, np.int32)[2])
        assert got == O.max_flow_ford_fulkerson(rowptr, adj, s, t)[0]
        assert got <= scipy_max_flow(V, rowptr, adj, s, t)


def test_run_tests_harness(tmp_path, ctx):
    """apps/run_tests.py (the reference's apps/run_tests.py + scripts/*_api.py): prepares the graph set with create_vgl_graphs, runs
    every app's argument sets in both storage formats, greps AVG_PERF / error count and exports JSON + CSV"""
    import json
    out = subprocess.run(["python", os.path.join(ROOT, "apps", "run_tests.py"), "-m", "smoke", "-a", "all", "-f", "csr,vcsr", "-b", "-v", "-t", "120",
                          "-n", "harness_smoke"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    res = json.load(open(os.path.join(BIN, "harness_smoke.json")))
    rows = res["rows"]
    assert {r["app"] for r in rows} == {"bfs", "sssp", "pr", "cc", "sswp", "hits", "scc", "coloring", "rw", "tc", "mf"}
    assert all(r["errors"] == 0 and isinstance(r["perf"], float) and r["perf"] > 0 for r in rows), [r for r in rows if r["errors"] != 0]
    assert os.path.exists(os.path.join(BIN, "harness_smoke.csv")) and "VERIFIED 20 TESTS" in out.stdout


def test_api_performance_stats(tmp_path, ctx):
    """VGL byte accounting (settings.h:140-155, performance_stats.hpp): operator-API runs report per-abstraction times and the bandwidth
    the reference's model charges them; a top-down BFS visits exactly the edges of the reached vertices, 4 ints per edge in the bfs app"""
    out, _ = run_app("bfs", ["-s", 12, "-e", 16, "-type", "rmat", "-seed", 3, "-source", 1], tmp_path)
    m = re.search(r"edges visited: (\d+)", out)
    assert m and "total bandwidth:" in out and "Advance" in out and "GNF" in out and "Compute" in out
    visited = int(m.group(1))
    assert 0 < visited <= 16 << 12
    bw = float(re.search(r"total bandwidth: ([0-9.e+-]+) GB/s", out).group(1))
    rate = float(re.search(r"edges rate: ([0-9.e+-]+) MTEPS", out).group(1))
    # bytes = 16 B per visited edge + 8 B per computed vertex + 4 B per GNF vertex, over the same inner wall time as the edges rate
    assert bw * 1e9 / (rate * 1e6) >= 16.0
    fused, _ = run_app("bfs", ["-s", 12, "-e", 16, "-type", "rmat", "-seed", 3, "-source", 1, "-fused", "-do"], tmp_path)
    assert "total bandwidth:" not in fused                     # the fused path does not go through the operator primitives


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
@pytest.mark.parametrize("kind,scale,ef", [("rmat", 12, 16), ("ru", 11, 24)])
def test_advance_six_functor_contract(kind, scale, ef, fmt, ctx):
    """the collective functor set of scatter / gather (common/advance.hpp:6-115): never called on CSR_GRAPH, called for rows shorter
    than VECTOR_CORE_THRESHOLD_VALUE on VECTOR_CSR_GRAPH (advance_worker.hpp:204-319), for all-active / sparse / dense frontiers in
    both directions, with the local / global edge position contract and the direction check; the app counts every call."""
    out = subprocess.run([os.path.join(BIN, "advance_contract_hip"), "-s", str(scale), "-e", str(ef), "-type", kind, "-format", fmt],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "error count: 0" in out.stdout, out.stdout + out.stderr


def test_corrupt_el_container_and_bad_source_are_refused(tmp_path, oracle, ctx):
    """-import trusts nothing in the file: counts the file cannot hold, negative counts and ids outside [0, V) are refused on
    the host; so are -source / -sink ids outside the graph."""
    O = oracle
    rng = np.random.default_rng(3)
    V, E = 500, 4000
    src, dst = rng.integers(0, V, E).astype(np.int32), rng.integers(0, V, E).astype(np.int32)
    good = str(tmp_path / "good.el_container")
    O.write_el_container(good, V, src, dst)
    raw = bytearray(open(good, "rb").read())

    def run(data, extra=()):
        path = str(tmp_path / "bad.el_container")
        open(path, "wb").write(bytes(data))
        return subprocess.run([os.path.join(BIN, "bfs_hip"), "-import", path, *extra], capture_output=True, text=True, timeout=120)

    bad = bytearray(raw); bad[4:12] = np.int64(1 << 40).tobytes()          # edge count far beyond the file
    out = run(bad); assert out.returncode == 1 and "corrupt header" in out.stdout
    bad = bytearray(raw); bad[4:12] = np.int64(-5).tobytes()
    out = run(bad); assert out.returncode == 1 and "corrupt header" in out.stdout
    bad = bytearray(raw); bad[0:4] = np.int32(0).tobytes()
    out = run(bad); assert out.returncode == 1 and "corrupt header" in out.stdout
    bad = bytearray(raw); bad[16 + 4 * 7:16 + 4 * 8] = np.int32(V).tobytes()          # a source id == V
    out = run(bad); assert out.returncode == 1 and "id out of range" in out.stdout
    bad = bytearray(raw); bad[16 + 4 * E + 4 * 9:16 + 4 * E + 4 * 10] = np.int32(-1).tobytes()   # a negative destination id
    out = run(bad); assert out.returncode == 1 and "id out of range" in out.stdout
    for fmt in ("csr", "vcsr"):
        out = run(raw, ("-source", str(V), "-format", fmt)); assert out.returncode == 1 and "outside" in out.stdout
    out = run(raw, ("-source", "3")); assert out.returncode == 0
