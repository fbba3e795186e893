"""BASELINE.json configurations at their stated sizes (run with -m gpu on a MI355X): the HIP path through the C ABI against the
pinned CPU oracle on the SAME graph, bit-exact for BFS levels / SSSP f32 distances / CC labels and within 1e-6 relative for
PageRank (tolerance from BASELINE.json north_star).

The graphs are built on the device by the same generator + stable COO -> CSR that the small-size tests prove equal to the
oracle's (tests/test_gpu_parity.py::build_case; config 1 below repeats that proof at RMAT-18 x 32) and copied to the host for
the oracle, whose OpenMP team is sized to the CPUs the box grants (oracle.host_cpus()).  Reference entry points:
apps/bfs/bfs.cpp:41-46 (-check), algorithms/sssp/shortest_paths.hpp:85-163, algorithms/pr/pr.hpp:7-149,
algorithms/cc/shiloach_vishkin.hpp:7-88."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PR_RTOL = 1e-6
SEED = 1


def _relerr(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b.astype(np.float64)), 1e-300)))


def _sources(rowptr, n, seed):
    """n distinct non-isolated vertices at hashed positions (VGL_Graph::select_random_nz_vertex stand-in)"""
    rng = np.random.default_rng(seed)
    deg = np.diff(rowptr)
    out = []
    while len(out) < n:
        v = int(rng.integers(0, len(deg)))
        if deg[v] > 0 and v not in out:
            out.append(v)
    return out


@pytest.fixture(scope="module")
def rmat24(ctx, oracle):
    """BASELINE configs[1] / [2]: RMAT scale 24, edge factor 32, degree-sorted numbering (what bench.py times) + weights"""
    from vectorgraphlibrary_amd import api
    oracle.set_threads()
    scale, ef = 24, 32
    src, dst = ctx.gen_rmat(scale, ef, SEED)
    g = api.Graph.from_coo(ctx, 1 << scale, src, dst, with_incoming=True, want_perm=True, renumber="total")
    del src, dst
    w = ctx.gather_u32(g.perm, ctx.gen_weights(g.E, SEED))
    host = {"rowptr": g.out_rowptr.cpu().numpy(), "adj": g.out_adj.cpu().numpy(), "w": w.cpu().numpy()}
    yield g, w, host
    g.close()


def test_config1_bfs_top_down_rmat18x32(ctx, oracle):
    """configs[0]: BFS top-down on RMAT scale-18, edge factor 32; also the CSR-build proof at this size"""
    from vectorgraphlibrary_amd import api
    O = oracle
    scale, ef = 18, 32
    V = 1 << scale
    src, dst = ctx.gen_rmat(scale, ef, SEED)
    hs, hd = O.gen_rmat(scale, ef, SEED)
    assert (src.cpu().numpy() == hs).all() and (dst.cpu().numpy() == hd).all()
    g = api.Graph.from_coo(ctx, V, src, dst)
    rowptr, adj, _ = O.coo_to_csr(V, hs, hd)
    assert (g.out_rowptr.cpu().numpy() == rowptr).all() and (g.out_adj.cpu().numpy() == adj).all()
    for s in _sources(rowptr, 3, 18):
        ref, ref_st = O.bfs_top_down(rowptr, adj, s)
        lv, st = api.bfs(g, s, api.BFS_TOP_DOWN)
        assert (lv.cpu().numpy() == ref).all()
        assert st["levels"] == ref_st["levels"] and st["edges_examined"] == ref_st["edges_examined"]
        lv2, _ = api.bfs(g, s, api.BFS_DIRECTION_OPT)
        assert (lv2.cpu().numpy() == ref).all()
    g.close()


def test_config2_do_bfs_rmat24(rmat24, ctx, oracle):
    """configs[1]: direction-optimising BFS on RMAT-24 x 32 == the reference's top-down levels, 3 sources"""
    from vectorgraphlibrary_amd import api
    g, _, host = rmat24
    for s in _sources(host["rowptr"], 3, 24):
        ref, ref_st = oracle.bfs_top_down(host["rowptr"], host["adj"], s, parallel=True)
        for mode in (api.BFS_DIRECTION_OPT, api.BFS_TOP_DOWN):
            lv, st = api.bfs(g, s, mode, raw=True)
            assert (lv.cpu().numpy() == ref).all(), f"source {s} mode {mode}: levels differ from the oracle"
            assert st["discovered"] == ref_st["discovered"]
        assert st["edges_examined"] == ref_st["edges_examined"] and st["levels"] == ref_st["levels"]
    # the same traversals with the graph prepared for blocked top-down levels (vgl_hip_bfs_prepare_blocked): same levels, same statistics
    g.prepare_blocked_bfs()
    for s in _sources(host["rowptr"], 2, 24):
        ref, ref_st = oracle.bfs_top_down(host["rowptr"], host["adj"], s, parallel=True)
        lv, st = api.bfs(g, s, api.BFS_TOP_DOWN, raw=True)
        assert (lv.cpu().numpy() == ref).all(), f"source {s}: blocked top-down levels differ from the oracle"
        assert st["edges_examined"] == ref_st["edges_examined"] and st["levels"] == ref_st["levels"] and st["discovered"] == ref_st["discovered"]


def test_config3_bellman_ford_sssp_rmat24(rmat24, ctx, oracle):
    """configs[2]: Bellman-Ford SSSP on RMAT-24 x 32 with f32 weights: every schedule reaches the oracle's f32 bits"""
    from vectorgraphlibrary_amd import api
    g, w, host = rmat24
    # the pull / direction-optimising schedules run on ONE blocked plan (dense block pairs as fused tiles), as a caller would keep it
    plan = api.SsspPullPlan(g, w)
    nofuse = None
    for k, s in enumerate(_sources(host["rowptr"], 3, 240)):
        ref, iters = oracle.sssp_bellman_ford(host["rowptr"], host["adj"], host["w"], s, parallel=True)
        ref_bits = ref.view(np.int32)
        assert iters > 2 and int((ref < 3.0e38).sum()) > (1 << 22)
        modes = [("ALL_ACTIVE", dict(mode=api.SSSP_ALL_ACTIVE)), ("ACTIVE_TILES", dict(mode=api.SSSP_ACTIVE_TILES)),
                 ("DELTA_STEPPING", dict(mode=api.SSSP_DELTA_STEPPING, delta=10.0)), ("PULL", dict(mode=api.SSSP_PULL, plan=plan)),
                 ("DIRECTION_OPT", dict(mode=api.SSSP_DIRECTION_OPT, plan=plan))]
        if k > 0:
            modes = modes[1:]                        # the all-active sweep (16+ passes over all edges) once is enough
        for name, kw in modes:
            dist, st = api.sssp(g, w, s, raw=True, **kw)
            assert (dist.cpu().numpy().view(np.int32) == ref_bits).all(), f"source {s}, SSSP {name}: distances differ from the oracle"
            if name == "DIRECTION_OPT":
                assert st["pull_steps"] > 0 and st["push_steps"] > 0
        if k == 0:                                    # the two-pass layout without fused tiles reaches the same bits
            os.environ["VGL_BLK_FUSE_MIN"] = "0"
            try:
                nofuse = api.SsspPullPlan(g, w)
            finally:
                os.environ.pop("VGL_BLK_FUSE_MIN")
            dist, _ = api.sssp(g, w, s, raw=True, mode=api.SSSP_PULL, plan=nofuse)
            assert (dist.cpu().numpy().view(np.int32) == ref_bits).all()
            nofuse.close()
    plan.close()


def test_config4_pagerank_uniform25(ctx, oracle):
    """configs[3] (one GPU's worth: the whole graph): PageRank on uniform-random scale 25 x 32, 5 iterations, <= 1e-6 relative"""
    from vectorgraphlibrary_amd import api
    import torch
    oracle.set_threads()
    scale, ef, it = 25, 32, 5
    src, dst = ctx.gen_uniform(scale, ef, SEED)
    g = api.Graph.from_coo(ctx, 1 << scale, src, dst, with_incoming=True)
    del src, dst
    rowptr, adj = g.out_rowptr.cpu().numpy(), g.out_adj.cpu().numpy()
    ref = oracle.pagerank(rowptr, adj, it, 1, parallel=True)
    del rowptr, adj
    for name, mode in (("PR_EXACT_ORDER", api.PR_EXACT_ORDER), ("PR_BLOCKED", api.PR_BLOCKED)):
        ranks, st = api.page_rank(g, it, mode=mode)
        rk = ranks.cpu().numpy()
        assert _relerr(rk, ref) <= PR_RTOL, f"{name}: {_relerr(rk, ref)}"
        assert abs(st["ranks_sum"] - float(rk.astype(np.float64).sum())) < 1e-9
    g.close()
    torch.cuda.empty_cache()


def test_config5_cc_rmat24_symmetrised(ctx, oracle):
    """configs[4] at the single-GPU size: Shiloach-Vishkin labels on the symmetrised RMAT-24 x 16 (537 M stored edges)"""
    from vectorgraphlibrary_amd import api
    import torch
    oracle.set_threads()
    s, d = ctx.gen_rmat(24, 16, SEED)
    s, d = torch.cat([s, d]), torch.cat([d, s])
    g = api.Graph.from_coo(ctx, 1 << 24, s, d, with_incoming=False)
    del s, d
    ref, passes = oracle.cc_sv(g.out_rowptr.cpu().numpy(), g.out_adj.cpu().numpy(), parallel=True)
    assert passes >= 2
    for sym in (False, True):
        comp, st = api.connected_components(g, symmetric=sym)
        assert (comp.cpu().numpy() == ref).all(), f"CC (symmetric={sym}) labels differ from the oracle"
    g.close()
    torch.cuda.empty_cache()


def test_config5_cc_rmat27_on_one_gpu(ctx):
    """configs[4] at its stated scale on ONE MI355X: Shiloach-Vishkin (hook as blocked passes over row-range pieces: 4.29 G stored edges are
    more than one plan's 2^32) and the min-id union-find on the symmetrised RMAT-27 x 16, checked through properties no CPU oracle is
    needed for: the two algorithms agree on every label; every stored edge joins equal labels; labels are roots no larger than their
    vertices; the giant component holds most of the vertices that have edges."""
    from vectorgraphlibrary_amd import api
    from vectorgraphlibrary_amd import distributed as vd
    import torch
    scale, ef = 27, 16
    V = 1 << scale
    g, _, _ = vd.build_generated_shard(ctx, scale, ef, SEED, 0, 1, kind="rmat", renumber="total", placement="ranges", symmetric=True, with_incoming=False)
    assert g.E == 2 * ef * V and g.E >= (1 << 32)
    os.environ["VGL_CC_BLOCKED"] = "1"
    try:
        sv, st = api.connected_components(g, raw=True)
    finally:
        os.environ.pop("VGL_CC_BLOCKED")
    assert st["hook_passes"] >= 2
    uf, _ = api.connected_components(g, raw=True, symmetric=True)
    assert torch.equal(sv, uf), "Shiloach-Vishkin and union-find labels differ"
    lab = sv.long()
    assert bool((lab[lab] == lab).all()) and bool((lab <= torch.arange(V, device=ctx.device)).all())
    deg = g.out_rowptr[1:] - g.out_rowptr[:-1]
    step = 1 << 22                                               # rows per slice of the edge check
    for r0 in range(0, V, step):
        r1 = min(V, r0 + step)
        e0, e1 = int(g.out_rowptr[r0]), int(g.out_rowptr[r1])
        if e1 == e0:
            continue
        rows = torch.repeat_interleave(torch.arange(r0, r1, device=ctx.device), deg[r0:r1])
        assert bool((sv[rows] == sv[g.out_adj[e0:e1].long()]).all()), f"an edge of rows {r0}..{r1} joins two labels"
        del rows
    sizes = torch.bincount(lab, minlength=1)
    with_edges = int((deg > 0).sum())
    assert int(sizes.max()) > 0.9 * with_edges and int((deg == 0).sum()) == int(((sizes == 1) & (deg == 0)).sum())
    g.close()
    del g, sv, uf, lab, sizes
    torch.cuda.empty_cache()
    ctx.L.vgl_hip_ctx_trim(ctx.h)
