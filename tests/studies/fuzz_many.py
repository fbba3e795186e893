import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import oracle as O
from vectorgraphlibrary_amd import api
from vectorgraphlibrary_amd import sharded as vs
ctx = api.Context(0)
def relerr(a,b): return float(np.max(np.abs(a-b)/np.maximum(np.abs(b),1e-300)))
t0=time.time(); bad=0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(77000 + seed)
    V = int(rng.integers(1, 60000))
    E = int(rng.integers(0, 20 * V + 1))
    e1, e2 = 1 + seed % 4, 1 + (seed // 4) % 4
    src = np.minimum((rng.random(E) ** e1 * V).astype(np.int32), V - 1)
    dst = np.minimum((rng.random(E) ** e2 * V).astype(np.int32), V - 1)
    rowptr, adj, perm = O.coo_to_csr(V, src, dst)
    w_in = (rng.random(E) * 100).astype(np.float32); w = w_in[perm] if E else np.zeros(0, np.float32)
    dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device) if len(a) else torch.zeros(0, dtype=dt, device=ctx.device)
    ren = [None, "total", "out"][seed % 3]
    # round 3: layout switches of the blocked advance (fused tiles down to tiny block pairs, pieces of a third of the edges, tiny fused
    # pieces), the blocked Shiloach-Vishkin hook, giant-hub workgroups of the ordered PageRank -- all read when a plan / schedule is built
    env = {}
    if seed % 3 == 1: env["VGL_BLK_FUSE_MIN"] = "64"
    if seed % 3 == 2: env["VGL_BLK_FUSE_MIN"] = "3000"
    if seed % 5 == 0: env["VGL_BLK_PIECE_EDGES"] = str(max(4096, E // 3))
    if seed % 2 == 0: env["VGL_CC_BLOCKED"] = "1"
    if seed % 4 == 3: env["VGL_PULL_GIANT_DEGREE"] = "600"
    if seed % 7 == 0: env["VGL_BLK_FUSED_UNIT"] = "64"
    os.environ.update(env)
    g = api.Graph.from_coo(ctx, V, dev(src, torch.int32), dev(dst, torch.int32), want_perm=True, renumber=ren)
    w_d = ctx.gather_u32(g.perm, dev(w_in, torch.float32)) if E else torch.zeros(1, dtype=torch.float32, device=ctx.device)
    source = int(rng.integers(0, V))
    try:
        ref_levels,_ = O.bfs_top_down(rowptr, adj, source)
        for mode in (api.BFS_TOP_DOWN, api.BFS_DIRECTION_OPT):
            assert (api.bfs(g, source, mode)[0].cpu().numpy() == ref_levels).all(), f"bfs {mode}"
        if seed % 2:                                       # blocked top-down levels (every level: share 0), pull / direction-optimising SSSP
            g.prepare_blocked_bfs()
            os.environ["VGL_BFS_BLOCKED_SHARE"] = "0" if seed % 4 == 1 else "0.1"
            for mode in (api.BFS_TOP_DOWN, api.BFS_DIRECTION_OPT):
                assert (api.bfs(g, source, mode)[0].cpu().numpy() == ref_levels).all(), f"blocked bfs {mode}"
            os.environ.pop("VGL_BFS_BLOCKED_SHARE", None)
        ref_dist,_ = O.sssp_bellman_ford(rowptr, adj, w, source)
        for mode in (api.SSSP_PULL, api.SSSP_DIRECTION_OPT):
            d,_ = api.sssp(g, w_d, source, mode)
            assert (d.cpu().numpy().view(np.int32) == ref_dist.view(np.int32)).all(), f"sssp {mode}"
        for mode in (api.SSSP_ALL_ACTIVE, api.SSSP_ACTIVE_TILES, api.SSSP_DELTA_STEPPING):
            d,_ = api.sssp(g, w_d, source, mode, delta=float(rng.choice([0.5, 7.0, 16.0, 40.0])))
            assert (d.cpu().numpy().view(np.int32) == ref_dist.view(np.int32)).all(), f"sssp {mode}"
        assert (api.sswp(g, w_d, source)[0].cpu().numpy().view(np.int32) == O.sswp_bellman_ford(rowptr, adj, w, source)[0].view(np.int32)).all(), "sswp"
        wd_pull = api.sswp(g, w_d, source, api.SSSP_DIRECTION_OPT)[0]
        assert (wd_pull.cpu().numpy().view(np.int32) == O.sswp_bellman_ford(rowptr, adj, w, source)[0].view(np.int32)).all(), "sswp do"
        # the C++ super-step loops in a world of one (same code path as the sharded runs, no exchange)
        for mode in (api.BFS_TOP_DOWN, api.BFS_DIRECTION_OPT):
            assert (g.to_original(vs.bfs_run_sharded(g, None, g.vertex_id(source), mode, global_edges=E)[0]).cpu().numpy() == ref_levels).all(), f"sharded-loop bfs {mode}"
        assert (g.to_original(vs.sssp_run_sharded(g, None, w_d, g.vertex_id(source))[0]).cpu().numpy().view(np.int32) == ref_dist.view(np.int32)).all(), "sharded-loop sssp"
        pr = api.page_rank(g, 3)[0].cpu().numpy(); prr = O.pagerank(rowptr, adj, 3, 1)
        if ren is None: assert (pr.view(np.int32) == prr.view(np.int32)).all(), "pr bits"
        else: assert relerr(pr.astype(np.float64), prr.astype(np.float64)) < 1e-5, "pr"
        if ren is None: assert (api.connected_components(g)[0].cpu().numpy() == O.cc_sv(rowptr, adj)[0]).all(), "cc"   # directed min-label propagation is numbering-dependent
        assert (api.strongly_connected_components(g)[0].cpu().numpy() == O.scc_tarjan(rowptr, adj)).all(), "scc"
        if E:
            s2, d2 = np.concatenate([src, dst]), np.concatenate([dst, src])
            gs = api.Graph.from_coo(ctx, V, dev(s2, torch.int32), dev(d2, torch.int32), with_incoming=False, renumber=ren)
            rp2, adj2, _ = O.coo_to_csr(V, s2, d2, want_perm=False)
            assert (api.connected_components(gs, symmetric=True)[0].cpu().numpy() == O.cc_sv(rp2, adj2)[0]).all(), "cc sym"
            gs.close()
    except AssertionError as ex:
        bad += 1; print("FAIL seed", seed, "V", V, "E", E, "ren", ren, ex, flush=True)
    g.close()
    for k in env: os.environ.pop(k, None)
    if seed % 20 == 0: print("seed", seed, "elapsed", round(time.time()-t0,1), flush=True)
print("done, failures:", bad)
