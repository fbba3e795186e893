import sys, time; sys.path.insert(0, '.')
import torch
from vectorgraphlibrary_amd import api
from vectorgraphlibrary_amd import distributed as vd
scale = int(sys.argv[1]); ef = 32
ctx = api.Context(0)
t0 = time.time()
g, degrees, _ = vd.build_generated_shard(ctx, scale, ef, 1, 0, 1, kind="rmat", renumber="total", placement="ranges")
ctx.sync(); print("build s", round(time.time() - t0, 1), "V", g.V, "E", int(g.out_adj.numel()), flush=True)
V = g.V
assert int(g.out_rowptr[-1]) == (1 << scale) * ef == int(g.in_rowptr[-1])
assert torch.equal((g.out_rowptr[1:] - g.out_rowptr[:-1]).to(torch.int32), degrees)
src_ok = int(torch.nonzero(degrees > 0)[12345])
for s in (0, src_ok):
    lv_td, st_td = api.bfs(g, s, api.BFS_TOP_DOWN, raw=True)
    lv_do, st_do = api.bfs(g, s, api.BFS_DIRECTION_OPT, raw=True)
    assert torch.equal(lv_td, lv_do), "DO != TD"
    print("source", s, "levels", st_do["levels"], "reached", st_do["discovered"], "td edges", st_td["edges_examined"], "do edges", st_do["edges_examined"], flush=True)
    # level consistency on a sample of rows: every out-neighbour of a reached vertex is reached at most one level later
    rows = torch.randint(0, V, (200000,), device=ctx.device)
    rows = rows[lv_td[rows] > 0]
    b, e = g.out_rowptr[rows], g.out_rowptr[rows + 1]
    has = e > b
    first, last = g.out_adj[b[has]].long(), g.out_adj[(e[has] - 1)].long()
    for nb in (first, last):
        assert bool((lv_td[nb] > 0).all()) and bool((lv_td[nb] <= lv_td[rows[has]] + 1).all())
torch.cuda.synchronize(); t1 = time.perf_counter()
for k in range(4): api.bfs(g, src_ok, api.BFS_DIRECTION_OPT, raw=True)
torch.cuda.synchronize(); print("ms per BFS", round((time.perf_counter() - t1) / 4 * 1e3, 3), flush=True)
print("BIG_CHECK_OK", scale)
