"""Device-memory drift over graph / plan create-destroy cycles (graph, blocked BFS / PageRank / SSSP / CC plans): free memory must come back.
usage (GPU box): python tests/studies/leak_check.py"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vectorgraphlibrary_amd import api
ctx = api.Context(0)
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 2**20
s, d = ctx.gen_rmat(20, 32, 1)
w = ctx.gen_weights(s.numel(), 1)
base = None
for it in range(12):
    g = api.Graph.from_coo(ctx, 1 << 20, s, d, with_incoming=True, want_perm=True, renumber="total")
    wd = ctx.gather_u32(g.perm, w)
    g.prepare_blocked_bfs()
    api.bfs(g, 5, api.BFS_TOP_DOWN); api.bfs(g, 5, api.BFS_DIRECTION_OPT)
    api.page_rank(g, 2, mode=api.PR_BLOCKED)
    plan = api.SsspPullPlan(g, wd); api.sssp(g, wd, 5, api.SSSP_DIRECTION_OPT, plan=plan); plan.close()
    import os; os.environ["VGL_CC_BLOCKED"] = "1"; api.connected_components(g)
    g.close(); del g, wd
    torch.cuda.empty_cache()
    f = free()
    if it == 1: base = f
    print(it, round(f, 1), flush=True)
print("drift MiB since iteration 1:", round(base - f, 1))
