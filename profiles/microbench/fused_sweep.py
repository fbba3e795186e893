"""pull pass timing for several fused-tile thresholds (RMAT-24x32)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vectorgraphlibrary_amd import api
ctx = api.Context(0)
scale, ef = 24, 32
V, E = 1 << scale, (1 << scale) * ef
s, d = ctx.gen_rmat(scale, ef, 1)
g = api.Graph.from_coo(ctx, V, s, d, with_incoming=True, want_perm=True, renumber="total")
w = ctx.gather_u32(g.perm, ctx.gen_weights(E, 1))
del s, d
deg = g.out_rowptr[1:] - g.out_rowptr[:-1]
src = int(torch.nonzero(deg > 0).flatten()[12345])
ref = None
for fm in sys.argv[1:]:
    os.environ["VGL_BLK_FUSE_MIN"] = fm
    os.environ["VGL_BLK_BUILD_TRACE"] = "1"
    plan = api.SsspPullPlan(g, w)
    os.environ.pop("VGL_BLK_BUILD_TRACE")
    for mode, name in ((api.SSSP_PULL, "pull"), (api.SSSP_DIRECTION_OPT, "do")):
        api.sssp(g, w, src, raw=True, mode=mode, plan=plan)
        ctx.timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dist, st = api.sssp(g, w, src, raw=True, mode=mode, plan=plan)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        k = {n: ctx.timing_get(n) for n in ("sssp_pull_gather", "sssp_pull_accumulate", "sssp_pull_fused", "sssp_relax", "gnf")}
        ctx.timing(False)
        if ref is None: ref = dist.clone()
        assert torch.equal(ref.view(torch.int32), dist.view(torch.int32))
        per = {n: (c, round(ms / c, 4) if c else 0) for n, (c, ms) in k.items()}
        tot = sum(ms for c, ms in k.values())
        print(f"fuse_min {fm} {name}: {dt*1e3:.2f} ms, steps {st['iterations']} (push {st['push_steps']} pull {st['pull_steps']}), kernels {tot:.2f} ms, per launch {per}", flush=True)
    plan.close()
