import sys, time; sys.path.insert(0,'.')
import torch
from vectorgraphlibrary_amd import api
ctx = api.Context(0)
scale, ef, seed = 24, 32, 1
V=1<<scale; E=V*ef
src,dst = ctx.gen_rmat(scale, ef, seed)
g = api.Graph.from_coo(ctx, V, src, dst, with_incoming=False, want_perm=True, renumber='total')
del src,dst
w = ctx.gather_u32(g.perm, ctx.gen_weights(E, seed))
deg = g.out_rowptr[1:]-g.out_rowptr[:-1]
nz = torch.nonzero(deg>0).flatten()
srcs=[int(nz[i]) for i in (12345, 999, 5000000)]
ref = {}
for s in srcs: ref[s] = api.sssp(g, w, s, api.SSSP_ACTIVE_TILES, raw=True)[0].clone()
def run(mode, **kw):
    api.sssp(g, w, srcs[0], mode, raw=True, **kw); torch.cuda.synchronize(); t0=time.perf_counter(); sts=[]
    for s in srcs:
        d, st = api.sssp(g, w, s, mode, raw=True, **kw); sts.append(st)
        assert torch.equal(d.view(torch.int32), ref[s].view(torch.int32))
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/len(srcs)
    return round(dt*1e3,2), round(E/dt/1e9,2), sts[0]['iterations'], round(sts[0]['edges_relaxed']/E,2)
print('active_tiles', run(api.SSSP_ACTIVE_TILES))
for delta in (2,4,8,12,16,24,32,50):
    t0=time.perf_counter(); plan = api.SsspPlan(g, w, delta); torch.cuda.synchronize(); tp=time.perf_counter()-t0
    print('delta', delta, run(api.SSSP_DELTA_STEPPING, plan=plan), 'plan build ms', round(tp*1e3,1)); plan.close()
