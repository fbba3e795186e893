import sys, time; sys.path.insert(0,'.')
import torch
from vectorgraphlibrary_amd import api
ctx = api.Context(0)
def t(f, n=3):
    f(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): r=f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n, r
# config 4: PR on uniform-random scale 25 x 32 (single GPU here)
for kind, scale in (('ru',25),('rmat',24)):
    ef=32; V=1<<scale; E=V*ef
    src,dst = (ctx.gen_uniform if kind=='ru' else ctx.gen_rmat)(scale, ef, 1)
    g = api.Graph.from_coo(ctx, V, src, dst, with_incoming=True, renumber='total' if kind=='rmat' else None)   # in-CSR: in-degrees without atomics
    del src,dst
    ctx.timing(True)
    dt,(rk,st) = t(lambda: api.page_rank(g, 10, raw=True), 2)
    n,ms = ctx.timing_get('pr_pull'); ctx.timing(False)
    print(f"PR {kind}-{scale}: {dt*1e3:.1f} ms for 10 it -> {10*E/dt/1e9:.1f} GTEPS; pull kernel {ms/n:.2f} ms/launch -> {(8*E+28*V)/(ms/n*1e-3)/1e9:.0f} GB/s algorithmic ({(8*E+28*V)/(ms/n*1e-3)/8e12*100:.1f}% of peak); ranks_sum {st['ranks_sum']:.6f}")
    g.close(); del g
# config 5 (single GPU stand-in): CC on symmetrised RMAT-24 x 16 (E = 537M directed after symmetrisation)
scale, ef = 24, 16; V=1<<scale
src,dst = ctx.gen_rmat(scale, ef, 1)
s2,d2 = torch.cat([src,dst]), torch.cat([dst,src]); del src,dst
E = s2.numel()
for ren in (None,'total'):
    g = api.Graph.from_coo(ctx, V, s2, d2, with_incoming=False, renumber=ren)
    ctx.timing(True)
    dt,(c,st) = t(lambda: api.connected_components(g, raw=True), 2)
    n,ms = ctx.timing_get('cc_hook'); n2,ms2 = ctx.timing_get('cc_jump'); ctx.timing(False)
    print(f"CC rmat-24 sym renumber={ren}: {dt*1e3:.1f} ms, {st['hook_passes']} hook passes -> {E/dt/1e9:.1f} GTEPS; hook {ms/n:.2f} ms/launch -> {(8*E+12*V)/(ms/n*1e-3)/1e9:.0f} GB/s algorithmic; jump {ms2/max(n2,1):.2f} ms/launch; components {int(torch.unique(c).numel())}")
    dt2,(c2,st2) = t(lambda: api.connected_components(g, raw=True, symmetric=True), 3)
    print(f"CC rmat-24 sym renumber={ren} union-find path: {dt2*1e3:.2f} ms -> {E/dt2/1e9:.1f} GTEPS; same labels: {bool(torch.equal(c, c2))}")
    g.close(); del g
