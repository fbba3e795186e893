#!/usr/bin/env python3
"""Study (GPU box, not a test): what the bottom-up levels of a direction-optimising RMAT traversal look like -- candidates per
level, how many are found, through which in-neighbour (smallest id first), in-degree mix of the candidates, where the smallest
in-neighbour ids fall.  Feeds the design of the bottom-up kernels.
usage: python3 tests/studies/bu_levels_study.py [scale] [n_sources]"""
import sys

sys.path.insert(0, ".")
import torch

from vectorgraphlibrary_amd import api

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ef = 32
ctx = api.Context(0)
src, dst = ctx.gen_rmat(scale, ef, 1)
V = 1 << scale
g = api.Graph.from_coo(ctx, V, src, dst, with_incoming=True, renumber="total")
del src, dst
odeg = (g.out_rowptr[1:] - g.out_rowptr[:-1])
ideg = (g.in_rowptr[1:] - g.in_rowptr[:-1])
nz = torch.nonzero(odeg > 0).flatten()
rows = torch.repeat_interleave(torch.arange(V, device=ctx.device, dtype=torch.int64), ideg)      # row of every in-edge
in_adj = g.in_adj.long()
print("V", V, "rows with in-edges", int((ideg > 0).sum()), "E", g.E, flush=True)
# in-degree mix of all rows with in-edges
for lo, hi in ((1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 1 << 30)):
    m = (ideg >= lo) & (ideg <= hi)
    print(f"  in-degree {lo}..{hi}: rows {int(m.sum())}, edges {int(ideg[m].sum())}")
for k in range(nsrc):
    s = int(nz[(k * 7919 + 13) % len(nz)])
    lv, st = api.bfs(g, s, api.BFS_DIRECTION_OPT, raw=True)
    mx = int(lv.max())
    print(f"source {s} odeg {int(odeg[s])} td/bu {st['td_steps']}/{st['bu_steps']} levels {mx}", flush=True)
    for L in range(1, mx + 1):
        fr = lv == L
        F = int(fr.sum())
        M = int(odeg[fr].sum())
        unv = ((lv == -1) | (lv > L)) & (ideg > 0)
        found = lv == (L + 1)
        # per candidate row: number of in-neighbours in the frontier, smallest in-neighbour, is it in the frontier
        in_f = fr[in_adj]                                   # per in-edge: source in frontier L
        hits = torch.zeros(V, dtype=torch.int64, device=ctx.device).index_add_(0, rows, in_f.long())
        mn = torch.full((V,), 1 << 40, dtype=torch.int64, device=ctx.device).index_reduce_(0, rows, in_adj, "amin")
        first_hit = unv & (mn < (1 << 40)) & fr[mn.clamp(max=V - 1)]
        ncand = int(unv.sum())
        msg = f"  L{L}: F {F} M {M} | candidates {ncand} found {int(found.sum())} first-probe hits {int(first_hit.sum())}"
        msg += f" | cand in-edges {int(ideg[unv].sum())} | unfound cand in-edges {int(ideg[unv & ~found].sum())}"
        for b in (16, 20):
            msg += f" | cand min-nbr<2^{b}: {int((unv & (mn < (1 << b))).sum())}"
        # frontier mass by id range
        msg += f" | frontier ids<2^16: {int(fr[:1 << 16].sum())} <2^20: {int(fr[:1 << 20].sum())}"
        # found candidates by in-degree class
        for lo, hi in ((1, 1), (2, 4), (5, 1 << 30)):
            m = unv & (ideg >= lo) & (ideg <= hi)
            msg += f" | deg{lo}-{hi if hi < 99 else ''}: cand {int(m.sum())} found {int((m & found).sum())}"
        print(msg, flush=True)
        del in_f, hits, mn
