#!/usr/bin/env python3
"""Study (GPU box, not a test): how many bottom-up candidates a short first record resolves.  For every bottom-up level of a few
RMAT traversals: candidates resolved by the K0 smallest in-neighbours (all of them / only those with id < W tested from an LDS
window), candidates whose row is exhausted by those entries, and what is left for the head records.
usage: python3 tests/studies/bu_plane0_study.py [scale] [n_sources]"""
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
dev = ctx.device
odeg = (g.out_rowptr[1:] - g.out_rowptr[:-1])
ideg = (g.in_rowptr[1:] - g.in_rowptr[:-1])
nzv = torch.nonzero(odeg > 0).flatten()
KMAX = 8
# the KMAX smallest DISTINCT in-neighbours of every row, ascending (V = none)
rows = torch.repeat_interleave(torch.arange(V, device=dev, dtype=torch.int64), ideg)
key = torch.unique(rows * (1 << 32) + g.in_adj.long())            # sorted, duplicates removed
del rows
krow = key >> 32
kid = key & 0xFFFFFFFF
del key
dd = torch.zeros(V + 1, dtype=torch.int64, device=dev)
dd[1:] = torch.cumsum(torch.bincount(krow, minlength=V), 0)
pos = torch.arange(krow.numel(), device=dev) - dd[krow]
small = torch.full((V, KMAX), V, dtype=torch.int64, device=dev)
m = pos < KMAX
small[krow[m], pos[m]] = kid[m]
ddeg = dd[1:] - dd[:-1]                                            # distinct in-degree
del krow, kid, pos, m
print("V", V, "rows with in-edges", int((ideg > 0).sum()), flush=True)
for k in range(nsrc):
    s = int(nzv[(k * 7919 + 13) % len(nzv)])
    lv, st = api.bfs(g, s, api.BFS_DIRECTION_OPT, raw=True)
    mx = int(lv.max())
    print(f"source {s} odeg {int(odeg[s])} td/bu {st['td_steps']}/{st['bu_steps']} levels {mx}", flush=True)
    for L in range(2, min(mx, 6) + 1):
        fr = torch.zeros(V + 1, dtype=torch.bool, device=dev)
        fr[:V] = lv == L
        cand = ((lv == -1) | (lv > L)) & (ideg > 0)
        nc = int(cand.sum())
        if nc == 0:
            continue
        sm = small[cand]
        dg = ddeg[cand]
        hit = fr[sm]                                               # (nc, KMAX)
        anyhit = hit.any(1)
        print(f"  L{L}: F {int(fr.sum())} candidates {nc} found {int(anyhit.sum())} (within first {KMAX}); first-probe {int(hit[:, 0].sum())}")
        for W in (1 << 16, 1 << 18, 1 << 20, V):
            for K0 in (1, 2, 4):
                inw = sm[:, :K0] < W
                res = (hit[:, :K0] & inw).any(1)                   # found through a window entry of the short record
                # row exhausted by the short record: all its distinct in-neighbours are among the K0 entries AND inside the window
                done = ~res & (dg <= K0) & (inw | (sm[:, :K0] >= V)).all(1)
                left = nc - int(res.sum()) - int(done.sum())
                print(f"      W=2^{W.bit_length() - 1} K0={K0}: resolved {int(res.sum())} ({100.0 * int(res.sum()) / nc:.1f}%), exhausted {int(done.sum())}, "
                      f"to the head records {left} ({100.0 * left / nc:.1f}%)")
        del sm, dg, hit, anyhit, fr, cand
