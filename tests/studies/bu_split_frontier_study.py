#!/usr/bin/env python3
"""Study (GPU box, not a test): the FIRST bottom-up level of RMAT traversals if its frontier were split at an id T -- the frontier vertices with id >= T
(the low-degree tail of a degree-sorted graph) expanded top-down, the bottom-up probe looking for parents among ids < T only.  A candidate whose 8th
smallest in-neighbour is >= T then needs nothing beyond its head records (every in-neighbour it has left is >= T and is served by the top-down part).
Per T: out-edges of the tail part, rows still deferred and the entries their scan would read (in-neighbours < T, adjacency sorted) against today's.
usage: python3 tests/studies/bu_split_frontier_study.py [scale] [n_sources]"""
import sys

sys.path.insert(0, ".")
import torch

import bench
from vectorgraphlibrary_amd import api

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = api.Context(0)
src, dst = ctx.gen_rmat(scale, 32, 1)
V = 1 << scale
g = api.Graph.from_coo(ctx, V, src, dst, with_incoming=True, renumber="total")
del src, dst
dev = ctx.device
odeg = (g.out_rowptr[1:] - g.out_rowptr[:-1])
ideg = (g.in_rowptr[1:] - g.in_rowptr[:-1])
E = int(odeg.sum())
KMAX = 8
rows = torch.repeat_interleave(torch.arange(V, device=dev, dtype=torch.int64), ideg)
key = torch.unique(rows * (1 << 32) + g.in_adj.long())
del rows
krow, kid = key >> 32, key & 0xFFFFFFFF
del key
dd = torch.zeros(V + 1, dtype=torch.int64, device=dev)
dd[1:] = torch.cumsum(torch.bincount(krow, minlength=V), 0)
pos = torch.arange(krow.numel(), device=dev) - dd[krow]
small = torch.full((V, KMAX), V, dtype=torch.int64, device=dev)
m = pos < KMAX
small[krow[m], pos[m]] = kid[m]
ddeg = dd[1:] - dd[:-1]
del pos, m
factor = max(1, (E // V) // 2)
sources = bench.pick_sources(g.out_rowptr, nsrc + 4, 1)[4:]
for s in sources:
    lv, st = api.bfs(g, s, api.BFS_DIRECTION_OPT, raw=True)
    prevF, visited = 0, 0
    first_bu = None
    for L in range(1, int(lv.max()) + 1):
        fr = lv == L
        F, M = int(fr.sum()), int(odeg[fr].sum())
        visited += F
        if F > prevF and M >= ((V - visited) * factor + V) // 15:
            first_bu = L
            break
        prevF = F
    if first_bu is None:
        continue
    L = first_bu
    frb = torch.zeros(V + 1, dtype=torch.bool, device=dev)
    frb[:V] = lv == L
    cand = ((lv == -1) | (lv > L)) & (ideg > 0)
    sm, dg = small[cand], ddeg[cand]
    hit = frb[sm]
    anyhit = hit.any(1)
    deferred = (~anyhit) & (dg > KMAX)
    print(f"source {s}: first bottom-up level {L}: F {F} M {M} candidates {int(cand.sum())} found in heads {int(anyhit.sum())} "
          f"deferred now {int(deferred.sum())} rows / {int(ideg[cand][deferred].sum())} entries; stats: bu_probes {st['bu_edges']} bu_found {st['bu_found']}", flush=True)
    for T in (1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20):
        tail = frb[:V].clone()
        tail[:T] = False
        tail_edges = int(odeg[tail].sum())
        hitT = hit & (sm < T)
        anyT = hitT.any(1)
        # found by the top-down part: any in-neighbour in the tail (full knowledge from the traversal: the vertex is at level L + 1)
        lvl_next = (lv[cand] == L + 1)
        still = (~anyT) & (dg > KMAX) & (sm[:, KMAX - 1] < T)
        # entries below T per row (sorted adjacency: the scan stops at the first id >= T)
        below = torch.bincount(krow[kid < T], minlength=V)[cand]
        need_plane1 = (~hitT[:, :4].any(1)) & (sm[:, 3] < T)
        print(f"   T 2^{T.bit_length() - 1:2d}: tail frontier {int(tail.sum()):8d} vertices / {tail_edges:9d} out-edges | found by the probe part {int(anyT.sum()):8d} of {int(lvl_next.sum())} "
              f"| second plane needed {int(need_plane1.sum()):8d} (now {int((~hit[:, :4].any(1) & (dg > 4)).sum())}) | deferred {int(still.sum()):7d} rows / {int(below[still].sum()):9d} entries below T", flush=True)
