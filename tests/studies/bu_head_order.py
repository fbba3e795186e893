#!/usr/bin/env python3
"""CPU study (oracle only): how many bottom-up candidates enter each probe round when the head records hold the first eight in-neighbours in
adjacency order, the eight smallest ids, or the eight largest.   python tests/studies/bu_head_order.py [scale]"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import oracle as O
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ef = 32
V = 1 << scale
s, d = O.gen_rmat(scale, ef, 1)
# degree-sorted renumbering by total degree (descending)
deg = np.bincount(s, minlength=V) + np.bincount(d, minlength=V)
order = np.argsort(-deg, kind='stable')
newid = np.empty(V, np.int64); newid[order] = np.arange(V)
s2 = newid[s].astype(np.int32); d2 = newid[d].astype(np.int32)
rp, adj, _ = O.coo_to_csr(V, s2, d2)
irp, iadj, _ = O.coo_to_csr(V, d2, s2)
outdeg = np.diff(rp)
src = int(np.nonzero(outdeg > 0)[0][1234 % (outdeg > 0).sum()])
lev = O.bfs_top_down(rp, adj, src)[0]
lev = np.asarray(lev)
print('levels hist', np.bincount(lev[lev >= 0]) if lev.min() >= 0 else np.unique(lev, return_counts=True))
indeg = np.diff(irp)
E = len(adj)
maxl = lev.max()
for L in range(1, maxl + 1):
    # level L is being found; frontier = level L-1
    unvis = (lev >= L) | (lev < 0) if lev.min() < 0 else (lev >= L)
    cand = unvis & (indeg > 0)
    front = lev == (L - 1)
    cidx = np.nonzero(cand)[0]
    # first in-neighbour (adjacency order)
    first = iadj[irp[cidx]]
    hit_first = front[first]
    # smallest-id in-neighbour
    mn = np.minimum.reduceat(iadj, irp[cidx])
    hit_min = front[mn]
    found = lev[cidx] == L
    # of first probes, how many land in first 1M ids (scaled: V/16)
    lowfrac = (first < V // 16).mean(); lowfrac_min = (mn < V // 16).mean()
    print(f'L={L}: |F|={front.sum()} M(F)={outdeg[front].sum()/E:.3f}E cand={len(cidx)} found={found.sum()} hit_first={hit_first.sum()} hit_min={hit_min.sum()}  first<V/16 {lowfrac:.2f} min<V/16 {lowfrac_min:.2f}')

print("---- rounds: candidates entering round 1 (1 probe), round 2 (3 probes), round 3 (record 2: 4 probes), deferred to heavy pass; heavy probes")
def simulate(iadj_used, name):
    for L in (3, 4, 5):
        unvis = (lev >= L) | (lev < 0)
        cand = np.nonzero(unvis & (indeg > 0))[0]
        front = lev == (L - 1)
        n = indeg[cand]
        alive = np.ones(len(cand), bool)
        entered = []
        pos = 0
        for width in (1, 3, 4):
            entered.append(int(alive.sum()))
            hit = np.zeros(len(cand), bool)
            for q in range(pos, pos + width):
                ok = alive & (n > q)
                idx = irp[cand[ok]] + q
                h = np.zeros(len(cand), bool); h[ok] = front[iadj_used[idx]]
                hit |= h
            pos += width
            alive &= ~hit
            alive &= (n > pos) if True else alive   # rows exhausted are finished (not found)
        deferred = int(alive.sum())
        # heavy: sequential scan from position 8 until hit
        hp = 0
        for c in cand[alive]:
            row = iadj_used[irp[c] + 8: irp[c + 1]]
            f = np.nonzero(front[row])[0]
            hp += (f[0] + 1) if len(f) else len(row)
        print(f'{name} L={L}: cand {len(cand)} rounds {entered} deferred {deferred} heavy probes {hp}')
simulate(iadj, 'adjacency order')
# sorted ascending within each row
rows = np.repeat(np.arange(V), indeg)
o = np.lexsort((iadj, rows))
simulate(iadj[o], 'ascending ids  ')
o2 = np.lexsort((-iadj.astype(np.int64), rows))
simulate(iadj[o2], 'descending ids ')
print('---- hybrid: head = 8 smallest ids, heavy pass scans whole row in adjacency order')
sorted_adj = iadj[o]
for L in (3, 4, 5):
    unvis = (lev >= L) | (lev < 0)
    cand = np.nonzero(unvis & (indeg > 8))[0]
    front = lev == (L - 1)
    hp = 0; nd = 0; hp16 = 0
    for c in cand:
        if front[sorted_adj[irp[c]:irp[c] + 8]].any(): continue
        nd += 1
        row = iadj[irp[c]: irp[c + 1]]
        f = np.nonzero(front[row])[0]
        k = (f[0] + 1) if len(f) else len(row)
        hp += k; hp16 += (k + 15) // 16
    print(f'L={L}: deferred {nd} heavy probes {hp}  16-wide steps {hp16}')
print('---- now: head = first 8, heavy from 8')
for L in (3, 4, 5):
    unvis = (lev >= L) | (lev < 0)
    cand = np.nonzero(unvis & (indeg > 8))[0]
    front = lev == (L - 1)
    hp = 0; nd = 0; hp16 = 0
    for c in cand:
        if front[iadj[irp[c]:irp[c] + 8]].any(): continue
        nd += 1
        row = iadj[irp[c] + 8: irp[c + 1]]
        f = np.nonzero(front[row])[0]
        k = (f[0] + 1) if len(f) else len(row)
        hp += k; hp16 += (k + 15) // 16
    print(f'L={L}: deferred {nd} heavy probes {hp}  16-wide steps {hp16}')
