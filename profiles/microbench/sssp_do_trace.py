#!/usr/bin/env python3
"""Direction-optimising Bellman-Ford on RMAT-24x32 (the bench's SSSP leg alone): N runs from one source, wall time per run.
Under rocprofv3 --kernel-trace --stats the per-kernel totals divided by the runs show where a run's time goes.
usage: sssp_do_trace.py [runs] [mode: do|pull|tiles|all]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vectorgraphlibrary_amd import api

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mode = sys.argv[2] if len(sys.argv) > 2 else "do"
ctx = api.Context(0)
scale, ef = 24, 32
V, E = 1 << scale, (1 << scale) * ef
s, d = ctx.gen_rmat(scale, ef, 1)
g = api.Graph.from_coo(ctx, V, s, d, with_incoming=True, want_perm=True, renumber="total")
w = ctx.gather_u32(g.perm, ctx.gen_weights(E, 1))
del s, d
deg = g.out_rowptr[1:] - g.out_rowptr[:-1]
src = int(torch.nonzero(deg > 0).flatten()[12345])
plan = api.SsspPullPlan(g, w)
kw = {"do": dict(mode=api.SSSP_DIRECTION_OPT, plan=plan), "pull": dict(mode=api.SSSP_PULL, plan=plan),
      "tiles": dict(mode=api.SSSP_ACTIVE_TILES), "all": dict(mode=api.SSSP_ALL_ACTIVE)}[mode]
api.sssp(g, w, src, raw=True, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(runs):
    dist, st = api.sssp(g, w, src, raw=True, **kw)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / runs
print(f"{mode}: {dt * 1e3:.2f} ms per run, {st['iterations']} steps ({st['push_steps']} push, {st['pull_steps']} pull), runs = {runs} (+1 warm-up)")
