#!/usr/bin/env python3
"""Delta-stepping SSSP on RMAT-24x32 (the bench's leg alone): N runs from one source with one plan; under rocprofv3 --kernel-trace --stats the
per-kernel totals divided by the runs show where a run goes; VGL_HIP_DEBUG=1 prints the bucket steps.  usage: sssp_ds_trace.py [runs] [delta]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vectorgraphlibrary_amd import api

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
delta = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
ctx = api.Context(0)
scale, ef = 24, 32
V, E = 1 << scale, (1 << scale) * ef
s, d = ctx.gen_rmat(scale, ef, 1)
g = api.Graph.from_coo(ctx, V, s, d, with_incoming=False, want_perm=True, renumber="total")
w = ctx.gather_u32(g.perm, ctx.gen_weights(E, 1))
del s, d
deg = g.out_rowptr[1:] - g.out_rowptr[:-1]
src = int(torch.nonzero(deg > 0).flatten()[12345])
plan = api.SsspPlan(g, w, delta)
api.sssp(g, w, src, plan=plan, raw=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(runs):
    dist, st = api.sssp(g, w, src, plan=plan, raw=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / runs
print(f"delta-stepping (delta {delta}): {dt * 1e3:.2f} ms per run, {st['iterations']} steps, {st['edges_relaxed'] / E:.2f} E relaxed, runs = {runs} (+1 warm-up)")
