#!/usr/bin/env python3
"""Wall time of building the blocked-advance plans (PageRank on uniform-25x32, SSSP pull on RMAT-24x32), twice each, with the builder's
stage trace (VGL_BLK_BUILD_TRACE=1) on stderr.   gpurun -- python profiles/microbench/plan_build.py"""
import os
import sys
import time

os.environ.setdefault("VGL_BLK_BUILD_TRACE", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vectorgraphlibrary_amd import api

ctx = api.Context(0)
which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("both", "pr"):
    scale, ef = 25, 32
    V, E = 1 << scale, (1 << scale) * ef
    s, d = ctx.gen_uniform(scale, ef, 1)
    g = api.Graph.from_coo(ctx, V, s, d, with_incoming=True)
    del s, d
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        api.page_rank(g, 1, raw=True, mode=api.PR_BLOCKED)
        torch.cuda.synchronize()
        print(f"uniform-25x32 blocked PageRank, 1 iteration, call {rep}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    g.close()
if which in ("both", "sssp"):
    scale, ef = 24, 32
    V, E = 1 << scale, (1 << scale) * ef
    s, d = ctx.gen_rmat(scale, ef, 1)
    g = api.Graph.from_coo(ctx, V, s, d, with_incoming=True, want_perm=True, renumber="total")
    w = ctx.gather_u32(g.perm, ctx.gen_weights(E, 1))
    del s, d
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        plan = api.SsspPullPlan(g, w)
        torch.cuda.synchronize()
        print(f"RMAT-24x32 SSSP pull plan, build {rep}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
        t0 = time.perf_counter()
        plan.close()
        print(f"   destroy: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    g.close()
