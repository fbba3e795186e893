#!/usr/bin/env python3
"""PageRank pull: exact-order kernel vs the blocked (LDS-window) pass, per-kernel HIP-event times.
usage: blocked_bench.py [uniform|rmat] [scale] [edge factor]   (run on the GPU box: gpurun -- python profiles/microbench/blocked_bench.py)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vectorgraphlibrary_amd import api

kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 25
ef = int(sys.argv[3]) if len(sys.argv) > 3 else 32
ctx = api.Context(0)
V, E = 1 << scale, (1 << scale) * ef
s, d = (ctx.gen_uniform if kind == "uniform" else ctx.gen_rmat)(scale, ef, 1)
g = api.Graph.from_coo(ctx, V, s, d, with_incoming=True, renumber=None if kind == "uniform" else "total")
del s, d
t0 = time.perf_counter()
api.page_rank(g, 1, raw=True, mode=api.PR_BLOCKED)
torch.cuda.synchronize()
print(f"{kind}-{scale}x{ef}: first blocked iteration incl. plan build {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
for mode, name in ((api.PR_BLOCKED, "blocked"), (api.PR_EXACT_ORDER, "exact-order")):
    api.page_rank(g, 2, raw=True, mode=mode)
    ctx.timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 10
    r, st = api.page_rank(g, iters, raw=True, mode=mode)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    parts = {k: ctx.timing_get(k) for k in ("pr_pull", "pr_blk_gather", "pr_blk_accumulate")}
    ctx.timing(False)
    alg = 8 * E + 28 * V
    print(f"  {name:12s} {dt * 1e3:7.3f} ms/iteration  {E / dt / 1e9:7.1f} GTEPS  {alg / dt / 1e9:7.1f} GB/s algorithmic ({alg / dt / 8e12:.3f} of 8 TB/s)  "
          + "  ".join(f"{k} {v[1] / max(v[0], 1):.3f} ms x{v[0]}" for k, v in parts.items() if v[0]), flush=True)
    if mode == api.PR_BLOCKED:
        rb = r.clone()
    else:
        rel = ((rb - r).abs() / r).max().item()
        print(f"  max relative difference blocked vs exact-order: {rel:.3e}")
g.close()

if kind == "rmat":
    # SSSP on the same kind of graph: push (active tiles), pull (blocked, every step), direction-optimising
    s, d = ctx.gen_rmat(scale, ef, 1)
    g = api.Graph.from_coo(ctx, V, s, d, with_incoming=True, want_perm=True, renumber="total")
    w = ctx.gather_u32(g.perm, ctx.gen_weights(E, 1))
    del s, d
    deg = g.out_rowptr[1:] - g.out_rowptr[:-1]
    srcs = [int(x) for x in torch.nonzero(deg > 0).flatten()[torch.randint(0, int((deg > 0).sum()), (3,), generator=torch.Generator().manual_seed(5))]]
    t0 = time.perf_counter()
    plan = api.SsspPullPlan(g, w)
    torch.cuda.synchronize()
    print(f"SSSP pull plan build {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    ref = None
    for name, kw in (("push all-active", dict(mode=api.SSSP_ALL_ACTIVE)), ("push active tiles", dict(mode=api.SSSP_ACTIVE_TILES)),
                     ("pull (blocked)", dict(mode=api.SSSP_PULL, plan=plan)), ("direction-opt default", dict(mode=api.SSSP_DIRECTION_OPT, plan=plan)),
                     ("direction-opt 0.2", dict(mode=api.SSSP_DIRECTION_OPT, plan=plan, share="0.2")),
                     ("direction-opt 0.35", dict(mode=api.SSSP_DIRECTION_OPT, plan=plan, share="0.35")),
                     ("direction-opt 0.65", dict(mode=api.SSSP_DIRECTION_OPT, plan=plan, share="0.65")),
                     ("direction-opt 0.8", dict(mode=api.SSSP_DIRECTION_OPT, plan=plan, share="0.8")),
                     ("push only (share 2)", dict(mode=api.SSSP_DIRECTION_OPT, plan=plan, share="2"))):
        share = kw.pop("share", None)
        if share:
            os.environ["VGL_SSSP_PULL_SHARE"] = share
        api.sssp(g, w, srcs[0], raw=True, **kw)
        ctx.timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sts = []
        for sx in srcs:
            dd, st = api.sssp(g, w, sx, raw=True, **kw)
            sts.append(st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / len(srcs)
        parts = {k: ctx.timing_get(k) for k in ("sssp_relax", "sssp_pull_gather", "sssp_pull_accumulate")}
        ctx.timing(False)
        os.environ.pop("VGL_SSSP_PULL_SHARE", None)
        if ref is None:
            ref = dd.clone()
        same = bool((dd.view(torch.int32) == ref.view(torch.int32)).all())
        print(f"  {name:20s} {dt * 1e3:8.3f} ms  {E / dt / 1e9:6.1f} GTEPS  steps {sts[-1]['iterations']} (push {sts[-1]['push_steps']} pull {sts[-1]['pull_steps']})  same bits {same}  "
              + "  ".join(f"{k} {v[1] / max(v[0], 1):.3f} ms x{v[0]}" for k, v in parts.items() if v[0]), flush=True)
