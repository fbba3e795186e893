#!/usr/bin/env python3
"""Where a TOP-DOWN traversal of RMAT-24 spends its time (GPU box): per-level trace (VGL_BFS_TRACE=1, synchronising) of a few
sources without and with the blocked-level plan, then per-kernel event times and wall time of 16 sources.
usage: python3 tests/studies/td_levels_study.py [--scale 24]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from vectorgraphlibrary_amd import api

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=24)
ap.add_argument("--sources", type=int, default=16)
args = ap.parse_args()
ctx = api.Context(0)
V = 1 << args.scale
src, dst = ctx.gen_rmat(args.scale, 32, 1)
g = api.Graph.from_coo(ctx, V, src, dst, with_incoming=True, renumber="total")
del src, dst
sources = bench.pick_sources(g.out_rowptr, args.sources + 2, 1)
KERNELS = ("bfs_top_down", "bfs_small_levels", "bfs_bitmap_expand", "gnf", "bfs_blk_gather", "bfs_blk_accumulate")


def measure(tag):
    for s in sources[:2]:
        api.bfs(g, s, api.BFS_TOP_DOWN, raw=True)
    os.environ["VGL_BFS_TRACE"] = "1"
    for s in sources[2:4]:
        print(f"---- {tag}: trace of source {s}", flush=True)
        sys.stderr.flush()
        api.bfs(g, s, api.BFS_TOP_DOWN, raw=True)
        sys.stderr.flush()
    os.environ.pop("VGL_BFS_TRACE")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in sources[2:]:
        api.bfs(g, s, api.BFS_TOP_DOWN, raw=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.sources * 1e3
    ctx.timing(True)
    for s in sources[2:]:
        api.bfs(g, s, api.BFS_TOP_DOWN, raw=True)
    torch.cuda.synchronize()
    kern = {k: ctx.timing_get(k) for k in KERNELS}
    ctx.timing(False)
    print(f"==== {tag}: {ms:.3f} ms per traversal | " + "  ".join(f"{k.replace('bfs_', '')} {n / args.sources:.1f}x {t / args.sources * 1e3:.0f}us" for k, (n, t) in kern.items() if n), flush=True)


measure("plain")
g.prepare_blocked_bfs()
measure("blocked levels prepared")
