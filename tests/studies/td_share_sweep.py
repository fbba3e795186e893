#!/usr/bin/env python3
"""Top-down traversal time of RMAT-24 (blocked levels prepared) against VGL_BFS_BLOCKED_SHARE -- the share of the edges from which a level takes the
blocked pass (GPU box).  Levels of every source are checked against the plain top-down run.  usage: python3 tests/studies/td_share_sweep.py [shares ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from vectorgraphlibrary_amd import api

shares = [float(x) for x in sys.argv[1:]] or [0.1, 0.05, 0.02, 0.01, 0.005]
ctx = api.Context(0)
src, dst = ctx.gen_rmat(24, 32, 1)
g = api.Graph.from_coo(ctx, 1 << 24, src, dst, with_incoming=True, renumber="total")
del src, dst
sources = bench.pick_sources(g.out_rowptr, 18, 1)
ref = {s: api.bfs(g, s, api.BFS_TOP_DOWN, raw=True)[0].clone() for s in sources}
g.prepare_blocked_bfs()
KERNELS = ("bfs_top_down", "bfs_small_levels", "bfs_bitmap_expand", "gnf", "bfs_blk_gather", "bfs_blk_accumulate")
for share in shares:
    os.environ["VGL_BFS_BLOCKED_SHARE"] = str(share)
    bad = 0
    for s in sources[:2]:
        api.bfs(g, s, api.BFS_TOP_DOWN, raw=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in sources[2:]:
        api.bfs(g, s, api.BFS_TOP_DOWN, raw=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 16 * 1e3
    ctx.timing(True)
    for s in sources[2:]:
        lv, st = api.bfs(g, s, api.BFS_TOP_DOWN, raw=True)
        bad += int(api.count_not_equal(ctx, lv, ref[s]) != 0)
    torch.cuda.synchronize()
    kern = {k: ctx.timing_get(k) for k in KERNELS}
    ctx.timing(False)
    print(f"share {share:6.3f}: {ms:.3f} ms per traversal  wrong {bad} | " + "  ".join(f"{k.replace('bfs_', '')} {n / 16:.1f}x {t / 16 * 1e3:.0f}us" for k, (n, t) in kern.items() if n), flush=True)
