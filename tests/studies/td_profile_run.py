#!/usr/bin/env python3
"""16 top-down traversals of RMAT-24 with the blocked-level plan prepared, nothing else: the program to put behind rocprofv3 --kernel-trace --stats
(per-kernel averages of the top-down mode).  usage: rocprofv3 --kernel-trace --stats -d DIR -- python3 tests/studies/td_profile_run.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from vectorgraphlibrary_amd import api

ctx = api.Context(0)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
src, dst = ctx.gen_rmat(scale, 32, 1)
g = api.Graph.from_coo(ctx, 1 << scale, src, dst, with_incoming=True, renumber="total")
del src, dst
g.prepare_blocked_bfs()
sources = bench.pick_sources(g.out_rowptr, 18, 1)
for s in sources:
    api.bfs(g, s, api.BFS_TOP_DOWN, raw=True)
torch.cuda.synchronize()
print("done")
