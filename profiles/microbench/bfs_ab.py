#!/usr/bin/env python3
"""A/B of direction-optimising BFS variants on ONE graph build (GPU box): every argument is a variant "NAME:K=V,K=V" of environment
switches the library reads per call (DESIGN 'Run-time switches'); each variant runs the bench's sources (4 warm-up + N timed), is checked
against the top-down levels of every source, and is timed twice: wall time without event brackets, then per kernel with all brackets.
usage: python3 profiles/microbench/bfs_ab.py [--scale 24] [--steps 32] base: new:VGL_X=1 ..."""
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
ap.add_argument("--steps", type=int, default=32)
ap.add_argument("--warmup", type=int, default=4)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--per-source", action="store_true")
ap.add_argument("variants", nargs="+")
args = ap.parse_args()

ctx = api.Context(0)
V = 1 << args.scale
src, dst = ctx.gen_rmat(args.scale, 32, 1)
g = api.Graph.from_coo(ctx, V, src, dst, with_incoming=True, renumber="total")
E = g.E
del src, dst
sources = bench.pick_sources(g.out_rowptr, args.steps + args.warmup, 1)
ref = {}
for s in sources:
    lv, _ = api.bfs(g, s, api.BFS_TOP_DOWN, raw=True)
    ref[s] = lv.clone()
KERNELS = ("bfs_bottom_up", "bfs_bottom_up_heavy", "bfs_top_down", "bfs_small_levels", "bfs_bitmap_expand", "gnf")
variants = []
for v in args.variants:
    name, _, kv = v.partition(":")
    variants.append((name, dict(x.split("=", 1) for x in kv.split(",") if x)))
touched = set(k for _, env in variants for k in env)
results = {name: [] for name, _ in variants}
for rnd in range(args.rounds):
    for name, env in variants:
        for k in touched:
            os.environ.pop(k, None)
        os.environ.update(env)
        bad = 0
        for s in sources[:args.warmup]:
            api.bfs(g, s, api.BFS_DIRECTION_OPT, raw=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in sources[args.warmup:]:
            api.bfs(g, s, api.BFS_DIRECTION_OPT, raw=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        ctx.timing(True)
        stats = []
        prev = {k: 0.0 for k in KERNELS}
        for s in sources[args.warmup:]:
            lv, st = api.bfs(g, s, api.BFS_DIRECTION_OPT, raw=True)
            stats.append(st)
            bad += int(api.count_not_equal(ctx, lv, ref[s]) != 0)
            if args.per_source and rnd == 0:
                now = {k: ctx.timing_get(k)[1] for k in KERNELS}
                print(f"    {name} source {s}: td/bu {st['td_steps']}/{st['bu_steps']} td_edges {st['td_edges']} bu_probes {st['bu_edges']} bu_found {st['bu_found']} | "
                      + " ".join(f"{k.replace('bfs_', '')} {(now[k] - prev[k]) * 1e3:.0f}" for k in KERNELS if now[k] > prev[k]) + " us")
                prev = now
        torch.cuda.synchronize()
        kern = {k: ctx.timing_get(k) for k in KERNELS}
        ctx.timing(False)
        results[name].append(ms)
        ks = "  ".join(f"{k.replace('bfs_', '')} {n}x {t / args.steps * 1e3:.1f}us" for k, (n, t) in kern.items() if n)
        print(f"round {rnd} {name:24s} {ms:.4f} ms/traversal  {E / ms / 1e6:.0f} GTEPS  wrong {bad}  td/bu steps {sum(s['td_steps'] for s in stats)}/"
              f"{sum(s['bu_steps'] for s in stats)}  | per traversal: {ks}", flush=True)
print("summary (best of rounds): " + "  ".join(f"{n} {min(v):.4f}" for n, v in results.items()))
