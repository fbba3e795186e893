"""BFS-only fuzz at larger sizes: random graph shapes (skewed / uniform, sparse / dense, isolated vertices, self loops, multi-edges),
both traversal modes, random settings of the small-level / bitmap-expand bounds, all three renumberings -- levels against the CPU
oracle, and identical statistics between the settings.  usage: fuzz_bfs.py <first seed> <last seed>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import oracle as O
from vectorgraphlibrary_amd import api
ctx = api.Context(0)
t0 = time.time(); bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(991000 + seed)
    V = int(rng.integers(2, 1 << int(rng.integers(8, 22))))
    E = int(rng.integers(0, int(rng.choice([1, 2, 8, 32])) * V + 1))
    e1, e2 = float(rng.choice([1, 1, 2, 4, 8])), float(rng.choice([1, 1, 2, 4, 8]))
    src = np.minimum((rng.random(E) ** e1 * V).astype(np.int32), V - 1)
    dst = np.minimum((rng.random(E) ** e2 * V).astype(np.int32), V - 1)
    if seed % 5 == 0 and E:                                     # a path graph glued on: many tiny levels
        n = min(V - 1, 3000); src = np.concatenate([src, np.arange(n, dtype=np.int32)]); dst = np.concatenate([dst, np.arange(1, n + 1, dtype=np.int32)]); E = len(src)
    rowptr, adj, _ = O.coo_to_csr(V, src, dst)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device) if len(a) else torch.zeros(0, dtype=torch.int32, device=ctx.device)
    ren = [None, "total", "out", "in"][seed % 4]
    g = api.Graph.from_coo(ctx, V, dev(src), dev(dst), renumber=ren)
    try:
        for source in {int(rng.integers(0, V)), int(np.argmax(np.diff(rowptr))), 0}:
            ref = O.bfs_top_down(rowptr, adj, source)[0]
            for mode in (api.BFS_TOP_DOWN, api.BFS_DIRECTION_OPT):
                seen = []
                for k in range(3):
                    env = {} if k == 0 else {"VGL_BFS_SMALL_M": str(int(rng.choice([0, 64, 300, 8192, 1000000]))), "VGL_BFS_BM_EXPAND": str(int(rng.choice([0, 100, 262144, 1 << 30])))}
                    if k == 2 and seed % 2: env["VGL_BFS_NO_HINT"] = "1"      # (round 3: the count launch walks the new frontier instead of taking F and M from its producer)
                    if k == 1 and seed % 3 == 0: env["VGL_TD_EMIT_EDGES"] = str(int(rng.choice([0, 4096, 1 << 40])))
                    if k == 2 and seed % 3: env["VGL_BFS_NO_SCAN_BOUND"] = "1"    # (round 4: the scanning frontier generation with a lower bound of M)
                    if k == 1 and seed % 2: env["VGL_TD_FILTER_SHARE"] = str(rng.choice([0, 2]))      # (round 4: visited-bitmap probe always / never)
                    os.environ.update(env)
                    try:
                        lv, st = api.bfs(g, source, mode)
                    finally:
                        for name in env: os.environ.pop(name, None)
                    assert (lv.cpu().numpy() == ref).all(), f"levels source {source} mode {mode} env {env}"
                    seen.append((st["levels"], st["edges_examined"], st["frontier_total"], st["discovered"], st["td_steps"], st["bu_steps"]))
                assert seen[0] == seen[1] == seen[2], f"stats differ source {source} mode {mode}: {seen}"
    except AssertionError as e:
        bad += 1; print("FAIL seed", seed, "V", V, "E", E, "renumber", ren, e, flush=True)
    g.close()
    if seed % 20 == 0: print("seed", seed, "V", V, "E", E, "elapsed %.1f" % (time.time() - t0), flush=True)
print("done, failures:", bad)
