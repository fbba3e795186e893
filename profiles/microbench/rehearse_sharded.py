#!/usr/bin/env python3
"""Rehearsal of the P-rank sharded BFS (vgl_hip_bfs_run_sharded, the C++ super-step loop) on ONE GPU: P rank THREADS of one process, each with
its own context / stream / dealt shard of RMAT-<scale> and a communicator over the host-staged transport with VGL_HOSTED_SERIALIZE=1, so
that between two exchanges the ranks work one at a time and every rank's HIP-event kernel times are those of its own work.  What it
reports per traversal: kernel time per rank (step kernels, frontier generation, owner resolve), collectives and bytes received per rank --
i.e. everything of an 8-GPU traversal except the time of the RCCL collectives themselves.
usage: rehearse_sharded.py [scale=27] [ranks=8] [sources=4] [transport=hosted|peer]
transport peer (round 4): the rank threads write into each other's device windows (PEER transport; nothing serialises them, so the kernel
times include waiting for the other ranks that share the card): what it reports then is the protocol -- flag rounds and bytes per level."""
import os
import sys
import threading
import time
import uuid

TRANSPORT = sys.argv[4] if len(sys.argv) > 4 else "hosted"
if TRANSPORT == "hosted":
    os.environ["VGL_HOSTED_SERIALIZE"] = "1"
else:
    os.environ["GPU_MAX_HW_QUEUES"] = "16"            # every rank thread's stream on a hardware queue of its own: a polling kernel must not sit in front of a peer's put
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402
from vectorgraphlibrary_amd import api  # noqa: E402
from vectorgraphlibrary_amd import distributed as vd  # noqa: E402
from vectorgraphlibrary_amd import sharded as vs  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 27
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nsrc = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ef, seed = 32, 1
V, E = 1 << scale, (1 << scale) * ef
name = "/vgl_rehearse_%s" % uuid.uuid4().hex[:10]
KERNELS = ("bfs_bottom_up", "bfs_top_down", "gnf", "bfs_shard_resolve")

build_ctx = api.Context(0)
t0 = time.time()
shards = []
for r in range(P):
    shard, degrees, bounds = vd.build_generated_shard(build_ctx, scale, ef, seed, r, P, kind="rmat", renumber="total", placement="dealt")
    shards.append((shard.out_rowptr, shard.out_adj, shard.in_rowptr, shard.in_adj, shard.row_begin, shard.row_end))
    shard.close()
    if r == 0:
        nz = torch.nonzero(degrees > 0).flatten()
        g = torch.Generator(device="cpu").manual_seed(seed)
        sources = [int(nz[i]) for i in torch.randint(0, nz.numel(), (nsrc + 1,), generator=g)]
    del degrees
build_ctx.sync()
print(f"RMAT-{scale}x{ef}: {P} dealt shards built in {time.time() - t0:.1f} s ({shards[0][1].numel() / 1e6:.0f} M out-edges each)", flush=True)

results, certs, errors = [None] * P, [None] * P, []


def rank_main(r):
    try:
        stream = torch.cuda.Stream(device=0)
        with torch.cuda.stream(stream):
            ctx = api.Context(0)                       # binds this thread's stream
            orp, oadj, irp, iadj, lo, hi = shards[r]
            shard = api.Graph(ctx, V, orp, oadj, irp, iadj, lo, hi)
            comm = vs.Comm.hosted(ctx, r, P, name, slot_bytes=32 << 20) if TRANSPORT == "hosted" else vs.Comm.peer(ctx, r, P, name, window_bytes=48 << 20)
            levels = torch.empty(V, dtype=torch.int32, device=ctx.device)
            vs.bfs_run_sharded(shard, comm, sources[0], api.BFS_DIRECTION_OPT, global_edges=E, gather_levels=False, levels=levels, want_stats=False)   # warm-up
            per = []
            for s in sources[1:]:
                ctx.timing(True)
                _, st = vs.bfs_run_sharded(shard, comm, s, api.BFS_DIRECTION_OPT, global_edges=E, gather_levels=False, levels=levels, want_stats=True)
                k = {n: ctx.timing_get(n) for n in KERNELS}
                ctx.timing(False)
                per.append((st, k, comm.stats()))
            # the last source once more with the levels gathered: both halves of the breadth-first certificate on this rank's rows
            vs.bfs_run_sharded(shard, comm, sources[-1], api.BFS_DIRECTION_OPT, global_edges=E, gather_levels=True, levels=levels, want_stats=False)
            certs[r] = vd.bfs_levels_certificate(levels, shard, sources[-1])
            results[r] = per
            comm.barrier()
            comm.close()
            shard.close()
            ctx.close()
    except Exception as e:                             # noqa: BLE001
        errors.append((r, repr(e)))


threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
for t in threads:
    t.start()
for t in threads:
    t.join()
if errors:
    sys.exit("rank errors: %s" % errors)
for i, s in enumerate(sources[1:]):
    rows = [results[r][i] for r in range(P)]
    st0 = rows[0][0]
    kern = [sum(ms for _, ms in rows[r][1].values()) for r in range(P)]
    print(f"source {s}: {st0['levels']} levels ({st0['td_steps']} top-down, {st0['bu_steps']} bottom-up); kernel ms per rank: "
          + " ".join(f"{x:.3f}" for x in kern) + f"  (max {max(kern):.3f}, sum {sum(kern):.3f});  "
          + "rank 0 by kernel: " + ", ".join(f"{n} {c}x {ms:.3f}" for n, (c, ms) in rows[0][1].items())
          + f";  exchange per rank: {rows[0][2]['collectives']} collectives"
          + (f" in {rows[0][2]['exchanges']} flag rounds ({rows[0][2]['exchanges'] / st0['levels']:.2f} per level)" if TRANSPORT == "peer" else "")
          + f", {rows[0][2]['bytes_received'] / 2**20:.1f} MiB received, {rows[0][2]['sparse_levels']} id-list levels", flush=True)
print("breadth-first certificate of the last traversal (out-edges never skip a level / every reached vertex has a parent one level up), per rank:",
      " ".join("%s/%s" % ("ok" if e else "FAIL", "ok" if p else "FAIL") for e, p in certs), flush=True)
if not all(e and p for e, p in certs):
    sys.exit("certificate failed")
