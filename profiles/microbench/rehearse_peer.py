#!/usr/bin/env python3
"""Rehearsal of the P-rank sharded BFS over the PEER transport on ONE GPU: P rank PROCESSES (the parent only starts them), each with a dealt
shard of RMAT-<scale>, mapping the other ranks' device windows through hipIpc.  (Rank threads of one process cannot rehearse this transport:
a hipFree on one thread waits for every stream of the process, also for a peer thread's polling kernel that waits for this thread's next put.)
Reports per traversal: levels, collectives, FLAG ROUNDS (sequence numbers: the collectives of a group share one) and bytes received per rank,
wall time per traversal with all ranks sharing the card, and the breadth-first certificate of the last traversal on every rank.
usage: rehearse_peer.py [scale=26] [ranks=4] [sources=3]"""
import os
import subprocess
import sys
import time
import uuid

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")


def rank_main(scale, P, nsrc, name, r):
    sys.path.insert(0, ROOT)
    import torch
    from vectorgraphlibrary_amd import api
    from vectorgraphlibrary_amd import distributed as vd
    from vectorgraphlibrary_amd import sharded as vs
    ef, seed = 32, 1
    V, E = 1 << scale, (1 << scale) * ef
    dbg = os.environ.get("VGL_REHEARSE_DEBUG") == "1"
    if dbg: print(f"rank {r}: imports done", flush=True)
    ctx = api.Context(0)
    if dbg: print(f"rank {r}: context up", flush=True)
    # the shards are built ONE RANK AT A TIME (a file lock): four processes streaming and sorting 2^29 edges each at the same moment on one
    # card did not finish in minutes at scale 24 (each alone: seconds) -- the build is not what is rehearsed here
    import fcntl
    with open("/tmp/vgl_rehearse_build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        shard, degrees, bounds = vd.build_generated_shard(ctx, scale, ef, seed, r, P, kind="rmat", renumber="total", placement="dealt")
        ctx.sync()
        torch.cuda.empty_cache()
        fcntl.flock(lock, fcntl.LOCK_UN)
    nz = torch.nonzero(degrees > 0).flatten()
    g = torch.Generator(device="cpu").manual_seed(seed)
    sources = [int(nz[i]) for i in torch.randint(0, nz.numel(), (nsrc + 1,), generator=g)]
    del degrees
    if dbg: print(f"rank {r}: shard built", flush=True)
    comm = vs.Comm.peer(ctx, r, P, name, window_bytes=int(os.environ.get("VGL_REHEARSE_WINDOW", 48 << 20)))
    if dbg: print(f"rank {r}: communicator up", flush=True)
    levels = torch.empty(V, dtype=torch.int32, device=ctx.device)
    vs.bfs_run_sharded(shard, comm, sources[0], api.BFS_DIRECTION_OPT, global_edges=E, gather_levels=False, levels=levels, want_stats=False)
    if dbg: print(f"rank {r}: warm-up traversal done", flush=True)
    for s in sources[1:]:
        comm.barrier()
        t0 = time.perf_counter()
        _, st = vs.bfs_run_sharded(shard, comm, s, api.BFS_DIRECTION_OPT, global_edges=E, gather_levels=False, levels=levels, want_stats=True)
        ctx.sync()
        dt = time.perf_counter() - t0
        ex = comm.stats()
        if r == 0:
            print(f"source {s}: {st['levels']} levels ({st['td_steps']} top-down, {st['bu_steps']} bottom-up), {dt * 1e3:.2f} ms with {P} ranks sharing the card; "
                  f"rank 0: {ex['collectives']} collectives in {ex['exchanges']} flag rounds = {ex['exchanges'] / st['levels']:.2f} per level, "
                  f"{ex['bytes_received'] / 2**20:.1f} MiB received, {ex['sparse_levels']} id-list levels", flush=True)
    vs.bfs_run_sharded(shard, comm, sources[-1], api.BFS_DIRECTION_OPT, global_edges=E, gather_levels=True, levels=levels, want_stats=False)
    e_ok, p_ok = vd.bfs_levels_certificate(levels, shard, sources[-1])
    print(f"rank {r}: certificate of the last traversal (out-edges never skip a level / every reached vertex has a parent one level up): "
          f"{'ok' if e_ok else 'FAIL'}/{'ok' if p_ok else 'FAIL'}", flush=True)
    comm.barrier()
    comm.close()
    shard.close()
    ctx.close()
    sys.exit(0 if e_ok and p_ok else 3)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--rank":
        rank_main(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6]))
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    nsrc = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    name = "/vgl_rehearse_%s" % uuid.uuid4().hex[:10]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch  # noqa: F401  (no GPU call: only pages the libraries in, so that the ranks' imports do not take minutes on a fresh box)
    print("parent: libraries paged in, starting the ranks", flush=True)
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(scale), str(P), str(nsrc), name, str(r)], env=env) for r in range(P)]
    rc = [p.wait() for p in procs]
    print("ranks exited with", rc, flush=True)
    sys.exit(max(rc))
