"""Helper of tests/test_sharded_gpu.py (not a test module): ONE rank of a multi-rank run of the C-ABI super-step loops
(vgl_hip_*_run_sharded) and exchanges (vgl_hip_exchange_*).  Every rank builds the same deterministic graph on the GPU, computes the
single-GPU fused results itself and compares the sharded results with them bit for bit.

    python tests/sharded_ranks.py rccl   RANK WORLD ID_FILE     one GPU per rank (a world of one + VGL_SHARD_FORCE_COLLECTIVES=1 on the one-GPU box)
    python tests/sharded_ranks.py hosted RANK WORLD SHM_NAME    ranks share cuda:0 through the host-staged transport
    python tests/sharded_ranks.py peer   RANK WORLD SHM_NAME    ranks share cuda:0 and write into each other's hipIpc-mapped windows (PEER)

Prints 'SHARDED_RANK_OK' on success."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vectorgraphlibrary_amd import api  # noqa: E402
from vectorgraphlibrary_amd import sharded as vs  # noqa: E402


def bits(t):
    return t.view(torch.int32) if t.dtype == torch.float32 else t


def check_exchanges(ctx, comm):
    rank, P, dev = comm.rank, comm.world, ctx.device
    n = 70001
    base = torch.arange(n, device=dev, dtype=torch.int64)
    # all-reduce, every (operator, type) the C ABI exports
    t = ((base * 7 + rank * 13) % 1000).to(torch.int32)
    want = torch.stack([((base * 7 + r * 13) % 1000).to(torch.int32) for r in range(P)])
    assert torch.equal(comm.allreduce(t.clone(), "min"), want.min(0).values)
    assert torch.equal(comm.allreduce(t.clone(), "sum"), want.sum(0).to(torch.int32))
    f = want.to(torch.float32) * 0.25
    assert torch.equal(comm.allreduce(f[rank].clone(), "min"), f.min(0).values)
    assert torch.equal(comm.allreduce(f[rank].clone(), "max"), f.max(0).values)
    fs = f[0].clone()
    for r in range(1, P):
        fs = fs + f[r]                                   # rank order: the order of the hosted fold; RCCL is checked to a tolerance
    got = comm.allreduce(f[rank].clone(), "sum")
    assert torch.allclose(got, fs, rtol=1e-6, atol=0)
    d = want.to(torch.float64)
    assert torch.allclose(comm.allreduce(d[rank].clone(), "sum"), d.sum(0), rtol=1e-12, atol=0)
    assert torch.equal(comm.allreduce(want[rank].to(torch.int64), "sum"), want.sum(0))
    # all-gather, all-gather of owned slices (unequal), bitmap OR (divisible and not divisible word counts)
    send = (base[:999] + rank * 1000).to(torch.int32)
    recv = torch.empty(999 * P, device=dev, dtype=torch.int32)
    comm.allgather(send, recv)
    assert torch.equal(recv, torch.cat([(base[:999] + r * 1000).to(torch.int32) for r in range(P)]))
    bounds = [0] + [int(n * (r + 1) ** 2 / P ** 2) for r in range(P)]
    arr = torch.full((n,), -1, device=dev, dtype=torch.int32)
    arr[bounds[rank]:bounds[rank + 1]] = rank
    comm.allgather_slices(arr, bounds)
    want_arr = torch.cat([torch.full((bounds[r + 1] - bounds[r],), r, device=dev, dtype=torch.int32) for r in range(P)])
    assert torch.equal(arr, want_arr)
    for words in (4096 * P, 4097 * P + 1):
        g = torch.Generator(device="cpu").manual_seed(1234 + words)
        allb = torch.randint(-2 ** 62, 2 ** 62, (P, words), generator=g, dtype=torch.int64) & torch.randint(-2 ** 62, 2 ** 62, (P, words), generator=g, dtype=torch.int64)
        mine = allb[rank].to(dev)
        comm.bitmap_or(mine)
        ref = allb[0]
        for r in range(1, P):
            ref = ref | allb[r]
        assert torch.equal(mine.cpu(), ref), words
    # changed-entries exchange: nothing changed / a few entries (one all-gather) / many (sized lists) / most (whole-array all-reduce)
    for k, take_min in ((0, True), (50, True), (5000, True), (5000, False), (60000, True)):
        before = torch.full((n,), 1000000, device=dev, dtype=torch.int32) if take_min else torch.zeros(n, device=dev, dtype=torch.int32)
        copies = []
        for r in range(P):
            v = before.clone()
            if k:
                idx = (torch.arange(k, device=dev, dtype=torch.int64) * (7 + 2 * r) + r * 31) % n
                v[idx] = ((idx * (r + 3)) % 999 + 1).to(torch.int32)
            copies.append(v)
        merged = torch.stack(copies).min(0).values if take_min else torch.stack(copies).max(0).values
        mine = copies[rank].clone()
        changed = comm.exchange_changed(before, mine, take_min)
        assert changed == (k > 0), (k, changed)
        assert torch.equal(mine, merged), (k, take_min)
    comm.barrier()


def main():
    transport, rank, world, token = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    per_rank_device = transport == "rccl" or os.environ.get("VGL_TEST_PEER_DEVICE_PER_RANK") == "1"
    ctx = api.Context(int(os.environ.get("LOCAL_RANK", rank)) if per_rank_device else 0)
    if transport == "hosted":
        comm = vs.Comm.hosted(ctx, rank, world, token, slot_bytes=1 << 16)          # small slots: every large payload goes in pieces
    elif transport == "peer":
        comm = vs.Comm.peer(ctx, rank, world, token, window_bytes=int(os.environ.get("VGL_TEST_PEER_WINDOW", 1 << 16)))      # small windows: pieces
    else:
        if rank == 0:
            with open(token + ".tmp", "wb") as f:
                f.write(vs.Comm.unique_id())
            os.replace(token + ".tmp", token)
        t0 = time.time()
        while not os.path.exists(token):
            assert time.time() - t0 < 120, "rank 0's RCCL id did not appear"
            time.sleep(0.05)
        comm = vs.Comm.rccl(ctx, rank, world, open(token, "rb").read())
    try:
        run(ctx, comm, transport, rank, world)
    except BaseException:
        comm.abort()                    # the other ranks' next barrier fails at once (hosted / peer), not after its timeout
        raise


def run(ctx, comm, transport, rank, world):
    check_exchanges(ctx, comm)
    if os.environ.get("VGL_TEST_FAIL_RANK") == str(rank):
        raise RuntimeError("rank %d fails on purpose (VGL_TEST_FAIL_RANK)" % rank)

    scale, ef, seed = 13, 16, 11
    V, E = 1 << scale, (1 << scale) * ef
    src, dst = ctx.gen_rmat(scale, ef, seed)
    g = api.Graph.from_coo(ctx, V, src, dst, want_perm=True, renumber="total")
    w = ctx.gather_u32(g.perm, ctx.gen_weights(E, seed))
    deg = (g.out_rowptr[1:] - g.out_rowptr[:-1])
    source = int(torch.argmax(deg))
    far = int(torch.nonzero(deg == 1)[0]) if bool((deg == 1).any()) else source
    ref = {}
    for s in (source, far):
        ref["td", s], _ = api.bfs(g, s, api.BFS_TOP_DOWN, raw=True)
        do, _ = api.bfs(g, s, api.BFS_DIRECTION_OPT, raw=True)
        assert torch.equal(do, ref["td", s])
    d_ref, _ = api.sssp(g, w, source, api.SSSP_ALL_ACTIVE, raw=True)
    wd_ref, _ = api.sswp(g, w, source, raw=True)
    comp_ref, _ = api.connected_components(g, raw=True)
    pr_ref, _ = api.page_rank(g, 5, raw=True, mode=api.PR_EXACT_ORDER)
    prb_ref, _ = api.page_rank(g, 5, raw=True, mode=api.PR_BLOCKED)
    auth_ref, hub_ref = api.hits(g, 3, raw=True)

    equal = [p * (V // world) for p in range(world + 1)]
    balanced = ctx.partition_rows(g.out_rowptr, world)
    placements = [("equal", equal)] + ([("edge-balanced", balanced)] if balanced != equal and all(b % 64 == 0 for b in balanced[:-1]) else [])
    for name, bounds in placements:
        lo, hi = bounds[rank], bounds[rank + 1]
        shard = g.shard(lo, hi)
        e_lo, e_hi = g.out_edge_range(lo, hi)
        ws = w[e_lo:e_hi].clone()
        for cap in ("4096", "0", "8", "emit0"):           # id-list exchange of the tiny levels: default bound, off, overflowing;
            if cap == "emit0":                            # ... and every top-down level reading its candidate bitmap off `levels`
                cap = "4096"
                os.environ["VGL_SHARD_TD_EMIT_EDGES"] = "0"
            os.environ["VGL_SHARD_SPARSE_CAP"] = cap
            for s in (source, far):
                for mode in (api.BFS_DIRECTION_OPT, api.BFS_TOP_DOWN):
                    levels, st = vs.bfs_run_sharded(shard, comm, s, mode, global_edges=E, gather_levels=True)
                    assert torch.equal(levels, ref["td", s]), (name, cap, s, mode)
                    assert mode == api.BFS_DIRECTION_OPT or st["bu_steps"] == 0
                levels, _ = vs.bfs_run_sharded(shard, comm, s, api.BFS_DIRECTION_OPT, global_edges=E, gather_levels=False, want_stats=False)
                assert torch.equal(levels[lo:hi], ref["td", s][lo:hi]), (name, cap, s, "owned only")
        os.environ.pop("VGL_SHARD_SPARSE_CAP")
        os.environ.pop("VGL_SHARD_TD_EMIT_EDGES", None)
        d, st = vs.sssp_run_sharded(shard, comm, ws, source)
        assert torch.equal(bits(d), bits(d_ref)), name
        ex = comm.stats()
        assert world == 1 and os.environ.get("VGL_SHARD_FORCE_COLLECTIVES") != "1" or ex["collectives"] > 0, ex
        wd, _ = vs.sswp_run_sharded(shard, comm, ws, source)
        assert torch.equal(bits(wd), bits(wd_ref)), name
        comp, _ = vs.cc_run_sharded(shard, comm)
        assert torch.equal(comp, comp_ref), name
        ranks, _ = vs.pr_run_sharded(shard, comm, 5, api.PR_EXACT_ORDER)
        assert torch.equal(bits(ranks), bits(pr_ref)), name
        ranks, _ = vs.pr_run_sharded(shard, comm, 5, api.PR_AUTO)          # small graph: AUTO resolves to the ordered chain on every rank
        assert torch.equal(bits(ranks), bits(pr_ref)), name
        ranks, _ = vs.pr_run_sharded(shard, comm, 5, api.PR_BLOCKED)       # exact sums: independent of the cut into shards
        assert torch.equal(bits(ranks), bits(prb_ref)), name
        # HITS: the sum of squares is folded rank by rank (an ulp on the norm): 1e-12 relative, the bar of the single-GPU tests
        auth, hub = vs.hits_run_sharded(shard, comm, 3)
        for got, want in ((auth, auth_ref), (hub, hub_ref)):
            err = ((got - want).abs() / want.abs().clamp(min=1e-300)).max().item()
            assert err <= 1e-12, (name, err)
        shard.close()
    comm.barrier()
    comm.close()
    ctx.sync()
    print("SHARDED_RANK_OK rank %d of %d over %s (%s)" % (rank, world, transport, ", ".join(n for n, _ in placements)), flush=True)


if __name__ == "__main__":
    main()
