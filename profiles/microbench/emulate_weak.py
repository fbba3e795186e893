"""single-GPU rehearsal of the N-rank weak-scaling bench: build all P shards of RMAT scale S with the streaming builder,
drive them in lock-step (exchange = concatenated bitmaps) with the same direction rule as protocol_model.bfs_sharded, and
compare direction-optimising vs top-down levels."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vectorgraphlibrary_amd import api
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import protocol_model as vd  # the Python protocol model (tests/protocol_model.py)

S, P = int(sys.argv[1]), int(sys.argv[2])
PLACEMENT = sys.argv[3] if len(sys.argv) > 3 else "dealt"
ef, seed = 32, 1
VERBOSE = len(sys.argv) > 4
OWNED = len(sys.argv) > 5 and sys.argv[5] == "owned"      # levels kept per owner (bfs_sharded(owned_levels=True))
ctx = api.Context(0)
V, E = 1 << S, (1 << S) * ef
shards = []
for p in range(P):
    t = time.time()
    s, degrees, bounds = vd.build_generated_shard(ctx, S, ef, seed, p, P, placement=PLACEMENT)
    ctx.sync()
    print(f"shard {p}: rows [{bounds[p]},{bounds[p+1]}) out-edges {s.E} in-edges {s.in_adj.numel()} build {time.time()-t:.1f}s "
          f"mem {torch.cuda.memory_allocated()/2**30:.1f} GiB", flush=True)
    shards.append(s)
ops = [vd.HipShardOps(s) for s in shards]
words = (V + 63) // 64
nz = torch.nonzero(degrees > 0).flatten()
src_list = [int(nz[i]) for i in torch.randint(0, nz.numel(), (3,), generator=torch.Generator().manual_seed(5))]


def run(source, direction_opt):
    reps = [o.new_i32() for o in ops]
    vis, fr = [o.new_words(1) for o in ops], [o.new_words(1) for o in ops]
    for o, r, v_, f_ in zip(ops, reps, vis, fr):
        o.bfs_init(r, source)
        o.levels_to_bitmap(r, 1, f_)
        v_.copy_(f_)
    everyone = ops[0].new_words(P)
    F, M = 1, int(degrees[source])
    prevF, visited_total, bottom_up = 0, 0, False
    factor = max(1, (E // V) // 2)
    level, trace = 1, []
    while True:
        visited_total += F
        if direction_opt:
            if not bottom_up:
                if F > prevF and M >= ((V - visited_total) * factor + V) // vd.ALPHA:
                    bottom_up = True
            elif F <= prevF and F < ((V - visited_total) * factor + V) // (factor * vd.BETA):
                bottom_up = False
        prevF = F
        tstep, tbm, tap = [], [], []
        for p, (o, r) in enumerate(zip(ops, reps)):
            mine = everyone[p * words:(p + 1) * words]
            ctx.sync(); t0 = time.perf_counter()
            if bottom_up:
                o.bfs_step_bu(r, level, vis[p], fr[p], mine)
                ctx.sync(); t1 = t2 = time.perf_counter()
            else:
                o.bfs_step_bits(r, level, vis[p], fr[p], mine)
                ctx.sync(); t1 = t2 = time.perf_counter()
            tstep.append(round((t1 - t0) * 1e3, 2)); tbm.append(round((t2 - t1) * 1e3, 2))
        res = []
        parts, bits = P, everyone
        if bottom_up and PLACEMENT == "dealt":
            lo = [b // 64 for b in bounds]
            parts, bits = 1, torch.cat([everyone[p * words + lo[p]:p * words + lo[p + 1]] for p in range(P)])
        for p, (o, r) in enumerate(zip(ops, reps)):
            ctx.sync(); t0 = time.perf_counter()
            res.append((o.apply_bitmaps_owned if OWNED else o.apply_bitmaps)(parts, bits, r, level + 1, vis[p], fr[p], degrees))
            ctx.sync(); tap.append(round((time.perf_counter() - t0) * 1e3, 2))
        if VERBOSE:
            print(f"  level {level} {'BU' if bottom_up else 'TD'} F={F}: step {tstep} bitmap {tbm} apply {tap}", flush=True)
        if OWNED:                                   # per-owner counts: what the ranks would all-reduce
            F, M = sum(x[0] for x in res), sum(x[1] for x in res)
        else:
            assert len(set(res)) == 1
            F, M = res[0]
        trace.append(("BU" if bottom_up else "TD", F))
        if F == 0:
            break
        level += 1
    if OWNED:                                       # assemble the owners' slices (what a final all-gather would return)
        full = reps[0].clone()
        for p in range(1, P):
            a, b = bounds[p], bounds[p + 1]
            full[a:b] = reps[p][a:b]
        return full, trace
    for r in reps[1:]:
        assert torch.equal(r, reps[0])
    return reps[0], trace


for s in src_list:
    t = time.time(); a, tr = run(s, True); ctx.sync(); ta = time.time() - t
    t = time.time(); b, _ = run(s, False); ctx.sync(); tb = time.time() - t
    ok = torch.equal(a, b)
    print(f"source {s}: DO {ta*1e3:.1f} ms (all {P} shards serially) TD {tb*1e3:.1f} ms equal={ok} reached={(a>0).sum().item()} trace={tr}", flush=True)
    assert ok
print("OK")
