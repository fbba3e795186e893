"""Helper of tests/test_distributed_gpu.py (not a test module): ONE rank under torch.distributed.run with the nccl (= RCCL) backend
and VGL_SHARD_FORCE_COLLECTIVES=1, so that every collective call of the N > 1 super-step drivers (all_gather_into_tensor on bitmap
words, slices, (index, value) pair lists and rank slices, all_to_all_single, all_reduce MIN / MAX / SUM on i32 / f32 / i64) runs through RCCL on the one-GPU box, and the
results are compared with the single-GPU fused paths.  Prints 'RCCL_ONE_RANK_OK' on success."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vectorgraphlibrary_amd import api  # noqa: E402
import protocol_model as vd  # noqa: E402  (tests/protocol_model.py: the Python model + the package's shard builder)


def main():
    os.environ["VGL_SHARD_FORCE_COLLECTIVES"] = "1"
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert dist.get_world_size() == 1
    ctx = api.Context(local_rank)
    scale, ef, seed = 14, 16, 5
    V, E = 1 << scale, (1 << scale) * ef
    shard, degrees, bounds = vd.build_generated_shard(ctx, scale, ef, seed, 0, 1, kind="rmat", renumber="total", chunk_edges=1 << 16, placement="dealt")
    ops = vd.HipShardOps(shard)
    source = int(torch.argmax(degrees))
    ref, _ = api.bfs(shard, source, api.BFS_DIRECTION_OPT, raw=True)
    for kw in (dict(equal_ranges=True), dict(equal_ranges=True, two_phase=True), dict(equal_ranges=False), dict(degrees=None, edges=None),
               dict(equal_ranges=True, sparse_cap=0), dict(equal_ranges=True, sparse_cap=16), dict(sparse_cap=100000),
               dict(equal_ranges=True, owned_levels=True), dict(equal_ranges=True, owned_levels=True, sparse_cap=0, two_phase=True)):
        args = dict(degrees=degrees, edges=E)
        args.update(kw)
        levels, nlevels = vd.bfs_sharded(ops, source, **args)
        assert torch.equal(levels, ref), kw
    w = ctx.gen_weights(E, seed)[:shard.out_adj.numel()].contiguous()
    wops = vd.HipShardOps(shard, weights=w)
    d_ref, _ = api.sssp(shard, w, source, api.SSSP_ACTIVE_TILES, raw=True)
    wd_ref, _ = api.sswp(shard, w, source, raw=True)
    comp_ref, _ = api.connected_components(shard, raw=True)
    for dense_only in (False, True):                   # pair lists (all-gather of int32 lists of changing length) and whole-array all-reduce
        st = {}
        d, _ = vd.sssp_sharded(wops, source, stats=st, dense_only=dense_only)
        assert torch.equal(d.view(torch.int32), d_ref.view(torch.int32)), dense_only
        assert (st.get("list_steps", 0) > 0) == (not dense_only), st
        wd, _ = vd.sswp_sharded(wops, source, dense_only=dense_only)
        assert torch.equal(wd.view(torch.int32), wd_ref.view(torch.int32)), dense_only
        comp, _ = vd.cc_sharded(ops, dense_only=dense_only)
        assert torch.equal(comp, comp_ref), dense_only
    ranks = vd.page_rank_sharded(ops, 5, 0, V)
    ranks_ref, _ = api.page_rank(shard, 5, raw=True)
    assert torch.equal(ranks.view(torch.int32), ranks_ref.view(torch.int32))
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("RCCL_ONE_RANK_OK", flush=True)


if __name__ == "__main__":
    main()
