import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import oracle as O
from vectorgraphlibrary_amd import api
from vectorgraphlibrary_amd.distributed import HipShardOps
ctx = api.Context(0)
scale, ef, seed = 12, 16, 21
V=1<<scale
src,dst = ctx.gen_rmat(scale, ef, seed)
g = api.Graph.from_coo(ctx, V, src, dst, want_perm=True)
hs,hd = O.gen_rmat(scale, ef, seed)
rowptr, adj, perm = O.coo_to_csr(V, hs, hd)
P=3
bounds = ctx.partition_rows(g.out_rowptr, P); print(bounds)
ops=[HipShardOps(g.shard(bounds[p],bounds[p+1])) for p in range(P)]
indeg = torch.from_numpy(O.indegree_noloops(rowptr, adj)).to(ctx.device)
ranks, rdeg, contrib = ops[0].new_f32(), ops[0].new_f32(), ops[0].new_f32()
ops[0].pr_setup(indeg, ranks, rdeg)
ref1 = O.pagerank(rowptr, adj, 1, 1)
new = [ranks.clone() for _ in range(P)]
for p,o in enumerate(ops):
    o.pr_iteration(indeg, rdeg, new[p], contrib)
    ctx.sync()
    a = new[p].cpu().numpy()
    lo,hi = bounds[p],bounds[p+1]
    print(p, 'owned equal', (a[lo:hi].view(np.int32)==ref1[lo:hi].view(np.int32)).mean(), 'changed outside', (a[:lo]!=np.float32(1/V)).sum(), (a[hi:]!=np.float32(1/V)).sum(), a[lo:lo+4], ref1[lo:lo+4])
