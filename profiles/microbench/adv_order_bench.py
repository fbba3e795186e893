#!/usr/bin/env python3
"""What would the operator API's advance cost if it walked the edges of a large level in (destination block, source block) order instead of
CSR order?  The user's operator (bfs.hpp:28-36: levels[src], levels[dst], one conditional store) over an explicit edge list of RMAT-24 x 32,
degree-sorted, at the level of a top-down traversal that holds most of the edges -- CSR order against pair order for block sizes 2^13 .. 2^16.
usage (GPU box): python profiles/microbench/adv_order_bench.py [scale=24]"""
import ctypes
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402
from vectorgraphlibrary_amd import api  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "adv_order_kernel.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "adv_order_kernel.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.run_edge_bfs.argtypes = [ctypes.c_longlong] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ef = 32
V, E = 1 << scale, (1 << scale) * ef
ctx = api.Context(0)
s, d = ctx.gen_rmat(scale, ef, 1)
g = api.Graph.from_coo(ctx, V, s, d, with_incoming=False, renumber="total")
del s, d
deg = (g.out_rowptr[1:] - g.out_rowptr[:-1])
src = torch.repeat_interleave(torch.arange(V, device=ctx.device, dtype=torch.int32), deg)
dst = g.out_adj.clone()
source = int(torch.nonzero(deg > 0)[12345])
lv, _ = api.bfs(g, source, api.BFS_TOP_DOWN, raw=True)
work = torch.zeros(int(lv.max()) + 1, dtype=torch.int64, device=ctx.device).index_add_(0, lv[lv > 0].long(), deg[lv > 0])
cur = int(torch.argmax(work))
print(f"RMAT-{scale}: level {cur} holds {int(work[cur])} of {E} edges ({int((lv == cur).sum())} frontier vertices)", flush=True)
flags = (lv == cur).to(torch.int32)
before = torch.where((lv > 0) & (lv <= cur), lv, torch.full_like(lv, -1))
want = torch.where((lv > 0) & (lv <= cur + 1), lv, torch.full_like(lv, -1))
stream = torch.cuda.current_stream().cuda_stream


def run(name, ssrc, sdst, ept):
    best = 1e9
    for rep in range(4):
        levels = before.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = lib.run_edge_bfs(E, ssrc.data_ptr(), sdst.data_ptr(), flags.data_ptr(), levels.data_ptr(), cur, ept, stream)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
        assert rc == 0 and torch.equal(levels, want), name
    print(f"{name:34s} EPT {ept}: {best * 1e3:7.3f} ms  ({E / best / 1e9:6.1f} G edges/s)", flush=True)


for ept in (4, 8):
    run("CSR order", src, dst, ept)
for bits in (16, 15, 14, 13):
    key = (dst >> bits).to(torch.int64) * (V >> bits) + (src >> bits).to(torch.int64)
    order = torch.sort(key, stable=True)[1]
    del key
    ssrc, sdst = src[order], dst[order]
    del order
    for ept in (4, 8):
        run(f"(dst, src) blocks of 2^{bits} ids", ssrc, sdst, ept)
    del ssrc, sdst
    torch.cuda.empty_cache()
print("ADV_ORDER_OK")
