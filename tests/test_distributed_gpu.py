"""The N > 1 super-step drivers against a REAL RCCL process group on the one-GPU box: one rank, collectives forced on
(VGL_SHARD_FORCE_COLLECTIVES=1), launched the way the driver launches bench.py (python -m torch.distributed.run, 127.0.0.1).
World-size-2 semantics are covered on CPU with gloo (tests/test_distributed_cpu.py); this test covers what gloo cannot: that
every collective / dtype / slicing pattern the drivers use is accepted by the nccl backend with device tensors."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_drivers_through_rccl_one_rank(ctx):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29731",
           os.path.join(ROOT, "tests", "rccl_one_rank.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "RCCL_ONE_RANK_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def test_bench_sharded_path_one_rank(ctx):
    """bench.py's multi-GPU branch (streaming shard build, dealt placement, equal ranges) with one rank at a small scale"""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-sharded", "--scale", "16", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline", "--no-sssp", "--no-pr-cc"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["scaling"] == "weak"
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["launches"] > 0 and 0 < r["frac"] < 1 and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-4


def test_bench_sharded_pagerank_and_cc_legs_one_rank(ctx):
    """the PageRank (BASELINE configs[3]) and Shiloach-Vishkin (configs[4]) legs of bench.py's multi-GPU branch with one rank at small scales:
    streaming shard builds (uniform / symmetrised RMAT, outgoing lists only), super-step drivers, result checks and rank-0 rooflines"""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-sharded", "--scale", "15", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-sssp", "--pr-scale", "16", "--cc-scale", "15", "--chunk-edges", str(1 << 18)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    pr, cc = line["pagerank_uniform16_sharded"], line["cc_rmat_symmetrised_sharded"]
    assert abs(pr["ranks_sum"] - 1.0) < 1e-3 and pr["teps"] > 0 and pr["shard_edges"] == (1 << 16) * 32
    assert cc["labels_idempotent"] and cc["hook_passes"] >= 1 and cc["shard_edges"] == cc["stored_edges"] == (1 << 15) * 32
