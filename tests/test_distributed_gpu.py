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


def test_bench_single_gpu_line_contract(ctx):
    """the one-GPU bench line at a small scale (RMAT-18, CPU baseline on, SSSP leg on): one JSON line on stdout with the contract's keys,
    the roofline object of the dominant kernel measured with HIP events, the CPU port timed beside it, and the built-in verification"""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scale", "18", "--steps", "4", "--warmup", "1", "--cpu-sources", "2",
                          "--no-pr-cc"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [x for x in out.stdout.strip().splitlines() if x.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 4 and line["warmup"] == 1 and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert line["value"] > 0 and line["ms_per_step"] > 0 and "workload" in line["config"] and line["data"].startswith("synthetic")
    r = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-4
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    v = line["verified"]
    assert v["bfs_do_equals_top_down"] and v["bfs_equals_cpu_oracle"] and v["sssp_equals_cpu_oracle"]
    td = line["bfs_top_down_reference_algorithm"]
    assert td["ms"] > 0 and td["blocked_levels"]["ms"] > 0


def test_bench_sharded_legs_through_the_library_rccl_communicator(ctx):
    """bench.py's N > 1 form as the driver launches it (torch.distributed.run, 127.0.0.1) with ONE rank and VGL_SHARD_FORCE_COLLECTIVES=1: the
    library's RCCL communicator is created from the id rank 0 broadcasts through the torch group, and every collective of the BFS, PageRank
    and Shiloach-Vishkin legs is issued by libvgl_hip.so (vgl_hip_*_run_sharded) on the context's stream"""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", VGL_SHARD_FORCE_COLLECTIVES="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29741",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-sharded", "--scale", "15", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-sssp",
           "--pr-scale", "16", "--cc-scale", "15", "--chunk-edges", str(1 << 18)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = json.loads([x for x in out.stdout.strip().splitlines() if x.startswith("{")][-1])
    assert line["value"] > 0 and line["verified_sharded_levels_consistent_over_owned_edges"] is True
    assert line["bfs"]["exchange_last_traversal"]["collectives"] > 0                       # the RCCL path really ran
    assert line["pagerank_uniform16_sharded"]["collectives"] > 0 and abs(line["pagerank_uniform16_sharded"]["ranks_sum"] - 1.0) < 1e-3
    assert line["cc_rmat_symmetrised_sharded"]["labels_idempotent"] and line["cc_rmat_symmetrised_sharded"]["exchange"]["collectives"] > 0


def test_bench_two_ranks_sharing_the_gpu_over_the_peer_transport(ctx):
    """bench.py's N = 2 form as the driver launches it, rehearsed on ONE card (VGL_BENCH_SHARE_GPU=1: both ranks on cuda:0, the launcher's
    group is gloo): the data path of every leg -- weak-scaling BFS, PageRank, Shiloach-Vishkin -- goes through the PEER transport (device
    windows mapped through hipIpc), chosen by the bench's own self-test; the certificate and the PageRank residual are checked in the run."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", VGL_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29743",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "15", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-sssp",
           "--pr-scale", "16", "--cc-scale", "15", "--chunk-edges", str(1 << 18)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = json.loads([x for x in out.stdout.strip().splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["exchange_transport"] == "peer"
    assert line["value"] > 0 and line["verified_sharded_levels_consistent_over_owned_edges"] is True
    assert line["bfs"]["exchange_last_traversal"]["collectives"] > 0 and line["bfs"]["exchange_last_traversal"]["exchanges"] > 0
    assert abs(line["pagerank_uniform16_sharded"]["ranks_sum"] - 1.0) < 1e-3
    assert line["cc_rmat_symmetrised_sharded"]["labels_idempotent"]
