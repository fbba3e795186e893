"""Runnable drop-in proof: oracle/_ref/dropin_hip is oracle/dropin_driver.cpp -- a driver written here around the REFERENCE'S OWN algorithm
headers (algorithms/bfs/bfs.hpp, sswp/widest_paths.h, hits/hits.h, included unchanged from /root/reference when the binary was built in the
CPU container; `make -C oracle dropin`) -- compiled by hipcc for gfx950 against this repository's operator class
(vectorgraphlibrary_amd/hip/vgl_hip.hpp) and libvgl_hip.so.  Here it RUNS on the MI355X: the reference's lambdas execute through
GraphAbstractionsHIP::scatter / gather / compute / reduce / generate_new_frontier, and its results are compared with the CPU oracle on
the same seeded inputs, in both storage formats.  (tests/test_dropin_compile.py is the compile-only half of the proof; the reference
itself is not on the GPU box, the prebuilt binary is.)"""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "dropin_hip")
CASES = [("rmat", 12, 16, 3), ("ru", 11, 8, 5)]


def run(tmp_path, algo, kind, scale, ef, seed, arg, fmt):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/dropin_hip is built where /root/reference exists (make -C oracle dropin)")
    dump = str(tmp_path / "dropin.bin")
    out = subprocess.run([EXE, algo, kind, str(scale), str(ef), str(seed), str(arg), fmt, dump], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "DROPIN " + algo in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    return dump


def graph(O, kind, scale, ef, seed):
    src, dst = (O.gen_rmat if kind == "rmat" else O.gen_uniform)(scale, ef, seed)
    rowptr, adj, perm = O.coo_to_csr(1 << scale, src, dst)
    return src, rowptr, adj, perm


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
@pytest.mark.parametrize("kind,scale,ef,seed", CASES)
def test_reference_bfs_header_runs_on_the_backend(kind, scale, ef, seed, fmt, tmp_path, oracle, ctx):
    O = oracle
    src, rowptr, adj, perm = graph(O, kind, scale, ef, seed)
    source = O.pick_source(rowptr, seed)
    dump = run(tmp_path, "bfs", kind, scale, ef, seed, source, fmt)
    ref, _ = O.bfs_top_down(rowptr, adj, source)
    assert (np.fromfile(dump, np.int32) == ref).all()                  # BFS::vgl_top_down of algorithms/bfs/bfs.hpp, levels exact


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
@pytest.mark.parametrize("kind,scale,ef,seed", CASES)
def test_reference_sswp_header_runs_on_the_backend(kind, scale, ef, seed, fmt, tmp_path, oracle, ctx):
    O = oracle
    src, rowptr, adj, perm = graph(O, kind, scale, ef, seed)
    source = O.pick_source(rowptr, seed)
    dump = run(tmp_path, "sswp", kind, scale, ef, seed, source, fmt)
    ref, _ = O.sswp_bellman_ford(rowptr, adj, O.gen_weights(len(src), seed)[perm], source)
    assert (np.fromfile(dump, np.float32).view(np.int32) == ref.view(np.int32)).all()     # SSWP::vgl_dijkstra: only min / max of the inputs


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
@pytest.mark.parametrize("kind,scale,ef,seed", CASES)
def test_reference_hits_header_runs_on_the_backend(kind, scale, ef, seed, fmt, tmp_path, oracle, ctx):
    O = oracle
    src, rowptr, adj, perm = graph(O, kind, scale, ef, seed)
    dump = run(tmp_path, "hits", kind, scale, ef, seed, 4, fmt)
    auth, hub = O.hits(rowptr, adj, 4)
    got = np.fromfile(dump, np.float64)
    V = len(rowptr) - 1
    # HITS::vgl_hits accumulates with VGL_SRC_ID_ADD (atomics: the order of the f64 additions is not fixed); the reference's own runs
    # differ from its sequential checker in the last bits too
    assert np.max(np.abs(got[:V] - auth) / np.maximum(np.abs(auth), 1e-300)) <= 1e-9
    assert np.max(np.abs(got[V:] - hub) / np.maximum(np.abs(hub), 1e-300)) <= 1e-9


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
@pytest.mark.parametrize("kind,scale,ef,seed", CASES)
def test_reference_scc_header_runs_on_the_backend(kind, scale, ef, seed, fmt, tmp_path, oracle, ctx):
    O = oracle
    src, rowptr, adj, perm = graph(O, kind, scale, ef, seed)
    dump = run(tmp_path, "scc", kind, scale, ef, seed, 0, fmt)
    got = np.fromfile(dump, np.int32)                                  # SCC::vgl_forward_backward: tree ids, one per component
    # the partition must be Tarjan's: relabel every class by its smallest member, as the oracle does
    order = np.argsort(got, kind="stable")
    first = np.ones(len(got), bool)
    first[1:] = got[order][1:] != got[order][:-1]
    canon = np.empty(len(got), np.int64)
    starts = np.flatnonzero(first)
    ends = np.append(starts[1:], len(got))
    mins = np.minimum.reduceat(order, starts)
    canon[order] = np.repeat(mins, ends - starts)
    assert (canon == O.scc_tarjan(rowptr, adj)).all()
