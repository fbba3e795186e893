"""VGL graph files (.csr / .vcsr).  CPU part: the numpy restatement (oracle/graph_files.py) against
the files the reference wrote itself (tests/golden/rmat_s6_e8_seed1.{csr,vcsr}; sha256 of the larger ones in
tests/golden/graph_files.json, made by oracle/make_golden_graph_files.py).  Device part: the C++ reader / writer
(VGL_Graph::load_from_binary_file / save_to_binary_file in hip/vgl_hip.hpp) through apps/create_vgl_graphs.cpp and the
-load option of the apps: written files are the reference's byte for byte, loaded files give the golden results."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "apps", "bin")
SMALL = "rmat_s6_e8_seed1"
DIGESTS = json.load(open(os.path.join(GOLD, "graph_files.json")))


def case_edges(O, name):
    d = DIGESTS[name]
    return 1 << d["scale"], (O.gen_rmat if d["kind"] == "rmat" else O.gen_uniform)(d["scale"], d["edge_factor"], d["seed"])


def read_el_container(path):
    raw = open(path, "rb").read()
    V, E = int(np.frombuffer(raw, np.int32, 1, 0)[0]), int(np.frombuffer(raw, np.int64, 1, 4)[0])
    assert int(np.frombuffer(raw, np.int32, 1, 12)[0]) == 4
    return V, np.frombuffer(raw, np.int32, E, 16), np.frombuffer(raw, np.int32, E, 16 + 4 * E)


@pytest.mark.parametrize("ext", ["csr", "vcsr"])
def test_restatement_writes_the_reference_files(ext, oracle):
    from oracle import graph_files as GF
    fmt = GF.CSR_GRAPH if ext == "csr" else GF.VECTOR_CSR_GRAPH
    V, src, dst = read_el_container(os.path.join(GOLD, SMALL + ".el_container"))
    gs, gd = case_edges(oracle, SMALL)[1]
    assert np.array_equal(src, gs) and np.array_equal(dst, gd)          # the committed edge list is the fixture's
    assert GF.graph_file_bytes(GF.build_graph(V, src, dst, fmt)) == open(os.path.join(GOLD, SMALL + "." + ext), "rb").read()
    for name, d in DIGESTS.items():
        V, (src, dst) = case_edges(oracle, name)
        raw = GF.graph_file_bytes(GF.build_graph(V, src, dst, fmt))
        assert len(raw) == d[ext]["bytes"] and hashlib.sha256(raw).hexdigest() == d[ext]["sha256"], name


@pytest.mark.parametrize("ext", ["csr", "vcsr"])
def test_reference_file_contents(ext, oracle):
    """what the reference's file holds, read back: both containers describe the fixture's graph, the reorder indexes lead from CSR
    positions to input edges (outgoing) and to outgoing positions (incoming), conversions are inverse permutations sorted by degree"""
    from oracle import graph_files as GF
    O = oracle
    g = GF.read_graph_file(os.path.join(GOLD, SMALL + "." + ext))
    V, (src, dst) = case_edges(O, SMALL)
    assert (g["V"], g["E"]) == (V, len(src))
    s, d = GF.edges_in_original_ids(g, "out")
    perm = g["out"]["perm"]
    assert np.array_equal(s, src[perm]) and np.array_equal(d, dst[perm])
    ins, ind = GF.edges_in_original_ids(g, "in")                      # incoming rows are destinations
    assert np.array_equal(ins, d[g["in"]["perm"]]) and np.array_equal(ind, s[g["in"]["perm"]])
    if ext == "csr":
        rowptr, adj, operm = O.coo_to_csr(V, src, dst)
        assert np.array_equal(g["out"]["rowptr"], rowptr) and np.array_equal(g["out"]["adj"], adj) and np.array_equal(perm, operm)
        gold = np.load(os.path.join(GOLD, SMALL + ".npz"))
        assert O.fnv1a64(g["out"]["rowptr"]) == int(gold["pin_rowptr"]) and O.fnv1a64(g["out"]["adj"]) == int(gold["pin_adj"])
    else:
        for direction in ("out", "in"):
            c = g[direction]
            assert np.array_equal(c["bwd"][c["fwd"]], np.arange(V))
            deg = np.diff(c["rowptr"])
            assert (deg[:-1] >= deg[1:]).all()
    with pytest.raises(ValueError):
        bad = os.path.join(GOLD, SMALL + ".el_container")
        GF.read_graph_file(bad)


def tool(args, timeout=300):
    out = subprocess.run([os.path.join(BIN, "create_vgl_graphs_hip")] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("ext", ["csr", "vcsr"])
def test_device_writer_matches_reference_bytes(ext, tmp_path, oracle, ctx):
    O = oracle
    tool(["-import", os.path.join(GOLD, SMALL + ".el_container"), "-format", ext, "-file", tmp_path / "small"])
    assert open(tmp_path / ("small." + ext), "rb").read() == open(os.path.join(GOLD, SMALL + "." + ext), "rb").read()
    for name, d in DIGESTS.items():
        if name == SMALL:
            continue
        V, (src, dst) = case_edges(O, name)
        el = str(tmp_path / (name + ".el_container"))
        O.write_el_container(el, V, src, dst)
        tool(["-import", el, "-format", ext, "-file", tmp_path / name])
        raw = open(tmp_path / (name + "." + ext), "rb").read()
        assert len(raw) == d[ext]["bytes"] and hashlib.sha256(raw).hexdigest() == d[ext]["sha256"], name
        # load + save round trip of the file just written (for .vcsr: one numbering in memory, two in the file)
        tool(["-load", tmp_path / (name + "." + ext), "-file", tmp_path / (name + ".again")])
        assert open(tmp_path / (name + ".again." + ext), "rb").read() == raw


@pytest.mark.gpu
@pytest.mark.parametrize("ext", ["csr", "vcsr"])
@pytest.mark.parametrize("mode", [[], ["-fused", "-do"]], ids=["operator_api", "fused_do"])
def test_apps_load_reference_files(ext, mode, tmp_path, oracle, ctx):
    gold = np.load(os.path.join(GOLD, SMALL + ".npz"))
    path = os.path.join(GOLD, SMALL + "." + ext)
    dump = str(tmp_path / "levels.bin")
    # the -format given on the command line is overridden by the file's container type, as in the reference
    out = subprocess.run([os.path.join(BIN, "bfs_hip"), "-load", path, "-format", "csr", "-source", str(int(gold["source"])), "-check", "-dump", dump] + mode,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "error count: 0" in out.stdout, out.stdout + out.stderr
    assert np.array_equal(np.fromfile(dump, np.int32), gold["levels"])
    dump = str(tmp_path / "ranks.bin")
    out = subprocess.run([os.path.join(BIN, "pr_hip"), "-load", path, "-it", str(int(gold["pr_iters"])), "-dump", dump, "-fused"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    ranks = np.fromfile(dump, np.float32)
    ref = gold["pr_seq_csr"]
    assert np.abs(ranks - ref).max() <= 1e-6 * np.abs(ref).max()


@pytest.mark.gpu
def test_device_reader_rejects_bad_files(tmp_path, ctx):
    raw = bytearray(open(os.path.join(GOLD, SMALL + ".csr"), "rb").read())
    cases = {"truncated": bytes(raw[: len(raw) // 2]), "not_a_graph": open(os.path.join(GOLD, SMALL + ".el_container"), "rb").read()}
    bad = bytearray(raw)
    bad[16 + 16 + 8 * 65: 16 + 16 + 8 * 65 + 4] = np.int32(1 << 20).tobytes()      # first adjacency id far outside [0, V)
    cases["id_out_of_range"] = bytes(bad)
    for name, data in cases.items():
        p = tmp_path / (name + ".csr")
        open(p, "wb").write(data)
        out = subprocess.run([os.path.join(BIN, "bfs_hip"), "-load", str(p), "-source", "0"], capture_output=True, text=True, timeout=300)
        assert out.returncode == 1 and ("Error" in out.stdout), (name, out.stdout, out.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("ext", ["csr", "vcsr"])
def test_edge_properties_follow_loaded_files(ext, tmp_path, oracle, ctx):
    """edge properties are generated per INPUT edge and carried into the CSR orders by the edge reorder indexes, so a graph loaded
    from a file must give the same SSSP / SSWP results as the same edge list imported directly -- only if the indexes read from the
    file (and, for .vcsr, re-expressed in the stored numbering) are right.  Also pinned to the golden distances of the fixture."""
    gold = np.load(os.path.join(GOLD, SMALL + ".npz"))
    source = str(int(gold["source"]))
    for app, dtype, key in (("sssp", np.float32, "dist"), ("sswp", np.float32, "width")):
        outs = []
        for how in (["-load", os.path.join(GOLD, SMALL + "." + ext)], ["-import", os.path.join(GOLD, SMALL + ".el_container"), "-format", ext]):
            dump = str(tmp_path / ("%s_%d.bin" % (app, len(outs))))
            out = subprocess.run([os.path.join(BIN, app + "_hip")] + how + ["-seed", "1", "-source", source, "-check", "-dump", dump],
                                 capture_output=True, text=True, timeout=300)
            assert out.returncode == 0 and "error count: 0" in out.stdout, out.stdout + out.stderr
            outs.append(np.fromfile(dump, dtype))
        assert np.array_equal(outs[0].view(np.int32), outs[1].view(np.int32)), (app, ext)
        assert np.array_equal(outs[0].view(np.int32), gold[key].view(np.int32)), (app, ext, "golden")
