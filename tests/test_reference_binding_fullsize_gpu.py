"""The reference's OWN applications, HIP backend bound in, at the sizes BASELINE.json names: apps/{bfs,sssp,pr,cc}/*.cpp of the
reference on its own containers (oracle/_ref/vgl_hip_*, built by `make -C oracle binding` where /root/reference exists), `-format csr` AND `-format
vcsr`, each run with the reference's own `-check` -- its sequential implementation recomputes on the host from the same containers and
verify_results / verify_ranking_results / equal_components compare:

    bfs                 RMAT-24 x 32     BFS::vgl_top_down  vs  BFS::seq_top_down
    sssp -all-active    RMAT-24 x 32     ShortestPaths::vgl_dijkstra (gpu_shortest_paths.hpp, all-active push)  vs  seq_dijkstra
    pr                  uniform-25 x 32  PageRank::vgl_page_rank (gpu_pr.hpp)  vs  seq_page_rank
    cc                  RMAT-24 x 16, symmetrised   ConnectedComponents::vgl_shiloach_vishkin (gpu_shiloach_vishkin.hpp)  vs  seq_bfs_based

The graphs come from graph FILES (the reference's `-load`, vgl_runtime.hpp:52-60): apps/bin/create_vgl_graphs_hip writes the reference's own
`.csr` / `.vcsr` layouts on the device in seconds (byte-identical to the reference's writer: tests/test_graph_files.py), where the reference's host-side
generator + import would take minutes per graph.  Device memory is the default: user arrays are shadowed (integration/.../shadow_memory.h), nothing
is switched by the environment.  What a run costs here is the reference's sequential checker (seq_dijkstra ~1 min, seq_page_rank ~30 s per
iteration at these sizes), so every app runs one round."""
import os
import re
import subprocess
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CREATE = os.path.join(ROOT, "apps", "bin", "create_vgl_graphs_hip")


def exe(app):
    path = os.path.join(ROOT, "oracle", "_ref", "vgl_hip_" + app)
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/vgl_hip_* are built where /root/reference exists (make -C oracle binding)")
    return path


def graph_file(directory, name, fmt, args):
    base = os.path.join(directory, name)
    out = subprocess.run([CREATE, *args, "-format", fmt, "-file", base], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "saved" in out.stdout, (out.stdout + out.stderr)[-2000:]
    return base + "." + fmt


def run(app, *args):
    out = subprocess.run([exe(app), *args], capture_output=True, text=True, timeout=800)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-3000:]
    assert "rror in" not in text and "NOT equal" not in text and "not found" not in text, text[-3000:]
    m = re.search(r"AVG_PERF: ([\d.e+]+) MTEPS", text)
    assert m, text[-3000:]
    return text, float(m.group(1))


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
def test_reference_bfs_and_sssp_apps_at_rmat24(fmt):
    exe("bfs")
    with tempfile.TemporaryDirectory(prefix="vgl_fullsize_") as d:
        path = graph_file(d, "syn_rmat_24_32", fmt, ["-s", "24", "-e", "32", "-type", "rmat"])
        text, perf = run("bfs", "-load", path, "-format", fmt, "-check", "-it", "1")
        assert len(re.findall(r"error count: 0\b", text)) == 1, text[-3000:]
        print("reference bfs app, RMAT-24, %s: %.0f MTEPS" % (fmt, perf))
        text, perf = run("sssp", "-load", path, "-format", fmt, "-check", "-it", "1", "-all-active")
        assert len(re.findall(r"error count: 0\b", text)) == 1, text[-3000:]
        print("reference sssp app (all-active push), RMAT-24, %s: %.0f MTEPS" % (fmt, perf))


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
def test_reference_pr_app_at_uniform25(fmt):
    exe("pr")
    with tempfile.TemporaryDirectory(prefix="vgl_fullsize_") as d:
        path = graph_file(d, "syn_ru_25_32", fmt, ["-s", "25", "-e", "32", "-type", "ru"])
        text, perf = run("pr", "-load", path, "-format", fmt, "-check", "-it", "1")
        assert len(re.findall(r"error count: 0\b", text)) == 1, text[-3000:]
        print("reference pr app, uniform-25, %s: %.0f MTEPS" % (fmt, perf))


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
def test_reference_cc_app_at_rmat24x16_symmetrised(fmt):
    exe("cc")
    with tempfile.TemporaryDirectory(prefix="vgl_fullsize_") as d:
        path = graph_file(d, "syn_rmat_24_16_undirected", fmt, ["-s", "24", "-e", "16", "-type", "rmat", "-undirected"])
        text, perf = run("cc", "-load", path, "-format", fmt, "-check")
        assert len(re.findall(r"error count: 0\b", text)) == 1, text[-3000:]
        print("reference cc app, RMAT-24 x 16 symmetrised, %s: %.0f MTEPS" % (fmt, perf))
