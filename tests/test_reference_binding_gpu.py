"""The reference's OWN applications on the reference's OWN containers, run on the MI355X through the bound-in HIP backend class
(oracle/_ref/vgl_hip_{bfs,sswp,hits,scc,pr,sssp,cc,mf}: apps/<app>/<app>.cpp of the reference, compiled in the CPU container by `make -C oracle binding` with
-D __USE_HIP__ after integration/apply_hip_binding.py; tests/test_reference_binding.py is the build half).  Each run uses the reference's own
-check: its sequential implementation (BFS::seq_top_down, SSWP::seq_dijkstra, HITS::seq_hits, SCC::seq_tarjan) recomputes the result on the host
from the same containers and verify_results / verify_ranking_results / equal_components compare.  pr, sssp and cc run the reference's own GPU
variants of the algorithms (algorithms/pr/gpu_pr.hpp, sssp/gpu_shortest_paths.hpp, cc/gpu_shiloach_vishkin.hpp: device lambdas with atomics, written for
its CUDA backend), enabled for __USE_HIP__ by the binding's edits.  CSR_GRAPH exercises advance_worker(CSRGraph&,
FrontierCSR&); VECTOR_CSR_GRAPH exercises the three degree ranges with the vector-extension kernel and, for SSWP, weights addressed by VE-space
global_edge_pos in the collective range (EdgesArray_VectorCSR)."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_unchecked(app, *args, env=None):
    exe = os.path.join(ROOT, "oracle", "_ref", "vgl_hip_" + app)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/vgl_hip_* are built where /root/reference exists (make -C oracle binding)")
    out = subprocess.run([exe, *args], capture_output=True, text=True, timeout=600, env=env)
    text = out.stdout + out.stderr
    assert out.returncode in (0, 1) and "rror in" not in text, text[-3000:]
    return text


def run(app, *args, env=None):
    exe = os.path.join(ROOT, "oracle", "_ref", "vgl_hip_" + app)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/vgl_hip_* are built where /root/reference exists (make -C oracle binding)")
    out = subprocess.run([exe, *args], capture_output=True, text=True, timeout=600, env=env)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-3000:]
    assert "rror in" not in text and "NOT equal" not in text, text[-3000:]
    return text


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
@pytest.mark.parametrize("kind,scale,edges", [("rmat", 14, 16), ("ru", 13, 8), ("rmat", 6, 4), ("ru", 9, 1)])      # the last two: a few vector segments; mostly isolated vertices
def test_reference_bfs_app(kind, scale, edges, fmt):
    text = run("bfs", "-s", str(scale), "-e", str(edges), "-type", kind, "-format", fmt, "-check", "-it", "4")
    assert len(re.findall(r"error count: 0\b", text)) == 4, text[-3000:]          # one verify_results per round (apps/bfs/bfs.cpp:39-49)


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
def test_reference_apps_with_int_flags_for_every_frontier(fmt):
    """VGL_GNF_INT_FLAGS=1: generate_new_frontier writes the int32 flags of every result in its count pass (the reference's contract to the letter)
    instead of a bitmap + flags for DENSE / ALL_ACTIVE results only (round 5).  A dense uniform graph turns BFS frontiers DENSE on VECTOR_CSR_GRAPH
    (more than 0.7 V vertices on one level), so both ways of producing the flags of a frontier that is walked by them run here."""
    for env in (None, dict(os.environ, VGL_GNF_INT_FLAGS="1")):
        text = run("bfs", "-s", "12", "-e", "32", "-type", "ru", "-format", fmt, "-check", "-it", "3", env=env)
        assert len(re.findall(r"error count: 0\b", text)) == 3, text[-3000:]
        text = run("sswp", "-s", "11", "-e", "32", "-type", "ru", "-format", fmt, "-check", "-it", "1", env=env)
        assert len(re.findall(r"error count: 0\b", text)) == 1, text[-3000:]


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
@pytest.mark.parametrize("kind,scale,edges", [("rmat", 13, 16), ("ru", 12, 8), ("rmat", 7, 3)])
def test_reference_sswp_app(kind, scale, edges, fmt):
    text = run("sswp", "-s", str(scale), "-e", str(edges), "-type", kind, "-format", fmt, "-check", "-it", "2")
    assert len(re.findall(r"error count: 0\b", text)) == 1, text[-3000:]


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
def test_reference_hits_app(fmt):
    text = run("hits", "-s", "12", "-e", "16", "-type", "rmat", "-format", fmt, "-check", "-it", "5")
    assert len(re.findall(r"error count: 0\b", text)) == 2, text[-3000:]          # authorities and hubs (apps/hits/hits.cpp:43-51)


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
def test_reference_scc_app(fmt):
    """The app runs and checks itself.  On VECTOR_CSR_GRAPH its check passes; on CSR_GRAPH the reference's forward-backward algorithm itself differs
    from SCC::seq_tarjan by a few vertices on 1 - 8 % of small R-MAT graphs ON ANY BACKEND (its own multicore build included: found in round 5, when
    this test -- the app seeds its generator with time(NULL) -- failed once in 25 runs), so there the verdict of the run is not asserted here: the
    seeded comparison with the genuine reference below is the parity test."""
    text = run_unchecked("scc", "-s", "12", "-e", "8", "-type", "rmat", "-format", fmt, "-check")
    counts = re.findall(r"error count: (\d+)\b", text)
    assert len(counts) == 1, text[-3000:]
    if fmt == "vcsr":
        assert counts == ["0"], text[-3000:]
    else:
        assert int(counts[0]) <= 16, text[-3000:]             # (the reference's own misses are a handful of vertices)


def test_reference_scc_matches_the_reference_seed_for_seed():
    """integration/tests/scc_check.cpp (our seeded driver of the UNCHANGED SCC::vgl_forward_backward against SCC::seq_tarjan) on the HIP backend
    bound into the reference's tree, against tests/golden/scc_reference_disagreements.json = the same program on the genuine reference (multicore
    flavour, oracle/make_golden_scc_check.py): for every seed the number of vertices whose component differs from Tarjan's must be the reference's
    -- zero for most seeds, the reference's own 1 - 4 for the seeds its algorithm gets wrong."""
    import json
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "scc_reference_disagreements.json")))
    for case in golden["cases"]:
        text = run_unchecked("scc_check", str(case["scale"]), str(case["edge_factor"]), str(case["first_seed"]), str(case["seeds"]), "0", case["format"])
        got = {m.group(1): int(m.group(2)) for m in re.finditer(r"seed (\d+): (\d+) vertices disagree", text)}
        assert re.search(r"%d seeds, %d with components" % (case["seeds"], len(got)), text), text[-2000:]
        assert got == case["vertices_that_disagree_with_seq_tarjan"], (case["format"], case["scale"], got)


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
@pytest.mark.parametrize("kind", ["rmat", "ru"])
def test_reference_pr_app(kind, fmt):
    text = run("pr", "-s", "12", "-e", "16", "-type", kind, "-format", fmt, "-check", "-it", "5")      # PageRank::vgl_page_rank of gpu_pr.hpp (float atomics)
    assert len(re.findall(r"error count: 0\b", text)) == 1, text[-3000:]


@pytest.mark.parametrize("mode", [[], ["-all-active"], ["-pull"], ["-all-active", "-pull"]], ids=["partial_active_push", "all_active_push", "partial_active_pull", "all_active_pull"])
@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
def test_reference_sssp_app(fmt, mode):
    """ShortestPaths::vgl_dijkstra of gpu_shortest_paths.hpp in its frontier / traversal variants: weights through global_edge_pos -- on vcsr both
    the CSR and the vector-extension index space of EdgesArray_VectorCSR"""
    text = run("sssp", "-s", "12", "-e", "16", "-type", "rmat", "-format", fmt, "-check", "-it", "2", *mode)
    assert len(re.findall(r"error count: 0\b", text)) == 2, text[-3000:]


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
@pytest.mark.parametrize("kind", ["rmat", "ru"])
def test_reference_cc_app(kind, fmt):
    text = run("cc", "-s", "12", "-e", "16", "-type", kind, "-format", fmt, "-check")                  # ConnectedComponents::vgl_shiloach_vishkin of gpu_shiloach_vishkin.hpp
    assert len(re.findall(r"error count: 0\b", text)) == 1, text[-3000:]


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
@pytest.mark.parametrize("kind,scale,edges", [("rmat", 12, 8), ("ru", 11, 16)])
def test_reference_coloring_app(kind, scale, edges, fmt):
    """Coloring::vgl_coloring (algorithms/coloring/coloring.hpp, round 5: no longer left out of the bound tree): its bit helpers are callable from
    device code, and its scatter under enable_safe_stores() -- a plain read-modify-write of available_colors[src] per edge -- runs one lane per
    vertex in the HIP class (with edge tiles the clears of a row's lanes would overwrite each other).  verify_colors: no edge joins two vertices of
    one colour."""
    text = run("coloring", "-s", str(scale), "-e", str(edges), "-type", kind, "-format", fmt, "-check")
    assert len(re.findall(r"error count: 0\b", text)) == 1, text[-3000:]


def test_reference_mf_app():
    """MF::vgl_ford_fulkerson (algorithms/mf/mf.hpp) against MF::seq_ford_fulkerson on a dense uniform graph, CSR_GRAPH.  (Not on a sparse directed
    graph: the reference divides by its iteration count, which is zero when the random sink cannot be reached.  Not on VECTOR_CSR_GRAPH: the
    reference's own multicore build loops for ever there once a frontier turns dense -- the host-side flow updates go to the CSR copy of the
    edge array and a dense collective advance reads the vector-extension copy; this backend reproduces that.)
    Round 5: 1024 vertices instead of 64.  The app draws source and sink independently (apps/mf/mf.cpp:33-34) and vgl_ford_fulkerson never ends
    when they are the same vertex (mf_bfs finds the sink at once, the path is empty, mf.hpp:85-110) -- with 64 vertices and three rounds that is
    one run in 22, on any backend, and it stopped the GPU suite once (600 s).  A run that still draws such a pair is skipped, not failed."""
    exe = os.path.join(ROOT, "oracle", "_ref", "vgl_hip_mf")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/vgl_hip_* are built where /root/reference exists (make -C oracle binding)")
    try:
        out = subprocess.run([exe, "-s", "10", "-e", "64", "-type", "ru", "-format", "csr", "-check", "-it", "2"], capture_output=True, text=True, timeout=90)
    except subprocess.TimeoutExpired:
        pytest.skip("the reference's Ford-Fulkerson does not terminate when its random source equals its random sink (2 chances in 1024 per run)")
    text = out.stdout + out.stderr
    assert out.returncode == 0 and "rror in" not in text and "NOT equal" not in text, text[-3000:]
    assert len(re.findall(r"Results are equal", text)) == 2, text[-3000:]


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
def test_user_arrays_live_in_hbm_and_cross_pcie_only_when_the_other_side_touches_them(fmt):
    """Residency (integration/vgl_compute_api/hip/shadow_memory.h): a VerticesArray is a pinned host mirror + a buffer in HBM with an owner flag.
    The reference's bfs app with -check, three rounds: `levels` goes up once per round (move_to_device() in bfs.hpp:71, after verify_results read it on
    the host; the first round uploads nothing: the array is fresh) and comes down once per round (verify_results); `check_levels` is written and read by
    host code only and never crosses.  Without -check nothing crosses at all."""
    env = dict(os.environ, VGL_HIP_SHADOW_STATS="1")
    text = run("bfs", "-s", "12", "-e", "16", "-type", "rmat", "-format", fmt, "-check", "-it", "3", env=env)
    assert len(re.findall(r"error count: 0\b", text)) == 3, text[-3000:]
    m = re.search(r"shadowed arrays: (\d+) uploads \(([\d.]+) MB\), (\d+) downloads \(([\d.]+) MB\)", text)
    assert m, text[-3000:]
    assert (int(m.group(1)), int(m.group(3))) == (2, 3), m.group(0)
    text = run("bfs", "-s", "12", "-e", "16", "-type", "rmat", "-format", fmt, "-it", "3", env=env)
    m = re.search(r"shadowed arrays: (\d+) uploads \(([\d.]+) MB\), (\d+) downloads", text)
    assert m and (int(m.group(1)), int(m.group(3))) == (0, 0), text[-3000:]


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
@pytest.mark.parametrize("kind,scale,edges", [("rmat", 12, 16), ("ru", 10, 4)])
def test_host_rewrites_of_a_generated_frontier_void_its_advance_plan(kind, scale, edges, fmt):
    """integration/tests/plan_stamp_check.cpp (our program against the patched tree): generate_new_frontier leaves the advance plan of a sparse
    frontier behind and stamps the container; clear / add_vertex / add_group_of_vertices / set_all_active void the stamp.  A scatter on a frontier
    that was generated and then rewritten by host code -- same size, other degrees -- must reach what it reaches on a frontier the backend never
    generated.  (The program also alternates host writes, kernels and host reads of four VerticesArrays: every case crosses the shadow both ways.)"""
    text = run("plan_stamp_check", "-s", str(scale), "-e", str(edges), "-type", kind, "-format", fmt)
    assert "PLAN STAMP CHECK PASSED" in text, text[-3000:]
    assert len(re.findall(r", 0 differences", text)) == 4, text[-3000:]


@pytest.mark.parametrize("kind,scale,edges", [("rmat", 11, 8), ("ru", 10, 2), ("rmat", 8, 4), ("rmat", 16, 16)])     # the last: 512 tiles of copy_if_indexes, two rounds of its scan
def test_reference_tc_unchanged_against_a_sequential_search(kind, scale, edges):
    """TransitiveClosure::vgl_purdoms (algorithms/tc/tc.hpp) bound in (round 5: tc.h is no longer left out of the tree): the edge filter writes two
    EdgesArrays through global_edge_pos, ParallelPrimitives::copy_if_indexes evaluates its DEVICE condition in kernels
    (vgl_compute_api/hip/parallel_primitives_hip.h), the condensed graph is an EDGES_LIST_GRAPH the class's edges-list workers traverse with the
    reference's own BFS::fast_vgl_top_down.  integration/tests/tc_check.cpp compares every answer with a sequential search over the same graph (the
    app's own -check compares zero elements, apps/tc/tc.cpp:67)."""
    text = run("tc_check", "-s", str(scale), "-e", str(edges), "-type", kind, "-format", "csr", "-it", "24")
    assert "TC CHECK PASSED" in text and ", 0 wrong answers" in text, text[-3000:]


@pytest.mark.parametrize("fmt", ["csr", "vcsr"])
def test_reference_tc_app_runs(fmt):
    text = run("tc", "-s", "10", "-e", "8", "-type", "rmat", "-format", fmt, "-it", "8", "-check")      # (its -check is vacuous, see above: this is "the app runs")
    assert "AVG_PERF" in text, text[-3000:]
