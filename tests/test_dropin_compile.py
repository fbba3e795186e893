"""Drop-in compile proof (CPU container only: the reference stays at /root/reference, nothing is copied): the reference's own
algorithm sources are compiled UNCHANGED, as device-lambda code for gfx950, against this repository's operator-API header
(vectorgraphlibrary_amd/hip/vgl_hip.hpp) -- the header the reference's `VGL_GRAPH_ABSTRACTIONS` macro selects per architecture
(architecture_independent_api.h:3-43).  Every translation unit is generated here: the backend header, what the reference's umbrella
header would have put in scope before an algorithm header (`using namespace std`, the BFS level constants of
algorithms/bfs/change_state/change_state.h:21-23), the reference header, and one instantiation so that the kernels for its
lambdas are really emitted.

Headers that CANNOT be device code on any GPU backend, and why (checked by test_headers_that_cannot_compile):
  * algorithms/{pr/pr,sssp/shortest_paths,cc/shiloach_vishkin,cc/bfs_based,rw/random_walk}.hpp are wrapped in
    `#if defined(__USE_NEC_SX_AURORA__) || defined(__USE_MULTICORE__)`: their lambdas capture per-thread register arrays by
    reference inside `omp parallel` regions.  The reference ships separate gpu_*.hpp variants for them, which call the CUDA runtime by
    name (cudaMemset, cudaMallocManaged); this repository's counterparts are apps/algorithms/{pr,sssp,cc,rw}.hpp.
  * algorithms/coloring/coloring.hpp calls its host-only helpers clear_bit / smallest_bit_pos from device lambdas.
  * algorithms/tc/tc.hpp needs the EdgesListGraph container and the multicore copy helper openmp_reorder_gather_copy.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HIPCC = "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF) or not os.path.exists(HIPCC), reason="needs /root/reference and hipcc (CPU container)")

PRELUDE = f"""#include "{ROOT}/vectorgraphlibrary_amd/hip/vgl_hip.hpp"
#include <map>
#include <queue>
#include <set>
#include <stack>
using namespace std;                 // the reference's umbrella header does this globally
#define UNVISITED_VERTEX -1          // algorithms/bfs/change_state/change_state.h:21-23
#define FIRST_LEVEL_VERTEX 1
"""

# name -> (what follows the prelude, instantiation)
UNITS = {
    "bfs/bfs.hpp": ("""class BFS {                          // declaration of algorithms/bfs/bfs.h:31-46 (its other includes are the NEC-only direction-optimising pieces)
public:
    template <typename _T> static void fast_vgl_top_down(VGL_Graph &_graph, VerticesArray<_T> &_levels, int _source_vertex,
                                                         VGL_GRAPH_ABSTRACTIONS &_graph_API, VGL_FRONTIER &_frontier);
    template <typename _T> static double vgl_top_down(VGL_Graph &_graph, VerticesArray<_T> &_levels, int _source_vertex);
};
#include "/root/reference/algorithms/bfs/bfs.hpp"
""", "double use(VGL_Graph &g, VerticesArray<int> &a) { return BFS::vgl_top_down(g, a, 0); }"),
    "sswp/widest_paths.h": ('#include "/root/reference/algorithms/sswp/widest_paths.h"\n',
                            "double use(VGL_Graph &g, EdgesArray<float> &w, VerticesArray<float> &a) { return SSWP::vgl_dijkstra(g, w, a, 0); }"),
    "hits/hits.h": ('#include "/root/reference/algorithms/hits/hits.h"\n',
                    "void use(VGL_Graph &g, VerticesArray<double> &a, VerticesArray<double> &h) { HITS::vgl_hits(g, a, h, 3); }"),
    "scc/scc.h": ('#include "/root/reference/algorithms/scc/scc.h"\n',
                  "void use(VGL_Graph &g, VerticesArray<int> &c) { SCC::vgl_forward_backward(g, c); }"),
    "mf/mf.h": ('#include "/root/reference/algorithms/mf/mf.h"\n',
                "void use(VGL_Graph &g, EdgesArray<int> &f) { int flow = 0; MF::vgl_ford_fulkerson(g, f, 0, 1, flow); }"),
}


def compile_unit(tmp_path, name, body, use):
    src = tmp_path / (name.replace("/", "_").replace(".", "_") + ".cpp")
    src.write_text(PRELUDE + body + use + "\n")
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-x", "hip", "-c", str(src), "-o", str(src) + ".o"],
                         capture_output=True, text=True, timeout=600)
    return out, str(src) + ".o"


@pytest.mark.parametrize("name", sorted(UNITS))
def test_reference_algorithm_header_compiles_unchanged(name, tmp_path):
    body, use = UNITS[name]
    out, obj = compile_unit(tmp_path, name, body, use)
    assert out.returncode == 0, out.stderr[-3000:]
    # the lambdas became gfx950 kernels of this backend: the object holds device code with the templated advance / vertex kernels
    dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", obj], capture_output=True, text=True)
    assert "gfx950" in dis.stdout
    shutil.rmtree(tmp_path, ignore_errors=True)


@pytest.mark.parametrize("name,needle", [("coloring/coloring.h", "clear_bit"), ("tc/tc.h", "openmp_reorder_gather_copy")])
def test_headers_that_cannot_compile(name, needle, tmp_path):
    """the two architecture-independent headers that do not build as device code, for the reasons stated in the module docstring"""
    pre = '#include "/root/reference/algorithms/scc/scc.h"\n' if name.startswith("tc") else ""
    out, _ = compile_unit(tmp_path, name, pre + f'#include "/root/reference/algorithms/{name}"\n', "")
    assert out.returncode != 0 and needle in out.stderr
