"""The HIP backend class bound into the REFERENCE'S OWN source tree (CPU container only).
`make -C oracle binding` copies /root/reference to a scratch directory in /tmp, applies integration/apply_hip_binding.py (the edits of INTEGRATION.md
section 2: -D __USE_HIP__ selects GraphAbstractionsHIP, a class derived from GraphAbstractions in vgl_compute_api/hip/graph_abstractions_hip.h that
works on CSRGraph / VectorCSRGraph / FrontierCSR / FrontierVectorCSR through friend access) and compiles the reference's applications
apps/{bfs,sswp,hits,scc,pr,sssp,cc,mf,coloring,tc}/*.cpp -- main() unchanged, algorithms unchanged but for three CUDA runtime calls by name in the GPU variants -- with hipcc for gfx950 against libvgl_hip.so.  Nothing of the reference
enters the repository; the binaries go to oracle/_ref (git-ignored, they travel to the GPU box, where tests/test_reference_binding_gpu.py runs
them with the reference's own -check)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HIPCC = "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF) or not os.path.exists(HIPCC), reason="needs /root/reference and hipcc (CPU container)")
APPS = ("bfs", "sswp", "hits", "scc", "pr", "sssp", "cc", "mf", "coloring", "tc")


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(os.path.join(ROOT, "vectorgraphlibrary_amd", "libvgl_hip.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "vectorgraphlibrary_amd", "csrc")], check=True, capture_output=True, timeout=1800)
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "binding"], capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    return out


def test_the_reference_apps_compile_with_the_backend_bound_in(built):
    for app in APPS + ("plan_stamp_check", "tc_check", "scc_check"):  # (the last two: integration/tests/*.cpp, our own programs against the patched tree)
        assert os.path.exists(os.path.join(ROOT, "oracle", "_ref", "vgl_hip_" + app)), app


def test_the_binaries_call_the_c_abi(built):
    # graph handles over the containers' arrays, frontier handles over the frontier containers' arrays, the frontier generation halves
    exe = os.path.join(ROOT, "oracle", "_ref", "vgl_hip_bfs")
    syms = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True, check=True).stdout
    for s in ("vgl_hip_graph_create", "vgl_hip_frontier_create_on", "vgl_hip_frontier_set_state", "vgl_hip_gnf_begin", "vgl_hip_gnf_complete",
              "vgl_hip_frontier_advance_plan", "vgl_hip_graph_tile_rows"):
        assert re.search(r"\bU " + s + r"\b", syms), s


def test_every_edit_of_the_binding_applies_to_a_fresh_copy(tmp_path):
    # the script refuses to continue when an anchor is missing or ambiguous: it is also the check that the reference has the shape it was written for
    copy = tmp_path / "vgl"
    subprocess.run(["cp", "-r", REF, str(copy)], check=True)
    subprocess.run(["chmod", "-R", "u+w", str(copy)], check=True)
    out = subprocess.run(["python3", os.path.join(ROOT, "integration", "apply_hip_binding.py"), str(copy)], capture_output=True, text=True)
    assert out.returncode == 0 and "edits applied" in out.stdout, out.stdout + out.stderr
    assert (copy / "vgl_compute_api" / "hip" / "graph_abstractions_hip.h").exists()
    text = (copy / "architecture_independent_api.h").read_text()
    assert "#define VGL_GRAPH_ABSTRACTIONS GraphAbstractionsHIP" in text
    again = subprocess.run(["python3", os.path.join(ROOT, "integration", "apply_hip_binding.py"), str(copy)], capture_output=True, text=True)
    assert again.returncode != 0                                                                  # not twice
