#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the GENUINE reference (oracle/_ref drivers).

TEST INFRASTRUCTURE.  Run in the build container only (needs /root/reference):
    make -C oracle ref && python oracle/make_golden.py
For every case the deterministic generator (oracle/vgl_oracle.c) produces the input edge list,
the reference multicore build (vgl_compute_api/multicore via oracle/ref_driver.cpp) computes
BFS levels / SSSP distances / SSWP widths / PageRank / CC labels, and the outputs are stored as data:
full arrays for V <= 4096, FNV-1a-64 hashes + histograms above.  The script also asserts that
the C restatement agrees with the reference before writing (so a drifting oracle cannot
silently produce self-consistent goldens).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(ROOT, "tests", "golden")

# (name, kind, scale, edge_factor, seed, pr_iters, full_arrays)
CASES = [
    ("rmat_s6_e8_seed1", "rmat", 6, 8, 1, 5, True),
    ("rmat_s10_e8_seed2", "rmat", 10, 8, 2, 5, True),
    ("rmat_s12_e16_seed3", "rmat", 12, 16, 3, 10, True),
    ("ru_s10_e8_seed4", "ru", 10, 8, 4, 5, True),
    ("ru_s12_e16_seed5", "ru", 12, 16, 5, 10, True),
    ("rmat_s16_e32_seed6", "rmat", 16, 32, 6, 5, False),
    ("ru_s16_e32_seed7", "ru", 16, 32, 7, 5, False),
]


def run(app, *args):
    env = dict(os.environ, OMP_NUM_THREADS="4")   # the reference segfaults with 1 thread
    subprocess.check_call([os.path.join(REF, "ref_driver_" + app)] + [str(a) for a in args],
                          stdout=subprocess.DEVNULL, env=env)


def relerr(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b.astype(np.float64)), 1e-300)))


def main():
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="vgl_golden_")
    for name, kind, scale, ef, seed, pr_iters, full in CASES:
        V = 1 << scale
        src, dst = (O.gen_rmat if kind == "rmat" else O.gen_uniform)(scale, ef, seed)
        E = len(src)
        w_in = O.gen_weights(E, seed)
        rowptr, adj, perm = O.coo_to_csr(V, src, dst)
        w = w_in[perm]
        source = O.pick_source(rowptr, seed)
        g = os.path.join(tmp, name + ".el_container")
        O.write_el_container(g, V, src, dst)
        wfile = os.path.join(tmp, name + ".w")
        w.tofile(wfile)
        o = os.path.join(tmp, "out.bin")

        # --- BFS (reference: csr and vcsr must agree) ---
        run("bfs", g, "csr", o, source)
        levels = np.fromfile(o, np.int32)
        run("bfs", g, "vcsr", o, source)
        assert (np.fromfile(o, np.int32) == levels).all(), "reference csr/vcsr BFS disagree"
        my_levels, st = O.bfs_top_down(rowptr, adj, source)
        assert (my_levels == levels).all(), "oracle BFS != reference"

        # --- SSSP (push and pull bit-identical in the reference) ---
        run("sssp", g, "csr", o, source, wfile, "push")
        dist = np.fromfile(o, np.float32)
        run("sssp", g, "csr", o, source, wfile, "pull")
        assert (np.fromfile(o, np.float32).view(np.int32) == dist.view(np.int32)).all()
        my_dist, _ = O.sssp_bellman_ford(rowptr, adj, w, source)
        assert (my_dist.view(np.int32) == dist.view(np.int32)).all(), "oracle SSSP != reference"

        # --- SSWP (capacities = the same f32 edge values; reference vgl and its sequential checker must agree bit for bit) ---
        run("sswp", g, "csr", o, source, wfile)
        wd = np.fromfile(o, np.float32).reshape(2, V)
        width = wd[0].copy()
        assert (wd[1].view(np.int32) == width.view(np.int32)).all(), "reference SSWP vgl != seq"
        my_width, _ = O.sswp_bellman_ford(rowptr, adj, w, source)
        assert (my_width.view(np.int32) == width.view(np.int32)).all(), "oracle SSWP != reference"
        assert (O.sswp_seq(rowptr, adj, w, source).view(np.int32) == width.view(np.int32)).all(), "oracle SSWP checker != reference"

        # --- PageRank ---
        run("pr", g, "csr", o, pr_iters)
        r = np.fromfile(o, np.float32).reshape(2, V)
        pr_vgl_csr, pr_seq_csr = r[0].copy(), r[1].copy()
        run("pr", g, "vcsr", o, pr_iters)
        pr_vgl_vcsr = np.fromfile(o, np.float32).reshape(2, V)[0].copy()
        my_pr0 = O.pagerank(rowptr, adj, pr_iters, 0)
        my_pr1 = O.pagerank(rowptr, adj, pr_iters, 1)
        assert (my_pr0.view(np.int32) == pr_seq_csr.view(np.int32)).all(), "oracle PR(mode0) != seq_page_rank bits"
        e_csr, e_vcsr = relerr(my_pr1, pr_vgl_csr), relerr(my_pr1, pr_vgl_vcsr)

        # --- CC on the symmetrised graph ---
        ss, dd = O.symmetrize(src, dst)
        gs = os.path.join(tmp, name + ".sym.el_container")
        O.write_el_container(gs, V, ss, dd)
        rp2, adj2, _ = O.coo_to_csr(V, ss, dd, want_perm=False)
        run("cc", gs, "csr", o)
        c = np.fromfile(o, np.int32).reshape(2, V)
        comp_csr, comp_seq = c[0].copy(), c[1].copy()
        run("cc", gs, "vcsr", o)
        comp_vcsr = np.fromfile(o, np.int32).reshape(2, V)[0].copy()
        my_comp, _ = O.cc_sv(rp2, adj2)
        assert (my_comp == comp_csr).all(), "oracle CC labels != reference (csr numbering)"
        assert (O.cc_seq_bfs(rp2, adj2) == comp_seq).all()
        if V <= 4096:
            assert O.same_partition(comp_vcsr, comp_csr)

        rec = dict(
            kind=kind, scale=scale, edge_factor=ef, seed=seed, pr_iters=pr_iters, source=source,
            pin_src=np.uint64(O.fnv1a64(src)), pin_dst=np.uint64(O.fnv1a64(dst)), pin_w=np.uint64(O.fnv1a64(w_in)),
            pin_rowptr=np.uint64(O.fnv1a64(rowptr)), pin_adj=np.uint64(O.fnv1a64(adj)),
            bfs_fnv=np.uint64(O.fnv1a64(levels)), sssp_fnv=np.uint64(O.fnv1a64(dist)), cc_fnv=np.uint64(O.fnv1a64(comp_csr)),
            sswp_fnv=np.uint64(O.fnv1a64(width)),
            bfs_level_hist=np.bincount(levels + 1),          # index 0 = unvisited (-1)
            bfs_edges_examined=st["edges_examined"], bfs_frontier_total=st["frontier_total"],
            cc_num_components=len(np.unique(comp_csr)),
            cc_size_hist=np.sort(np.bincount(comp_csr)[np.bincount(comp_csr) > 0])[::-1][:64],
            pr_sum_vgl=np.float64(pr_vgl_csr.astype(np.float64).sum()),
            pr_ref_csr_vs_vcsr=relerr(pr_vgl_csr, pr_vgl_vcsr), pr_ref_vgl_vs_seq=relerr(pr_vgl_csr, pr_seq_csr),
        )
        if full:
            rec.update(levels=levels, dist=dist, width=width, pr_vgl_csr=pr_vgl_csr, pr_vgl_vcsr=pr_vgl_vcsr,
                       pr_seq_csr=pr_seq_csr, comp_csr=comp_csr, comp_seq=comp_seq)
        else:                                   # sampled arrays + hashes keep the fixture small
            idx = np.arange(0, V, V // 1024, dtype=np.int64)
            rec.update(sample_idx=idx, levels_s=levels[idx], dist_s=dist[idx], width_s=width[idx], pr_vgl_csr_s=pr_vgl_csr[idx],
                       pr_seq_csr_fnv=np.uint64(O.fnv1a64(pr_seq_csr)), comp_csr_s=comp_csr[idx])
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
        print(f"{name}: V={V} E={E} src={source} bfs_levels={st['levels']} "
              f"PR rel.err oracle(f64-dangling) vs ref-vgl csr={e_csr:.2e} vcsr={e_vcsr:.2e} "
              f"(ref vgl-vs-seq {rec['pr_ref_vgl_vs_seq']:.2e}, ref csr-vs-vcsr {rec['pr_ref_csr_vs_vcsr']:.2e})")


if __name__ == "__main__":
    main()
