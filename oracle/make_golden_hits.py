#!/usr/bin/env python3
"""Generate tests/golden/hits_*.npz from the GENUINE reference (oracle/_ref/ref_driver_hits).

TEST INFRASTRUCTURE.  Run in the build container only (needs /root/reference):
    make -C oracle ref && python oracle/make_golden_hits.py
Same deterministic inputs as make_golden.py (kept separate so that adding an algorithm does not re-run the slow scale-16 cases of
the others).  The reference's HITS::vgl_hits (csr and vcsr) and its sequential checker HITS::seq_hits are run; the script asserts
that the C restatement reproduces seq_hits BIT FOR BIT and stores authorities / hubs (f64) as data.
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
# (name, kind, scale, edge_factor, seed, steps)
CASES = [
    ("rmat_s6_e8_seed1", "rmat", 6, 8, 1, 4),
    ("rmat_s10_e8_seed2", "rmat", 10, 8, 2, 4),
    ("rmat_s12_e16_seed3", "rmat", 12, 16, 3, 6),
    ("ru_s10_e8_seed4", "ru", 10, 8, 4, 4),
    ("ru_s12_e16_seed5", "ru", 12, 16, 5, 6),
]


def relerr(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def main():
    tmp = tempfile.mkdtemp(prefix="vgl_golden_hits_")
    env = dict(os.environ, OMP_NUM_THREADS="4")
    for name, kind, scale, ef, seed, steps in CASES:
        V = 1 << scale
        src, dst = (O.gen_rmat if kind == "rmat" else O.gen_uniform)(scale, ef, seed)
        rowptr, adj, _ = O.coo_to_csr(V, src, dst)
        g = os.path.join(tmp, name + ".el_container")
        O.write_el_container(g, V, src, dst)
        o = os.path.join(tmp, "out.bin")
        res = {}
        for fmt in ("csr", "vcsr"):
            subprocess.check_call([os.path.join(REF, "ref_driver_hits"), g, fmt, o, str(steps)], stdout=subprocess.DEVNULL, env=env)
            res[fmt] = np.fromfile(o, np.float64).reshape(4, V)
        auth_vgl, hub_vgl, auth_seq, hub_seq = res["csr"]
        my_auth, my_hub = O.hits(rowptr, adj, steps)
        assert (my_auth.view(np.int64) == auth_seq.view(np.int64)).all(), "oracle HITS authorities != reference seq_hits bits"
        assert (my_hub.view(np.int64) == hub_seq.view(np.int64)).all(), "oracle HITS hubs != reference seq_hits bits"
        spread = max(relerr(auth_vgl, auth_seq), relerr(hub_vgl, hub_seq), relerr(res["vcsr"][0], auth_seq), relerr(res["vcsr"][1], hub_seq))
        np.savez_compressed(os.path.join(OUT, "hits_" + name + ".npz"), kind=kind, scale=scale, edge_factor=ef, seed=seed, steps=steps,
                            pin_rowptr=np.uint64(O.fnv1a64(rowptr)), pin_adj=np.uint64(O.fnv1a64(adj)),
                            auth_seq=auth_seq, hub_seq=hub_seq, auth_vgl_csr=auth_vgl, hub_vgl_csr=hub_vgl,
                            ref_vgl_vs_seq=spread)
        print(f"hits_{name}: V={V} steps={steps} oracle == seq_hits bit for bit; reference vgl (csr, vcsr) vs seq rel. spread {spread:.2e}")


if __name__ == "__main__":
    main()
