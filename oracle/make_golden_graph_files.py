#!/usr/bin/env python3
"""Write tests/golden/<case>.{el_container,csr,vcsr} with the GENUINE reference (oracle/_ref/ref_graphfile).

TEST INFRASTRUCTURE.  Build container only: make -C oracle ref && python oracle/make_golden_graph_files.py
The edge lists are the ones of the fixtures of the same name (deterministic generator in oracle/vgl_oracle.c).  The
script asserts, before writing, that oracle/graph_files.py reproduces the reference's files byte for byte, and that the
reference reads a file written by the restatement and computes the golden BFS levels from it.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from oracle import graph_files as GF  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "ref_graphfile")
OUT = os.path.join(ROOT, "tests", "golden")
COMMIT = [("rmat_s6_e8_seed1", "rmat", 6, 8, 1)]                       # small enough to keep as files
CHECK_ONLY = [("rmat_s10_e8_seed2", "rmat", 10, 8, 2), ("ru_s10_e8_seed4", "ru", 10, 8, 4), ("ru_s12_e16_seed5", "ru", 12, 16, 5)]


def ref(*args):
    env = dict(os.environ, OMP_NUM_THREADS="4")
    subprocess.check_call([REF] + [str(a) for a in args], stdout=subprocess.DEVNULL, env=env)


def main():
    tmp = tempfile.mkdtemp(prefix="vgl_gf_")
    digests = {}
    for name, kind, scale, ef, seed in COMMIT + CHECK_ONLY:
        V = 1 << scale
        src, dst = (O.gen_rmat if kind == "rmat" else O.gen_uniform)(scale, ef, seed)
        keep = (name, kind, scale, ef, seed) in COMMIT
        base = os.path.join(OUT if keep else tmp, name)
        O.write_el_container(base + ".el_container", V, src, dst)
        gold = np.load(os.path.join(OUT, name + ".npz"))
        for ext, fmt in (("csr", GF.CSR_GRAPH), ("vcsr", GF.VECTOR_CSR_GRAPH)):
            ref("save", base + ".el_container", ext, base + "." + ext)
            theirs = open(base + "." + ext, "rb").read()
            ours = GF.graph_file_bytes(GF.build_graph(V, src, dst, fmt))
            assert ours == theirs, "restatement differs from the reference file: %s.%s" % (name, ext)
            mine = os.path.join(tmp, name + ".mine." + ext)
            open(mine, "wb").write(ours)
            lv = os.path.join(tmp, "levels.bin")
            ref("bfs", mine, ext, int(gold["source"]), lv)
            levels = np.fromfile(lv, np.int32)
            assert np.array_equal(levels, gold["levels"]), "reference BFS on a restated file: %s.%s" % (name, ext)
            digests.setdefault(name, {"kind": kind, "scale": scale, "edge_factor": ef, "seed": seed})[ext] = {
                "bytes": len(theirs), "sha256": hashlib.sha256(theirs).hexdigest()}
            print("%-22s .%-4s %7d bytes  identical to the reference, reference BFS from it matches" % (name, ext, len(theirs)))

    # sha256 of the reference-written file of every case: lets the device tests compare files too large to commit
    with open(os.path.join(OUT, "graph_files.json"), "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
