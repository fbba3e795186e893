#!/usr/bin/env python3
"""<dir>/pmc_*.json (profiles/pmc_reduce.py) -> profiles/rNN_pmc_traffic.json in the layout bench.py's pmc_traffic() reads:
per kernel FETCH_SIZE / WRITE_SIZE per launch (KiB, from separate passes), hbm_bytes_raw = (FETCH + WRITE) * 1024 and
hbm_bytes_stream_corrected = (2 * FETCH + WRITE) * 1024 (gfx950 reports half the bytes of wide coalesced reads: MI355X_MICROARCH.md,
HBM section; an upper bound for kernels that mix streams and gathers).
usage: pmc_to_traffic.py <out.json> <pmc_a.json> [<pmc_b.json> ...]"""
import json
import sys

out = {}
for path in sys.argv[2:]:
    for k, v in json.load(open(path)).items():
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        f, w = v["FETCH_SIZE"]["per_launch"], v["WRITE_SIZE"]["per_launch"]
        out[k] = {"launches": v["FETCH_SIZE"]["launches"], "fetch_KiB_per_launch": round(f, 1), "write_KiB_per_launch": round(w, 1),
                  "hbm_bytes_raw": int((f + w) * 1024), "hbm_bytes_stream_corrected": int((2 * f + w) * 1024), "source": path.split("/")[-1]}
# what bench.py checks before it quotes a figure: the hash of the kernel sources the counters were collected on
import hashlib, os
h = hashlib.sha256()
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vectorgraphlibrary_amd", "csrc")
for name in ("bfs.hip", "vgl_hip_internal.h", "vgl_gnf.h"):
    h.update(open(os.path.join(root, name), "rb").read())
out["_kernel_source_sha"] = h.hexdigest()[:16]
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(len(out), "kernels ->", sys.argv[1])
