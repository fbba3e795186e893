#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs of the same bench command, as
MI355X_MICROARCH.md 'rocprofv3 PMC slots' requires) into per-kernel, per-launch figures.
usage: pmc_summarize.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
Units: FETCH_SIZE / WRITE_SIZE are KiB.  gfx950 caveat (MI355X_MICROARCH.md 'HBM'): FETCH_SIZE reports HALF of the bytes of
a wide coalesced stream (16 B/lane); calibration on this repo's own kernels: vgl_k_gnf_count reads 64 MiB of levels with
dwordx4 and shows 37.7 MiB (half + row offsets); vgl_k_gather (random 4-byte gathers) shows 64 B per gathered element,
not halved.  `hbm_bytes_raw` = (FETCH+WRITE)*1024; `hbm_bytes_stream_corrected` doubles the fetch side (upper bound for
kernels that mix streams and gathers)."""
import collections
import csv
import json
import sys


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[name][0] += 1
        agg[name][1] += float(r["Counter_Value"])
    return agg


def main():
    f = load(sys.argv[1], "FETCH_SIZE")
    w = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in f:
        if not k.startswith("vgl_k_"):
            continue
        n, tot = f[k]
        wn, wt = w.get(k, [0, 0.0])
        fk, wk = tot / n, wt / max(wn, 1)
        out[k] = {"launches": n, "fetch_KiB_per_launch": round(fk, 1), "write_KiB_per_launch": round(wk, 1),
                  "hbm_bytes_raw": int((fk + wk) * 1024), "hbm_bytes_stream_corrected": int((2 * fk + wk) * 1024)}
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_raw"]):
        print(f"{k:45s} {v}")


if __name__ == "__main__":
    main()
