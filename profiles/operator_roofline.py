#!/usr/bin/env python3
"""Reduce profiles/collect_r04_operator.sh's output to roofline rows of the operator path's dominant kernels: per launch, average duration from the
kernel trace, algorithmic bytes of SURVEY 8(d) for the pass the launch is (SSSP relax 12 E + 28 V, SV hook 8 E + 12 V, PageRank pull 8 E + 28 V),
fraction of the 8 TB/s HBM peak, and the HBM traffic of the launch from the PMC passes (FETCH_SIZE + WRITE_SIZE: KiB units, raw -- the guide's x2
correction applies to wide coalesced streams only; these kernels gather 4-byte words)."""
import csv
import glob
import json
import os
import re
import sys

root = sys.argv[1]
PEAK = 8000.0
CASES = {   # name -> (V, E, bytes per pass, substring of the kernel's lambda type that picks the edge pass)
    "sssp": (1 << 24, (1 << 24) * 32, lambda V, E: 12 * E + 28 * V, "advance_static"),
    "cc": (1 << 24, (1 << 24) * 16 * 2, lambda V, E: 8 * E + 12 * V, "advance_static"),          # the app symmetrises: 2 x 16 x V stored edges
    "pr_atomics": (1 << 25, (1 << 25) * 32, lambda V, E: 8 * E + 28 * V, "advance_static"),
    "pr_pull": (1 << 25, (1 << 25) * 32, lambda V, E: 8 * E + 28 * V, "vertex_op"),
    "pr_rows": (1 << 25, (1 << 25) * 32, lambda V, E: 8 * E + 28 * V, "advance_rows"),
}
out = {}
for name, (V, E, fbytes, pick) in CASES.items():
    rows = []
    for f in glob.glob(os.path.join(root, "trace_" + name, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    if not rows:
        out[name] = {"error": "no kernel trace"}
        continue
    # the edge pass = the instantiation of the picked kernel with the largest total time
    tot = {}
    for k, d in rows:
        if pick in k:
            tot.setdefault(k, []).append(d)
    if not tot:
        out[name] = {"error": "kernel not found"}
        continue
    kern = max(tot, key=lambda k: sum(tot[k]))
    durs = tot[kern]
    if name == "pr_pull":                                   # (the pull is the longest of the compute launches: the others are V-sized passes)
        durs = [d for d in durs if d > 0.5 * max(durs)]
    avg_us = sum(durs) / len(durs) / 1e3
    alg = fbytes(V, E)
    rec = {"kernel": re.sub(r"\(.*", "", kern)[:160], "launches": len(durs), "avg_us": round(avg_us, 1), "algorithmic_bytes_per_launch": alg,
           "algorithmic_GBps": round(alg / (avg_us * 1e-6) / 1e9, 1), "frac_of_hbm_peak": round(alg / (avg_us * 1e-6) / 1e9 / PEAK, 4)}
    traffic = {}
    for i, cname in ((1, "FETCH_SIZE"), (2, "WRITE_SIZE")):
        vals = []
        for f in glob.glob(os.path.join(root, f"pmc_{name}_{i}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Kernel_Name"] == kern and r["Counter_Name"] == cname:
                    vals.append(float(r["Counter_Value"]))
        if name == "pr_pull" and vals:
            vals = [v for v in vals if v > 0.5 * max(vals)]
        if vals:
            traffic[cname + "_KiB_per_launch"] = round(sum(vals) / len(vals), 1)
    if len(traffic) == 2:
        hbm = (traffic["FETCH_SIZE_KiB_per_launch"] + traffic["WRITE_SIZE_KiB_per_launch"]) * 1024
        rec["hbm_bytes_raw_per_launch"] = int(hbm)
        rec["traffic_over_algorithmic"] = round(hbm / alg, 2)
    rec.update(traffic)
    out[name] = rec
json.dump(out, sys.stdout, indent=1)
