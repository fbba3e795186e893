#!/usr/bin/env python3
"""Per-launch kernel durations of the BFS traversals from a rocprofv3 kernel trace (the bottom-up levels differ a lot: an average hides it).
usage (GPU box):  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bfstrace -- python3 bench.py --steps 4 --warmup 1 \
                      --no-cpu-baseline --no-sssp --no-pr-cc && python3 profiles/microbench/bfs_launches.py gpurun_out/bfstrace"""
import csv
import glob
import sys

files = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
short = {"vgl_k_bu_probe": "probe", "vgl_k_bu_heavy": "heavy", "vgl_k_bm_advance": "adv", "vgl_k_td_expand": "td"}
# the last traversals: everything after the last but N-th vgl_k_bfs_init
inits = [i for i, r in enumerate(rows) if "bfs_init" in r[2]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
spans = [(a, b) for a, b in zip(inits[:-1], inits[1:]) if any("bu_probe" in r[2] for r in rows[a:b])]
for a, b in spans[-n:]:
    t0 = rows[a][0]
    parts, prev_end, busy = [], None, 0
    for s, e, name in rows[a:b]:
        label = next((v for k, v in short.items() if k in name), name.replace('void ', '').replace('vgl_k_', '')[:14])
        gap = "" if prev_end is None else f"(+{(s - prev_end) / 1e3:.1f})"          # idle time of the stream before this launch
        parts.append(f"{label} {(e - s) / 1e3:.1f}{gap}")
        busy += e - s
        prev_end = e
    span = rows[b - 1][1] - t0
    print("traversal: " + "  ".join(parts) + f"   | span {span / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us")
