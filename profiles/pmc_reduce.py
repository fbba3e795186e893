#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc output directories (one counter_collection.csv per pass) to per-kernel, per-launch averages.
usage: pmc_reduce.py <dir holding pmc_*/ sub-directories>   -> writes <dir>/pmc_<name>.json per group of passes"""
import collections
import csv
import glob
import json
import os
import re
import sys


def reduce(paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for path in paths:
        for r in csv.DictReader(open(path)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            if not name.startswith("vgl_k_"):
                continue
            a = agg[name][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return {k: {c: {"launches": n, "per_launch": tot / n} for c, (n, tot) in v.items()} for k, v in agg.items()}


def main():
    root = sys.argv[1]
    groups = collections.defaultdict(list)
    for d in sorted(glob.glob(os.path.join(root, "pmc_*_[0-9]*"))):
        if os.path.isdir(d):
            groups[re.sub(r"_[0-9]+$", "", os.path.basename(d))] += glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    for g, paths in groups.items():
        out = reduce(paths)
        json.dump(out, open(os.path.join(root, g + ".json"), "w"), indent=1, sort_keys=True)
        print(g, len(paths), "files", len(out), "kernels")


if __name__ == "__main__":
    main()
