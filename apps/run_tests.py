#!/usr/bin/env python3
"""Benchmark / verification driver for the apps of this backend -- the counterpart of the reference's apps/run_tests.py with
apps/scripts/{benchmarking_api,verification_api,create_graphs_api,settings}.py: the same graph naming (syn_<type>_<scale>_<edge
factor>, stored as bin/input_graphs/<name>.vgraph.el_container by create_vgl_graphs), the same per-app argument sets, the same
two lines grepped from the output (AVG_PERF, "error count:").  Results go to a JSON file and a CSV table (the reference writes an
xlsx sheet and can post to a rating server; neither exists here).

    python apps/run_tests.py -a bfs,pr -f csr,vcsr -m fastest -b -v
"""
import argparse
import csv
import json
import os
import re
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "bin")
GRAPHS_DIR = os.path.join(BIN, "input_graphs")

# per-app argument sets (one benchmark / verification row each)
APP_ARGS = {
    "bfs": [["-top-down"], ["-fused", "-td"], ["-fused", "-do"]],
    "sssp": [["-push", "-all-active"], ["-fused"]],
    "pr": [["-pull"], ["-fused"]],
    "cc": [["-cv"], ["-fused"]],
    "sswp": [["-push"], ["-fused"]],
    "hits": [[], ["-fused"]],
    "scc": [[], ["-fused"]],
    "coloring": [[]],
    "rw": [["-it", "100", "-wv", "20"]],
    "tc": [["-bfs-based", "-it", "500"], ["-purdoms", "-it", "500"]],
    "mf": [[]],
}
MODES = {                                     # graph sets by run mode
    "smoke": ["syn_rmat_10_8", "syn_ru_10_8"],
    "fastest": ["syn_rmat_18_32", "syn_ru_18_32"],
    "tiny-only": ["syn_rmat_18_32", "syn_ru_18_32", "syn_rmat_20_32", "syn_ru_20_32"],
    "small-only": ["syn_rmat_22_32", "syn_ru_22_32"],
    "medium-only": ["syn_rmat_24_32", "syn_ru_24_32"],
    "large-only": ["syn_rmat_25_32", "syn_ru_25_32"],
    "scaling": ["syn_rmat_%d_32" % s for s in range(18, 24)],
}
MODES["tiny-small"] = MODES["tiny-only"] + MODES["small-only"]
MODES["tiny-small-medium"] = MODES["tiny-small"] + MODES["medium-only"]
UNDIRECTED_APPS = {"coloring"}               # need every edge in both directions (requires_undir_graphs, settings.py:30-35)
UNDIRECTED_PREFIX = "undir_"
COMMON_ITERATIONS = 10
PERF_PATTERN = "AVG_PERF"
CORRECTNESS_PATTERN = "error count:"


def graph_path(name, undirected=False):
    return os.path.join(GRAPHS_DIR, (UNDIRECTED_PREFIX if undirected else "") + name + ".vgraph.el_container")


def binary(app):
    return os.path.join(BIN, app + "_hip")


def create_graphs(names, undirected=False):
    os.makedirs(GRAPHS_DIR, exist_ok=True)
    for name in names:
        path = graph_path(name, undirected)
        if os.path.exists(path):
            continue
        _, kind, scale, edge_factor = name.split("_")
        cmd = [binary("create_vgl_graphs"), "-s", scale, "-e", edge_factor, "-type", kind, "-undirected" if undirected else "-directed",
               "-format", "el_container", "-file", path[: -len(".el_container")]]
        print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
        if not os.path.exists(path):
            raise RuntimeError("graph %s can not be created" % path)


def run_app(cmd, timeout):
    print(" ".join(cmd), flush=True)
    start = time.time()
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout).stdout
    except subprocess.TimeoutExpired:
        out = None
    return out, time.time() - start


def perf_value(out):
    if out is None:
        return "TIMED OUT"
    lines = [line for line in out.split("\n") if PERF_PATTERN in line]
    if not lines:
        return "NO PERF LINE"
    m = re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", lines[-1].split(PERF_PATTERN, 1)[1])
    return float(m[0]) if m else "NO PERF VALUE"


def correctness_value(out):
    if out is None:
        return "TIMED OUT"
    counts = [int(x) for x in re.findall(CORRECTNESS_PATTERN + r"\s*(\d+)", out)]
    return "NO CORRECTNESS LINE" if not counts else sum(counts)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-a", "--apps", default="all", help="comma-separated apps (default all)")
    ap.add_argument("-f", "--formats", default="vcsr", help="comma-separated storage formats: csr, vcsr, all (default vcsr)")
    ap.add_argument("-m", "--mode", default="fastest", choices=sorted(MODES), help="graph set")
    ap.add_argument("-b", "--benchmark", action="store_true", help="measure AVG_PERF")
    ap.add_argument("-v", "--verify", action="store_true", help="run every app with -check and collect the error counts")
    ap.add_argument("-t", "--timeout", type=int, default=600, help="seconds per run")
    ap.add_argument("-n", "--name", default="vgl_hip_results", help="output file stem (under apps/bin/)")
    opt = ap.parse_args()
    apps = sorted(APP_ARGS) if opt.apps == "all" else opt.apps.split(",")
    formats = ["csr", "vcsr"] if opt.formats == "all" else opt.formats.split(",")
    for app in apps:
        if app not in APP_ARGS:
            sys.exit("unknown app " + app)
        if not os.path.exists(binary(app)):
            sys.exit("binary %s is missing: run `make -C apps`" % binary(app))
    graphs = MODES[opt.mode]
    create_graphs(graphs)
    if UNDIRECTED_APPS & set(apps):
        create_graphs(graphs, undirected=True)
    rows = []
    for fmt in formats:
        for app in apps:
            for args in APP_ARGS[app]:
                own_it = "-it" in args
                for graph in graphs:
                    row = {"app": app, "args": " ".join(args), "format": fmt, "graph": graph}
                    base = [binary(app), "-import", graph_path(graph, app in UNDIRECTED_APPS)] + args + ["-format", fmt]
                    if opt.benchmark:
                        out, secs = run_app(base + ([] if own_it else ["-it", str(COMMON_ITERATIONS)]), opt.timeout)
                        row["perf"], row["perf_seconds"] = perf_value(out), round(secs, 2)
                    if opt.verify:
                        out, secs = run_app(base + ["-check"] + ([] if own_it else ["-it", "1"]), opt.timeout)
                        row["errors"], row["verify_seconds"] = correctness_value(out), round(secs, 2)
                    rows.append(row)
    os.makedirs(BIN, exist_ok=True)
    stem = os.path.join(BIN, opt.name)
    with open(stem + ".json", "w") as f:
        json.dump({"mode": opt.mode, "formats": formats, "rows": rows}, f, indent=1)
    keys = ["app", "args", "format", "graph", "perf", "errors", "perf_seconds", "verify_seconds"]
    with open(stem + ".csv", "w", newline="") as f:
        w = csv.DictWriter(f, keys, extrasaction="ignore")
        w.writeheader()
        w.writerows(rows)
    width = max(len(r["app"] + " " + r["args"]) for r in rows) + 2
    print("\n%-*s %-5s %-16s %14s %8s" % (width, "test", "fmt", "graph", "perf", "errors"))
    for r in rows:
        print("%-*s %-5s %-16s %14s %8s" % (width, r["app"] + " " + r["args"], r["format"], r["graph"], r.get("perf", "-"), r.get("errors", "-")))
    evaluated = len({(r["app"], r["args"]) for r in rows if isinstance(r.get("perf"), float)})
    verified = len({(r["app"], r["args"]) for r in rows if r.get("errors") == 0})
    print("\nEVALUATED PERFORMANCE OF %d TESTS\nVERIFIED %d TESTS" % (evaluated, verified))
    bad = [r for r in rows if ("errors" in r and r["errors"] != 0) or ("perf" in r and not isinstance(r["perf"], float))]
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
