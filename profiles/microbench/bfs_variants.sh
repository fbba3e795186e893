#!/bin/bash
# BFS on RMAT-24 under the bottom-up probe variants (env switches read by bfs.hip); prints value / ms_per_step / probe time per launch
cd "$(dirname "$0")/../.."
for v in "0 0"; do
    set -- $v
    VGL_BFS_PROBE_NT=$1 VGL_BFS_PROBE_LDS=$2 python3 bench.py --no-cpu-baseline --no-sssp --no-pr-cc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['bfs']['kernels']
print('NT=$1 LDS=$2  %.4f ms/BFS  %.1f GTEPS  probe %.2f us x%d  heavy %.2f us  td %.2f us  small %.2f us  gnf %.2f us  kernels/wall %.3f' % (d['ms_per_step'], d['value']/1e9,
      1e3*k['bfs_bottom_up']['total_ms']/k['bfs_bottom_up']['launches'], k['bfs_bottom_up']['launches'], 1e3*k['bfs_bottom_up_heavy']['total_ms']/k['bfs_bottom_up_heavy']['launches'],
      1e3*k['bfs_top_down']['total_ms']/max(1,k['bfs_top_down']['launches']), 1e3*k['bfs_small_levels']['total_ms']/max(1,k['bfs_small_levels']['launches']),
      1e3*k['gnf']['total_ms']/max(1,k['gnf']['launches']), d['bfs']['timed_kernels_over_wall_time']))"
done
