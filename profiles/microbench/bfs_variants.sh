#!/bin/bash
# BFS on RMAT-24 under run-time switches of bfs.hip (VGL_BFS_SMALL_M: edge bound of the single-workgroup levels kernel); prints value / ms / kernel times
cd "$(dirname "$0")/../.."
for v in ${BFS_VARIANTS:-8192}; do
    VGL_BFS_SMALL_M=$v python3 bench.py --no-cpu-baseline --no-sssp --no-pr-cc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['bfs']['kernels']
print('SMALL_M=$v  %.4f ms/BFS  %.1f GTEPS  ' % (d['ms_per_step'], d['value']/1e9) + '  '.join('%s %.1f us x%d' % (n, 1e3*x['total_ms']/max(1,x['launches']), x['launches']) for n,x in k.items()) + '  kernels/wall %.3f' % d['bfs']['timed_kernels_over_wall_time'])"
done
