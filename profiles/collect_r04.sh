#!/bin/bash
# Round-4 profile collection, run on the GPU box from the repository root:  gpurun -- bash profiles/collect_r04.sh
# Writes under gpurun_out/r4prof/ (progress lines in progress.txt); the summaries are then copied into profiles/ (see profiles/README.md).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4prof
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
BFS="python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-sssp --no-pr-cc --no-operator-api"
# 1. per-kernel times of the default bench command (CPU baseline and the operator-API apps off: they only add host / child-process time)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-operator-api > $OUT/trace_bench_line.json 2> $OUT/trace.err
echo "trace done" > $OUT/progress.txt
# 2. HBM traffic of the BFS kernels (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE in passes of their own, no tracing in the same run)
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_bfs_$i -- $BFS > $OUT/pmc_bfs_$i.log 2>&1
    echo "pmc bfs $i done ($set)" >> $OUT/progress.txt
done
# 3. the top-down mode with blocked levels (the reference's algorithm): per-kernel averages of 18 traversals
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_td -- python3 tests/studies/td_profile_run.py > $OUT/trace_td.log 2>&1
echo "td trace done" >> $OUT/progress.txt
python3 profiles/pmc_reduce.py $OUT > $OUT/summary.log 2>&1
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +0 -delete
echo "all done" >> $OUT/progress.txt
