#!/bin/bash
# Round-5 profile collection, run on the GPU box from the repository root:  gpurun -- bash profiles/collect_r05.sh
# Writes under gpurun_out/r5prof/ (progress lines in progress.txt); the summaries are then copied into profiles/ (see profiles/README.md).
#   1. the DRIVER'S bench command (python bench.py --steps 20 --warmup 5): its JSON line -> r05_bench_line_driver_cmd.json
#   2. rocprofv3 --kernel-trace --stats of the same command (CPU baseline / operator apps off: child processes; the RMAT-27 leg off: its launches of the same kernels would mix into the averages) -> r05_bench_kernel_stats.csv
#   3. FETCH_SIZE / WRITE_SIZE of the BFS kernels in passes of their own (MI355X_MICROARCH.md) -> r05_pmc_bfs.json via pmc_reduce.py
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5prof
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_line_driver_cmd.json 2> $OUT/bench_driver_cmd.err
echo "driver command done" > $OUT/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-operator-api --no-bfs-big > $OUT/trace_bench_line.json 2> $OUT/trace.err
echo "trace done" >> $OUT/progress.txt
BFS="python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-sssp --no-pr-cc --no-operator-api"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_bfs_$i -- $BFS > $OUT/pmc_bfs_$i.log 2>&1
    echo "pmc bfs $i done ($set)" >> $OUT/progress.txt
done
python3 profiles/pmc_reduce.py $OUT > $OUT/summary.log 2>&1
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +0 -delete
echo "all done" >> $OUT/progress.txt
