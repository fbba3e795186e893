#!/bin/bash
# Round-2 profile collection, run on the GPU box from the repository root:  gpurun -- bash profiles/collect_r02.sh
# Writes under gpurun_out/r2prof/; the summaries are then copied into profiles/ (see profiles/README.md).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r2prof
mkdir -p $OUT
export TMPDIR=/tmp
BFS="python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-sssp --no-pr-cc"
# 1. per-kernel times of the default bench command (CPU baseline off: it only adds host time)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline > $OUT/trace_bench_line.json 2> $OUT/trace.err
echo "trace done" > $OUT/progress.txt
# 2. counters for the BFS kernels, one pass per block group (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE in passes of their own)
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_bfs_$i -- $BFS > $OUT/pmc_bfs_$i.log 2>&1
    echo "pmc bfs $i done ($set)" >> $OUT/progress.txt
done
# 3. the blocked passes (PageRank uniform-25; SSSP pull on RMAT-24): times incl. the plan build kernels, then traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_blocked_u25 -- python3 profiles/microbench/blocked_bench.py uniform 25 > $OUT/blocked_u25.log 2>&1
echo "blocked trace done" >> $OUT/progress.txt
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_u25_$i -- python3 profiles/microbench/blocked_bench.py uniform 25 > $OUT/pmc_u25_$i.log 2>&1
    rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_r24_$i -- python3 profiles/microbench/blocked_bench.py rmat 24 > $OUT/pmc_r24_$i.log 2>&1
    echo "pmc blocked $i done ($set)" >> $OUT/progress.txt
done
# 4. the 8-rank lock-step rehearsal of the RMAT-27 weak-scaling traversal on this one GPU (levels kept per owner)
python3 profiles/microbench/emulate_weak.py 27 8 dealt v owned > $OUT/emulate_weak_s27_p8_owned.log 2>&1
echo "rehearsal done" >> $OUT/progress.txt
# keep what is small: stats tables and per-kernel counter sums
python3 profiles/pmc_reduce.py $OUT > $OUT/summary.log 2>&1
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +0 -delete
echo "all done" >> $OUT/progress.txt
