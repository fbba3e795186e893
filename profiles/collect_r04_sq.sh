#!/bin/bash
# Round-4 issue / wait counters of the BFS kernels (SQ block: 8 counters per pass), run on the GPU box from the repository root.
# usage: gpurun -- bash profiles/collect_r04_sq.sh [extra env assignments for the traversal, e.g. VGL_BU_FILTER_F=0]
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4sq
mkdir -p $OUT
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
CMD="python3 profiles/microbench/bfs_ab.py --steps 8 --rounds 1 run:"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_sq_$i -- $CMD > $OUT/pmc_sq_$i.log 2>&1
    echo "pass $i rc=$? ($set)" >> $OUT/progress.txt
done
python3 profiles/pmc_reduce.py $OUT > $OUT/summary.log 2>&1
find $OUT -name "*counter_collection.csv" -size +0 -delete
echo "all done" >> $OUT/progress.txt
