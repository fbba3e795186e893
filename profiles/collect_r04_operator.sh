#!/bin/bash
# Round-4 evidence for the drop-in operator path: kernel times (rocprofv3 --kernel-trace --stats) and HBM traffic (--pmc FETCH_SIZE / WRITE_SIZE,
# passes of their own) of the generic advance kernels under the SSSP / CC / PageRank lambdas of apps/algorithms/*.hpp at the bench's sizes.
# Run on the GPU box from the repository root:  gpurun -- bash profiles/collect_r04_operator.sh ; then profiles/operator_roofline.py reduces it.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4op
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
run() {   # name, app, args...
    name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -- "$@" > $OUT/trace_$name.log 2>&1
    echo "trace $name rc=$?" >> $OUT/progress.txt
    i=0
    for set in FETCH_SIZE WRITE_SIZE; do
        i=$((i+1))
        rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_${name}_$i -- "$@" > $OUT/pmc_${name}_$i.log 2>&1
        echo "pmc $name $set rc=$?" >> $OUT/progress.txt
    done
}
run sssp ./apps/bin/sssp_hip -s 24 -e 32 -type rmat -it 1 -format vcsr
run cc   ./apps/bin/cc_hip   -s 24 -e 16 -type rmat -it 1 -format vcsr
run pr_atomics ./apps/bin/pr_hip -s 25 -e 32 -type ru -it 5 -format csr
run pr_pull    ./apps/bin/pr_hip -s 25 -e 32 -type ru -it 5 -format csr -pull
run pr_rows    ./apps/bin/pr_hip -s 25 -e 32 -type ru -it 5 -format csr -deterministic
python3 profiles/operator_roofline.py $OUT > $OUT/operator_roofline.json 2> $OUT/operator_roofline.err
find $OUT -name "*_kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +0 -delete
echo "all done" >> $OUT/progress.txt
