#!/bin/bash
# The REFERENCE'S OWN apps, HIP backend bound in, at the BASELINE sizes (round 5; device memory is the default: shadowed user arrays):
#   * AVG_PERF of bfs / sssp -all-active / pr / cc, csr and vcsr, graph files from apps/bin/create_vgl_graphs_hip  -> gpurun_out/r05_binding_perf_baseline_sizes.log
#   * rocprofv3 --kernel-trace --stats of the sssp app on RMAT-24 vcsr: the row of vgl_k_advance_vector_extension
#     -> gpurun_out/r05_binding_sssp_vcsr_rmat24_kernel_stats.csv, and of the bfs app on RMAT-24 csr / vcsr
# usage: gpurun -- bash profiles/collect_r05_binding.sh
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp VGL_HIP_SHADOW_STATS=1
D=/tmp/vgl_r05_graphs; mkdir -p $D
LOG=gpurun_out/r05_binding_perf_baseline_sizes.log; : > $LOG
run() { echo "=== $*" >> $LOG; "$@" 2>&1 | grep -i "AVG_PERF\|graph load\|shadowed\|rror" >> $LOG; }
trace() { # <tag> <app> <args...>
  local tag=$1 app=$2; shift 2
  rm -rf gpurun_out/bindprof_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bindprof_$tag -- oracle/_ref/vgl_hip_$app "$@" < /dev/null > gpurun_out/bindprof_$tag.log 2>&1
  local f; f=$(ls gpurun_out/bindprof_$tag/*/*kernel_stats.csv 2>/dev/null | tail -1)
  if [ -n "$f" ]; then cp "$f" gpurun_out/r05_binding_${tag}_kernel_stats.csv; fi
  rm -rf gpurun_out/bindprof_$tag
}
for fmt in csr vcsr; do
  apps/bin/create_vgl_graphs_hip -s 24 -e 32 -type rmat -format $fmt -file $D/rmat24 > /dev/null
  run oracle/_ref/vgl_hip_bfs -load $D/rmat24.$fmt -format $fmt -it 16
  run oracle/_ref/vgl_hip_sssp -load $D/rmat24.$fmt -format $fmt -it 2 -all-active
  run oracle/_ref/vgl_hip_sssp -load $D/rmat24.$fmt -format $fmt -it 2 -all-active -pull
  trace bfs_${fmt}_rmat24 bfs -load $D/rmat24.$fmt -format $fmt -it 8
  if [ $fmt = vcsr ]; then trace sssp_vcsr_rmat24 sssp -load $D/rmat24.$fmt -format $fmt -it 1 -all-active; fi
  rm -f $D/rmat24.$fmt
  apps/bin/create_vgl_graphs_hip -s 25 -e 32 -type ru -format $fmt -file $D/ru25 > /dev/null
  run oracle/_ref/vgl_hip_pr -load $D/ru25.$fmt -format $fmt -it 5
  rm -f $D/ru25.$fmt
  apps/bin/create_vgl_graphs_hip -s 24 -e 16 -type rmat -undirected -format $fmt -file $D/rmat24u > /dev/null
  run oracle/_ref/vgl_hip_cc -load $D/rmat24u.$fmt -format $fmt
  rm -f $D/rmat24u.$fmt
done
cat $LOG
