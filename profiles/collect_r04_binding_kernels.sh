#!/bin/bash
# Kernel trace of the REFERENCE'S OWN apps with the HIP backend bound in (which kernels run, how long): rocprofv3 runs the app binary directly.
# usage: bash profiles/collect_r04_binding_kernels.sh [vcsr]  (writes gpurun_out/bindprof_<app>_<format>_kernel_stats.csv)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp VGL_HIP_DEVICE_ARRAYS=1
one() { # <tag> <app> <args...>
  local tag=$1 app=$2; shift 2
  rm -rf gpurun_out/bindprof_$tag
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bindprof_$tag -- oracle/_ref/vgl_hip_$app "$@" < /dev/null > gpurun_out/bindprof_$tag.log 2>&1
  local f; f=$(ls gpurun_out/bindprof_$tag/*/*kernel_stats.csv 2>/dev/null | tail -1)
  if [ -n "$f" ]; then cp "$f" gpurun_out/bindprof_${tag}_kernel_stats.csv; fi
  grep AVG_PERF gpurun_out/bindprof_$tag.log
  rm -rf gpurun_out/bindprof_$tag
}
if [ "$1" = "vcsr" ]; then one bfs_vcsr bfs -s 20 -e 32 -type rmat -format vcsr -it 8; exit 0; fi
one bfs_csr bfs -s 20 -e 32 -type rmat -format csr -it 8
one bfs_vcsr bfs -s 20 -e 32 -type rmat -format vcsr -it 8
one sssp_csr sssp -s 20 -e 32 -type rmat -format csr -it 4
one pr_csr pr -s 20 -e 32 -type ru -format csr -it 5
