#!/bin/bash
# how often does each bound-in app of the reference fail its own -check on time-seeded graphs (a flaky test would stop the round-end suite)
cd "$(dirname "$0")/.." || exit 1
N=${1:-40}
run() { # name, expected "error count: 0" occurrences, command...
  local name=$1 want=$2; shift 2
  local bad=0
  for i in $(seq 1 $N); do
    out=$("$@" 2>&1)
    got=$(echo "$out" | grep -c "error count: 0")
    if [ "$got" -ne "$want" ] || echo "$out" | grep -q "rror in\|NOT equal"; then bad=$((bad+1)); echo "$out" | tail -5 > gpurun_out/flaky_${name}_$i.txt; fi
  done
  echo "$name: $bad failures in $N runs" | tee -a gpurun_out/flaky.log
}
: > gpurun_out/flaky.log
R=oracle/_ref
for fmt in csr vcsr; do
  run bfs_$fmt 4 $R/vgl_hip_bfs -s 14 -e 16 -type rmat -format $fmt -check -it 4
  run sswp_$fmt 1 $R/vgl_hip_sswp -s 13 -e 16 -type rmat -format $fmt -check -it 2
  run hits_$fmt 2 $R/vgl_hip_hits -s 12 -e 16 -type rmat -format $fmt -check -it 5
  run pr_$fmt 1 $R/vgl_hip_pr -s 12 -e 16 -type rmat -format $fmt -check -it 5
  run sssp_push_$fmt 2 $R/vgl_hip_sssp -s 12 -e 16 -type rmat -format $fmt -check -it 2 -all-active
  run sssp_partial_$fmt 2 $R/vgl_hip_sssp -s 12 -e 16 -type rmat -format $fmt -check -it 2
  run sssp_pull_$fmt 2 $R/vgl_hip_sssp -s 12 -e 16 -type rmat -format $fmt -check -it 2 -all-active -pull
  run cc_$fmt 1 $R/vgl_hip_cc -s 12 -e 16 -type rmat -format $fmt -check
  run coloring_$fmt 1 $R/vgl_hip_coloring -s 12 -e 8 -type rmat -format $fmt -check
done


bad=0; for i in $(seq 1 $N); do if ! oracle/_ref/vgl_hip_tc_check -s 11 -e 8 -type rmat -format csr -it 24 2>&1 | grep -q "TC CHECK PASSED"; then bad=$((bad+1)); fi; done; echo "tc_check: $bad failures in $N runs" | tee -a gpurun_out/flaky.log
bad=0; for i in $(seq 1 $N); do if ! oracle/_ref/vgl_hip_plan_stamp_check -s 12 -e 16 -type rmat -format csr 2>&1 | grep -q "PLAN STAMP CHECK PASSED"; then bad=$((bad+1)); fi; done; echo "plan_stamp_check csr: $bad failures in $N runs" | tee -a gpurun_out/flaky.log
bad=0; for i in $(seq 1 $N); do if ! oracle/_ref/vgl_hip_plan_stamp_check -s 12 -e 16 -type rmat -format vcsr 2>&1 | grep -q "PLAN STAMP CHECK PASSED"; then bad=$((bad+1)); fi; done; echo "plan_stamp_check vcsr: $bad failures in $N runs" | tee -a gpurun_out/flaky.log
