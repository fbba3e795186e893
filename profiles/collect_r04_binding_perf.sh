#!/bin/bash
# Rates of the REFERENCE'S OWN apps (oracle/_ref/vgl_hip_<app>, built by `make -C oracle binding`) with the HIP backend bound in.
# usage: bash profiles/collect_r04_binding_perf.sh <scale> <out.log>
S=${1:-20}; OUT=${2:-gpurun_out/binding_perf.log}
: > "$OUT"
run() { # <device arrays 0|1> <app> <args...>
  local d=$1 app=$2; shift 2
  echo "=== device arrays $d: $app $*" >> "$OUT"
  VGL_HIP_DEVICE_ARRAYS=$d timeout -k 10 240 oracle/_ref/vgl_hip_$app "$@" < /dev/null 2>&1 | grep AVG_PERF >> "$OUT" || echo "no AVG_PERF" >> "$OUT"
}
run 1 bfs -s $S -e 32 -type rmat -format csr -it 8
run 1 bfs -s $S -e 32 -type rmat -format vcsr -it 8
run 1 pr -s $S -e 32 -type ru -format csr -it 5
run 1 sssp -s $S -e 32 -type rmat -format csr -it 4
run 1 cc -s $S -e 16 -type rmat -format vcsr
run 1 hits -s $S -e 32 -type rmat -format csr -it 3
run 1 sswp -s $S -e 32 -type rmat -format csr -it 3
if [ -n "${BIG:-}" ]; then run 1 bfs -s $BIG -e 32 -type rmat -format csr -it 8; fi
cat "$OUT"
