#!/bin/bash
set -u
mkdir -p gpurun_out/r04
step() {
  local name=$1 lim=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/r04/progress.log
  timeout -k 10 "$lim" "$@" > "gpurun_out/r04/$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/r04/progress.log
  tail -n 5 "gpurun_out/r04/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT in $name: stopping" | tee -a gpurun_out/r04/progress.log; exit 3; fi
  return 0
}
step pytest_all_gpu 1100 python -m pytest tests -x -q -m gpu
