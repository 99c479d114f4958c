#!/bin/bash
# one gpurun call: new tests, then the API sweep.  A step that was killed at its time limit ends the call (no GPU work after a hang).
set -u
mkdir -p gpurun_out/r04
step() {  # name, limit, command...
  local name=$1 lim=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/r04/progress.log
  timeout -k 10 "$lim" "$@" > "gpurun_out/r04/$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/r04/progress.log
  tail -n 6 "gpurun_out/r04/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT in $name: stopping" | tee -a gpurun_out/r04/progress.log; exit 3; fi
  return 0
}
step pytest_parity_new 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sub_batches or rendezvous or converts or overflow"
step pytest_fullsize_new 500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "bench_launch or captured_chains"
OUT=gpurun_out/r04/sweep_api.jsonl tools/r04_sweep.sh 2000 200 2> gpurun_out/r04/sweep_api.progress
cat gpurun_out/r04/sweep_api.jsonl | cut -c1-200
