#!/bin/bash
# lead x lag grid of the cohort-pipelined multi-step launch (one chain, 64 steps per launch, 4096 x 180 polygons50): rate per pair.
# (AUV_HIP_LIB=<variant> in the environment measures a variant build.)  Output: gpurun_out/r05/lead_lag_grid.jsonl
mkdir -p gpurun_out/r05
python bench.py --steps 20 --bank-cache /tmp/bank --cpu-baseline 0 > /dev/null 2>&1
rm -f gpurun_out/r05/lead_lag_grid.jsonl
for lead in 10 12 14 16 20; do for lag in 22 26 30 34 40; do
  v=$(python bench.py --steps 2560 --warmup 256 --cpu-baseline 0 --bank-cache /tmp/bank --sub-batches 1 --multi 64 --multi-lead $lead --multi-lag $lag --probe-streams 0 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])")
  echo "{\"lead\": $lead, \"lag\": $lag, \"value\": $v}" | tee -a gpurun_out/r05/lead_lag_grid.jsonl
done; done
