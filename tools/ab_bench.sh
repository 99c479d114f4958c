#!/bin/bash
# A/B of library builds on ONE box, alternating (boxes differ by 1-2 %): tools/ab_bench.sh <out.jsonl> <rounds> <sub> <extra bench args> -- <name=lib.so> ...
#   e.g. tools/ab_bench.sh gpurun_out/r04/ab.jsonl 3 4 "" -- base=gym_auv_amd/csrc_base/libauv_hip.so new=gym_auv_amd/csrc/libauv_hip.so
OUT=$1; R=$2; SUB=$3; EXTRA=$4; shift 5
for r in $(seq 1 $R); do
  for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    v=$(AUV_HIP_LIB=$lib python bench.py --steps 3000 --warmup 300 --cpu-baseline 0 --bank-cache /tmp/bank --sub-batches $SUB $EXTRA 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])")
    echo "{\"round\": $r, \"sub\": $SUB, \"extra\": \"$EXTRA\", \"lib\": \"$name\", \"value\": $v}" | tee -a $OUT
  done
done
