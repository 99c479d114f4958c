#!/bin/bash
# Interleaved A/B timing of two builds of libauv_hip.so on the GPU box (run-to-run spread of a
# single bench is ~5 %, so variants are compared by alternating runs and taking medians).
# Usage: bash tools/ab_bench.sh <libA.so> <libB.so> [rounds] [extra bench args...]
A=$1; B=$2; R=${3:-5}; shift 3
python bench.py --bank-cache /tmp/bank --steps 50 --cpu-baseline 0 "$@" > /dev/null 2>&1   # builds the world cache
for i in $(seq $R); do
  for L in $A $B; do
    AUV_HIP_LIB=$L python bench.py --bank-cache /tmp/bank --cpu-baseline 0 --steps 1500 --warmup 150 "$@" 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('$L', d['value'], d['ms_per_step'], k['k1_dynamics']['avg_ms'], k['k2_lidar']['avg_ms'], k['k3_nav_reward']['avg_ms'])"
  done
done | python -c "
import sys, collections, statistics
v = collections.defaultdict(list)
for line in sys.stdin:
    p = line.split(); v[p[0]].append([float(x) for x in p[1:]])
for k, rows in v.items():
    cols = list(zip(*rows))
    print('%-40s env-steps/s median %.3e (min %.3e max %.3e)  ms/step %.5f  K1/K2/K3 ms %.4f %.4f %.4f' % (k, statistics.median(cols[0]), min(cols[0]), max(cols[0]), statistics.median(cols[1]), statistics.median(cols[2]), statistics.median(cols[3]), statistics.median(cols[4])))
"
