#!/bin/bash
# Interleaved timing of step launch shapes (bench.py --step-mode) on the GPU box.
# Usage: bash tools/ab_modes.sh "<mode1> <mode2> ..." [rounds] [extra bench args...]
MODES=$1; R=${2:-5}; shift 2
python bench.py --bank-cache /tmp/bank --steps 50 --cpu-baseline 0 "$@" > /dev/null 2>&1   # builds the world cache
for i in $(seq $R); do
  for M in $MODES; do
    python bench.py --bank-cache /tmp/bank --cpu-baseline 0 --steps 1500 --warmup 150 --step-mode $M "$@" 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$M', d['value'], d['ms_per_step'])"
  done
done | python -c "
import sys, collections, statistics
v = collections.defaultdict(list)
for line in sys.stdin:
    p = line.split(); v[p[0]].append([float(x) for x in p[1:]])
for k, rows in v.items():
    cols = list(zip(*rows))
    print('%-16s env-steps/s median %.3e (min %.3e max %.3e)  ms/step %.5f' % (k, statistics.median(cols[0]), min(cols[0]), max(cols[0]), statistics.median(cols[1])))
"
