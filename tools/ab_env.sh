#!/bin/bash
# Interleaved timing of bench.py under different values of one environment variable.
# Usage: bash tools/ab_env.sh VAR "<v1> <v2> ..." [rounds] [extra bench args...]
VAR=$1; VALS=$2; R=${3:-5}; shift 3
python bench.py --bank-cache /tmp/bank --steps 50 --cpu-baseline 0 "$@" > /dev/null 2>&1   # builds the world cache
for i in $(seq $R); do
  for V in $VALS; do
    env $VAR=$V python bench.py --bank-cache /tmp/bank --cpu-baseline 0 --steps 1500 --warmup 150 "$@" 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('$VAR=$V', d['value'], d['ms_per_step'], ' '.join('%s=%.4f' % (n, x['avg_ms']) for n, x in k.items()))"
  done
done | python -c "
import sys, collections, statistics
v = collections.defaultdict(list); extra = {}
for line in sys.stdin:
    p = line.split(); v[p[0]].append([float(x) for x in p[1:3]]); extra[p[0]] = ' '.join(p[3:])
for k, rows in v.items():
    cols = list(zip(*rows))
    print('%-20s env-steps/s median %.3e (min %.3e max %.3e)  ms/step %.5f   last: %s' % (k, statistics.median(cols[0]), min(cols[0]), max(cols[0]), statistics.median(cols[1]), extra[k]))
"
