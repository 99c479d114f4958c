#!/bin/bash
# Round-4 sweep of the ways of driving the batch (open loop, VecEnv rendezvous by each mechanism, captured chains), one
# JSON line per run into $OUT (default gpurun_out/r04/sweep_api.jsonl).  Usage: tools/archive/r04_sweep.sh [steps] [warmup]
set -u
STEPS=${1:-2000}; WARM=${2:-200}
OUT=${OUT:-gpurun_out/r04/sweep_api.jsonl}
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
run() {
  echo "== $*" >&2
  timeout -k 10 300 python bench.py --steps "$STEPS" --warmup "$WARM" --cpu-baseline 0 "$@" 2>>"$OUT.err" | tail -n 1 |
    python -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); c=d['config']
    print(json.dumps(dict(args='$*', value_M=round(d['value']/1e6,2), ms_per_step=d['ms_per_step'], api=c['api'], sub=c['sub_batches'], loop=c['loop'], launch_ms=d['roofline']['kernels'], cmp=d.get('comparison'))))
except Exception as e:
    print(json.dumps(dict(args='$*', error=str(e), raw=l[:200])))
" >> "$OUT"
  tail -n 1 "$OUT" | cut -c1-260 >&2
}
run
run --api step
for rdv in events device cp; do
  for k in 4 2; do
    for inl in 0 1; do
      run --api async --sub-batches $k --rendezvous $rdv --inline-first $inl
    done
  done
  run --api async --sub-batches 1 --rendezvous $rdv --inline-first 0
done
run --api async --sub-batches 1 --inline-first 1
run --actions pilot --api step
run --actions pilot --sub-batches 2
run --actions pilot --sub-batches 4
for rdv in events device; do
  run --actions pilot --api async --sub-batches 4 --rendezvous $rdv --inline-first 1
  run --actions pilot --api async --sub-batches 2 --rendezvous $rdv --inline-first 1
done
run --graph 16 --sub-batches 1
run --graph 16 --sub-batches 4
run --graph 16 --sub-batches 4 --one-graph 1
run --graph 4 --sub-batches 4
run --workload mixed47 --envs 8192
run --workload mixed47 --envs 8192 --graph 16 --sub-batches 1
run --workload mixed47 --envs 8192 --graph 16 --sub-batches 4
run --workload mixed47 --envs 8192 --graph 16 --sub-batches 4 --one-graph 1
echo done >&2
