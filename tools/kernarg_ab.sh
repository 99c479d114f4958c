#!/bin/bash
# Does it matter where the runtime keeps kernel arguments?  HIP_FORCE_DEV_KERNARG=1 (device memory) against =0 (host memory) for the
# launch-bound shapes: four one-step chains, env.step() on one chain, the policy-in-the-loop rollout, and the 64-steps-per-launch shape.
# (An environment variable of the HIP runtime, read at its initialisation: the library itself reads none.)
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/ab_dev_kernarg.jsonl
rm -f $OUT
python bench.py --steps 20 --bank-cache /tmp/bank --cpu-baseline 0 > /dev/null 2>&1
for r in 1 2; do for k in 0 1; do
  for spec in "chains4:--multi 1 --sub-batches 4" "step1:--api step --sub-batches 1" "multi64:--multi 64 --sub-batches 1"; do
    name=${spec%%:*}; flags=${spec#*:}
    HIP_FORCE_DEV_KERNARG=$k python bench.py --steps 1920 --warmup 192 --cpu-baseline 0 --bank-cache /tmp/bank $flags 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); c=d.get('comparison') or {}
print(json.dumps(dict(round=$r, dev_kernarg=$k, shape='$name', value=d['value'], policy_rollout={k:v for k,v in c.items() if k.startswith('policy_rollout_sub')}, one_chain_step=c.get('one_chain_step'))))" | tee -a $OUT
  done
done; done
