#!/bin/bash
# gpurun call B: policy tests, PPO rates (fused / torch), captured chains at 16 steps per replay
set -u
mkdir -p gpurun_out/r04
step() {
  local name=$1 lim=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/r04/progress.log
  timeout -k 10 "$lim" "$@" > "gpurun_out/r04/$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/r04/progress.log
  tail -n 8 "gpurun_out/r04/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT in $name: stopping" | tee -a gpurun_out/r04/progress.log; exit 3; fi
  return 0
}
step pytest_policy 400 python -m pytest tests/test_gpu_policy.py -x -q -m gpu
step pytest_vecenv 300 python -m pytest tests/test_gpu_vecenv.py -x -q -m gpu
step ppo_colav_fused 300 python examples/ppo.py --envs 4096 --updates 6 --rollout 64 --fused-policy 1
step ppo_colav_torch 300 python examples/ppo.py --envs 4096 --updates 4 --rollout 64 --fused-policy 0
step ppo_colav_torch_graph 300 python examples/ppo.py --envs 4096 --updates 4 --rollout 64 --fused-policy 0 --graph-rollout 1
step ppo_colav_fused_sub2 300 python examples/ppo.py --envs 4096 --updates 4 --rollout 64 --fused-policy 1 --sub-batches 2
step ppo_colav_fused_sub1 300 python examples/ppo.py --envs 4096 --updates 4 --rollout 64 --fused-policy 1 --sub-batches 1
step pytest_ppo 900 python -m pytest tests/test_gpu_ppo.py -x -q -m gpu
OUT=gpurun_out/r04/sweep_graph.jsonl
: > $OUT
for a in "--graph 16 --sub-batches 4" "--graph 16 --sub-batches 4 --one-graph 1" "--graph 8 --sub-batches 4" "--graph 64 --sub-batches 4" \
         "--workload mixed47 --envs 8192 --graph 16 --sub-batches 4" "--workload mixed47 --envs 8192 --graph 16 --sub-batches 4 --one-graph 1" \
         "--workload mixed47 --envs 8192 --graph 4 --sub-batches 4" "--workload mixed47 --envs 8192"; do
  echo "== $a"
  timeout -k 10 300 python bench.py --steps 1920 --warmup 192 --cpu-baseline 0 $a 2>>$OUT.err | tail -n 1 | python -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); print(json.dumps(dict(args='$a', value_M=round(d['value']/1e6,2), ms_per_step=d['ms_per_step'], loop=d['config']['loop'])))
except Exception as e:
    print(json.dumps(dict(args='$a', error=str(e), raw=l[:300])))
" | tee -a $OUT
done
