#!/bin/bash
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
step pytest_policy3 400 python -m pytest tests/test_gpu_policy.py -x -q -m gpu
step policy_bench2 300 python tools/policy_bench.py 4096
step pytest_hooks 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hook or sub_batches or rendezvous"
step ppo_colav_fused3 300 python examples/ppo.py --envs 4096 --updates 5 --rollout 64 --fused-policy 1
step ppo_gu_colav_stackfree 600 python examples/ppo.py --envs 2048 --updates 60 --rollout 64 --graph-update 1 --log-every 10
step ppo_gu_colav_stacked 600 python examples/ppo.py --envs 2048 --updates 30 --rollout 64 --graph-update 2 --log-every 10
step bench_default 300 python bench.py --steps 20 --warmup 5
step bench_long 300 python bench.py --cpu-baseline 0
