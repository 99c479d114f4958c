#!/bin/bash
set -u
mkdir -p gpurun_out/r04
step() {
  local name=$1 lim=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/r04/progress.log
  timeout -k 10 "$lim" "$@" > "gpurun_out/r04/$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/r04/progress.log
  tail -n 5 "gpurun_out/r04/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIME LIMIT in $name: stopping" | tee -a gpurun_out/r04/progress.log; exit 3; fi
  return 0
}
step pytest_policy4 400 python -m pytest tests/test_gpu_policy.py -x -q -m gpu
step policy_bench3 300 python tools/policy_bench.py 4096
step pytest_nav 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edge.py -x -q -m gpu
OUT=gpurun_out/r04/ab_navsqrt.jsonl; : > $OUT
for rep in 1 2; do
for lib in "" "gym_auv_amd/csrc_navsqrt/libauv_hip.so"; do
for a in "--sub-batches 4" "--sub-batches 1" "--envs 32768 --steps 400 --warmup 50"; do
  AUV_HIP_LIB=$lib timeout -k 10 300 python bench.py --cpu-baseline 0 $a 2>>$OUT.err | tail -n 1 | python -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); print(json.dumps(dict(lib='$lib' or 'product(no sqrt)', args='$a', value_M=round(d['value']/1e6,2), ms_per_step=d['ms_per_step'])))
except Exception as e:
    print(json.dumps(dict(args='$a', error=str(e), raw=l[:300])))
" | tee -a $OUT
done; done; done
step ppo_colav_fused4 300 python examples/ppo.py --envs 4096 --updates 5 --rollout 64 --fused-policy 1
step ppo_gu_mode3 600 python examples/ppo.py --envs 2048 --updates 40 --rollout 64 --graph-update 3 --log-every 10
step ppo_gu_mode1_churn 600 python examples/ppo.py --envs 2048 --updates 40 --rollout 64 --graph-update 1 --log-every 10
