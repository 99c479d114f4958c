#!/bin/bash
set -u
mkdir -p gpurun_out/r04
for lib in "" gym_auv_amd/csrc_w4/libauv_hip.so gym_auv_amd/csrc_w16/libauv_hip.so; do
  echo "== lib: ${lib:-product}"
  AUV_HIP_LIB=$lib timeout -k 10 200 python tools/policy_bench.py 4096 2>&1 | grep envs
done | tee gpurun_out/r04/policy_bench5.log
timeout -k 10 200 python -m pytest tests/test_gpu_policy.py -x -q -m gpu 2>&1 | tail -2
timeout -k 10 300 python examples/ppo.py --envs 4096 --updates 5 --rollout 64 --fused-policy 1 2>&1 | grep "^update" | sed -e 's/.*| rollout/rollout/' | tail -2
