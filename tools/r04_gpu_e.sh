#!/bin/bash
set -u
mkdir -p gpurun_out/r04
timeout -k 10 300 env AUV_HIP_LIB=gym_auv_amd/csrc/libauv_hip_hooks.so python tests/hooks_runner.py > gpurun_out/r04/hooks_runner.log 2>&1
echo "hooks rc=$?"
timeout -k 10 200 python tools/stack_capture_repro.py > gpurun_out/r04/stack_capture_repro.log 2>&1
echo "repro rc=$?"
tail -8 gpurun_out/r04/stack_capture_repro.log
