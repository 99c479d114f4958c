#!/bin/bash
# one_launch against side_by_side by launch size (one chain): where should AUV_STEP_AUTO switch?
B="python bench.py --bank-cache /tmp/bank --cpu-baseline 0 --worlds-per-env 1 --sub-batches 1"
for n in 8192 16384 32768; do for m in one_launch side_by_side; do
  $B --envs $n --step-mode $m --steps 200 --warmup 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('envs %6d %-13s %.1f M  %.4f ms/step' % (d['config']['envs_per_gpu'], d['config']['step_mode'], d['value']/1e6, d['ms_per_step']))"
done; done
