mkdir -p gpurun_out/r05
one() { tag=$1; shift
  timeout -k 10 300 python bench.py --bank-cache /tmp/bank --cpu-baseline 0 --workload mixed47 --envs 8192 "$@" > gpurun_out/r05/sweep_mixed_$tag.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r05/sweep_mixed_$tag.json')); print('%-26s %6.1f M ms/step %.5f' % ('$tag', d['value']/1e6, d['ms_per_step']))"; }
one sub4_T1 --sub-batches 4 --multi 1
one sub1_T64 --sub-batches 1 --multi 64
one sub2_T64 --sub-batches 2 --multi 64
one sub4_T64 --sub-batches 4 --multi 64
one sub2_T16 --sub-batches 2 --multi 16
one sub1_T64_l24_g60 --sub-batches 1 --multi 64 --multi-lead 24 --multi-lag 60
one sub1_T64_l6_g16 --sub-batches 1 --multi 64 --multi-lead 6 --multi-lag 16
for i in 1 2 3; do timeout -k 10 200 python -m pytest tests/test_gpu_policy.py -x -q -m gpu -k noise 2>&1 | tail -1; done
