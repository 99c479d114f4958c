#!/bin/bash
# One GPU-box call that answers "is it still right and how fast is it": the -m gpu suite, then the headline bench,
# the driver's short command and the configs[4] shard.  bash tools/gpu_check.sh [tag]
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/${1:-check}; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for a in "" "--steps 20 --warmup 5" "--workload mixed47 --envs 8192"; do
python bench.py --bank-cache /tmp/bank --cpu-baseline 0 $a 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print('$a', b['value'], b['ms_per_step'], b['config']['episodes_finished'], {k:v['avg_ms'] for k,v in b['roofline']['kernels'].items()})"
done
