#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/${TAG:-r02_j}; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
python bench.py --bank-cache /tmp/bank > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"; cut -c1-900 $O/bench.json
for a in "--graph 16" "--graph 64" "--worlds-per-env 1" "--graph 0"; do
python bench.py --bank-cache /tmp/bank --cpu-baseline 0 $a 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print('$a', b['value'], b['ms_per_step'], b['config']['episodes_finished'], {k:v['avg_ms'] for k,v in b['roofline']['kernels'].items()})"
done
