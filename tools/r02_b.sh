#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/r02_b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for m in two_kernels side_by_side two_kernels side_by_side; do
python bench.py --bank-cache /tmp/bank --cpu-baseline 0 --step-mode $m > $O/bench_$m.json 2>$O/bench_$m.err; python - <<PY
import json; b=json.load(open("$O/bench_$m.json")); print("$m", b["value"], b["ms_per_step"], {k:v["avg_ms"] for k,v in b["roofline"]["kernels"].items()})
PY
done
