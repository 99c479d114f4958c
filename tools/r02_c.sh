#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=gpurun_out/${TAG:-r02_c}; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for m in two_kernels side_by_side two_kernels side_by_side; do
python bench.py --bank-cache /tmp/bank --cpu-baseline 0 --step-mode $m > $O/bench_$m.json 2>$O/bench_$m.err; python - <<PY
import json; b=json.load(open("$O/bench_$m.json")); print("$m", b["value"], b["ms_per_step"], {k:v["avg_ms"] for k,v in b["roofline"]["kernels"].items()})
PY
done
STEPS=2000 AUV_HIP_LIB=gym_auv_amd/csrc_stamps/libauv_hip.so python tools/phase_stamps2.py > $O/stamps.log 2>&1; cat $O/stamps.log
