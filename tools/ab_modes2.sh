#!/bin/bash
# Interleaved A/B of step modes (and library builds): bash tools/ab_modes2.sh "<lib:mode> ..." [rounds] [extra bench args]
#   e.g. bash tools/ab_modes2.sh "csrc:side_by_side csrc:paired csrc_nowt:paired" 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
SPECS=$1; R=${2:-3}; shift 2
python bench.py --bank-cache /tmp/bank --steps 20 --cpu-baseline 0 "$@" > /dev/null 2>&1
for rep in $(seq $R); do
for sp in $SPECS; do
lib=${sp%%:*}; mode=${sp##*:}
AUV_HIP_LIB=gym_auv_amd/$lib/libauv_hip.so python bench.py --bank-cache /tmp/bank --cpu-baseline 0 --step-mode $mode "$@" 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print('$sp', b['value'], b['ms_per_step'], {k:v['avg_ms'] for k,v in b['roofline']['kernels'].items()})"
done; done
