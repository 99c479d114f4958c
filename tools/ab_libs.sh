#!/bin/bash
# Interleaved A/B of library builds: bash tools/ab_libs.sh "<dir1> <dir2> ..." [rounds] [extra bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
LIBS=$1; R=${2:-3}; shift 2
python bench.py --bank-cache /tmp/bank --steps 20 --cpu-baseline 0 "$@" > /dev/null 2>&1
for rep in $(seq $R); do
for v in $LIBS; do
AUV_HIP_LIB=gym_auv_amd/$v/libauv_hip.so python bench.py --bank-cache /tmp/bank --cpu-baseline 0 "$@" 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print('$v', b['value'], b['ms_per_step'], {k:v['avg_ms'] for k,v in b['roofline']['kernels'].items()})"
done; done
