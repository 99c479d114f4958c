#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python bench.py --bank-cache /tmp/bank --steps 20 --cpu-baseline 0 > /dev/null 2>&1
for v in csrc csrc_e4 csrc_e2; do
AUV_HIP_LIB=gym_auv_amd/$v/libauv_hip.so python bench.py --bank-cache /tmp/bank --cpu-baseline 0 --step-mode two_kernels 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print('$v', b['value'], b['ms_per_step'], {k:v['avg_ms'] for k,v in b['roofline']['kernels'].items()})"
done
for v in stamps stamps4 stamps2; do
echo "== $v"; STEPS=2000 AUV_HIP_LIB=gym_auv_amd/csrc_$v/libauv_hip.so python tools/phase_stamps2.py 2>&1 | grep -E "k1n|nav\."
done
