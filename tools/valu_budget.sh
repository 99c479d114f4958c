#!/bin/bash
# Per-phase VALU budget of the one-launch step on the GPU box (see tools/valu_budget.py).
# usage: bash tools/valu_budget.sh <tag>      (needs tools/build_variant.sh cuts "-DAUV_CUTS" run beforehand, here or there)
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
[ -f gym_auv_amd/csrc_cuts/libauv_hip.so ] || bash tools/build_variant.sh cuts "-DAUV_CUTS"
python bench.py --bank-cache /tmp/bank --steps 20 --cpu-baseline 0 > /dev/null 2>&1     # (generates / caches the bank)
export AUV_HIP_LIB=$ROOT/gym_auv_amd/csrc_cuts/libauv_hip.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/valu_budget_raw -- python3 $ROOT/tools/valu_budget.py > $OUT/valu_budget_run.json 2> $OUT/valu_budget_run.err
cd $ROOT
python tools/valu_budget_summary.py $OUT/valu_budget_raw > $OUT/valu_budget.json
python tools/trace_summary.py $OUT/valu_budget_raw/*/*_kernel_trace.csv > $OUT/valu_budget_kernel_durations.txt
rm -rf $OUT/valu_budget_raw
cat $OUT/valu_budget.json
