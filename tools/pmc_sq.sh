#!/bin/bash
# SQ counters for the step kernels (one pass, 8 SQ slots): the VALU leg of the roofline and where
# the wave-cycles go.  Usage: bash tools/pmc_sq.sh <tag> [extra bench.py flags]
# Output: gpurun_out/<tag>/pmc_sq/ (raw csv) and gpurun_out/<tag>/pmc_sq_summary.json
TAG=${1:-r02}
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
python $ROOT/bench.py --bank-cache /tmp/bank --steps 20 --cpu-baseline 0 "$@" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --bank-cache /tmp/bank --cpu-baseline 0 --steps 60 --warmup 10 "$@" > $OUT/pmc_sq_bench.json 2> $OUT/pmc_sq_bench.err
cd $ROOT
python tools/pmc_sq_summary.py $OUT/pmc_sq > $OUT/pmc_sq_summary.json
cat $OUT/pmc_sq_summary.json
