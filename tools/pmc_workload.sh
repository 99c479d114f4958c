#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE, SQ counters: three separate runs) for ONE bench workload, published into
# profiles/pmc_traffic.json and profiles/pmc_sq.json under that workload's key.
# Usage (GPU box): bash tools/pmc_workload.sh <tag> <workload> [extra bench flags]
TAG=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python bench.py --bank-cache /tmp/bank --workload $WL --steps 20 --cpu-baseline 0 "$@" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --bank-cache /tmp/bank --workload $WL --cpu-baseline 0 --steps 60 "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --bank-cache /tmp/bank --workload $WL --cpu-baseline 0 --steps 60 "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --bank-cache /tmp/bank --workload $WL --cpu-baseline 0 --steps 300 --warmup 1900 "$@" > /dev/null 2>&1
cd $ROOT
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.json
python tools/pmc_sq_summary.py $OUT/pmc_sq > $OUT/pmc_sq_summary.json
python tools/trace_summary.py $OUT/pmc_sq/*/*_kernel_trace.csv > $OUT/kernel_trace_summary.txt
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
head -3 $OUT/kernel_trace_summary.txt
