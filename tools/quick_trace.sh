#!/bin/bash
# Kernel trace of a short headline bench (GPU box): per-kernel durations and the gaps between
# consecutive kernels of a step.  Usage: bash tools/quick_trace.sh <tag> [bench args...]
TAG=${1:-qt}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python bench.py --bank-cache /tmp/bank --steps 50 --cpu-baseline 0 "$@" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --bank-cache /tmp/bank --cpu-baseline 0 --steps 600 --warmup 100 "$@" > $OUT/bench.json 2>/dev/null
cd $ROOT
python tools/trace_summary.py $OUT/trace/*/*_kernel_trace.csv > $OUT/summary.txt
python tools/trace_overlap.py $OUT/trace/*/*_kernel_trace.csv >> $OUT/summary.txt 2>&1
head -30 $OUT/summary.txt
