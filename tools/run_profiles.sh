#!/bin/bash
# Round profile run (GPU box): headline bench, the other BASELINE workload shapes, rocprofv3
# kernel trace of the headline command, and separate PMC passes for HBM traffic.
# Usage: bash tools/run_profiles.sh <tag>      (outputs under gpurun_out/<tag>/)
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python bench.py --bank-cache /tmp/bank > $OUT/bench_polygons50.json 2> $OUT/bench_polygons50.err
echo "headline: $(cut -c1-120 $OUT/bench_polygons50.json)"
python bench.py --bank-cache /tmp/bank --graph 1 --cpu-baseline 0 > $OUT/bench_polygons50_graph.json 2>/dev/null
python bench.py --bank-cache /tmp/bank --step-mode two_streams --cpu-baseline 0 > $OUT/bench_polygons50_two_streams.json 2>/dev/null
python bench.py --bank-cache /tmp/bank --step-mode one_kernel --cpu-baseline 0 > $OUT/bench_polygons50_one_kernel.json 2>/dev/null
python bench.py --bank-cache /tmp/bank --workload circles20 --cpu-baseline 0 > $OUT/bench_circles20.json 2>/dev/null
python bench.py --bank-cache /tmp/bank --workload moving28 --cpu-baseline 0 > $OUT/bench_moving28.json 2>/dev/null
python bench.py --bank-cache /tmp/bank --workload mixed47 --envs 8192 --cpu-baseline 0 > $OUT/bench_mixed47_8192.json 2>/dev/null
python bench.py --bank-cache /tmp/bank --envs 32768 --steps 100 --cpu-baseline 0 > $OUT/bench_polygons50_32768.json 2>/dev/null
echo "benches done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --bank-cache /tmp/bank --cpu-baseline 0 > $OUT/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --bank-cache /tmp/bank --cpu-baseline 0 --steps 60 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --bank-cache /tmp/bank --cpu-baseline 0 --steps 60 > /dev/null 2>&1
cd $ROOT
python tools/trace_summary.py $OUT/trace/*/*_kernel_trace.csv > $OUT/kernel_trace_summary.txt
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.json
head -8 $OUT/kernel_trace_summary.txt
cat $OUT/pmc_summary.json
