#!/bin/bash
# Round profile run (GPU box): headline bench, the other BASELINE workload shapes, rocprofv3 kernel trace of the
# headline command, and SEPARATE PMC passes (FETCH_SIZE / WRITE_SIZE for HBM traffic, SQ counters for the VALU leg).
# Usage: bash tools/run_profiles.sh <tag>      (outputs under gpurun_out/<tag>/; copy what is to be judged to profiles/<tag>/)
set -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
B="python bench.py --bank-cache /tmp/bank"
$B > $OUT/bench_polygons50.json 2> $OUT/bench_polygons50.err
echo "headline: $(cut -c1-120 $OUT/bench_polygons50.json)"
python bench.py --gpus 1 --steps 20 --warmup 5 --bank-cache /tmp/bank --cpu-baseline 0 > $OUT/bench_driver_command.json 2>/dev/null
$B --graph 16 --cpu-baseline 0 > $OUT/bench_polygons50_graph16.json 2>/dev/null
$B --worlds-per-env 1 --cpu-baseline 0 > $OUT/bench_polygons50_worlds1.json 2>/dev/null
$B --step-mode paired --cpu-baseline 0 > $OUT/bench_polygons50_paired.json 2>/dev/null
$B --step-mode side_by_side --cpu-baseline 0 > $OUT/bench_polygons50_side_by_side.json 2>/dev/null
$B --step-mode two_kernels --cpu-baseline 0 > $OUT/bench_polygons50_two_kernels.json 2>/dev/null
$B --step-mode two_streams --cpu-baseline 0 > $OUT/bench_polygons50_two_streams.json 2>/dev/null
$B --step-mode one_kernel --cpu-baseline 0 > $OUT/bench_polygons50_one_kernel.json 2>/dev/null
$B --actions pilot --cpu-baseline 0 > $OUT/bench_polygons50_pilot.json 2>/dev/null
$B --workload circles20 --cpu-baseline 0 > $OUT/bench_circles20.json 2>/dev/null
$B --workload moving28 --cpu-baseline 0 > $OUT/bench_moving28.json 2>/dev/null
$B --workload mixed47 --envs 8192 --cpu-baseline 0 > $OUT/bench_mixed47_8192.json 2>/dev/null
$B --workload mixed47 --envs 8192 --graph 16 --cpu-baseline 0 > $OUT/bench_mixed47_8192_graph16.json 2>/dev/null
$B --workload mixed47 --envs 8192 --cpu-baseline 0 > $OUT/bench_mixed47_8192_b.json 2>/dev/null
$B --workload mixed47 --envs 8192 --graph 16 --cpu-baseline 0 > $OUT/bench_mixed47_8192_graph16_b.json 2>/dev/null
$B --envs 32768 --steps 100 --warmup 20 --worlds-per-env 1 --cpu-baseline 0 > $OUT/bench_polygons50_32768.json 2>/dev/null
# two ranks started by bench.py itself, sharing the one GPU of this box (gloo instead of RCCL): the N > 1 code path on hardware
python bench.py --gpus 2 --rehearse 1 --steps 500 --warmup 100 --bank-cache /tmp/bank --cpu-baseline 0 > $OUT/bench_2ranks_rehearsal_1gpu.json 2> $OUT/bench_2ranks_rehearsal_1gpu.err
python bench.py --gpus 2 --steps 10 > $OUT/bench_2ranks_refused.out 2>&1; echo "exit code $? (two ranks without --rehearse on a 1-GPU box must be refused)" >> $OUT/bench_2ranks_refused.out
echo "benches done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --bank-cache /tmp/bank --cpu-baseline 0 > $OUT/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --bank-cache /tmp/bank --cpu-baseline 0 --steps 60 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --bank-cache /tmp/bank --cpu-baseline 0 --steps 60 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --bank-cache /tmp/bank --cpu-baseline 0 --steps 300 --warmup 1900 > /dev/null 2>&1
cd $ROOT
python tools/trace_summary.py $OUT/trace/*/*_kernel_trace.csv > $OUT/kernel_trace_summary.txt
cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.json
python tools/pmc_sq_summary.py $OUT/pmc_sq > $OUT/pmc_sq_summary.json
python tools/trace_summary.py $OUT/pmc_sq/*/*_kernel_trace.csv > $OUT/pmc_sq_kernel_durations.txt
head -8 $OUT/kernel_trace_summary.txt
cat $OUT/pmc_summary.json
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
