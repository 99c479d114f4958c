#!/bin/bash
# Round profile run (GPU box): headline bench and its variants, the other BASELINE workload shapes, a rocprofv3 kernel
# trace of the headline command (kernel stats + how much the sub-batch chains overlap), and SEPARATE PMC passes
# (FETCH_SIZE / WRITE_SIZE for HBM traffic, SQ counters for the VALU leg) of the headline configuration.
# Usage: bash tools/run_profiles.sh <tag>   (outputs under gpurun_out/<tag>/; tools/publish_profiles.py copies what is judged)
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
sha256sum gym_auv_amd/csrc/libauv_hip.so > $OUT/lib_sha256.txt
B="python bench.py --bank-cache /tmp/bank"
$B > $OUT/bench_polygons50.json 2> $OUT/bench_polygons50.err
echo "headline: $(cut -c1-140 $OUT/bench_polygons50.json)"
python bench.py --gpus 1 --steps 20 --warmup 5 --bank-cache /tmp/bank --cpu-baseline 0 > $OUT/bench_driver_command.json 2>/dev/null
for k in 1 2 3; do $B --sub-batches $k --cpu-baseline 0 > $OUT/bench_polygons50_sub$k.json 2>/dev/null; done
$B --step-mode side_by_side --sub-batches 1 --cpu-baseline 0 > $OUT/bench_polygons50_side_by_side_sub1.json 2>/dev/null
$B --step-mode side_by_side --sub-batches 4 --cpu-baseline 0 > $OUT/bench_polygons50_side_by_side_sub4.json 2>/dev/null
$B --graph 16 --cpu-baseline 0 > $OUT/bench_polygons50_graph16.json 2>/dev/null
$B --worlds-per-env 1 --cpu-baseline 0 > $OUT/bench_polygons50_worlds1.json 2>/dev/null
$B --actions pilot --cpu-baseline 0 > $OUT/bench_polygons50_pilot_sub1.json 2>/dev/null
$B --actions pilot --sub-batches 2 --cpu-baseline 0 > $OUT/bench_polygons50_pilot_sub2.json 2>/dev/null
$B --workload circles20 --cpu-baseline 0 > $OUT/bench_circles20.json 2>/dev/null
$B --workload moving28 --cpu-baseline 0 > $OUT/bench_moving28.json 2>/dev/null
$B --workload mixed47 --envs 8192 --cpu-baseline 0 > $OUT/bench_mixed47_8192.json 2>/dev/null
$B --workload mixed47 --envs 8192 --graph 16 --cpu-baseline 0 > $OUT/bench_mixed47_8192_graph16.json 2>/dev/null
$B --envs 32768 --steps 100 --warmup 20 --worlds-per-env 1 --cpu-baseline 0 > $OUT/bench_polygons50_32768.json 2>/dev/null
# two ranks started by bench.py itself, sharing the one GPU of this box (gloo instead of RCCL): the N > 1 code path on hardware
python bench.py --gpus 2 --rehearse 1 --steps 500 --warmup 100 --bank-cache /tmp/bank --cpu-baseline 0 > $OUT/bench_2ranks_rehearsal_1gpu.json 2> $OUT/bench_2ranks_rehearsal_1gpu.err
python bench.py --gpus 2 --steps 10 > $OUT/bench_2ranks_refused.out 2>&1; echo "exit code $? (two ranks without --rehearse on a 1-GPU box must be refused)" >> $OUT/bench_2ranks_refused.out
echo "benches done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --bank-cache /tmp/bank --cpu-baseline 0 > $OUT/bench_under_rocprof.json 2>/dev/null
cd $ROOT
python tools/trace_summary.py $OUT/trace/*/*_kernel_trace.csv > $OUT/kernel_trace_summary.txt
python tools/trace_overlap.py $OUT/trace/*/*_kernel_trace.csv k_step_roles 4000 > $OUT/kernel_trace_overlap.txt
cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace
head -6 $OUT/kernel_trace_summary.txt; head -12 $OUT/kernel_trace_overlap.txt
bash tools/pmc_workload.sh $TAG polygons50 4
