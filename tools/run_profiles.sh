#!/bin/bash
# Round profile run (GPU box): headline bench and its variants, the other BASELINE workload shapes, a rocprofv3 kernel
# trace of the headline command (kernel stats + how much the sub-batch chains overlap), and SEPARATE PMC passes
# (FETCH_SIZE / WRITE_SIZE for HBM traffic, SQ counters for the VALU leg) of the headline configuration.
# Usage: bash tools/run_profiles.sh <tag>   (outputs under gpurun_out/<tag>/; tools/publish_profiles.py copies what is judged)
set -o pipefail
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
sha256sum gym_auv_amd/csrc/libauv_hip.so > $OUT/lib_sha256.txt
# Every pass says when it starts AND when it ends (a long pass otherwise looks hung to the GPU runner: round 4 lost two calls
# to 420 s of silence), and the benches' stderr is APPENDED to $OUT/stderr.log under the pass's name -- never /dev/null.
ERR=$OUT/stderr.log
[ -n "$ONLY$NOT" ] || : > $ERR
run() {   # run <name> <stdout file> <command ...>
  local name=$1 out=$2; shift 2
  if [ -n "$ONLY" ] && ! [[ $name =~ $ONLY ]]; then return 0; fi     # ONLY=<regex> / NOT=<regex>: a round's passes split over GPU calls
  if [ -n "$NOT" ] && [[ $name =~ $NOT ]]; then return 0; fi
  echo "  start $name $(date +%T)"
  echo "==== $name: $*" >> $ERR
  "$@" > $out 2>> $ERR
  local rc=$?
  echo "  done  $name rc=$rc $(date +%T)"
  [ $rc = 0 ] || tail -3 $ERR
  return 0
}
B="python bench.py --bank-cache /tmp/bank"
run bench_polygons50 $OUT/bench_polygons50.json $B
echo "headline: $(cut -c1-140 $OUT/bench_polygons50.json)"
run bench_driver_command $OUT/bench_driver_command.json python bench.py --gpus 1 --steps 20 --warmup 5 --bank-cache /tmp/bank --cpu-baseline 0
# several steps per launch (round 5): one chain x 64 / 16 steps, two chains x 64, one step per launch on four chains; the rate table by order
run bench_polygons50_multi64_sub1 $OUT/bench_polygons50_multi64_sub1.json $B --multi 64 --sub-batches 1 --cpu-baseline 0
run bench_polygons50_multi16_sub1 $OUT/bench_polygons50_multi16_sub1.json $B --multi 16 --sub-batches 1 --cpu-baseline 0
run bench_polygons50_multi64_sub2 $OUT/bench_polygons50_multi64_sub2.json $B --multi 64 --sub-batches 2 --cpu-baseline 0
run bench_polygons50_multi1_sub4 $OUT/bench_polygons50_multi1_sub4.json $B --multi 1 --sub-batches 4 --cpu-baseline 0
run multi_bench_polygons50 $OUT/multi_bench_polygons50.jsonl python tools/multi_bench.py polygons50 4096
for k in 1 2; do run bench_polygons50_sub$k $OUT/bench_polygons50_sub$k.json $B --sub-batches $k --cpu-baseline 0; done
# the VecEnv protocol: a full rendezvous per step (round 4), by mechanism and chain count; one chain on the caller's stream
run bench_polygons50_async_inline_sub1 $OUT/bench_polygons50_async_inline_sub1.json $B --api async --sub-batches 1 --inline-first 1 --cpu-baseline 0
for k in 2 4; do for r in device events cp; do
  run bench_polygons50_async_${r}_sub$k $OUT/bench_polygons50_async_${r}_sub$k.json $B --api async --sub-batches $k --rendezvous $r --inline-first 1 --cpu-baseline 0
done; done
run bench_polygons50_pilot_step $OUT/bench_polygons50_pilot_step.json $B --actions pilot --api step --cpu-baseline 0
run bench_polygons50_pilot_async_device_sub2 $OUT/bench_polygons50_pilot_async_device_sub2.json $B --actions pilot --api async --sub-batches 2 --inline-first 1 --cpu-baseline 0
run bench_polygons50_pilot_per_chain_sub2 $OUT/bench_polygons50_pilot_per_chain_sub2.json $B --actions pilot --sub-batches 2 --cpu-baseline 0
run bench_polygons50_side_by_side_sub1 $OUT/bench_polygons50_side_by_side_sub1.json $B --step-mode side_by_side --sub-batches 1 --cpu-baseline 0
# captured steps: one chain (three-launch shape with the fused launch), captured chains (round 4), one graph with four branches
run bench_polygons50_graph16_sub1 $OUT/bench_polygons50_graph16_sub1.json $B --graph 16 --sub-batches 1 --steps 1920 --warmup 192 --cpu-baseline 0
run bench_polygons50_graph16_chains4 $OUT/bench_polygons50_graph16_chains4.json $B --graph 16 --sub-batches 4 --steps 1920 --warmup 192 --cpu-baseline 0
run bench_polygons50_graph16_one_graph4 $OUT/bench_polygons50_graph16_one_graph4.json $B --graph 16 --sub-batches 4 --one-graph 1 --steps 1920 --warmup 192 --cpu-baseline 0
run bench_polygons50_worlds1 $OUT/bench_polygons50_worlds1.json $B --worlds-per-env 1 --cpu-baseline 0
run bench_circles20 $OUT/bench_circles20.json $B --workload circles20 --cpu-baseline 0
run bench_moving28 $OUT/bench_moving28.json $B --workload moving28 --cpu-baseline 0
# a fresh world on every reset (round 5): the generator inside the timed region, the bank-cycling rate of the same loop beside it
run bench_moving28_fresh_worlds $OUT/bench_moving28_fresh_worlds.json $B --workload moving28 --fresh-worlds 1 --steps 2000 --warmup 3000 --cpu-baseline 0
run bench_moving28_fresh_worlds_depth4_period8 $OUT/bench_moving28_fresh_worlds_depth4_period8.json $B --workload moving28 --fresh-worlds 1 --worlds-per-env 4 --fresh-period 8 --steps 2000 --warmup 3000 --cpu-baseline 0
run host_bound_probe $OUT/host_bound_probe.jsonl python tools/host_bound_probe.py
run side_queue_probe $OUT/side_queue_probe_default.json python tools/side_queue_probe.py
run side_queue_probe2 $OUT/side_queue_probe2.json python tools/side_queue_probe2.py
run fma_issue_all_cus $OUT/fma_issue_all_cus.jsonl ./tools/fma_issue_bench
run fma_issue_one_cu $OUT/fma_issue_one_cu.jsonl ./tools/fma_issue_bench 1
run bench_mixed47_8192 $OUT/bench_mixed47_8192.json $B --workload mixed47 --envs 8192 --cpu-baseline 0
run bench_mixed47_8192_graph16_sub1 $OUT/bench_mixed47_8192_graph16_sub1.json $B --workload mixed47 --envs 8192 --graph 16 --sub-batches 1 --steps 1920 --warmup 192 --cpu-baseline 0
run bench_mixed47_8192_graph16_chains4 $OUT/bench_mixed47_8192_graph16_chains4.json $B --workload mixed47 --envs 8192 --graph 16 --sub-batches 4 --steps 1920 --warmup 192 --cpu-baseline 0
run bench_polygons50_32768 $OUT/bench_polygons50_32768.json $B --envs 32768 --steps 100 --warmup 20 --worlds-per-env 1 --cpu-baseline 0
# two ranks started by bench.py itself, sharing the one GPU of this box (gloo instead of RCCL): the N > 1 code path on hardware
run bench_2ranks_rehearsal_1gpu $OUT/bench_2ranks_rehearsal_1gpu.json python bench.py --gpus 2 --rehearse 1 --steps 500 --warmup 100 --bank-cache /tmp/bank --cpu-baseline 0
python bench.py --gpus 2 --steps 10 > $OUT/bench_2ranks_refused.out 2>&1; echo "exit code $? (two ranks without --rehearse on a 1-GPU box must be refused)" >> $OUT/bench_2ranks_refused.out
run policy_bench $OUT/policy_bench.log python tools/policy_bench.py 4096
run ppo_colav_fused_4096x128 $OUT/ppo_colav_fused_4096x128.log python examples/ppo.py --envs 4096 --updates 6 --rollout 128
run ppo_colav_fused_4096x128_generated_bank $OUT/ppo_colav_fused_4096x128_generated_bank.log python examples/ppo.py --envs 4096 --updates 6 --rollout 128 --worlds generated
echo "benches done"
[ "${SKIP_TRACE:-0}" = 1 ] && exit 0
cd /tmp && export TMPDIR=/tmp
run bench_under_rocprof $OUT/bench_under_rocprof.json rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --bank-cache /tmp/bank --cpu-baseline 0
cd $ROOT
python tools/trace_summary.py $OUT/trace/*/*_kernel_trace.csv > $OUT/kernel_trace_summary.txt
python tools/trace_overlap.py $OUT/trace/*/*_kernel_trace.csv k_step_roles 4000 > $OUT/kernel_trace_overlap.txt
python tools/trace_overlap.py $OUT/trace/*/*_kernel_trace.csv k_step_multi 64 >> $OUT/kernel_trace_overlap.txt   # the timed loop when it runs 64 steps per launch
cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace
# the same for the driver's own command (20 steps: its calibration issues launches of several lengths; the summary lists k_step_multi per length)
cd /tmp
run bench_driver_command_under_rocprof $OUT/bench_driver_command_under_rocprof.json rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_drv -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --bank-cache /tmp/bank --cpu-baseline 0
cd $ROOT
python tools/trace_summary.py $OUT/trace_drv/*/*_kernel_trace.csv > $OUT/kernel_trace_summary_driver_command.txt
cp $OUT/trace_drv/*/*_kernel_stats.csv $OUT/kernel_stats_driver_command.csv 2>/dev/null
rm -rf $OUT/trace_drv
head -6 $OUT/kernel_trace_summary.txt; head -12 $OUT/kernel_trace_overlap.txt
# (SKIP_PMC=1: the counter passes go in a GPU call of their own -- tools/pmc_workload.sh <tag> polygons50 4 -- when one call cannot hold both)
[ "${SKIP_PMC:-0}" = 1 ] || bash tools/pmc_workload.sh $TAG polygons50 4
