#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE, SQ counters: three SEPARATE runs, kernel trace only beside them) for ONE bench
# configuration; tools/pmc_step_summary.py turns them into per-step figures, tools/publish_profiles.py publishes them.
# Usage (GPU box): [MULTI=64] bash tools/pmc_workload.sh <tag> <workload> <sub-batches> [extra bench flags]
# (MULTI=T: the passes run T steps per launch -- k_step_multi, the default command's kernel; step counts are multiples of T)
TAG=$1; WL=$2; SUB=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
sha256sum gym_auv_amd/csrc/libauv_hip.so > $OUT/lib_sha256.txt
B="$ROOT/bench.py --bank-cache /tmp/bank --workload $WL --sub-batches $SUB --probe-streams 0 --cpu-baseline 0 --multi ${MULTI:-1}"
# Every pass says when it starts AND when it ends, and its stderr is APPENDED to $OUT/pmc_stderr.log (round 4: both went to
# /dev/null, and a pass that sat behind a polling kernel for 300 s was killed for silence with nothing to read afterwards)
ERR=$OUT/pmc_stderr.log
: > $ERR
pass() {   # pass <name> <stdout file> <command ...>
  local name=$1 out=$2; shift 2
  echo "  start $name $(date +%T)"
  echo "==== $name: $*" >> $ERR
  "$@" > $out 2>> $ERR
  local rc=$?
  echo "  done  $name rc=$rc $(date +%T)"
  [ $rc = 0 ] || { tail -5 $ERR; exit $rc; }
}
T=${MULTI:-1}
if [ $T -gt 1 ]; then N1=$((10 * T)); N2=$((5 * T)); N3=$((30 * T)); else N1=100; N2=300; N3=1900; fi
pass warm_bank /dev/null python $B --steps 20 "$@"
cd /tmp && export TMPDIR=/tmp
pass "pmc pass 1/3 (FETCH_SIZE)" $OUT/pmc_fetch_bench.json rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $B --steps $N1 --warmup $N1 "$@"
pass "pmc pass 2/3 (WRITE_SIZE)" $OUT/pmc_write_bench.json rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $B --steps $N1 --warmup $N1 "$@"
pass "pmc pass 3/3 (SQ counters)" $OUT/pmc_sq_bench.json rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $B --steps $N2 --warmup $N3 "$@"
echo "  pmc passes done"
cd $ROOT
python tools/pmc_step_summary.py $OUT $SUB $T > $OUT/pmc_step_summary.json   # (checks $SUB against config.sub_batches of the passes)
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
python -c "
import json; d=json.load(open('$OUT/pmc_step_summary.json')); print({k: (round(v) if isinstance(v, float) else v) for k, v in d['per_step'].items()})"
