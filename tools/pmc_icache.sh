#!/bin/bash
# Instruction-cache counters of the step kernels.  Usage: bash tools/pmc_icache.sh <tag> [extra bench.py flags]
TAG=${1:-ic}
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
python $ROOT/bench.py --bank-cache /tmp/bank --steps 20 --cpu-baseline 0 "$@" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_ic -- python3 $ROOT/bench.py --bank-cache /tmp/bank --cpu-baseline 0 --steps 60 --warmup 10 "$@" > $OUT/pmc_ic_bench.json 2> $OUT/pmc_ic_bench.err
cd $ROOT
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/pmc_ic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k in acc:
    if cnt[k] >= 50: print(k, {c: round(v / cnt[k]) for c, v in acc[k].items()})
PY
