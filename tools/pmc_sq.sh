#!/bin/bash
# SQ counters for the step kernels (one pass, 8 SQ slots): where do wave-cycles go?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_sq
python $ROOT/bench.py --bank-cache /tmp/bank --steps 20 --cpu-baseline 0 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT -- python3 $ROOT/bench.py --bank-cache /tmp/bank --cpu-baseline 0 --steps 60 --warmup 10 > /dev/null 2>&1
cd $ROOT
python - <<'PY'
import csv, glob, collections, numpy as np, os
f = glob.glob(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc_sq/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split()[-1]
    if any(t in k for t in ("k1_", "k2_lidar", "k3_nav", "k3_reward")) and "fresh" not in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    m = {n: np.median(v) for n, v in c.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print("%-16s wave_cycles %.3e  wait_any %.0f%%  wait_inst %.0f%%  active_inst %.0f%% | per-launch insts: VALU %.3e SALU %.3e LDS %.3e VMEM_RD %.3e" % (
        k, wc, 100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_SALU", 0), m.get("SQ_INSTS_LDS", 0), m.get("SQ_INSTS_VMEM_RD", 0)))
PY
