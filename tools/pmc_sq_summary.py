"""Per-kernel medians of the SQ counter pass (tools/pmc_sq.sh) and the VALU leg derived from them.

Units (MI355X_MICROARCH.md, cycle-constants table): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles (4 shader cycles) summed over waves; SQ_INSTS_* count wave-instructions; SQ_BUSY_CYCLES
counts per-SE busy cycles.  VALU issue peak: 256 CUs x 4 SIMDs, one wave64 VALU instruction per SIMD
every 2 cycles for 4-byte types, every 4 for fp64 (half rate).  `issue_cycles` below = 4 x
SQ_ACTIVE_INST_VALU (cycles in which some wave was executing a VALU instruction, summed over waves);
valu_frac = issue_cycles / (duration x clock x 1024 SIMDs) needs the duration, which bench.py supplies."""
import collections
import csv
import glob
import json
import sys

import numpy as np

root = sys.argv[1]
files = glob.glob(root + "/*/*counter_collection.csv") + glob.glob(root + "/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split()[-1].split("<")[0]
        if any(t in k for t in ("k1_", "k2_", "k3_", "k23_", "k_step")) and "fresh" not in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, c in agg.items():
    m = {n: float(np.median(v)) for n, v in c.items()}
    m["launches"] = len(next(iter(c.values())))
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        m["wait_any_frac"] = m.get("SQ_WAIT_ANY", 0.0) / wc
        m["wait_inst_frac"] = m.get("SQ_WAIT_INST_ANY", 0.0) / wc
        m["active_inst_frac"] = m.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
        m["active_valu_frac_of_wave_cycles"] = m.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
    m["valu_issue_cycles"] = 4.0 * m.get("SQ_ACTIVE_INST_VALU", 0.0)
    res[k] = m
print(json.dumps(res, indent=1))
