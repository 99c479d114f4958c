#!/usr/bin/env python3
"""Reads the counter CSV of a tools/valu_budget.py run (dispatch order) and prints, per cut level, the group medians of
the SQ counters per launch and what each phase adds.  usage: valu_budget_summary.py <rocprof out dir> [warmup group]"""
import collections
import csv
import glob
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from valu_budget import LEVELS  # noqa: E402

root = sys.argv[1]
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 1900
group = int(sys.argv[3]) if len(sys.argv) > 3 else 16
files = glob.glob(root + "/*/*counter_collection.csv") + glob.glob(root + "/*counter_collection.csv")
rows = collections.defaultdict(dict)
for f in files:
    for r in csv.DictReader(open(f)):
        if "k_step_roles" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
assert len(ids) >= warm + group * len(LEVELS), (len(ids), warm, group)
ids = ids[len(ids) - group * len(LEVELS):]          # the groups are the LAST launches of the run
names = sorted(next(iter(rows.values())))
med = []
for gi in range(len(LEVELS)):
    sel = ids[gi * group + 2:(gi + 1) * group]        # (the first two launches of a group still see the previous level's state)
    med.append({c: float(np.median([rows[i][c] for i in sel])) for c in names})
waves = dict(lidar=4096, nav=4096, dyn=512)
out = dict(counters=names, levels=[], per_phase=[])
for (cl, cn, label), m in zip(LEVELS, med):
    out["levels"].append(dict(cut_lidar=cl, cut_nav=cn, label=label, **{c: round(v) for c, v in m.items()}))
for gi in range(2, len(LEVELS)):
    prev, cur = med[gi - 1], med[gi]
    label = LEVELS[gi][2]
    role = "nav" if "navigation" in label else "lidar"
    dv = cur["SQ_INSTS_VALU"] - prev["SQ_INSTS_VALU"]
    dq = 4.0 * (cur["SQ_ACTIVE_INST_VALU"] - prev["SQ_ACTIVE_INST_VALU"])
    out["per_phase"].append(dict(phase=label, valu_insts_per_launch=round(dv), valu_insts_per_wave=round(dv / waves[role], 1),
                                 issue_cycles_per_launch=round(dq), cycles_per_inst=round(dq / dv, 2) if dv else None,
                                 lds_insts_per_wave=round((cur.get("SQ_INSTS_LDS", 0) - prev.get("SQ_INSTS_LDS", 0)) / waves[role], 1),
                                 share_of_valu_issue=round(dq / (4.0 * med[0]["SQ_ACTIVE_INST_VALU"]), 4)))
base = med[1]
out["base"] = dict(label=LEVELS[1][2], valu_insts_per_launch=round(base["SQ_INSTS_VALU"]),
                   issue_cycles_per_launch=round(4.0 * base["SQ_ACTIVE_INST_VALU"]),
                   share_of_valu_issue=round(base["SQ_ACTIVE_INST_VALU"] / med[0]["SQ_ACTIVE_INST_VALU"], 4))
out["check_everything_again_vs_first"] = round(med[-1]["SQ_INSTS_VALU"] / med[0]["SQ_INSTS_VALU"], 4)
print(json.dumps(out, indent=1))
