"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel count / min / median / mean / p95 / max (us).  Launches of k_step_multi are
listed per launch LENGTH as well (its grid says how many steps a launch holds: 112 or 144 workgroups per cohort position), since a
run that calibrates its shape issues launches of several lengths."""
import csv, sys, collections
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-40:]
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[name].append(us)
    if name.endswith("k_step_multi"):
        grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
        agg["k_step_multi [grid %s threads]" % grid].append(us)
print("%-42s %6s %8s %8s %8s %8s %8s" % ("kernel", "calls", "min", "median", "mean", "p95", "max"))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v = np.array(v)
    print("%-42s %6d %8.1f %8.1f %8.1f %8.1f %8.1f" % (k, len(v), v.min(), np.median(v), v.mean(), np.percentile(v, 95), v.max()))
