"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel count / min / median / mean / p95 / max (us)."""
import csv, sys, collections
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    agg[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-40:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-42s %6s %8s %8s %8s %8s %8s" % ("kernel", "calls", "min", "median", "mean", "p95", "max"))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v = np.array(v)
    print("%-42s %6d %8.1f %8.1f %8.1f %8.1f %8.1f" % (k, len(v), v.min(), np.median(v), v.mean(), np.percentile(v, 95), v.max()))
