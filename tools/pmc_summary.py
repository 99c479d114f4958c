"""Per-kernel HBM traffic from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes.
Units: the counters are in KiB (MI355X_MICROARCH.md: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024);
on gfx950 FETCH_SIZE under-reports wide coalesced reads by exactly 2x, other access widths are
uncalibrated, so both the raw and the doubled-read figures are given."""
import csv, glob, json, sys, collections
import numpy as np
out = sys.argv[1]
res = collections.defaultdict(dict)
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    files = glob.glob("%s/%s/*/*counter_collection.csv" % (out, sub))
    if not files:
        continue
    rows = list(csv.DictReader(open(files[0])))
    agg = collections.defaultdict(list)
    for r in rows:
        if r.get("Counter_Name") != name:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split()[-1]
        agg[k].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if any(t in k for t in ("k1_", "k2_", "k3_", "k23_", "k_step")):
            res[k][name + "_KiB_per_launch_median"] = float(np.median(v))
            res[k]["launches"] = len(v)
for k, v in res.items():
    f, w = v.get("FETCH_SIZE_KiB_per_launch_median", 0.0), v.get("WRITE_SIZE_KiB_per_launch_median", 0.0)
    v["bytes_raw"] = int((f + w) * 1024)
    v["bytes_reads_doubled"] = int((2 * f + w) * 1024)
print(json.dumps(res, indent=1))
