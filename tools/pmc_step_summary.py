"""Per-STEP counters of the step's launches from the separate rocprofv3 passes of tools/run_profiles.sh /
tools/pmc_workload.sh (--pmc FETCH_SIZE, --pmc WRITE_SIZE, --pmc SQ_*): for each pass the counters of all step-kernel
dispatches of the steady window are summed and divided by the number of steps (a step of K sub-batches is K dispatches
of k_step_roles), plus the per-dispatch medians and the shader clock of the SQ pass (SQ_BUSY_CYCLES / 32 shader
engines / the dispatch's duration in the same pass's kernel trace).
Units: FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read
(MI355X_MICROARCH.md), so both the raw and the reads-doubled byte counts are given.
With a third argument T > 1 the passes ran T steps per launch (k_step_multi): a step is then 1 / T of K dispatches.
usage: pmc_step_summary.py <out dir> <chains> [steps per launch]"""
import collections
import csv
import glob
import json
import sys

import numpy as np

out, per_step = sys.argv[1], int(sys.argv[2])
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1
import os
for nm in ("pmc_fetch", "pmc_write", "pmc_sq"):      # the passes must have stepped with that many launches per step
    f = os.path.join(out, nm + "_bench.json")
    if os.path.exists(f) and os.path.getsize(f):
        got = json.load(open(f))["config"]["sub_batches"]
        assert got == per_step, "%s ran with %d sub-batches, summary asked for %d" % (nm, got, per_step)
        gotT = json.load(open(f))["config"].get("steps_per_launch", 1)
        assert gotT == T, "%s ran %d steps per launch, summary asked for %d" % (nm, gotT, T)
STEP_KERNELS = ("k_step_multi", "k_step_roles", "k1_dynamics", "k23_lidar_nav", "k3_reward")


def kname(r):
    return r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split()[-1].split("<")[0]


def load(sub):
    files = glob.glob("%s/%s/*/*counter_collection.csv" % (out, sub))
    rows = collections.defaultdict(dict)
    names = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            k = kname(r)
            if any(k.startswith(t) for t in STEP_KERNELS) and "fresh" not in k:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
                names[int(r["Dispatch_Id"])] = k
    dur = {}
    for f in glob.glob("%s/%s/*/*kernel_trace.csv" % (out, sub)):
        for r in csv.DictReader(open(f)):
            dur[int(r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return rows, names, dur


MAIN = "k_step_multi" if T > 1 else "k_step_roles"
res = dict(launches_per_step=per_step / float(T), chains=per_step, steps_per_launch=T, kernels={}, per_step={})
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    rows, names, dur = load(sub)
    if not rows:
        continue
    ids = sorted(rows)
    main = [i for i in ids if names[i] == MAIN] or ids
    main = main[len(main) // 2:]                     # steady state: the second half of the run
    lo = main[0]
    ids = [i for i in ids if i >= lo]
    n_steps = len(main) * T / float(per_step)
    for c in sorted(next(iter(rows.values()))):
        res["per_step"][c] = sum(rows[i].get(c, 0.0) for i in ids) / n_steps
    by_k = collections.defaultdict(list)
    for i in ids:
        by_k[names[i]].append(i)
    for k, lst in by_k.items():
        e = res["kernels"].setdefault(k, {})
        for c in sorted(rows[lst[0]]):
            e[c + "_median"] = float(np.median([rows[i][c] for i in lst]))
        e["dispatches_" + sub] = len(lst)
        if sub == "pmc_sq":
            ds = [dur[i] for i in lst if i in dur]
            if ds:
                e["duration_us_median_under_pmc"] = float(np.median(ds)) / 1e3
                ghz = [rows[i]["SQ_BUSY_CYCLES"] / 32.0 / dur[i] for i in lst if i in dur and "SQ_BUSY_CYCLES" in rows[i]]
                if ghz:
                    e["clock_ghz"] = float(np.median(ghz))
ps = res["per_step"]
f, w = ps.get("FETCH_SIZE", 0.0), ps.get("WRITE_SIZE", 0.0)
ps["bytes_raw"] = int((f + w) * 1024)
ps["bytes_reads_doubled"] = int((2 * f + w) * 1024)
main_k = res["kernels"].get(MAIN) or next(iter(res["kernels"].values()), {})
ps["clock_ghz"] = round(main_k.get("clock_ghz", 0.0), 4)
print(json.dumps(res, indent=1))
