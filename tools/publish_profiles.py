"""Copy the judged summaries of a tools/run_profiles.sh run from gpurun_out/<run>/ to profiles/<tag>/ and refresh
the two files bench.py reads for the dominant kernel of the headline workload:
  profiles/pmc_traffic.json  HBM bytes per launch from the separate FETCH_SIZE / WRITE_SIZE passes
  profiles/pmc_sq.json       SQ counters per launch (VALU leg)
usage: python tools/publish_profiles.py gpurun_out/<run> <tag> [workload]"""
import glob
import json
import os
import shutil
import sys

run, tag = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "polygons50"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(run, "bench_*.json")) + [os.path.join(run, n) for n in (
        "kernel_trace_summary.txt", "kernel_stats.csv", "pmc_summary.json", "pmc_sq_summary.json", "pmc_sq_kernel_durations.txt",
        "bench_2ranks_refused.out")]:
    if os.path.exists(f) and os.path.getsize(f) > 0:
        shutil.copy(f, dst)
clean = lambda k: k.split("<")[0]   # noqa: E731
pmc = json.load(open(os.path.join(run, "pmc_summary.json")))
traffic = {clean(k): v["bytes_reads_doubled"] for k, v in pmc.items() if "fresh" not in k}
raw = {clean(k): v["bytes_raw"] for k, v in pmc.items() if "fresh" not in k}
tpath = os.path.join(root, "profiles", "pmc_traffic.json")
t = json.load(open(tpath)) if os.path.exists(tpath) else {}
t.setdefault(workload, {}).update(traffic)            # (kernels of launch shapes profiled in earlier runs stay)
t.setdefault(workload + "_raw", {}).update(raw)
t["_note"] = ("HBM bytes per launch at 4096 envs per GPU from SEPARATE rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes "
              "(tools/run_profiles.sh; profiles/%s/pmc_summary.json).  <workload>: (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 "
              "correction of MI355X_MICROARCH.md for wide coalesced reads (FETCH_SIZE counts 64 B per 128-B request); "
              "<workload>_raw: (FETCH_SIZE + WRITE_SIZE) x 1024.  K1's 8-byte accesses read 1:1 (274 KiB fetched vs 288 KiB "
              "algorithmic reads), the kernels with 16/32-byte loads lie between the two figures." % tag)
json.dump(t, open(tpath, "w"), indent=1)
sq = json.load(open(os.path.join(run, "pmc_sq_summary.json")))
spath = os.path.join(root, "profiles", "pmc_sq.json")
s = json.load(open(spath)) if os.path.exists(spath) else {}
s.setdefault(workload, {}).update({clean(k): {c: v[c] for c in v if c.startswith("SQ_") or c == "launches"} for k, v in sq.items()})
s["_note"] = ("SQ counters per launch (medians) at 4096 envs per GPU from one rocprofv3 --pmc pass, steady state (steps 1900-2200 of "
              "the headline run), tools/run_profiles.sh; profiles/%s/pmc_sq_summary.json.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* "
              "count quad-cycles summed over waves, SQ_INSTS_* wave-instructions, SQ_BUSY_CYCLES busy cycles summed over the 32 shader "
              "engines.  VALU leg = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x SQ_BUSY_CYCLES / 32)." % tag)
json.dump(s, open(spath, "w"), indent=1)
print("published", dst, "| traffic", traffic, "| valu frac",
      {k: round(4 * v["SQ_ACTIVE_INST_VALU"] / (1024 * v["SQ_BUSY_CYCLES"] / 32), 3) for k, v in s[workload].items()})
