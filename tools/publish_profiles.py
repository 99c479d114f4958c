"""Copy the judged summaries of a tools/run_profiles.sh (or pmc_workload.sh) run from gpurun_out/<run>/ to
profiles/<tag>/ and refresh the two files bench.py reads:
  profiles/pmc_traffic.json  HBM bytes per STEP from the separate FETCH_SIZE / WRITE_SIZE passes
  profiles/pmc_sq.json       SQ counters per STEP (VALU leg, wait share) and the shader clock of the pass
both keyed "<workload>/sub<K>" and carrying the sha256 of the libauv_hip.so they were measured on (recorded on the
GPU box by the run script), so that bench.py can mark a leg stale when the library has changed since.
usage: python tools/publish_profiles.py gpurun_out/<run> <tag> [workload] [sub-batches] [steps per launch]"""
import glob
import json
import os
import shutil
import sys

run, tag = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "polygons50"
sub = int(sys.argv[4]) if len(sys.argv) > 4 else 4
T = int(sys.argv[5]) if len(sys.argv) > 5 else 1
sfx = "sub%d" % sub + ("_T%d" % T if T > 1 else "")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
headline = (workload, sub, T) == ("polygons50", 4, 1)       # the headline run keeps the plain names; other passes get a suffix
for f in glob.glob(os.path.join(run, "bench_*.json")) + glob.glob(os.path.join(run, "*.txt")) + glob.glob(os.path.join(run, "pmc_*.json")) + \
        [os.path.join(run, n) for n in ("kernel_stats.csv", "kernel_stats_driver_command.csv", "bench_2ranks_refused.out", "lib_sha256.txt")]:
    if os.path.exists(f) and os.path.getsize(f) > 0:
        if headline:
            shutil.copy(f, dst)
        elif os.path.basename(f).startswith("pmc_") and f.endswith("_bench.json"):
            shutil.copy(f, os.path.join(dst, os.path.basename(f)[:-5] + "_%s_%s.json" % (workload, sfx)))
sha = open(os.path.join(run, "lib_sha256.txt")).read().split()[0]
summ = json.load(open(os.path.join(run, "pmc_step_summary.json")))
ps = summ["per_step"]
key = "%s/%s" % (workload, sfx)
tpath = os.path.join(root, "profiles", "pmc_traffic.json")
t = json.load(open(tpath)) if os.path.exists(tpath) else {}
t = {k: v for k, v in t.items() if "/" in k or k == "_note"}     # (keys of earlier rounds' layout are dropped)
t[key] = dict(bytes_reads_doubled=ps["bytes_reads_doubled"], bytes_raw=ps["bytes_raw"], FETCH_SIZE_KiB=ps.get("FETCH_SIZE"),
              WRITE_SIZE_KiB=ps.get("WRITE_SIZE"), launches_per_step=summ["launches_per_step"], lib_sha256=sha, profile="profiles/%s/pmc_step_summary_%s_%s.json" % (tag, workload, sfx))
t["_note"] = ("HBM bytes per STEP (all launches of one step of 4096 envs per GPU) from SEPARATE rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE "
              "passes (tools/run_profiles.sh, tools/pmc_workload.sh, tools/pmc_step_summary.py).  bytes_reads_doubled = (2 x FETCH_SIZE + "
              "WRITE_SIZE) x 1024, the gfx950 correction of MI355X_MICROARCH.md for wide coalesced reads; bytes_raw = (FETCH_SIZE + WRITE_SIZE) x "
              "1024; the kernels' 16/32-byte loads lie between the two.  lib_sha256: the library the pass ran on.")
json.dump(t, open(tpath, "w"), indent=1)
spath = os.path.join(root, "profiles", "pmc_sq.json")
s = json.load(open(spath)) if os.path.exists(spath) else {}
s = {k: v for k, v in s.items() if "/" in k or k == "_note"}
s[key] = dict({c: ps[c] for c in ps if c.startswith("SQ_")}, clock_ghz=ps["clock_ghz"], launches_per_step=summ["launches_per_step"],
              lib_sha256=sha, profile="profiles/%s/pmc_step_summary_%s_%s.json" % (tag, workload, sfx))
s["_note"] = ("SQ counters per STEP (summed over the launches of one step of 4096 envs per GPU, steady state) from one rocprofv3 --pmc pass.  "
              "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_INSTS_* wave-instructions.  clock_ghz = "
              "SQ_BUSY_CYCLES / 32 shader engines / dispatch duration in the same pass.  VALU leg of bench.py = 4 x SQ_ACTIVE_INST_VALU / "
              "(1024 SIMDs x clock x measured time per step); wait share = SQ_WAIT_ANY / SQ_WAVE_CYCLES.")
json.dump(s, open(spath, "w"), indent=1)
shutil.copy(os.path.join(run, "pmc_step_summary.json"), os.path.join(dst, "pmc_step_summary_%s_%s.json" % (workload, sfx)))
print("published", dst, key, "| traffic", t[key]["bytes_raw"], t[key]["bytes_reads_doubled"], "| valu issue cycles per step",
      int(4 * s[key]["SQ_ACTIVE_INST_VALU"]), "clock", s[key]["clock_ghz"], "| sha", sha[:12])
