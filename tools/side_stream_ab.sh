# A/B of the fresh-world refill pass (round 5): chains, period, batch, depth -- moving28, 4096 envs.
# (The *_nopace / *_nograph rows under profiles/r05 came from a diagnostic build that read AUV_FW_NO_PACE / AUV_FW_NO_GRAPH; the
# shipped library reads no environment variable.  Neither made a difference: 122-125 M env-steps/s in all four combinations.)
mkdir -p gpurun_out/r05
run() {  # run <tag> <env assignments...> -- <bench flags...>
  tag=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --workload moving28 --cpu-baseline 0 "$@" > gpurun_out/r05/side_$tag.json 2>> gpurun_out/r05/side_stderr.log
  python - <<PY
import json
d=json.load(open("gpurun_out/r05/side_$tag.json"))
fw=d["config"].get("fresh_worlds", {})
print("%-28s %6.1f M  sub %d slices %s  cycling %s  fresh %s gen %.2fs" % ("$tag", d["value"]/1e6, d["config"]["sub_batches"], d["roofline"]["kernels"]["k_step_roles"].get("per_slice_ms"), d.get("comparison", {}).get("bank_cycling_same_loop"), {k: fw.get(k) for k in ("regenerated","reused","queued","passes_issued","passes_published")}, d["config"]["world_gen_s"]))
PY
}
run sub3_p8 X=1 -- --fresh-worlds 1 --steps 2000 --warmup 3000 --fresh-period 8
run sub3_p16 X=1 -- --fresh-worlds 1 --steps 2000 --warmup 3000
run sub3_p16_d2 X=1 -- --fresh-worlds 1 --steps 2000 --warmup 3000 --worlds-per-env 4
run nofresh_sub3 X=1 -- --steps 2000 --warmup 3000 --sub-batches 3
