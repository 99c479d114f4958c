#!/usr/bin/env python3
"""Sub-batch sweep: the headline workload (4096 envs x 180 sensors, 50 polygons) stepped as K launch chains on K
streams (BatchedAuvEnv.set_sub_batches / step_slice), K from the command line; for every K the rate over `--steps`
steps and whether the final state equals the K = 1 run bit for bit (same actions, same start).

    python tools/archive/subbatch_sweep.py --ks 1,2,3,4,6,8 --steps 2000 [--envs 4096] [--workload polygons50]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ks", default="1,2,3,4,6,8")
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--workload", default="polygons50")
    ap.add_argument("--worlds-per-env", type=int, default=2)
    ap.add_argument("--bank-cache", default="")
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--modes", default="auto")
    args = ap.parse_args()
    import bench
    from gym_auv_amd.config import effective_reference_config
    from gym_auv_amd.world import build_bank_parallel
    gen, kwargs, ns, nps, desc = bench.WORKLOADS[args.workload]
    cfg = effective_reference_config(use_lidar=True)
    cfg.vessel.n_sectors, cfg.vessel.n_sensors_per_sector = ns, nps
    n = args.envs
    seeds = bench.world_seeds(0, n, n, args.worlds_per_env)
    cache = args.bank_cache and "%s.%s.%d.%d.npz" % (args.bank_cache, args.workload, n, args.worlds_per_env)
    if cache and os.path.exists(cache):
        z = np.load(cache)
        bank = {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
    else:
        bank = build_bank_parallel(gen, seeds, procs=max(1, min(16, bench.host_cores())), **kwargs)
        if cache:
            np.savez(cache, **bank)
    dev = torch.device("cuda:0")
    from gym_auv_amd.batched_env import BatchedAuvEnv
    env = BatchedAuvEnv(cfg, bank, n, device=dev, auto_reset=True)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    low = torch.tensor([-1.0, -0.15], device=dev)
    high = torch.tensor([1.0, 0.15], device=dev)
    pool = low + (high - low) * torch.rand((64, n, 2), generator=g, device=dev)
    w0 = torch.arange(n, device=dev, dtype=torch.int32)
    ref = None
    fields = ("STATE", "OBS64", "REWARD64", "LIDAR_D", "WORLD_IDX", "INFO64", "NAV64", "EPISODE")
    for mode, k in [(m, int(x)) for m in args.modes.split(",") for x in args.ks.split(",")]:
        env.set_step_mode(mode)
        env.set_sub_batches(k)
        rates = []
        for rep in range(args.repeat):
            env.reset(world_idx=w0)
            torch.cuda.synchronize(dev)

            def run(i0, cnt):
                if k == 1:
                    for i in range(cnt):
                        env.step(pool[(i0 + i) % 64])
                else:
                    for i in range(cnt):
                        env.step_pipelined(pool[(i0 + i) % 64])
            run(0, args.warmup)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            run(args.warmup, args.steps)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            rates.append(n * args.steps / dt)
        snap = {f: env.read(f).cpu().numpy() for f in fields}
        cnt2 = env.read("COUNTERS").cpu().numpy()[:, :2]
        snap["CNT"] = cnt2
        if ref is None:
            ref = snap
            same = True
        else:
            same = all(np.array_equal(ref[f], snap[f], equal_nan=True) if snap[f].dtype.kind == "f" else np.array_equal(ref[f], snap[f])
                       for f in snap)
        print(json.dumps(dict(mode=mode, sub_batches=env.sub_batches, slices=env._slices[:2], env_steps_per_s=[round(r / 1e6, 2) for r in rates],
                              ms_per_step=round(1e3 * n / max(rates), 5), bitwise_equal_to_first=bool(same))), flush=True)
    env.close()


if __name__ == "__main__":
    main()
