#!/usr/bin/env python3
"""Dynamic per-phase instruction budget of the one-launch step, measured, not estimated.

Runs under `rocprofv3 --pmc SQ_INSTS_VALU ...` with the DIAGNOSTIC build of the library (-DAUV_CUTS: the LiDAR and
navigation roles can be told to stop after phase n; the shipped library has no such switch).  After a steady-state
warm-up with everything on, groups of G launches follow, one per cut level; tools/valu_budget_summary.py reads the
counter CSV in dispatch order and differences the group medians: what a phase adds is what it executes.

    tools/build_variant.sh cuts "-DAUV_CUTS"
    AUV_HIP_LIB=gym_auv_amd/csrc_cuts/libauv_hip.so rocprofv3 --kernel-trace --pmc ... -- python3 tools/valu_budget.py
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

# (cut_lidar, cut_nav, label): cumulative -- a level runs everything BEFORE phase n of that role
LEVELS = [
    (0, 0, "everything"),
    (1, 1, "dynamics role + both roles' preparation, polls, free LiDAR row, reward/finish"),
    (2, 1, "+ LiDAR B0/B: nearby refresh, cull windows, compaction, segment prefix"),
    (3, 1, "+ LiDAR C: ray table; E: free-beam pass"),
    (4, 1, "+ LiDAR S: staging (front-face test, spans, point-in-polygon predicates)"),
    (5, 1, "+ LiDAR D(i) + work-item prefix"),
    (6, 1, "+ LiDAR D(ii): pair sweep"),
    (0, 1, "+ LiDAR E: returns (sqrt, log, exp)"),
    (0, 2, "+ navigation: chunk circles, hint chunk, survivor list"),
    (0, 3, "+ navigation: exact distances, (distance, index) reduction"),
    (0, 0, "+ navigation tail (finish role, eight environments per wave): spline evaluation, atan2, features, path reward  (= everything again)"),
]
WARMUP = int(os.environ.get("WARMUP", "1900"))
GROUP = int(os.environ.get("GROUP", "16"))


def main():
    import bench
    from gym_auv_amd.batched_env import _LIB, BatchedAuvEnv, _check
    from gym_auv_amd.config import effective_reference_config
    assert hasattr(_LIB, "auv_diag_cuts"), "needs the -DAUV_CUTS build (AUV_HIP_LIB)"
    _LIB.auv_diag_cuts.restype, _LIB.auv_diag_cuts.argtypes = C.c_int, [C.c_void_p, C.c_int32, C.c_int32]
    workload = os.environ.get("WORKLOAD", "polygons50")
    gen, kwargs, ns, nps, desc = bench.WORKLOADS[workload]
    cfg = effective_reference_config(use_lidar=True)
    cfg.vessel.n_sectors, cfg.vessel.n_sensors_per_sector = ns, nps
    n = 4096
    z = np.load(os.environ.get("BANK") or sorted(__import__("glob").glob("/tmp/bank.%s.0.4096.4096.2*.npz" % workload))[0])
    bank = {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
    dev = torch.device("cuda:0")
    env = BatchedAuvEnv(cfg, bank, n, device=dev, auto_reset=True)
    env.set_step_mode("one_launch")
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    low, high = torch.tensor([-1.0, -0.15], device=dev), torch.tensor([1.0, 0.15], device=dev)
    pool = low + (high - low) * torch.rand((64, n, 2), generator=g, device=dev)
    env.reset()
    for i in range(WARMUP):
        env.step(pool[i % 64])
    torch.cuda.synchronize()
    k = WARMUP
    for cl, cn, _ in LEVELS:
        _check(_LIB.auv_diag_cuts(env._h, cl, cn), "auv_diag_cuts")
        for i in range(GROUP):
            env.step(pool[k % 64])
            k += 1
        torch.cuda.synchronize()
    print(json.dumps(dict(warmup=WARMUP, group=GROUP, levels=[l[2] for l in LEVELS], workload=workload)))


if __name__ == "__main__":
    main()
