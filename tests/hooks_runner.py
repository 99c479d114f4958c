#!/usr/bin/env python3
"""Cases that need the TEST-HOOK build of the library (gym_auv_amd/csrc/libauv_hip_hooks.so, `make hooks`): roles skewed
onto different XCDs, and a sweep that withholds its word so that a poll runs out.  The shipped library has no such
hooks, and one process can only hold one build of it, so tests/test_gpu_parity.py runs this file in a child process
with AUV_HIP_LIB pointing at the hook build and reads one JSON line per case.

    AUV_HIP_LIB=gym_auv_amd/csrc/libauv_hip_hooks.so python tests/hooks_runner.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FIELDS = ("STATE", "LIDAR_D", "OBS64", "INFO64", "NAV64", "MOVER_STATE", "EPISODE", "COUNTERS", "NEARBY", "STEP_INFO",
          "WORLD_IDX", "CULL_LIMITS", "COLLISION", "REWARD64")


def mixed_bank(n):
    from gym_auv_amd.scenarios import moving_obstacles_world, polygon_world, static_circles_world
    from gym_auv_amd.world import build_world, pack_bank
    specs = []
    for i in range(n):
        if i % 3 == 0:
            specs.append(moving_obstacles_world(1000 + i))
        elif i % 3 == 1:
            specs.append(static_circles_world(1000 + i, 20))
        else:
            specs.append(polygon_world(1000 + i, 12, n_circles=4, n_moving=3))
    return pack_bank([build_world(s) for s in specs])


def main():
    from gym_auv_amd import _capi
    from gym_auv_amd.batched_env import _LIB, BatchedAuvEnv, _check
    from gym_auv_amd.config import effective_reference_config
    assert hasattr(_LIB, "auv_test_hooks"), "not the hook build: %s" % os.environ.get("AUV_HIP_LIB")
    bank = mixed_bank(32)

    def env_(cfg, n, mode, skew=0, fault=0):
        e = BatchedAuvEnv(cfg, bank, n, device="cuda:0", auto_reset=True)
        e.set_step_mode(mode)
        _check(_LIB.auv_test_hooks(e._h, skew, fault), "auv_test_hooks")
        e.reset()
        return e

    # ---- the hand-overs with an environment's waves on DIFFERENT XCDs (three idle workgroups between the roles):
    # bit for bit the three-launch shape over short episodes (every environment restored >= 10 times)
    for mode in ("one_launch",):
        n = 1024
        cfg = effective_reference_config(use_lidar=True)
        cfg.episode.max_timesteps = 5
        ref, par = env_(cfg, n, "side_by_side"), env_(cfg, n, mode, skew=3)
        rs = np.random.RandomState(20)
        ok, n_done = True, 0
        for k in range(60):
            a = torch.as_tensor(rs.uniform([-1, -0.15], [1, 0.15], (n, 2)), dtype=torch.float32, device="cuda:0")
            o0, r0, d0, _ = ref.step(a)
            o1, r1, d1, _ = par.step(a)
            torch.cuda.synchronize()
            ok = ok and torch.equal(o0, o1) and torch.equal(r0, r1) and torch.equal(d0, d1)
            n_done += int(d0.sum())
            if k % 7 == 0 or k == 59:
                ok = ok and all(torch.equal(ref.read(f), par.read(f)) for f in FIELDS)
        print(json.dumps(dict(case="skew", mode=mode, bitwise=bool(ok), n_done=n_done, n=n, effective=par.effective_step_mode())), flush=True)
        ref.close(), par.close()

    # ---- a poll that runs out: the launch ENDS, launches queued behind it do nothing (ABORT packets), the next call
    # reports once and resets exactly the environments whose step was left unfinished; every other environment keeps the
    # state of the one step it completed; the handle goes on in the three-launch shape -- bit for bit what a fresh handle
    # does that steps once, resets the same environments and carries on.  Two sub-batches of 128: the hook hits the first
    # environment of EVERY launch, so both chains report.
    for mode, fault in (("one_launch", 1), ("one_launch", 2), ("one_launch", 3)):   # sweep's word / state packet / search record withheld
        n = 256
        cfg = effective_reference_config(use_lidar=True)
        env = env_(cfg, n, mode, fault=fault)
        slices = env.set_sub_batches(2, strict=True)   # (streams measured to run side by side: a chain queued BEHIND the other's faulty
                                                       # launch on one hardware queue would be aborted before its first step)
        a = torch.zeros((n, 2), dtype=torch.float32, device="cuda:0")
        a[:, 0] = 0.7
        h0 = env.health()
        for _ in range(3):
            env.step_pipelined(a)                    # the faulty launches and two more behind each of them ...
        torch.cuda.synchronize()                     # ... all END (no hang)
        h1 = env.health()
        marked = torch.nonzero(env.read("BROKEN")).flatten().tolist()   # (before the recovery clears the marks)
        msg = ""
        try:
            env.step(a)
        except RuntimeError as exc:
            msg = str(exc)
        h2, lt = env.health(), env.last_timeout()
        # the eight environments of the finish wave that covers a launch's first environment: e0 + 8 g
        broken = sorted(lo + 8 * g for lo, _ in slices for g in range(8))
        mask = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
        mask[torch.as_tensor(broken, device="cuda:0")] = 1
        t_step = env.read("COUNTERS")[:, 0]
        steps_as_expected = bool((t_step[mask == 0] == 1).all()) and bool((t_step[mask == 1] == 0).all())
        ref = env_(cfg, n, "side_by_side")
        ref.step(a)
        ref.reset(mask=mask)
        same_state = all(torch.equal(ref.read(f), env.read(f)) for f in FIELDS if f not in ("EPISODE", "STEP_INFO", "REWARD64"))
        obs_rows_reset = torch.equal(ref.obs[mask == 1], env.obs[mask == 1])     # the recovery wrote the reset observation rows
        ok = True
        for _ in range(5):
            o0, r0, d0, _ = ref.step(a)
            o1, r1, d1, _ = env.step(a)              # (fault hook still on: the three-launch shape has no hand-over to fail)
            torch.cuda.synchronize()
            ok = ok and torch.equal(o0, o1) and torch.equal(r0, r1) and torch.equal(d0, d1)
        ok = ok and all(torch.equal(ref.read(f), env.read(f)) for f in ("STATE", "LIDAR_D", "OBS64", "NAV64", "MOVER_STATE", "NEARBY", "CULL_LIMITS", "COUNTERS"))
        print(json.dumps(dict(case="fault", mode=mode, fault=fault, before=h0, after_launch=h1, after_recovery=h2, last_timeout=lt, message=msg,
                              n_broken_expected=len(broken), marked=marked, t_step=t_step.tolist(), steps_as_expected=steps_as_expected, state_equal=bool(same_state),
                              obs_rows_reset=bool(obs_rows_reset), continues_bitwise=bool(ok),
                              effective=env.effective_step_mode())), flush=True)
        ref.close(), env.close()

    # ---- the same inside a launch of SEVERAL steps (auv_step_multi): the sweep of the first environment withholds its word in
    # step 0; its finish wave gives up, the abort flag goes up, every later step of the launch does nothing (ABORT packets; waves
    # that wait for a carry record look at the flag) -- the launch of 16 steps ENDS in milliseconds, the next call reports once
    # and resets the marked environments, stepping goes on
    import time as _time
    for fault in (1, 2):
        n = 256
        cfg = effective_reference_config(use_lidar=True)
        env = env_(cfg, n, "one_launch", fault=fault)
        env.set_sub_batches(1)
        ring = torch.zeros((4, n, 2), dtype=torch.float32, device="cuda:0")
        ring[:, :, 0] = 0.7
        torch.cuda.synchronize()
        t0 = _time.perf_counter()
        env.step_multi(ring, 0, 16)
        torch.cuda.synchronize()
        dt = _time.perf_counter() - t0
        h1 = env.health()
        t_step = env.read("COUNTERS")[:, 0]
        marked = int((env.read("BROKEN") != 0).sum())
        msg = ""
        try:
            env.step(ring[0])
        except RuntimeError as exc:
            msg = str(exc)
        h2 = env.health()
        ok = True
        for _ in range(5):
            o, r, dn, _ = env.step(ring[0])
            ok = ok and bool(torch.isfinite(o).all()) and bool(torch.isfinite(r).all())
        torch.cuda.synchronize()
        print(json.dumps(dict(case="multi_fault", fault=fault, seconds=round(dt, 4), after_launch=h1, after_recovery=h2, message=msg, marked=marked,
                              t_step_min=int(t_step.min()), t_step_max=int(t_step.max()), continues=bool(ok), effective=env.effective_step_mode())), flush=True)
        env.close()

    # ---- the dispatch-order probe in the shape production runs: four launches in flight on four streams, a foreign kernel
    # behind each (set_sub_batches runs it); and a rendezvous kernel of step_async that gives up (nobody publishes)
    cfg = effective_reference_config(use_lidar=True)
    env = env_(cfg, 1024, "one_launch")
    env.set_sub_batches(4)
    print(json.dumps(dict(case="probe", sub_batches=env.sub_batches, health=env.health(), effective=env.effective_step_mode(256))), flush=True)
    env.close()


if __name__ == "__main__":
    main()
