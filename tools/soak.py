"""One-off long parity soak on the GPU box: E envs x T steps of the mixed world with auto-reset, HIP path against the
CPU oracle, integer fields bit for bit EVERY step (cull limits, nearby flags, collision, done, world binding), fp64
fields every 25th step.  Counts what was compared.  usage: python tools/soak.py [envs] [steps] [mode] [sub-batches]
(sub-batches > 1: the batch is stepped as that many chains on their own streams, BatchedAuvEnv.step_async / step_wait;
SKEW=k in the environment + AUV_HIP_LIB=.../libauv_hip_hooks.so: the one-launch step's roles k workgroups apart, i.e. an
environment's waves on different XCDs)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gym_auv_amd._capi import make_config
from gym_auv_amd.batched_env import BatchedAuvEnv
from gym_auv_amd.config import effective_reference_config
from gym_auv_amd.scenarios import moving_obstacles_world, polygon_world, static_circles_world
from gym_auv_amd.world import build_world, pack_bank
from oracle import pyoracle

E = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
mode = sys.argv[3] if len(sys.argv) > 3 else "auto"
SUB = int(sys.argv[4]) if len(sys.argv) > 4 else 1
specs = []
for i in range(96):
    specs.append([moving_obstacles_world, lambda s: static_circles_world(s, 20), lambda s: polygon_world(s, 50),
                  lambda s: polygon_world(s, 10, n_circles=20, n_moving=17)][i % 4](7000 + i))
bank = pack_bank([build_world(s) for s in specs])
cfg = effective_reference_config(use_lidar=True)
cfg.episode.max_timesteps = 400
env = BatchedAuvEnv(cfg, bank, E, device="cuda:0", auto_reset=True)
env.set_step_mode(mode)
if SUB > 1:
    env.set_sub_batches(SUB)
if os.environ.get("SKEW"):
    # roles skewed onto different XCDs: needs the hook build (AUV_HIP_LIB=gym_auv_amd/csrc/libauv_hip_hooks.so)
    from gym_auv_amd.batched_env import _LIB, _check
    _check(_LIB.auv_test_hooks(env._h, int(os.environ["SKEW"]), 0), "auv_test_hooks")
    print("roles skewed by %s workgroups (hook build)" % os.environ["SKEW"], flush=True)
ora = pyoracle.Oracle(make_config(cfg, auto_reset=True), E, bank)
pyoracle.set_threads(min(16, os.cpu_count() or 1))
env.reset(), ora.reset()
rs = np.random.RandomState(99)
np_ = lambda t: t.detach().cpu().numpy()
n_lim = n_beams = n_done = 0
worst = 0.0
t0 = time.time()
MULTI = int(os.environ.get("MULTI", "0"))      # MULTI=k: the batch is stepped k steps per launch (auv_step_multi), compared with the oracle every k steps
if MULTI:
    n_cmp = 0
    for t0_ in range(0, T, MULTI):
        acts = rs.uniform([-1, -0.15], [1, 0.15], (MULTI, E, 2))
        acts[..., 0] = np.abs(acts[..., 0]) ** 0.3
        env.step_multi(torch.as_tensor(acts, dtype=torch.float64, device="cuda:0").contiguous(), 0, MULTI)
        for k in range(MULTI):
            o_obs, o_rew, o_done = ora.step(acts[k])
            n_done += int(o_done.sum())
        for f in ("CULL_LIMITS", "NEARBY", "COLLISION", "WORLD_IDX", "COUNTERS"):
            g, o = np_(env.read(f)), ora.read(f)
            if not np.array_equal(g, o):
                bad = np.argwhere(g != o)[0]
                raise SystemExit("MISMATCH %s after step %d at %s: gpu %s oracle %s" % (f, t0_ + MULTI, bad, g[tuple(bad)], o[tuple(bad)]))
        for f in ("STATE", "LIDAR_D", "OBS64", "REWARD64", "INFO64", "NAV64", "MOVER_STATE", "EPISODE"):
            dlt = float(np.abs(np_(env.read(f)) - ora.read(f)).max())
            worst = max(worst, dlt)
            if dlt > 1e-8:
                raise SystemExit("MISMATCH %s after step %d: %.3e" % (f, t0_ + MULTI, dlt))
        n_cmp += 1
        if (t0_ // MULTI) % 50 == 49:
            print("step %d: %d comparisons, %d episodes, worst fp64 delta %.2e (%.0f s)" % (t0_ + MULTI, n_cmp, n_done, worst, time.time() - t0), flush=True)
    assert env.health() == dict(handover_ok=1, probe_failures=0, timeouts=0, pending=0), env.health()
    print("SOAK OK multi=%d sub-batches=%d: %d envs x %d steps in launches of %d, every field against the oracle after every launch (%d comparisons), "
          "integer fields bit-exact, %d episodes ended, worst fp64 field delta %.2e" % (MULTI, env.sub_batches, E, T, MULTI, n_cmp, n_done, worst))
    sys.exit(0)
for t in range(T):
    a = rs.uniform([-1, -0.15], [1, 0.15], (E, 2))
    a[:, 0] = np.abs(a[:, 0]) ** 0.3                     # mostly forward: obstacles are met
    if SUB > 1:
        env.step_async(torch.as_tensor(a, device="cuda:0"))
        obs, rew, done, _ = env.step_wait()
    else:
        obs, rew, done, _ = env.step(torch.as_tensor(a, device="cuda:0"))
    o_obs, o_rew, o_done = ora.step(a)
    for f in ("CULL_LIMITS", "NEARBY", "COLLISION", "WORLD_IDX"):
        g, o = np_(env.read(f)), ora.read(f)
        if not np.array_equal(g, o):
            bad = np.argwhere(g != o)[0]
            raise SystemExit("MISMATCH %s step %d at %s: gpu %s oracle %s" % (f, t, bad, g[tuple(bad)], o[tuple(bad)]))
    if not np.array_equal(np_(done), o_done):
        raise SystemExit("MISMATCH done step %d" % t)
    lim = ora.read("CULL_LIMITS")
    n_lim += int((lim[..., 0] != np.iinfo(np.int32).min).sum())
    n_beams += E * cfg.vessel.n_sensors
    n_done += int(o_done.sum())
    if t % 25 == 24:
        for f in ("STATE", "LIDAR_D", "OBS64", "REWARD64", "INFO64", "NAV64"):
            dlt = float(np.abs(np_(env.read(f)) - ora.read(f)).max())
            worst = max(worst, dlt)
            if dlt > 1e-8:
                raise SystemExit("MISMATCH %s step %d: %.3e" % (f, t, dlt))
    if t % 500 == 499:
        print("step %d: %d cull windows, %d beams, %d episodes, worst fp64 delta %.2e (%.0f s)" % (t + 1, n_lim, n_beams, n_done, worst, time.time() - t0), flush=True)
print("SOAK OK mode=%s sub-batches=%d: %d envs x %d steps, %d cull windows and %d beams bit-exact on the integer side, %d episodes ended, worst fp64 field delta %.2e"
      % (mode, env.sub_batches, E, T, n_lim, n_beams, n_done, worst))
