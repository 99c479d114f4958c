"""Times the REFERENCE's own Python step() -- /root/reference/gym_auv, imported through the harness
(bootstrap.py: inert gym / pygame / turtle, the numpy shim for Shapely/GEOS) -- in the build container
and writes oracle/ref_harness/reference_timing.json.  TEST / MEASUREMENT TOOLING, build container only:
the reference cannot travel to the GPU box, so bench.py only COPIES the committed result into
`cpu_baseline.reference_python` as a stated constant with its provenance; it never runs this.

Scenario: MovingObstaclesNoRules (17 moving + 11 static obstacles, envs/movingobstacles.py:98-103), LiDAR on,
180 sensors, effective dt 0.5 s -- BASELINE configs[0].  One core, float64, the reference's single-threaded
per-object Python loops (vessel.py:399-409, sensor.py:140-159).  Geometry primitives are the harness' numpy
shim, not real GEOS (absent here), so this is indicative of the reference's cost, not a GEOS measurement.

    python oracle/ref_harness/time_reference.py [--steps 300] [--seeds 3]
"""
import argparse
import contextlib
import copy
import io
import json
import os
import platform
import random
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import bootstrap  # noqa: E402

gym_auv = bootstrap.install()
import gym_auv.envs.movingobstacles as mo  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--seeds", type=int, default=3)
    args = ap.parse_args()
    cfg = copy.deepcopy(gym_auv.DEFAULT_CONFIG)
    cfg.simulation.t_step_size = 0.5
    cfg.episode.min_goal_distance = 0.1
    cfg.vessel.use_lidar = True
    cfg.vessel.n_sectors, cfg.vessel.n_sensors_per_sector = 9, 20
    per_seed, n_total, t_total = [], 0, 0.0
    for seed in range(100, 100 + args.seeds):
        with contextlib.redirect_stdout(io.StringIO()):
            np.random.seed(seed)
            random.seed(seed)
            env = mo.MovingObstaclesNoRules(env_config=cfg, renderer=None)
            env.seed(seed)
            env.reset()
            rs = np.random.RandomState(seed)
            n, t0 = 0, time.perf_counter()
            for _ in range(args.steps):
                _, _, done, _ = env.step(rs.uniform([0.0, -0.15], [1.0, 0.15]))
                n += 1
                if done:
                    env.reset()
            dt = time.perf_counter() - t0
        per_seed.append(round(1e3 * dt / n, 3))
        n_total, t_total = n_total + n, t_total + dt
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    res = dict(ms_per_step=round(1e3 * t_total / n_total, 3), env_steps_per_s_per_core=round(n_total / t_total, 1),
               ms_per_step_by_seed=per_seed, steps=n_total, cores=1,
               scenario="MovingObstaclesNoRules (17 moving + 11 static obstacles), LiDAR on, 180 sensors, dt 0.5 s",
               what="the reference's own gym_auv step() (environment.py:292-366) incl. its resets on done, imported from "
                    "/root/reference through oracle/ref_harness/bootstrap.py",
               geometry="numpy shim for Shapely/GEOS (oracle/ref_harness/shim): real GEOS is not installable here",
               host="build container (not the GPU box): %s, python %s, numpy %s" % (cpu or platform.processor(), platform.python_version(), np.__version__),
               script="oracle/ref_harness/time_reference.py")
    with open(os.path.join(HERE, "reference_timing.json"), "w") as f:
        json.dump(res, f, indent=1)
        f.write("\n")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
