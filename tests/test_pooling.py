"""Feasibility pooling: host sector partition and the CPU oracle against
golden vectors produced by the reference's own functions (G6); the HIP kernel in -m gpu."""
import numpy as np
import pytest

from gym_auv_amd._capi import make_config
from gym_auv_amd.config import effective_reference_config
from gym_auv_amd.pooling import sector_of_sensor, sector_starts
from gym_auv_amd.scenarios import moving_obstacles_world
from gym_auv_amd.world import build_world, pack_bank
from helpers import load

CASES = [(180, 9, 20), (64, 8, 8), (256, 16, 16)]


@pytest.mark.parametrize("S,ns,nps", CASES)
def test_sector_partition_matches_reference(S, ns, nps):
    z = load("g6_pooling.npz")
    np.testing.assert_array_equal(sector_of_sensor(ns, nps), z["S%d_sector_of_sensor" % S])
    np.testing.assert_array_equal(sector_starts(ns, nps), z["S%d_starts" % S])


def _cfg(ns, nps):
    cfg = effective_reference_config(use_lidar=True)
    cfg.vessel.n_sectors, cfg.vessel.n_sensors_per_sector = ns, nps
    return cfg


@pytest.mark.parametrize("S,ns,nps", CASES)
def test_oracle_pooling_matches_reference(S, ns, nps):
    from oracle.pyoracle import Oracle
    z = load("g6_pooling.npz")
    d, ref = z["S%d_d" % S], z["S%d_feasible" % S]
    width = float(z["S%d_width_theta" % S][0])
    o = Oracle(make_config(_cfg(ns, nps)), len(d), pack_bank([build_world(moving_obstacles_world(0, 0, 0))]))
    o.reset()
    o.write("LIDAR_D", d)
    np.testing.assert_array_equal(o.feasibility_pooling(z["S%d_starts" % S], width), ref)   # bit-exact


@pytest.mark.gpu
@pytest.mark.parametrize("S,ns,nps", CASES)
def test_hip_pooling_matches_reference(S, ns, nps):
    from gym_auv_amd.batched_env import BatchedAuvEnv
    z = load("g6_pooling.npz")
    d, ref = z["S%d_d" % S], z["S%d_feasible" % S]
    cfg = _cfg(ns, nps)
    env = BatchedAuvEnv(cfg, pack_bank([build_world(moving_obstacles_world(0, 0, 0))]), len(d), auto_reset=False)
    env.reset()
    env.write("LIDAR_D", d)
    dist, clos = env.feasibility_pooling()
    np.testing.assert_array_equal(dist.cpu().numpy(), ref)                                    # bit-exact
    exp = 1 - np.clip(np.log(1 + ref) / np.log(1 + 150.0), 0, 1)
    np.testing.assert_allclose(clos.cpu().numpy(), exp, rtol=0, atol=1e-6)
