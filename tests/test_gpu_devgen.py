"""-m gpu: on-device scenario generation (auv_generate_worlds) against the host
builder consuming the same random draws, and a rollout on generated worlds against the oracle.

The device builds each world with block-parallel prefix sums, so the cumulative arclengths
differ from NumPy's sequential cumsum in the last bits (~1e-13 relative); tables are compared
at 1e-9, the rollout at the usual 1e-9 on fp64 fields after loading the HOST-built bank of the
same draws into the oracle (flags bit-exact)."""
import numpy as np
import pytest
import torch

from gym_auv_amd import devgen
from gym_auv_amd._capi import make_config
from gym_auv_amd.config import effective_reference_config
from gym_auv_amd.devgen import GeneratedWorlds
from gym_auv_amd.world import build_world, pack_bank

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _env(cfg, spec, n, **kw):
    from gym_auv_amd.batched_env import BatchedAuvEnv
    return BatchedAuvEnv(cfg, spec, n, device="cuda:0", **kw)


def _host_worlds(cfg, spec):
    draws = _np(devgen.sample_draws(spec.n_worlds, spec.n_moving, spec.n_static, seed=spec.seed, device="cuda:0"))
    return [build_world(devgen.world_from_draws(r, spec.n_moving, spec.n_static, dt=cfg.simulation.t_step_size,
                                                vessel_width=cfg.vessel.vessel_width)) for r in draws]


@pytest.mark.parametrize("nm,ns,seed", [(17, 11, 3), (0, 5, 4), (6, 0, 5), (0, 0, 6)])
def test_generated_tables_match_host_builder(nm, ns, seed):
    cfg = effective_reference_config(use_lidar=True)
    spec = GeneratedWorlds(6, nm, ns, seed)
    env = _env(cfg, spec, 6, auto_reset=False)
    host = _host_worlds(cfg, spec)
    cnt = _np(env.read_bank("POLY_CNT"))
    xy, cum = _np(env.read_bank("POLY_XY")), _np(env.read_bank("POLY_CUM"))
    ks, kc = _np(env.read_bank("KNOT_S")), _np(env.read_bank("KNOT_COEF"))
    sc = _np(env.read_bank("WORLD_SCALAR"))
    meta, cull, seg = _np(env.read_bank("OBS_META")), _np(env.read_bank("OBS_CULL")), _np(env.read_bank("SEG"))
    mp, mi, mv = _np(env.read_bank("MV_PARAM")), _np(env.read_bank("MV_INIT")), _np(env.read_bank("MV_VTAB"))
    cb = _np(env.read_bank("CHUNK_BOUND"))
    for w, hw in enumerate(host):
        p = hw.path
        P = len(p.points)
        assert cnt[w] == P
        np.testing.assert_allclose(ks[w], p.knot_s, rtol=0, atol=1e-9)
        np.testing.assert_allclose(kc[w, :-1, 0:4], p.cx.T, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(kc[w, :-1, 4:8], p.cy.T, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(xy[w, :P], p.points, rtol=0, atol=1e-9)
        np.testing.assert_allclose(cum[w, :P], p._cum, rtol=0, atol=1e-9)
        np.testing.assert_allclose(sc[w], hw.scalar, rtol=0, atol=1e-9)
        # every chunk circle really bounds its vertices
        for c in range((P - 1 + 63) // 64):
            v = p.points[c * 64:min(c * 64 + 64, P - 1) + 1]
            assert np.all(np.hypot(v[:, 0] - cb[w, c, 0], v[:, 1] - cb[w, c, 1]) <= cb[w, c, 2])
        K = nm + ns
        if K:
            hm = hw.obs_meta
            np.testing.assert_array_equal(meta[w, :K, 0], hm[:, 0])
            np.testing.assert_array_equal(meta[w, :K, 2], hm[:, 2])
            # movers: index within the world; circles: -3 = simple clockwise ring (back-face flag set on the device)
            np.testing.assert_array_equal(meta[w, :K, 3], np.where(hm[:, 0] == 0, -3, hm[:, 3]))
            np.testing.assert_allclose(cull[w, :K], hw.obs_cull, rtol=0, atol=1e-9)
            for k in range(ns):
                so = meta[w, k, 1] - w * seg.shape[1]          # absolute slot offset -> world-relative
                assert so == 64 * k
                np.testing.assert_allclose(seg[w, so:so + hm[k, 2]], hw.seg[hm[k, 1]:hm[k, 1] + hm[k, 2]], rtol=0, atol=1e-9)
        if nm:
            np.testing.assert_allclose(mp[w], hw.mv_param, rtol=0, atol=1e-9)
            np.testing.assert_allclose(mi[w], hw.mv_init, rtol=0, atol=1e-9)
            np.testing.assert_allclose(mv[w], np.concatenate(hw.mv_vtab), rtol=0, atol=1e-9)
    env.close()


def test_rollout_on_generated_worlds_vs_oracle():
    from oracle.pyoracle import Oracle
    n = 48
    cfg = effective_reference_config(use_lidar=True)
    cfg.episode.max_timesteps = 40
    spec = GeneratedWorlds(24, seed=11)
    env = _env(cfg, spec, n, auto_reset=True)
    ora = Oracle(make_config(cfg, auto_reset=True), n, pack_bank(_host_worlds(cfg, spec)))
    np.testing.assert_allclose(_np(env.reset()), ora.reset(), rtol=0, atol=1e-6)
    rs = np.random.RandomState(2)
    n_done = 0
    for t in range(90):
        a = rs.uniform([-1, -0.15], [1, 0.15], (n, 2))
        obs, rew, done, _ = env.step(torch.as_tensor(a, device="cuda:0"))
        o_obs, o_rew, o_done = ora.step(a)
        np.testing.assert_array_equal(_np(done), o_done)
        for f in ("STATE", "LIDAR_D", "OBS64", "REWARD64", "INFO64", "NAV64", "MOVER_STATE"):
            np.testing.assert_allclose(_np(env.read(f)), ora.read(f), rtol=0, atol=1e-8, err_msg="%s step %d" % (f, t))
        for f in ("WORLD_IDX", "NEARBY", "COLLISION"):
            np.testing.assert_array_equal(_np(env.read(f)), ora.read(f), err_msg="%s step %d" % (f, t))
        np.testing.assert_allclose(_np(obs), o_obs, rtol=0, atol=1e-6)
        n_done += int(o_done.sum())
    assert n_done >= n


def test_regenerate_in_place_and_determinism():
    cfg = effective_reference_config(use_lidar=True)
    spec = GeneratedWorlds(16, seed=21)
    env = _env(cfg, spec, 32)
    first = {k: _np(env.read_bank(k)).copy() for k in ("KNOT_S", "OBS_CULL", "MV_INIT", "WORLD_SCALAR")}
    obs_a = _np(env.reset()).copy()
    env.step(torch.zeros((32, 2), device="cuda:0"))
    # other draws -> other worlds, every env reset onto them
    env.generate(GeneratedWorlds(16, seed=22))
    assert not np.allclose(_np(env.read_bank("KNOT_S")), first["KNOT_S"])
    assert int(_np(env.read("COUNTERS"))[:, 0].max()) == 0
    # the first seed again -> bit-identical tables and first observation
    env.generate(spec)
    for k, v in first.items():
        np.testing.assert_array_equal(_np(env.read_bank(k)), v)
    np.testing.assert_array_equal(_np(env.reset()), obs_a)
    # a different shape re-allocates
    env.generate(GeneratedWorlds(5, 3, 2, seed=1))
    assert env.read_bank("OBS_META").shape == (5, 5, 4)
    obs, _, _, _ = env.step(torch.zeros((32, 2), device="cuda:0"))
    assert torch.isfinite(obs).all()


def test_generate_argument_errors():
    cfg = effective_reference_config(use_lidar=True)
    env = _env(cfg, GeneratedWorlds(4, seed=1), 4)
    with pytest.raises(ValueError):
        env.generate(GeneratedWorlds(4, seed=1), draws=torch.zeros((4, 7), device="cuda:0", dtype=torch.float64))
    from gym_auv_amd.scenarios import moving_obstacles_world
    host_env = _env(cfg, [moving_obstacles_world(5)], 2)
    with pytest.raises(RuntimeError):
        host_env.read_bank("KNOT_S")


def test_exhausted_candidate_pool_device_equals_host():
    """Draws whose whole candidate pool is rejected (6 km radii): the device takes further candidates from the
    keyed generator exactly as the host mirror does (devgen.extra_candidate), and no obstacle ends up on the
    vessel or the goal (utils/helpers.py:13-33)."""
    cfg = effective_reference_config(use_lidar=True)
    spec = GeneratedWorlds(4, 17, 11, seed=31)
    draws = devgen.sample_draws(4, 17, 11, seed=31, device="cuda:0")
    C = devgen.CAND
    for w, base in ((0, 11), (1, 11 + 5 * (3 * C + 2)), (2, 11 + 17 * (3 * C + 2) + 2 * 3 * C), (2, 11 + 17 * (3 * C + 2))):
        draws[w, base + 2: base + 3 * C: 3] = 6000.0
    env = _env(cfg, spec, 4, auto_reset=False)
    env.generate(spec, draws=draws)
    host = [build_world(devgen.world_from_draws(r, 17, 11, dt=cfg.simulation.t_step_size, vessel_width=cfg.vessel.vessel_width))
            for r in _np(draws)]
    mp, cull, sc = _np(env.read_bank("MV_PARAM")), _np(env.read_bank("OBS_CULL")), _np(env.read_bank("WORLD_SCALAR"))
    for w, hw in enumerate(host):
        np.testing.assert_allclose(mp[w], hw.mv_param, rtol=0, atol=1e-8)
        np.testing.assert_allclose(cull[w, :28], hw.obs_cull, rtol=0, atol=1e-8)
        assert np.all(mp[w, :, 0] < 100) and np.all(cull[w, :11, 2] < 100)        # no 6 km obstacle was kept
        goal, start = sc[w, 1:3], sc[w, 3:5]
        for pos, rad in list(zip(mp[w, :, 1:3], mp[w, :, 0])) + list(zip(cull[w, :11, 0:2], cull[w, :11, 2])):
            assert np.hypot(*(pos - start)) - cfg.vessel.vessel_width - rad > 0
            assert np.hypot(*(pos - goal)) - rad > 0
    env.close()
