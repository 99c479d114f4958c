"""-m gpu: BASELINE.json's full sizes (4096 envs x 64 / 180 / 256 sensors) checked through
size-independent properties, plus oracle parity on a random subset of the batch.

Worlds are small banks cycled over the envs (world generation is host work, not what is
tested here); every env still has its own state, actions and episode."""
import numpy as np
import pytest
import torch

from gym_auv_amd._capi import make_config
from gym_auv_amd.config import effective_reference_config
from gym_auv_amd.scenarios import moving_obstacles_world, polygon_world, static_circles_world
from gym_auv_amd.world import build_world, pack_bank

pytestmark = pytest.mark.gpu
N = 4096


def _np(t):
    return t.detach().cpu().numpy()


def _bank(kind, n_worlds=96):
    gen = {"circles20": lambda s: static_circles_world(s, 20),
           "polygons50": lambda s: polygon_world(s, 50),
           "moving28": lambda s: moving_obstacles_world(s),
           "mixed47": lambda s: polygon_world(s, 10, n_circles=20, n_moving=17)}[kind]
    return pack_bank([build_world(gen(3000 + i)) for i in range(n_worlds)])


CASES = [("circles20", 8, 8), ("polygons50", 9, 20), ("moving28", 9, 20), ("mixed47", 16, 16)]


@pytest.mark.parametrize("kind,ns,nps", CASES)
def test_fullsize_subset_parity_and_invariants(kind, ns, nps):
    from gym_auv_amd.batched_env import BatchedAuvEnv
    from oracle.pyoracle import Oracle
    cfg = effective_reference_config(use_lidar=True)
    cfg.vessel.n_sectors, cfg.vessel.n_sensors_per_sector = ns, nps
    cfg.episode.max_timesteps = 23                     # force episode turnover at full batch size
    S = ns * nps
    bank = _bank(kind)
    W = int(bank["n_worlds"])
    env = BatchedAuvEnv(cfg, bank, N, device="cuda:0", auto_reset=True)
    rs = np.random.RandomState(7)
    sub = np.sort(rs.choice(N, 64, replace=False))
    # the oracle only simulates the subset; bind its env i to the world GPU env sub[i] uses, and
    # give it N-compatible auto-reset stepping (next world = (w + N) % W)
    ora = Oracle(make_config(cfg, auto_reset=False), len(sub), bank)
    w_now = (sub % W).astype(np.int32)
    obs = env.reset()
    o_obs = ora.reset(world_idx=w_now)
    np.testing.assert_allclose(_np(obs)[sub], o_obs, rtol=0, atol=1e-6)
    acts = torch.as_tensor(rs.uniform([-1, -0.15], [1, 0.15], (40, N, 2)), device="cuda:0")
    a_np = _np(acts)
    total_done = 0
    for t in range(40):
        obs, rew, done, _ = env.step(acts[t])
        o_obs, o_rew, o_done = ora.step(a_np[t][sub])
        g_done = _np(done)
        np.testing.assert_array_equal(g_done[sub], o_done)
        # emulate the batch's auto-reset for the oracle's finished envs
        if o_done.any():
            w_now = np.where(o_done > 0, (w_now + N) % W, w_now).astype(np.int32)
            o_obs_r = ora.reset(mask=o_done, world_idx=w_now)
            o_obs = np.where(o_done[:, None] > 0, o_obs_r, o_obs)
        g_obs = _np(obs)
        np.testing.assert_allclose(g_obs[sub], o_obs, rtol=0, atol=1e-6, err_msg="obs step %d" % t)
        np.testing.assert_allclose(_np(rew)[sub], o_rew, rtol=1e-6, atol=1e-4)
        np.testing.assert_allclose(_np(env.read("STATE"))[:, sub], ora.read("STATE"), rtol=0, atol=1e-9)
        np.testing.assert_allclose(_np(env.read("LIDAR_D"))[sub], ora.read("LIDAR_D"), rtol=0, atol=1e-9)
        np.testing.assert_array_equal(_np(env.read("WORLD_IDX"))[sub], w_now)
        # invariants over the WHOLE batch
        d = _np(env.read("LIDAR_D"))
        assert (d >= 0).all() and (d <= cfg.vessel.sensor_range).all()
        assert np.isfinite(g_obs).all() and (np.abs(g_obs) <= 1.0).all() and g_obs.shape == (N, 6 + S)
        psi = _np(env.read("STATE"))[2]
        assert (psi >= -np.pi).all() and (psi < np.pi).all()
        total_done += int(g_done.sum())
    assert total_done >= N                              # every env turned over at least once
    ep = env.episode_stats()
    assert int(ep["episodes"].sum().item()) == total_done
    assert float(ep["episode_length"].max().item()) <= cfg.episode.max_timesteps
    env.close()


def test_fullsize_permutation_invariance_and_determinism():
    """Env results do not depend on where in the batch an env sits nor on the run."""
    from gym_auv_amd.batched_env import BatchedAuvEnv
    cfg = effective_reference_config(use_lidar=True)
    bank = _bank("mixed47", 64)
    W = int(bank["n_worlds"])
    rs = np.random.RandomState(3)
    perm = rs.permutation(N)
    acts = rs.uniform([-1, -0.15], [1, 0.15], (12, N, 2))
    world = (np.arange(N) % W).astype(np.int32)
    outs = []
    for order in (np.arange(N), perm, np.arange(N)):
        env = BatchedAuvEnv(cfg, bank, N, device="cuda:0", auto_reset=False)
        env.reset(world_idx=torch.as_tensor(world[order]))
        for t in range(12):
            obs, rew, done, _ = env.step(torch.as_tensor(acts[t][order], device="cuda:0"))
        inv = np.empty(N, dtype=np.int64)
        inv[order] = np.arange(N)
        outs.append((_np(env.read("OBS64"))[inv], _np(env.read("REWARD64"))[inv], _np(env.read("STATE"))[:, inv]))
        env.close()
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)             # bitwise: same arithmetic wherever the env sits
    for a, b in zip(outs[0], outs[2]):
        np.testing.assert_array_equal(a, b)             # bitwise run-to-run


def test_fullsize_exact_cull_never_sees_less():
    """cull="exact" ranges <= cull="reference" ranges on every beam of the full batch."""
    from gym_auv_amd.batched_env import BatchedAuvEnv
    cfg = effective_reference_config(use_lidar=True)
    bank = _bank("polygons50", 64)
    a = BatchedAuvEnv(cfg, bank, N, device="cuda:0", cull="reference", auto_reset=False)
    b = BatchedAuvEnv(cfg, bank, N, device="cuda:0", cull="exact", auto_reset=False)
    a.reset(), b.reset()
    act = torch.zeros((N, 2), device="cuda:0")
    act[:, 0] = 1.0
    for _ in range(5):
        a.step(act), b.step(act)
    da, db = _np(a.read("LIDAR_D")), _np(b.read("LIDAR_D"))
    assert (db <= da + 1e-12).all()
    assert (db < da - 1e-6).any()                       # the modulo bug hides something somewhere


def test_config4_shard_graph_captured_8192x256():
    """BASELINE configs[4] at its per-GPU size: 8192 envs x 256 sensors, mixed world (20 circles + 10 polygons
    + 17 movers), the step captured in a hipGraph.  (a) a one-step graph and a five-step graph over an action
    ring replay bit for bit what eager launches compute, 30 steps, auto-reset on; (b) oracle parity on a 64-env
    subset of the eager run (pattern of test_fullsize_subset_parity_and_invariants).  VERDICT r1 next #1(c)."""
    from gym_auv_amd.batched_env import BatchedAuvEnv
    from oracle.pyoracle import Oracle
    n, ns, nps, steps, ring = 8192, 16, 16, 30, 10
    cfg = effective_reference_config(use_lidar=True)
    cfg.vessel.n_sectors, cfg.vessel.n_sensors_per_sector = ns, nps
    cfg.episode.max_timesteps = 17                         # episode turnover inside the 30 steps
    bank = _bank("mixed47", 64)
    W = int(bank["n_worlds"])
    rs = np.random.RandomState(11)
    acts = torch.as_tensor(rs.uniform([-1, -0.15], [1, 0.15], (ring, n, 2)), dtype=torch.float32, device="cuda:0")
    eager = BatchedAuvEnv(cfg, bank, n, device="cuda:0", auto_reset=True)
    g1 = BatchedAuvEnv(cfg, bank, n, device="cuda:0", auto_reset=True)
    g5 = BatchedAuvEnv(cfg, bank, n, device="cuda:0", auto_reset=True)
    for e in (eager, g1, g5):
        e.reset()
    g1.capture_graph(torch.float32, slots=ring, steps=1).copy_(acts)
    g5.capture_graph(torch.float32, slots=ring, steps=5).copy_(acts)
    sub = np.sort(rs.choice(n, 64, replace=False))
    ora = Oracle(make_config(cfg, auto_reset=False), len(sub), bank)
    w_now = (sub % W).astype(np.int32)
    ora.reset(world_idx=w_now)
    a_np = _np(acts)
    n_done = 0
    for t in range(steps):
        obs, rew, done, _ = eager.step(acts[t % ring])
        o1, r1, d1, _ = g1.step_graph()
        torch.cuda.synchronize()
        assert torch.equal(obs, o1) and torch.equal(rew, r1) and torch.equal(done, d1), "one-step graph, step %d" % t
        if t % 5 == 4:
            o5, r5, d5, _ = g5.step_graph()
            torch.cuda.synchronize()
            assert torch.equal(obs, o5) and torch.equal(rew, r5) and torch.equal(done, d5), "five-step graph, step %d" % t
        o_obs, o_rew, o_done = ora.step(a_np[t % ring][sub])
        g_done = _np(done)
        np.testing.assert_array_equal(g_done[sub], o_done)
        if o_done.any():
            w_now = np.where(o_done > 0, (w_now + n) % W, w_now).astype(np.int32)
            o_obs_r = ora.reset(mask=o_done, world_idx=w_now)
            o_obs = np.where(o_done[:, None] > 0, o_obs_r, o_obs)
        np.testing.assert_allclose(_np(obs)[sub], o_obs, rtol=0, atol=1e-6, err_msg="obs step %d" % t)
        np.testing.assert_allclose(_np(rew)[sub], o_rew, rtol=1e-6, atol=1e-4)
        n_done += int(g_done.sum())
    for f in ("STATE", "LIDAR_D", "OBS64", "INFO64", "NAV64", "MOVER_STATE", "EPISODE", "COUNTERS", "NEARBY", "WORLD_IDX"):
        a = eager.read(f)
        assert torch.equal(a, g1.read(f)), f
        assert torch.equal(a, g5.read(f)), f
    np.testing.assert_allclose(_np(eager.read("STATE"))[:, sub], ora.read("STATE"), rtol=0, atol=1e-9)
    np.testing.assert_allclose(_np(eager.read("LIDAR_D"))[sub], ora.read("LIDAR_D"), rtol=0, atol=1e-9)
    assert n_done >= n                                     # every env turned over at least once
    for e in (eager, g1, g5):
        e.close()


@pytest.mark.parametrize("interleave", [False, True])
def test_bench_launch_shape_4x1024_polygons50(interleave):
    """The launch configuration bench.py times: 4096 x 180, 50 polygons, the bench's own bank of two
    worlds per environment (world_seeds), the batch stepped OPEN-LOOP as four chains of 1024 on probe-selected streams
    (set_sub_batches(4) + step_pipelined), in stretches of several steps without any synchronisation in between so that
    the chains drift apart as they do in the timed loop; >= 60 steps with forced episode turnover.  Against (a) ONE chain
    over the same environments, bit for bit after every stretch, and (b) the CPU oracle on a 64-environment subset, every
    step.  `interleave`: a torch kernel is launched on every chain's stream between its steps (the PPO / pilot
    situation: foreign kernels beside the one-launch step's in-launch hand-overs)."""
    import bench
    from gym_auv_amd.batched_env import BatchedAuvEnv
    from gym_auv_amd.world import build_bank_parallel
    from oracle.pyoracle import Oracle
    cfg = effective_reference_config(use_lidar=True)
    cfg.vessel.n_sectors, cfg.vessel.n_sensors_per_sector = 9, 20
    cfg.episode.max_timesteps = 23                          # force episode turnover at full batch size
    seeds = bench.world_seeds(0, N, N, 2)
    bank = build_bank_parallel("polygon_world", seeds, procs=min(16, bench.host_cores()), n_polygons=50)
    W = int(bank["n_worlds"])
    assert W == 2 * N
    one = BatchedAuvEnv(cfg, bank, N, device="cuda:0", auto_reset=True)
    four = BatchedAuvEnv(cfg, bank, N, device="cuda:0", auto_reset=True)
    slices = four.set_sub_batches(4, strict=True)
    assert [c for _, c in slices] == [1024] * 4 and four.effective_step_mode(1024) == "one_launch"
    rs = np.random.RandomState(17)
    sub = np.sort(np.concatenate([lo + rs.choice(cnt, 16, replace=False) for lo, cnt in slices]))   # 16 of every chain
    ora = Oracle(make_config(cfg, auto_reset=False), len(sub), bank)
    w_now = (sub % W).astype(np.int32)
    one.reset(), four.reset()
    ora.reset(world_idx=w_now)
    n_steps, pool_n = 66, 16
    pool = torch.as_tensor(rs.uniform([-1, -0.15], [1, 0.15], (pool_n, N, 2)), dtype=torch.float32, device="cuda:0")
    a_np = _np(pool).astype(np.float64)
    scratch = torch.zeros(N, device="cuda:0")
    fields = ("STATE", "LIDAR_D", "OBS64", "INFO64", "NAV64", "EPISODE", "COUNTERS", "NEARBY", "STEP_INFO", "WORLD_IDX",
              "CULL_LIMITS", "COLLISION", "REWARD64")
    t, n_done, stretch = 0, 0, 0
    while t < n_steps:
        L = min(n_steps - t, 1 + (stretch * 5) % 7)         # stretches of 1, 6, 4, 2, 7, 5, 3, ... steps
        stretch += 1
        torch.cuda.synchronize()
        for j in range(L):
            four.step_pipelined(pool[(t + j) % pool_n])
            if interleave:
                for i, (lo, cnt) in enumerate(slices):
                    with torch.cuda.stream(four._sub_streams[i]):
                        torch.mul(four.obs[lo:lo + cnt, 4], 0.15, out=scratch[lo:lo + cnt])
        for j in range(L):
            obs, rew, done, _ = one.step(pool[(t + j) % pool_n])
            o_obs, o_rew, o_done = ora.step(a_np[(t + j) % pool_n][sub])
            g_done = _np(done)
            np.testing.assert_array_equal(g_done[sub], o_done)
            if o_done.any():
                w_now = np.where(o_done > 0, (w_now + N) % W, w_now).astype(np.int32)
                o_obs_r = ora.reset(mask=o_done, world_idx=w_now)
                o_obs = np.where(o_done[:, None] > 0, o_obs_r, o_obs)
            np.testing.assert_allclose(_np(obs)[sub], o_obs, rtol=0, atol=1e-6, err_msg="obs step %d" % (t + j))
            np.testing.assert_allclose(_np(rew)[sub], o_rew, rtol=1e-6, atol=1e-4)
            n_done += int(g_done.sum())
        t += L
        torch.cuda.synchronize()
        assert torch.equal(one.obs, four.obs) and torch.equal(one.reward, four.reward) and torch.equal(one.done, four.done), t
        for f in fields:
            assert torch.equal(one.read(f), four.read(f)), (t, f)
    np.testing.assert_allclose(_np(four.read("STATE"))[:, sub], ora.read("STATE"), rtol=0, atol=1e-9)
    np.testing.assert_allclose(_np(four.read("LIDAR_D"))[sub], ora.read("LIDAR_D"), rtol=0, atol=1e-9)
    assert n_done >= 2 * N                                  # every environment turned over, most of them twice
    assert four.health() == dict(handover_ok=1, probe_failures=0, timeouts=0, pending=0)   # (incl. the four-stream probe of set_sub_batches)
    one.close(), four.close()


@pytest.mark.parametrize("one_graph", [False, True])
def test_config4_shard_captured_chains_8192x256(one_graph):
    """BASELINE configs[4] names a hipGraph-captured step: the 8192 x 256 mixed shard stepped by CAPTURED CHAINS -- five steps
    of each of four sub-batches per replay, as four linear graphs on the sub-batches' streams or as one graph with four
    branches -- bit for bit what eager launches compute over 40 steps with auto-reset (every chain walks the action ring
    with a position of its own)."""
    from gym_auv_amd.batched_env import BatchedAuvEnv
    n, ns, nps, steps, ring, per = 8192, 16, 16, 40, 10, 5
    cfg = effective_reference_config(use_lidar=True)
    cfg.vessel.n_sectors, cfg.vessel.n_sensors_per_sector = ns, nps
    cfg.episode.max_timesteps = 17
    bank = _bank("mixed47", 64)
    rs = np.random.RandomState(11)
    acts = torch.as_tensor(rs.uniform([-1, -0.15], [1, 0.15], (ring, n, 2)), dtype=torch.float32, device="cuda:0")
    eager = BatchedAuvEnv(cfg, bank, n, device="cuda:0", auto_reset=True)
    chains = BatchedAuvEnv(cfg, bank, n, device="cuda:0", auto_reset=True)
    eager.reset(), chains.reset()
    chains.set_sub_batches(4, strict=True)
    chains.capture_graph_chains(torch.float32, slots=ring, steps=per, one_graph=one_graph).copy_(acts)
    for t in range(steps):
        obs, rew, done, _ = eager.step(acts[t % ring])
        if t % per == per - 1:
            o, r, d, _ = chains.step_graph()
            if t % (2 * per) == per - 1:
                continue                                     # (two replays back to back: the chains run ten steps unsynchronised)
            torch.cuda.synchronize()
            assert torch.equal(obs, o) and torch.equal(rew, r) and torch.equal(done, d), "captured chains, step %d" % t
    torch.cuda.synchronize()
    for f in ("STATE", "LIDAR_D", "OBS64", "INFO64", "NAV64", "MOVER_STATE", "EPISODE", "COUNTERS", "NEARBY", "WORLD_IDX"):
        assert torch.equal(eager.read(f), chains.read(f)), f
    assert int(eager.read("COUNTERS")[:, 2].sum()) >= n     # every env turned over at least once
    assert chains.health()["timeouts"] == 0
    eager.close(), chains.close()


def test_captured_chains_in_the_side_by_side_shape_walk_their_own_ring_positions():
    """a captured chain whose slice steps in the three-launch shape (set_step_mode('side_by_side'), no LiDAR, or
    hand-overs disabled after a failed probe) must advance ITS ring position -- the k23 launch used to advance it only for
    the chain that holds environment 0, the other chains then replayed slot 0 for good.  Four captured chains, four steps per
    replay, distinct actions per ring slot: bitwise what eager steps of the same shape compute."""
    from gym_auv_amd.batched_env import BatchedAuvEnv
    n, steps, ring, per = 1024, 24, 8, 4
    cfg = effective_reference_config(use_lidar=True)
    cfg.vessel.n_sectors, cfg.vessel.n_sensors_per_sector = 8, 8
    cfg.episode.max_timesteps = 11
    bank = _bank("mixed47", 32)
    rs = np.random.RandomState(5)
    acts = torch.as_tensor(rs.uniform([-1, -0.15], [1, 0.15], (ring, n, 2)), dtype=torch.float32, device="cuda:0")
    eager = BatchedAuvEnv(cfg, bank, n, device="cuda:0", auto_reset=True)
    chains = BatchedAuvEnv(cfg, bank, n, device="cuda:0", auto_reset=True)
    eager.set_step_mode("side_by_side"), chains.set_step_mode("side_by_side")
    eager.reset(), chains.reset()
    chains.set_sub_batches(4, strict=True)
    assert chains.effective_step_mode(n // 4) == "side_by_side"
    chains.capture_graph_chains(torch.float32, slots=ring, steps=per).copy_(acts)
    for t in range(steps):
        obs, rew, done, _ = eager.step(acts[t % ring])
        if t % per == per - 1:
            o, r, d, _ = chains.step_graph()
            torch.cuda.synchronize()
            assert torch.equal(obs, o) and torch.equal(rew, r) and torch.equal(done, d), "captured side-by-side chains, step %d" % t
    for f in ("STATE", "LIDAR_D", "OBS64", "INFO64", "COUNTERS", "WORLD_IDX"):
        assert torch.equal(eager.read(f), chains.read(f)), f
    assert int(eager.read("COUNTERS")[:, 2].sum()) >= n
    eager.close(), chains.close()


@pytest.mark.parametrize("kind,n,ns,nps,T", [("polygons50", 4096, 9, 20, 64), ("mixed47", 8192, 16, 16, 48)])
def test_fullsize_several_steps_per_launch_bitwise_and_against_the_oracle(kind, n, ns, nps, T):
    """BASELINE's full shapes through `auv_step_multi`: two launches of T steps are bit for bit 2 T one-step
    launches (every output of every step, every field at the end), and a 64-env subset agrees with the oracle."""
    from gym_auv_amd.batched_env import BatchedAuvEnv
    from oracle.pyoracle import Oracle
    cfg = effective_reference_config(use_lidar=True)
    cfg.vessel.n_sectors, cfg.vessel.n_sensors_per_sector = ns, nps
    cfg.episode.max_timesteps = 37                     # every env turns over inside each launch
    bank = _bank(kind)
    W = int(bank["n_worlds"])
    one = BatchedAuvEnv(cfg, bank, n, device="cuda:0", auto_reset=True)
    mul = BatchedAuvEnv(cfg, bank, n, device="cuda:0", auto_reset=True)
    rs = np.random.RandomState(11)
    sub = np.sort(rs.choice(n, 64, replace=False))
    ora = Oracle(make_config(cfg, auto_reset=False), len(sub), bank)
    w_now = (sub % W).astype(np.int32)
    one.reset(), mul.reset(), ora.reset(world_idx=w_now)
    a_np = rs.uniform([-1, -0.15], [1, 0.15], (T, n, 2))
    a_np[..., 0] = np.abs(a_np[..., 0]) ** 0.3
    ring = torch.as_tensor(a_np, device="cuda:0").contiguous()
    n_done = 0
    for rep in range(2):
        for t in range(T):
            o, r, dn, _ = one.step(ring[t])
            o_obs, o_rew, o_done = ora.step(a_np[t][sub])
            g_done = _np(dn)[sub]
            np.testing.assert_array_equal(g_done, o_done)
            np.testing.assert_allclose(_np(r)[sub], o_rew, rtol=1e-6, atol=1e-4)
            n_done += int(o_done.sum())
            if o_done.any():                           # the oracle's subset follows the batch's world rotation
                w_now = np.where(o_done > 0, (w_now + n) % W, w_now).astype(np.int32)
                ora.reset(mask=o_done, world_idx=w_now)
        mul.step_multi(ring, 0, T)
        torch.cuda.synchronize()
        # (a multi-step launch leaves the LAST step's outputs in the env's buffers)
        assert torch.equal(one.obs, mul.obs) and torch.equal(one.reward, mul.reward) and torch.equal(one.done, mul.done), rep
        for f in ("STATE", "LIDAR_D", "OBS64", "REWARD64", "INFO64", "NAV64", "MOVER_STATE", "NEARBY", "COLLISION", "COUNTERS", "EPISODE",
                  "CULL_LIMITS", "STEP_INFO", "WORLD_IDX"):
            assert torch.equal(one.read(f), mul.read(f)), (rep, f)
    assert n_done >= 64 and int(one.read("COUNTERS")[:, 2].sum()) >= 2 * n
    la, lb = one.episode_log().cpu().numpy(), mul.episode_log().cpu().numpy()
    np.testing.assert_array_equal(la[np.lexsort(la.T[::-1])], lb[np.lexsort(lb.T[::-1])])
    assert mul.health() == dict(handover_ok=1, probe_failures=0, timeouts=0, pending=0)
    one.close(), mul.close()


@pytest.mark.parametrize("kind,ns,nps,auto_max", [("mixed47", 16, 16, 34), ("polygons50", 9, 20, 96)])
def test_lidar_stage_capacity_is_picked_for_occupancy_and_changes_no_result(kind, ns, nps, auto_max):
    """The LiDAR wave's LDS stage (auv_lidar_stage) is sized per bank so that the one-launch step keeps 16 waves per CU: 34 segments
    at 256 beams + 47 obstacles + 17 movers, 96 at 180 beams + 50 polygons.  A smaller stage only means more batches in a crowded
    environment's pair sweep: every field of every step is bit for bit the same with 32, the picked value and 96 segments."""
    from gym_auv_amd.batched_env import BatchedAuvEnv
    cfg = effective_reference_config(use_lidar=True)
    cfg.vessel.n_sectors, cfg.vessel.n_sensors_per_sector = ns, nps
    cfg.episode.max_timesteps = 31
    bank = _bank(kind, 48)
    n = 1024
    envs = [BatchedAuvEnv(cfg, bank, n, device="cuda:0", auto_reset=True) for _ in range(3)]
    picked = envs[0].lidar_stage()
    assert 32 <= picked <= auto_max and picked % 2 == 0, picked
    if auto_max == 96:
        assert picked == 96
    assert envs[1].lidar_stage(32) == 32 and envs[2].lidar_stage(96) == 96
    with pytest.raises(RuntimeError):
        envs[1].lidar_stage(31)
    rs = np.random.RandomState(5)
    a_np = rs.uniform([-1, -0.15], [1, 0.15], (70, n, 2))
    a_np[..., 0] = np.abs(a_np[..., 0]) ** 0.3
    acts = torch.as_tensor(a_np, device="cuda:0")
    for e in envs:
        e.reset()
    for t in range(70):
        outs = [e.step(acts[t]) for e in envs]
        for o in outs[1:]:
            assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) and torch.equal(outs[0][2], o[2]), t
        if t % 10 == 9:
            for f in ("STATE", "LIDAR_D", "OBS64", "REWARD64", "INFO64", "NAV64", "MOVER_STATE", "NEARBY", "COLLISION", "COUNTERS", "EPISODE",
                      "CULL_LIMITS", "STEP_INFO", "WORLD_IDX"):
                for e in envs[1:]:
                    assert torch.equal(envs[0].read(f), e.read(f)), (t, f)
    assert int(envs[0].read("COUNTERS")[:, 2].sum()) >= n
    # several steps per launch and the three-launch shape with the smallest stage
    envs[1].set_step_mode("side_by_side")
    ring = acts[:16].contiguous()
    for t in range(16):
        envs[1].step(ring[t]), envs[2].step(ring[t])
    envs[0].step_multi(ring, 0, 16)
    torch.cuda.synchronize()
    for f in ("STATE", "LIDAR_D", "OBS64", "REWARD64", "NAV64", "MOVER_STATE", "COUNTERS", "CULL_LIMITS", "WORLD_IDX"):
        for e in envs[1:]:
            assert torch.equal(envs[0].read(f), e.read(f)), f
    for e in envs:
        assert e.health()["timeouts"] == 0
        e.close()
