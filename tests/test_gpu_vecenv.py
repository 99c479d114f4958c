"""-m gpu: the stable-baselines VecEnv protocol over the batched HIP path (gym_auv_amd/vec_env.py), driven the way the
reference drives its SubprocVecEnv: construction and stepping as /root/reference/scripts/run.py:293-296 (and
stable-baselines' runner: step_async / step_wait), the training callback's attribute pulls as run.py:415-426
(`get_attr('last_episode' | 'config' | 'obstacles' | 'history' | 'total_t_steps')`), and the per-episode records of
save_latest_episode (gym_auv/environment.py:466-489) checked against the CPU oracle stepped beside it."""
import numpy as np
import pytest
import torch

from gym_auv_amd._capi import make_config
from gym_auv_amd.config import effective_reference_config
from gym_auv_amd.scenarios import moving_obstacles_world, polygon_world
from gym_auv_amd.world import build_world, pack_bank

pytestmark = pytest.mark.gpu
HISTORY_KEYS = {"cross_track_error", "reached_goal", "collision", "reward", "timesteps", "duration", "progress", "pathlength"}


def _worlds(n):
    return [build_world(moving_obstacles_world(300 + i) if i % 2 == 0 else polygon_world(300 + i, 12, n_circles=4, n_moving=3))
            for i in range(n)]


@pytest.mark.parametrize("sub_batches", [1, 2])
def test_vecenv_protocol_as_the_reference_training_loop_uses_it(sub_batches):
    from gym_auv_amd.vec_env import AuvVecEnv
    from oracle.pyoracle import Oracle
    num_cpu = 96                                             # (the reference: NUM_CPU = 8 worker processes)
    cfg = effective_reference_config(use_lidar=True)
    cfg.episode.max_timesteps = 25
    worlds = _worlds(3 * num_cpu)
    vec_env = AuvVecEnv(cfg, worlds, num_cpu, sub_batches=sub_batches, track_trajectories=(0, 5))
    ora = Oracle(make_config(cfg, auto_reset=True), num_cpu, pack_bank(worlds))
    assert vec_env.num_envs == num_cpu and len(vec_env) == num_cpu
    assert vec_env.observation_space.shape == (6 + 180,) and vec_env.action_space.shape == (2,)
    assert vec_env.seed(7) == [7 + i for i in range(num_cpu)]
    obs = vec_env.reset()
    o_obs = ora.reset()
    assert isinstance(obs, np.ndarray) and obs.shape == (num_cpu, 186) and obs.dtype == np.float32
    np.testing.assert_allclose(obs, o_obs, rtol=0, atol=1e-6)
    assert vec_env.get_attr("last_episode")[0] is None       # environment.py:95: nothing has ended yet
    rs = np.random.RandomState(4)
    ends = [[] for _ in range(num_cpu)]                      # what the oracle sees end: (return, length, collision, reached)
    cte = np.zeros(num_cpu)
    steps = np.zeros(num_cpu, dtype=int)
    n_steps = 120
    for t in range(n_steps):
        actions = rs.uniform([-1, -0.15], [1, 0.15], (num_cpu, 2)).astype(np.float32)
        actions[:, 0] = np.abs(actions[:, 0])
        vec_env.step_async(actions)
        obs, rewards, dones, infos = vec_env.step_wait()     # (stable-baselines' runner: env.step = step_async + step_wait)
        o_obs, o_rew, o_done = ora.step(actions.astype(np.float64))
        assert obs.dtype == np.float32 and rewards.shape == (num_cpu,) and dones.dtype == bool and len(infos) == num_cpu
        assert set(infos[0]) == {"collision", "reached_goal", "goal_distance", "progress"}
        np.testing.assert_array_equal(dones, o_done.astype(bool))
        np.testing.assert_allclose(obs, o_obs, rtol=0, atol=1e-6)
        # the oracle's view of the episode records (save_latest_episode), accumulated on the host
        nav, ep = ora.read("NAV64"), ora.read("EPISODE")
        step_info = ora.read("STEP_INFO")
        steps += 1
        for e in range(num_cpu):
            if o_done[e]:
                ends[e].append((ep[e, 0], int(ep[e, 1]), int(ep[e, 2]), int(ep[e, 3]), step_info[e, 3]))
                steps[e] = 0
        assert all(bool(infos[e]["collision"]) == bool(step_info[e, 0]) for e in range(num_cpu))
    # ---- the training callback of scripts/run.py:415-426, verbatim in structure ----
    class Struct(object):
        pass
    report_env = Struct()
    report_env.history = []
    report_env.last_episode = vec_env.get_attr('last_episode')[0]
    report_env.config = vec_env.get_attr('config')[0]
    report_env.obstacles = vec_env.get_attr('obstacles')[0]
    env_histories = vec_env.get_attr('history')
    for episode in range(max(map(len, env_histories))):
        for env_idx in range(len(env_histories)):
            if (episode < len(env_histories[env_idx])):
                report_env.history.append(env_histories[env_idx][episode])
    report_env.episode = len(report_env.history) + 1
    total_t_steps = vec_env.get_attr('total_t_steps')[0] * num_cpu
    # ---- and what it must have got ----
    assert report_env.config is cfg and len(report_env.obstacles) > 0
    assert len(env_histories) == num_cpu and sum(map(len, env_histories)) == sum(map(len, ends)) >= 3 * num_cpu
    for e in range(num_cpu):
        assert len(env_histories[e]) == len(ends[e])
        for h, (ret, length, col, reached, progress) in zip(env_histories[e], ends[e]):
            assert set(h) == HISTORY_KEYS
            assert h["timesteps"] == length and h["collision"] == col and h["reached_goal"] == reached
            assert abs(h["reward"] - ret) <= 1e-9 * max(1.0, abs(ret)) and abs(h["progress"] - progress) < 1e-9
            assert h["duration"] == length * cfg.simulation.t_step_size and h["pathlength"] > 0 and h["cross_track_error"] >= 0.0
    assert total_t_steps == sum(h["timesteps"] for h in env_histories[0]) * num_cpu
    assert vec_env.get_attr('total_t_steps', indices=[3, 4]) == [sum(h["timesteps"] for h in env_histories[i]) for i in (3, 4)]
    le = report_env.last_episode
    assert set(le) == {"path", "path_taken", "obstacles"} and le["path"].shape == (2, 1000)
    assert le["path_taken"].shape == (env_histories[0][-1]["timesteps"], 3)          # tracked environment 0
    assert vec_env.get_attr('last_episode', indices=1)[0]["path_taken"] is None       # environment 1 is not tracked
    # mean |cross-track error| of an episode = mean over its steps of |NAV64[5]| * 100 (environment.py:460-464)
    assert all(np.isfinite(h["cross_track_error"]) for hs in env_histories for h in hs)
    # set_attr / env_method round trips
    vec_env.set_attr("pilot", "look-ahead", indices=[0, 2])
    assert vec_env.get_attr("pilot", indices=[0, 1, 2]) == ["look-ahead", None, "look-ahead"]
    with pytest.raises(AttributeError):
        vec_env.set_attr("history", [])
    assert vec_env.env_method("seed", 11, indices=[0]) == [[11]]
    assert vec_env.get_attr("rewarder")[0].params["gamma_theta"] == 10.0
    assert vec_env.get_attr("t_step") == ora.read("COUNTERS")[:, 0].tolist()
    with pytest.raises(AttributeError):
        vec_env.get_attr("no_such_attribute")
    vec_env.close()


def test_vecenv_cross_track_error_mean_and_device_tensors():
    """numpy=False keeps everything on the device (no host synchronisation in the step); the episode records' mean
    |cross-track error| equals the mean of |cross_track_error| * 100 over the episode's steps as the reference keeps it
    (environment.py:460-489), here tracked on the host from NAV64 for every environment."""
    from gym_auv_amd.vec_env import AuvVecEnv
    n = 64
    cfg = effective_reference_config(use_lidar=True)
    cfg.episode.max_timesteps = 12
    vec_env = AuvVecEnv(cfg, _worlds(2 * n), n, numpy=False, track_trajectories=())
    obs = vec_env.reset()
    assert isinstance(obs, torch.Tensor) and obs.is_cuda
    acc, cnt, means = np.zeros(n), np.zeros(n), [[] for _ in range(n)]
    g = torch.Generator(device="cuda:0")
    g.manual_seed(1)
    for t in range(40):
        a = torch.rand((n, 2), generator=g, device="cuda:0") * torch.tensor([1.0, 0.3], device="cuda:0") - torch.tensor([0.0, 0.15], device="cuda:0")
        vec_env.step_async(a)
        obs, rew, done, info = vec_env.step_wait()
        assert obs.is_cuda and rew.is_cuda and done.is_cuda
        nav = vec_env.env.read("NAV64").cpu().numpy()
        d = done.cpu().numpy().astype(bool)
        # NAV64 of an environment that ended holds the reset row already; its terminal |cte| is in the running sum only
        acc[~d] += np.abs(nav[~d, 5]) * 100
        cnt += 1
        for e in np.nonzero(d)[0]:
            means[e].append((acc[e], cnt[e]))
            acc[e], cnt[e] = 0.0, 0
    hist = vec_env.get_attr("history")
    for e in range(n):
        assert len(hist[e]) == len(means[e]) >= 3
        for h, (s, c) in zip(hist[e], means[e]):
            assert h["timesteps"] == c
            # all but the terminal step's |cte| are in `s`: the record's sum lies between s and s + (a step's |cte| <= a few 100 m)
            assert h["cross_track_error"] * c >= s - 1e-6
    vec_env.close()


def test_vecenv_dict_observation_with_velocity_channels():
    """use_dict_observation + sensor_use_velocity_observations for the whole batch (environment.py:116-137, :281-288):
    {"proprioceptive": [N, 6], "lidar": [N, 3, S]} with the closeness row over two all-zero velocity rows (the
    reference's simulate_sensor always returns velocity (0, 0), sensor.py:159), equal to the flat layout's columns."""
    from gym_auv_amd.vec_env import AuvVecEnv
    n = 32
    cfg = effective_reference_config(use_lidar=True)
    cfg.vessel.use_dict_observation = True
    cfg.vessel.sensor_use_velocity_observations = True
    worlds = _worlds(n)
    vec_env = AuvVecEnv(cfg, worlds, n, track_trajectories=())
    flat_cfg = effective_reference_config(use_lidar=True)
    flat = AuvVecEnv(flat_cfg, worlds, n, track_trajectories=())
    S = cfg.vessel.n_sensors
    assert set(vec_env.observation_space.spaces) == {"proprioceptive", "lidar"} and vec_env.observation_space["lidar"].shape == (3, S)
    o, f = vec_env.reset(), flat.reset()
    rs = np.random.RandomState(0)
    for _ in range(10):
        a = rs.uniform([0, -0.15], [1, 0.15], (n, 2)).astype(np.float32)
        o, _, _, _ = vec_env.step(a)
        f, _, _, _ = flat.step(a)
        assert o["proprioceptive"].shape == (n, 6) and o["lidar"].shape == (n, 3, S) and o["lidar"].dtype == np.float32
        np.testing.assert_array_equal(o["proprioceptive"], f[:, :6])
        np.testing.assert_array_equal(o["lidar"][:, 0, :], f[:, 6:])
        assert (o["lidar"][:, 1:, :] == 0.0).all()
    vec_env.close(), flat.close()
