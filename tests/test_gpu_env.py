"""-m gpu: the single-environment gym.Env-style adapter, mirroring the reference's own
tests/test_end_to_end.py:20-58 (every registered scenario id: reset, one step([0.5, 0.6]),
observation inside the space, reward/done/info types, observation changed) and
tests/test_config.py (observation length follows the LiDAR setting)."""
import numpy as np
import pytest

from gym_auv_amd.config import Config, effective_reference_config

pytestmark = pytest.mark.gpu


def _ids():
    from gym_auv_amd.env import SCENARIOS
    return list(SCENARIOS.keys())


@pytest.mark.parametrize("scenario_name", ["MovingObstaclesNoRules-v0", "PathFollowNoObstacles-v0", "TestScenario1-v0",
                                           "TestScenario2-v0", "TestScenario3-v0", "TestScenario4-v0", "TestHeadOn-v0",
                                           "TestCrossing-v0", "TestCrossing1-v0", "DebugScenario-v0", "EmptyScenario-v0"])
def test_single_step(scenario_name):
    from gym_auv_amd.env import make
    assert scenario_name in _ids()
    env = make(scenario_name)
    first_obs = env.reset()
    obs, reward, done, info = env.step(np.array([0.5, 0.6]))
    space = env.observation_space
    assert isinstance(obs, np.ndarray) and obs.shape == space.shape == (6,)      # LiDAR is off by default
    assert np.all(space.low <= obs) and np.all(space.high >= obs)
    assert isinstance(reward, float) and isinstance(done, bool) and isinstance(info, dict)
    assert set(info) == {"collision", "reached_goal", "goal_distance", "progress"}
    assert np.any(first_obs != obs)
    env.close()


def test_lidar_observation_length_and_episode_bookkeeping():
    from gym_auv_amd.env import make
    cfg = effective_reference_config(use_lidar=True)
    cfg.episode.max_timesteps = 6
    env = make("MovingObstaclesNoRules-v0", env_config=cfg)
    env.seed(3)
    obs = env.reset()
    assert obs.shape == (6 + cfg.vessel.n_lidar_observations,) == (186,)
    done, steps = False, 0
    while not done:
        obs, reward, done, info = env.step(env.action_space.sample())
        steps += 1
    assert steps == 6                       # done once t_step (before its increment) reaches max_timesteps - 1
    env.reset()
    assert len(env.history) == 1 and env.history[0]["timesteps"] == 6 and env.episode == 3
    assert env.total_t_steps == 6
    # accepts the RLlib-style {"config": Config} wrapper too (environment.py:66-74)
    from gym_auv_amd.env import AuvEnv
    e2 = AuvEnv({"config": Config()})
    assert e2.config.simulation.t_step_size == 1.0
    e2.close(), env.close()


def test_seed_reproduces_world_and_rollout():
    from gym_auv_amd.env import make
    outs = []
    for _ in range(2):
        env = make("MovingObstaclesNoRules-v0", env_config=effective_reference_config(use_lidar=True))
        env.seed(11)
        o0 = env.reset()
        o1, r1, _, _ = env.step(np.array([0.9, 0.1]))
        outs.append((o0, o1, r1))
        env.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2]


def test_velocity_channels_and_dict_observation():
    """sensor_use_velocity_observations appends 2*S zeros (the reference's velocity channel is
    hard-wired to (0, 0), sensor.py:159); use_dict_observation splits the same data
    (environment.py:116-137, :281-288).  Mirrors the intent of the reference's tests/test_config.py."""
    import torch
    from gym_auv_amd.batched_env import BatchedAuvEnv
    from gym_auv_amd.env import AuvEnv
    from gym_auv_amd.scenarios import moving_obstacles_world
    from gym_auv_amd.world import build_world, pack_bank
    cfg = effective_reference_config(use_lidar=True)
    cfg.vessel.sensor_use_velocity_observations = True
    S = cfg.vessel.n_sensors
    assert cfg.vessel.n_lidar_observations == 3 * S
    bank = pack_bank([build_world(moving_obstacles_world(3))])
    env = BatchedAuvEnv(cfg, bank, 5, device="cuda:0")
    plain_cfg = effective_reference_config(use_lidar=True)
    plain = BatchedAuvEnv(plain_cfg, bank, 5, device="cuda:0")
    o, p = env.reset(), plain.reset()
    assert o.shape == (5, 6 + 3 * S) and env.observation_space.shape == (6 + 3 * S,)
    a = torch.rand((5, 2), device="cuda:0")
    for _ in range(4):
        o, r, d, _ = env.step(a)
        p, r2, d2, _ = plain.step(a)
        assert torch.equal(o[:, :6 + S], p) and torch.equal(r, r2)
        assert (o[:, 6 + S:] == 0).all()
    cfg.vessel.use_dict_observation = True
    e = AuvEnv(cfg)
    ob = e.reset()
    assert set(ob) == {"proprioceptive", "lidar"} and ob["proprioceptive"].shape == (6,) and ob["lidar"].shape == (3, S)
    ob, rew, done, info = e.step([0.5, 0.6])
    assert (ob["lidar"][1:] == 0).all() and e.observation_space.contains(ob)
    e.close()


def test_adapter_attributes_and_persistent_handle():
    """What the reference's callers read off the environment (scripts/run.py:415-426, environment.py:444-489):
    vessel / path / obstacles / rewarder.params / last_episode / history -- and reset() keeps ONE library handle
    (it used to tear the handle down and rebuild it on every reset)."""
    from gym_auv_amd.env import make
    cfg = effective_reference_config(use_lidar=True)
    cfg.episode.max_timesteps = 5
    env = make("MovingObstaclesNoRules-v0", env_config=cfg)
    env.seed(5)
    env.reset()
    handle = env._env._h.value
    assert env.last_episode is None or isinstance(env.last_episode, dict)
    assert env.rewarder.params["lambda"] == 0.5 and env.rewarder.params["collision"] == -10000.0
    assert env.path.length > 0 and env.path(np.array([0.0, env.path.length])).shape == (2, 2)
    obst = env.obstacles
    assert len(obst) == 28 and sum(1 for o in obst if not o.static) == 17
    p0 = env.vessel.position.copy()
    mv0 = [o.position.copy() for o in obst if not o.static]
    done = False
    while not done:
        _, _, done, _ = env.step([1.0, 0.05])
    assert env.t_step == 5 and env.vessel.path_taken.shape == (6, 2)
    assert np.linalg.norm(env.vessel.position - p0) > 0 and env.vessel.speed > 0
    np.testing.assert_array_equal(env.vessel.path_taken[0], p0)
    mv1 = [o.position for o in env.obstacles if not o.static]
    assert all(np.linalg.norm(a - b) > 0 for a, b in zip(mv0, mv1))          # movers moved
    env.reset()
    assert env._env._h.value == handle                                         # same handle, new world
    le = env.last_episode
    assert le["path"].shape in ((1000, 2), (2, 1000)) and le["path_taken"].shape == (6, 2) and len(le["obstacles"]) == 28
    assert env.history[-1]["timesteps"] == 5 and env.history[-1]["pathlength"] > 0
    # a third episode on the same handle still steps correctly against a freshly built one
    obs_a = env.reset()
    fresh = make("MovingObstaclesNoRules-v0", env_config=cfg)
    fresh.seed(5)
    fresh.reset(), fresh.reset(save_history=False)
    obs_b = fresh.reset(save_history=False)
    np.testing.assert_array_equal(obs_a, obs_b)
    a = env.step([0.7, -0.1])
    b = fresh.step([0.7, -0.1])
    np.testing.assert_array_equal(a[0], b[0])
    assert a[1] == b[1]
    env.close(), fresh.close()
