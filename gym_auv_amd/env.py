"""AuvEnv — single-environment adapter with the reference's `gym.Env` surface, on top of
`BatchedAuvEnv(n_envs=1)`.

Mirrors /root/reference/gym_auv/environment.py: ctor `Env(env_config, test_mode=False,
renderer=None, verbose=False)` accepting a `Config` or `{"config": Config}` (:29-36, :66-74),
`reset() -> obs` (:176), `step(action) -> (obs, reward: float, done: bool, info: dict)`
(:292-366; types asserted by the reference's tests/test_end_to_end.py:44-46), `seed(seed) ->
[seed]` (:439-442), `observation_space` / `action_space` (:101-106, :139-143), `close()`,
`render()` (no-op: rendering is out of scope), and the attributes callers read: `config`,
`episode`, `t_step`, `total_t_steps`, `cumulative_reward`, `history`, `last_episode`, `last_reward`,
`collision`, `reached_goal`, `progress`, `vessel`, `path`, `obstacles`, `rewarder.params`
(scripts/run.py:415-426, environment.py:444-489).

One library handle lives as long as the adapter: `reset()` builds the new scenario on the host, swaps it in
with `auv_load_worlds` and resets; `step()` gathers what it returns on the device and makes ONE blocking
device-to-host copy.

`make(id)` resolves the reference's registered scenario ids (gym_auv/__init__.py:43-121).
A fresh world is generated on every `reset()` from the env-local RNG stream, as the reference
does in `_generate()`; no auto-reset inside `step()` (callers reset on `done`).
"""
from typing import Callable, Dict, Optional, Union

import numpy as np
import torch

from . import scenarios
from .batched_env import BatchedAuvEnv
from .config import Config, effective_reference_config
from .seeding import np_random
from .world import build_world, pack_bank
from .worldspec import WorldSpec


class _VesselView:
    """What callers read off `env.vessel` (objects/vessel/vessel.py:100-187), served from the last packed
    read of the device state -- no extra transfer."""

    def __init__(self, env):
        self._env = env

    @property
    def config(self):
        return self._env.config

    @property
    def width(self) -> float:
        return self._env.config.vessel.vessel_width

    @property
    def n_sensors(self) -> int:
        return self._env.config.vessel.n_sensors

    @property
    def _state(self) -> np.ndarray:
        return self._env._state.copy()

    @property
    def position(self) -> np.ndarray:
        return self._env._state[0:2].copy()

    @property
    def heading(self) -> float:
        return float(self._env._state[2])

    @property
    def velocity(self) -> np.ndarray:
        return self._env._state[3:5].copy()

    @property
    def speed(self) -> float:
        return float(np.linalg.norm(self._env._state[3:5]))

    @property
    def yaw_rate(self) -> float:
        return float(self._env._state[5])

    @property
    def max_speed(self) -> float:
        return 2.0

    @property
    def course(self) -> float:
        u, v = self._env._state[3:5]
        return float(self._env._state[2] + np.arctan2(v, u))

    @property
    def path_taken(self) -> np.ndarray:
        return np.asarray(self._env._trajectory, dtype=np.float64).reshape(-1, 3)[:, 0:2]

    @property
    def heading_taken(self) -> np.ndarray:
        return np.asarray(self._env._trajectory, dtype=np.float64).reshape(-1, 3)[:, 2]

    @property
    def progress(self) -> float:
        return self._env.progress

    @property
    def max_progress(self) -> float:
        return self._env._max_progress


class _ObstacleView:
    """One entry of `env.obstacles`: kind, the boundary polygon / circle as built for the device, and -- for
    moving obstacles -- the position / heading of the moment it was asked for (objects/obstacles.py)."""

    def __init__(self, kind, static, width=None, position=None, heading=None, radius=None, points=None):
        self.kind, self.static = kind, static
        self.width, self.position, self.heading, self.radius, self.points = width, position, heading, radius, points

    def __repr__(self):
        return "<%s obstacle at %s>" % (self.kind, None if self.position is None else np.round(self.position, 2))


class _RewarderView:
    """`env.rewarder.params`: the dicts of the reference's rewarders, key for key (objects/rewarder.py:56-70 PathFollowRewarder,
    :143-159 ColavRewarder).  The reward kernels apply these constants (csrc/k3_nav_reward.hip: reward_apply,
    reward_path_term; csrc/k2_lidar.hip: k2_back) -- and, like the reference's calculate(), a few literals that are
    not in the dict (PathFollow: slow-speed threshold 0.1 and penalty -2, max_speed 2, rewarder.py:78-140)."""

    def __init__(self, kind: str):
        self.params = {"gamma_theta": 10.0, "gamma_x": 0.1, "gamma_v_y": 1.0, "gamma_y_e": 5.0, "penalty_yawrate": 10.0,
                       "penalty_torque_change": 0.0}
        if kind == "colav":
            self.params.update({"penalty_slow": -2, "cruise_speed": 0.1, "slow_speed": 0.04})
        else:
            self.params.update({"cruise_speed": 0.1})
        self.params.update({"neutral_speed": 0.05, "negative_multiplier": 2.0, "collision": -10000.0, "lambda": 0.5, "eta": 0})


class AuvEnv:
    metadata = {"render.modes": ["human", "rgb_array", "state_pixels"]}

    def __init__(self, env_config: Union[Config, dict], test_mode: bool = False, renderer: Optional[str] = None,
                 verbose: bool = False, world_fn: Optional[Callable[[int], WorldSpec]] = None,
                 rewarder: str = "colav", device: str = "cuda:0"):
        if isinstance(env_config, dict):
            env_config = env_config["config"]
        assert isinstance(env_config, Config), "Expected gym_auv_amd.Config, got %s" % type(env_config)
        self.config = env_config
        self.test_mode = test_mode
        self.renderer = None            # rendering is out of scope
        self.verbose = verbose
        self._world_fn = world_fn or (lambda seed: scenarios.moving_obstacles_world(
            seed, dt=env_config.simulation.t_step_size, vessel_width=env_config.vessel.vessel_width))
        self._rewarder = rewarder
        self._device = device
        self._env: Optional[BatchedAuvEnv] = None
        self._pack = None
        self.episode = 0
        self.total_t_steps = 0
        self.t_step = 0
        self.cumulative_reward = 0.0
        self.history = []
        self.last_reward = 0.0
        self.last_episode = None
        self.collision = self.reached_goal = False
        self.progress = 0.0
        self._max_progress = 0.0
        self.goal_distance = None
        self.world: Optional[WorldSpec] = None
        self.path = None
        self._state = np.zeros(6)
        self._trajectory = []
        self._cte = []
        self.rng = None
        self.vessel = _VesselView(self)
        self.rewarder = _RewarderView(rewarder)
        self.seed()
        v = env_config.vessel
        n_obs = 6 + (v.n_lidar_observations if v.use_lidar else 0)         # environment.py:112-114
        from .spaces import Box, Dict
        self._action_space = Box(low=np.array([-1, -0.15]), high=np.array([1, 0.15]), dtype=np.float32)
        if v.use_dict_observation:                                          # environment.py:116-137
            self._observation_space = Dict({
                "proprioceptive": Box(low=-1.0, high=1.0, shape=(6,), dtype=np.float32),
                "lidar": Box(low=-1.0, high=1.0, shape=v.lidar_shape, dtype=np.float32)})
        else:
            self._observation_space = Box(low=np.array([-1] * n_obs), high=np.array([1] * n_obs), dtype=np.float32)
        self.reset()

    @property
    def action_space(self):
        return self._action_space

    @property
    def observation_space(self):
        return self._observation_space

    def seed(self, seed=None):
        self.rng, seed = np_random(seed)
        return [seed]

    @property
    def obstacles(self):
        """The scenario's obstacles (environment.py:86-89): static ones from the world spec, moving ones with the
        pose the device holds right now (one read of MOVER_STATE)."""
        if self.world is None:
            return []
        out = [_ObstacleView("circle", True, position=np.array(c[:2], dtype=np.float64), radius=float(c[2]))
               for c in np.asarray(self.world.circles, dtype=np.float64).reshape(-1, 3)]
        out += [_ObstacleView("polygon", True, points=np.asarray(p, dtype=np.float64)) for p in self.world.polygons]
        if self.world.movers:
            ms = self._env.read("MOVER_STATE")[0].cpu().numpy()
            out += [_ObstacleView("vessel", False, width=float(m.width), position=ms[k, 0:2].copy(), heading=float(ms[k, 2]))
                    for k, m in enumerate(self.world.movers)]
        return out

    def save_latest_episode(self, save_history: bool = True):
        """environment.py:466-489."""
        self.last_episode = {
            "path": self.path(np.linspace(0, self.path.length, 1000)) if self.path is not None else None,
            "path_taken": self.vessel.path_taken,
            "obstacles": self.obstacles,
        }
        if save_history:
            self.history.append({
                "cross_track_error": float(np.mean(self._cte)) if self._cte else 0.0,
                "reached_goal": int(self.reached_goal), "collision": int(self.collision),
                "reward": self.cumulative_reward, "timesteps": self.t_step,
                "duration": self.t_step * self.config.simulation.t_step_size, "progress": self.progress,
                "pathlength": float(self.path.length),
            })

    def reset(self, save_history: bool = True) -> np.ndarray:
        if self.t_step:                                                     # environment.py:199-200
            self.save_latest_episode(save_history=save_history)
        self.episode += 1
        self.total_t_steps += self.t_step
        self.cumulative_reward, self.t_step, self.last_reward = 0.0, 0, 0.0
        self.reached_goal = self.collision = False
        self.progress = self._max_progress = 0.0
        self._cte = []
        # new scenario from the env-local stream (the reference's _generate())
        world_seed = int(self.rng.randint(0, 2 ** 31 - 1))
        self.world = self._world_fn(world_seed)
        built = build_world(self.world)
        self.path = built.path
        bank = pack_bank([built])
        if self._env is None:
            # ONE handle for the life of the adapter; later resets only replace its one-world bank
            self._env = BatchedAuvEnv(self.config, bank, 1, device=self._device, rewarder=self._rewarder,
                                      test_mode=self.test_mode, auto_reset=False)
        else:
            self._env.load_worlds(bank)
        self._env.reset()
        S = self.config.vessel.n_sensors
        self._n_obs64 = 6 + S
        # packed read-back of a step: OBS64 row | REWARD64 | INFO64 | NAV64 | STATE | done
        self._pack = torch.zeros(self._n_obs64 + 1 + 8 + 8 + 6 + 1, dtype=torch.float64, device=self._env.device)
        self._read_pack()
        self._trajectory = [self._state[0:3].copy()]
        return self._obs()

    def _read_pack(self):
        """ONE blocking device-to-host copy per step: the fields are gathered on the device first."""
        e, p, n = self._env, self._pack, self._n_obs64
        e.read_into("OBS64", p[0:n])
        e.read_into("REWARD64", p[n:n + 1])
        e.read_into("INFO64", p[n + 1:n + 9])
        e.read_into("NAV64", p[n + 9:n + 17])
        e.read_into("STATE", p[n + 17:n + 23])
        p[n + 23:n + 24].copy_(e.done)
        self._host = p.cpu().numpy()
        self._state = self._host[n + 17:n + 23]

    def _obs(self):
        # the reference returns float64 although the space says float32 (environment.py:276-280)
        v = self.config.vessel
        S = v.n_sensors
        row = self._host[:self._n_obs64]
        flat = np.concatenate([row[:6 + (S if v.use_lidar else 0)],
                               np.zeros(2 * S if (v.use_lidar and v.sensor_use_velocity_observations) else 0)])
        if not v.use_dict_observation:
            return flat
        # environment.py:281-288: closeness row stacked over the (zero) velocity rows
        return {"proprioceptive": flat[:6], "lidar": flat[6:].reshape(v.lidar_shape)}

    def step(self, action):
        a = torch.as_tensor(np.asarray(action, dtype=np.float64).reshape(1, 2), device=self._env.device)
        self._env.step(a)
        self._read_pack()
        n = self._n_obs64
        reward = float(self._host[n])
        info64, nav64 = self._host[n + 1:n + 9], self._host[n + 9:n + 17]
        self.collision, self.reached_goal = bool(info64[0]), bool(info64[1])
        self.goal_distance, self.progress = float(info64[2]), float(info64[3])
        self.cumulative_reward, self._max_progress = float(info64[4]), float(info64[5])
        self.last_reward = reward
        self._cte.append(abs(float(nav64[5])) * 100)                        # environment.py:460-464
        self._trajectory.append(self._state[0:3].copy())
        self.t_step += 1
        info = {"collision": self.collision, "reached_goal": self.reached_goal,
                "goal_distance": self.goal_distance, "progress": self.progress}
        return self._obs(), reward, bool(self._host[n + 23] != 0.0), info

    def render(self, mode="rgb_array", **kwargs):
        return None

    def close(self):
        if self._env is not None:
            self._env.close()
            self._env = None


def _scenario_table() -> Dict[str, dict]:
    eff = effective_reference_config
    return {
        # id: world generator(seed) , rewarder          (gym_auv/__init__.py:43-121)
        "MovingObstaclesNoRules-v0": dict(world=lambda s: scenarios.moving_obstacles_world(s), rewarder="colav", config=eff),
        "PathFollowNoObstacles-v0": dict(world=lambda s: scenarios.moving_obstacles_world(s, 0, 0), rewarder="pathfollow", config=eff),
        "TestScenario1-v0": dict(world=lambda s: scenarios.test_scenario1(), rewarder="colav", config=eff),
        "TestScenario2-v0": dict(world=lambda s: scenarios.test_scenario2(), rewarder="colav", config=eff),
        "TestScenario3-v0": dict(world=lambda s: scenarios.test_scenario3(), rewarder="colav", config=eff),
        "TestScenario4-v0": dict(world=lambda s: scenarios.test_scenario4(), rewarder="colav", config=eff),
        "TestHeadOn-v0": dict(world=lambda s: scenarios.test_head_on(s), rewarder="colav", config=eff),
        "TestCrossing-v0": dict(world=lambda s: scenarios.test_crossing(), rewarder="colav", config=eff),
        "TestCrossing1-v0": dict(world=lambda s: scenarios.test_crossing1(), rewarder="colav", config=eff),
        "DebugScenario-v0": dict(world=lambda s: scenarios.debug_scenario(s), rewarder="colav", config=eff),
        "EmptyScenario-v0": dict(world=lambda s: scenarios.empty_scenario(), rewarder="colav", config=eff),
    }


SCENARIOS = _scenario_table()


def make(env_id: str, env_config: Optional[Config] = None, **kwargs) -> AuvEnv:
    """`gym.make(id)` for the reference's registered ids (same defaults: LiDAR off unless the
    config says otherwise, effective dt 0.5 s / min_goal_distance 0.1 m)."""
    sc = SCENARIOS[env_id]
    cfg = env_config if env_config is not None else sc["config"]()
    return AuvEnv(cfg, world_fn=sc["world"], rewarder=sc["rewarder"], **kwargs)
