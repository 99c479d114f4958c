"""AuvVecEnv — the stable-baselines VecEnv protocol over one BatchedAuvEnv.

The reference trains through stable-baselines' SubprocVecEnv (/root/reference/scripts/run.py:293-296: eight worker
processes, one gym env each) and its training callback reaches into the workers with `vec_env.get_attr(...)`
(run.py:415-426: 'last_episode', 'config', 'obstacles', 'history', 'total_t_steps').  This class offers that protocol
-- `num_envs`, `observation_space` / `action_space`, `reset`, `step_async` / `step_wait` / `step`, `get_attr` /
`set_attr` / `env_method`, `seed`, `close` -- on top of N environments stepped by the HIP library, so that code written
against a VecEnv drives the batched path unchanged:

    vec_env = AuvVecEnv(config, worlds, n_envs=4096)            # instead of SubprocVecEnv([make_mp_env(...)] * 8)
    obs = vec_env.reset()
    vec_env.step_async(actions); obs, rewards, dones, infos = vec_env.step_wait()
    histories = vec_env.get_attr('history')                     # a list (episode dicts) per environment

`step_async` enqueues the step (BatchedAuvEnv.step_async: by default ONE launch on the caller's stream; with `sub_batches`
> 1 the other sub-batches on streams of their own, ordered against the caller's inside the library) and returns at once;
`step_wait` orders the caller's stream behind all of it.  Arrays cross the boundary as NumPy (`numpy=True`, the
stable-baselines convention: float32 observations / rewards, bool dones, a list of info dicts) or stay torch tensors
on the device (`numpy=False`: no host synchronisation in the step; `infos` is then the lazy batched info).

Per-environment attributes (reference: gym_auv/environment.py)
    history         list of the environment's finished episodes, the dicts of save_latest_episode (:466-489):
                    cross_track_error, reached_goal, collision, reward, timesteps, duration, progress, pathlength
                    -- from the library's episode log (auv_episode_log), so nothing is recomputed on the host
    last_episode    {"path": [2, 1000] samples of the path, "path_taken": [T, 3] poses, "obstacles": ...} of the
                    environment's last finished episode (:468-474); path_taken only for environments listed in
                    `track_trajectories` (their pose is copied to a device ring after every step), else None; the pose
                    after the terminal step is not in it (the auto-reset overwrites it inside that step)
    total_t_steps   steps of the environment's finished episodes (:204)
    config, obstacles, episode, t_step, cumulative_reward, collision, reached_goal, progress, path, rewarder
"""
from typing import Any, Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from .batched_env import BatchedAuvEnv
from .config import Config
from .env import _ObstacleView, _RewarderView
from .world import BuiltWorld, build_world, pack_bank
from .worldspec import WorldSpec

_HISTORY_KEYS = ("cross_track_error", "reached_goal", "collision", "reward", "timesteps", "duration", "progress", "pathlength")


class AuvVecEnv:
    def __init__(self, config: Config, worlds: Sequence[Union[WorldSpec, BuiltWorld]], n_envs: int,
                 device: Union[str, torch.device] = "cuda:0", rewarder: str = "colav", test_mode: bool = False,
                 sub_batches: int = 1, numpy: bool = True, track_trajectories: Sequence[int] = (0,),
                 max_trajectory: int = 20000):
        self._built: List[BuiltWorld] = [w if isinstance(w, BuiltWorld) else build_world(w) for w in worlds]
        self.env = BatchedAuvEnv(config, pack_bank(self._built), n_envs, device=device, rewarder=rewarder,
                                 test_mode=test_mode, auto_reset=True)
        self.num_envs = int(n_envs)
        self.config = config
        self.observation_space, self.action_space = self.env.observation_space, self.env.action_space
        v = config.vessel
        self._dict_obs = bool(v.use_dict_observation and v.use_lidar)
        if self._dict_obs:                                                    # environment.py:116-137
            from .spaces import Box, Dict
            self.observation_space = Dict({"proprioceptive": Box(low=-1.0, high=1.0, shape=(6,), dtype=np.float32),
                                           "lidar": Box(low=-1.0, high=1.0, shape=v.lidar_shape, dtype=np.float32)})
        self.numpy = bool(numpy)
        self._rewarder = _RewarderView(rewarder)
        # (the VecEnv protocol waits for ALL environments every step: one launch on the caller's stream is then the fastest
        # shape -- 118-121 M env-steps/s at 4096 x 180 against 108-110 / 89-105 M with two / four chains and a rendezvous per step;
        # chains pay where slices are consumed independently: BatchedAuvEnv.step_slice, examples/ppo.py)
        if sub_batches > 1:
            self.env.set_sub_batches(sub_batches, inline_first=True)
        self._history: List[List[dict]] = [[] for _ in range(self.num_envs)]
        self._total_t_steps = np.zeros(self.num_envs, dtype=np.int64)
        self._last_world = np.full(self.num_envs, -1, dtype=np.int64)       # world of each env's last finished episode
        self._last_len = np.zeros(self.num_envs, dtype=np.int64)
        self._track = [int(i) for i in track_trajectories if 0 <= int(i) < self.num_envs]
        self._track_idx = torch.as_tensor(self._track, dtype=torch.int64, device=self.env.device)
        self._traj_cap = int(max_trajectory)
        self._traj = torch.zeros((self._traj_cap, max(1, len(self._track)), 3), dtype=torch.float64, device=self.env.device)
        self._traj_n = 0                                                     # poses written so far (ring position)
        self._extra: Dict[str, list] = {}
        self._seeds: List[Optional[int]] = [None] * self.num_envs
        self._waiting = False
        self.reset()

    # ------------------------------------------------------------------------------ VecEnv protocol
    def _out(self, t: torch.Tensor, dtype=None):
        if not self.numpy:
            return t
        a = t.detach().cpu().numpy()
        return a.astype(dtype) if dtype is not None else a

    def _obs(self, obs: torch.Tensor):
        """The batched observation as the caller wants it: flat [N, 6 + channels * S], or -- use_dict_observation,
        environment.py:281-288 -- {"proprioceptive": [N, 6], "lidar": [N, channels, S]} (closeness row over the velocity
        rows, which the reference hard-wires to zero, sensor.py:159)."""
        if not self._dict_obs:
            return self._out(obs)
        c, S = self.config.vessel.lidar_shape
        return {"proprioceptive": self._out(obs[:, :6]), "lidar": self._out(obs[:, 6:6 + c * S].reshape(-1, c, S))}

    def _record_pose(self):
        if self._track:
            st = self.env.read("STATE")                                      # [6, N] on the device
            self._traj[self._traj_n % self._traj_cap] = st[0:3].index_select(1, self._track_idx).t()
            self._traj_n += 1

    def reset(self):
        """Every environment back to the first observation of the world it is bound to (VecEnv.reset)."""
        obs = self.env.reset()
        self._traj_n = 0
        self._record_pose()
        return self._obs(obs.clone() if not self.numpy else obs)

    def step_async(self, actions):
        a = torch.as_tensor(np.asarray(actions) if isinstance(actions, (list, tuple)) else actions)
        self.env.step_async(a.to(self.env.device))
        self._waiting = True

    def step_wait(self):
        if not self._waiting:
            raise RuntimeError("step_wait() without step_async()")
        obs, rew, done, info = self.env.step_wait()
        self._waiting = False
        self._record_pose()
        if not self.numpy:
            return self._obs(obs), rew, done, info
        step_info = self.env.read("STEP_INFO").cpu().numpy()                 # one read for all four keys
        infos = [dict(collision=bool(r[0]), reached_goal=bool(r[1]), goal_distance=float(r[2]), progress=float(r[3]))
                 for r in step_info]
        return self._obs(obs), self._out(rew), self._out(done).astype(bool), infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def seed(self, seed: Optional[int] = None):
        """VecEnv.seed: environment i gets seed + i.  The scenarios of this adapter are the worlds handed to it (the
        bank), so the seeds are only recorded (`get_attr('seed_value')`), as for a gym env whose scenario is fixed."""
        self._seeds = [None if seed is None else int(seed) + i for i in range(self.num_envs)]
        return list(self._seeds)

    def close(self):
        self.env.close()

    def render(self, *args, **kwargs):
        return None                                                          # rendering is out of scope

    def get_images(self):
        return [None] * self.num_envs

    # ------------------------------------------------------------------------------ attributes of the environments
    def _indices(self, indices) -> List[int]:
        if indices is None:
            return list(range(self.num_envs))
        if isinstance(indices, int):
            return [indices]
        return [int(i) for i in indices]

    def _pull_log(self):
        """Move what the library's episode log holds into the per-environment histories (one read)."""
        rows = self.env.episode_log()
        if rows.numel() == 0:
            return
        dt = self.config.simulation.t_step_size
        for r in rows.cpu().numpy():
            e, w = int(r[0]), int(r[7])
            self._history[e].append({"cross_track_error": float(r[6]), "reached_goal": int(r[4]), "collision": int(r[3]),
                                     "reward": float(r[1]), "timesteps": int(r[2]), "duration": int(r[2]) * dt,
                                     "progress": float(r[5]), "pathlength": float(self._built[w].path.length)})
            self._total_t_steps[e] += int(r[2])
            self._last_world[e], self._last_len[e] = w, int(r[2])

    def _obstacles(self, w: int, mover_state: Optional[np.ndarray] = None):
        spec = self._built[w].spec
        out = [_ObstacleView("circle", True, position=np.array(c[:2], dtype=np.float64), radius=float(c[2]))
               for c in np.asarray(spec.circles, dtype=np.float64).reshape(-1, 3)]
        out += [_ObstacleView("polygon", True, points=np.asarray(p, dtype=np.float64)) for p in spec.polygons]
        for k, m in enumerate(spec.movers):
            pos = None if mover_state is None else mover_state[k, 0:2].copy()
            out.append(_ObstacleView("vessel", False, width=float(m.width), position=pos,
                                     heading=None if mover_state is None else float(mover_state[k, 2])))
        return out

    def _last_episode(self, e: int, t_now: int):
        w = int(self._last_world[e])
        if w < 0:
            return None                                                      # no episode has ended yet (environment.py:95)
        path = self._built[w].path
        taken = None
        if e in self._track:
            L, col = int(self._last_len[e]), self._track.index(e)
            # the ring holds the pose after reset() and after every step; the pose after an episode's LAST step is the
            # next episode's reset pose already (the auto-reset happens inside the step), so a finished episode of L steps
            # left L poses -- its start and L - 1 steps -- just before the current episode's t_now + 1
            hi = self._traj_n - (t_now + 1)
            lo = hi - L
            if lo >= max(0, self._traj_n - self._traj_cap) and hi > lo:
                idx = torch.arange(lo, hi, device=self.env.device) % self._traj_cap
                taken = self._traj.index_select(0, idx)[:, col, :].cpu().numpy()
        return {"path": path(np.linspace(0, path.length, 1000)), "path_taken": taken, "obstacles": self._obstacles(w)}

    def get_attr(self, attr_name: str, indices=None) -> List[Any]:
        """VecEnv.get_attr: the attribute of each (selected) environment, as a list."""
        idx = self._indices(indices)
        name = attr_name
        if name in ("history", "total_t_steps", "last_episode"):
            self._pull_log()
        if name == "history":
            return [self._history[i] for i in idx]
        if name == "total_t_steps":
            return [int(self._total_t_steps[i]) for i in idx]
        if name == "config":
            return [self.config for _ in idx]
        if name == "rewarder":
            return [self._rewarder for _ in idx]
        if name == "seed_value":
            return [self._seeds[i] for i in idx]
        if name in self._extra:
            return [self._extra[name][i] for i in idx]
        cnt = self.env.read("COUNTERS").cpu().numpy()
        if name == "last_episode":
            return [self._last_episode(i, int(cnt[i, 0])) for i in idx]
        if name == "t_step":
            return [int(cnt[i, 0]) for i in idx]
        if name == "episode":
            return [int(cnt[i, 2]) + 1 for i in idx]                         # (the reference counts from 1, environment.py:202)
        world = self.env.read("WORLD_IDX").cpu().numpy()
        if name == "obstacles":
            ms = self.env.read("MOVER_STATE").cpu().numpy()
            return [self._obstacles(int(world[i]), ms[i]) for i in idx]
        if name == "path":
            return [self._built[int(world[i])].path for i in idx]
        info = self.env.read("INFO64").cpu().numpy()
        col = {"collision": 0, "reached_goal": 1, "goal_distance": 2, "progress": 3, "cumulative_reward": 4, "max_path_prog": 5}
        if name in col:
            cast = bool if name in ("collision", "reached_goal") else float
            return [cast(info[i, col[name]]) for i in idx]
        if name == "last_reward":
            r = self.env.read("REWARD64").cpu().numpy()
            return [float(r[i]) for i in idx]
        raise AttributeError("AuvVecEnv: environments have no attribute %r" % attr_name)

    def set_attr(self, attr_name: str, value, indices=None):
        """VecEnv.set_attr.  Attributes the device owns (state, counters, ...) are not settable this way -- use
        BatchedAuvEnv.write; anything else is kept per environment and handed back by get_attr."""
        if attr_name in ("history", "total_t_steps", "last_episode", "config", "obstacles", "path", "t_step", "episode"):
            raise AttributeError("AuvVecEnv: %r is read-only" % attr_name)
        store = self._extra.setdefault(attr_name, [None] * self.num_envs)
        for i in self._indices(indices):
            store[i] = value

    def env_method(self, method_name: str, *args, indices=None, **kwargs) -> List[Any]:
        """VecEnv.env_method for the methods of the reference environment that make sense per environment."""
        idx = self._indices(indices)
        if method_name == "seed":
            base = args[0] if args else kwargs.get("seed")
            for i in idx:
                self._seeds[i] = None if base is None else int(base)
            return [[self._seeds[i]] for i in idx]
        if method_name == "save_latest_episode":                             # done by the library at every episode end
            self._pull_log()
            return [None for _ in idx]
        if method_name == "reset":
            mask = torch.zeros(self.num_envs, dtype=torch.uint8, device=self.env.device)
            mask[torch.as_tensor(idx, device=self.env.device)] = 1
            obs = self.env.reset(mask=mask)
            return [self._out(obs[i]) for i in idx]
        if method_name == "observe":
            obs = self.env.obs
            return [self._out(obs[i]) for i in idx]
        raise AttributeError("AuvVecEnv: environments have no method %r" % method_name)

    # (stable-baselines 3 asks this of a VecEnv)
    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False for _ in self._indices(indices)]

    def __len__(self):
        return self.num_envs
