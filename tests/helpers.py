"""Shared test helpers: golden loading, config reconstruction, world construction."""
import os

import numpy as np

from gym_auv_amd.config import Config
from gym_auv_amd.worldspec import MoverSpec, WorldSpec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def cfg_from_scalars(keys, vals) -> Config:
    cs = dict(zip([str(k) for k in keys], vals))
    cfg = Config()
    cfg.simulation.t_step_size = float(cs["dt"])
    cfg.episode.min_goal_distance = float(cs["min_goal_distance"])
    cfg.episode.max_timesteps = int(cs["max_timesteps"])
    cfg.episode.min_cumulative_reward = float(cs["min_cumulative_reward"])
    cfg.episode.min_path_progress = float(cs["min_path_progress"])
    cfg.vessel.use_lidar = bool(cs["use_lidar"])
    cfg.vessel.n_sectors = int(cs["n_sectors"])
    cfg.vessel.n_sensors_per_sector = int(cs["n_sensors_per_sector"])
    cfg.vessel.sensor_range = float(cs["sensor_range"])
    cfg.vessel.vessel_width = float(cs["vessel_width"])
    cfg.vessel.look_ahead_distance = int(cs["look_ahead_distance"])
    cfg.vessel.sensor_interval_load_obstacles = int(cs["sensor_interval_load_obstacles"])
    return cfg


def scene_world(z, i) -> WorldSpec:
    """G3 scene i -> WorldSpec (dummy straight path; movers frozen in their recorded pose).
    Obstacle order in the built world is circles, polygons, movers; `scene_order` maps the
    reference's obstacle list onto it."""
    pre = "s%d_" % i
    off = z[pre + "poly_off"]
    pts = z[pre + "poly_pts"]
    polys = [pts[off[j]:off[j + 1]] for j in range(len(off) - 1)]
    movers = [MoverSpec(width=float(m[0]), pos0=m[1:3].copy(), vel=np.array([[1.0, 0.0]]), n_vel=9999,
                        pos=m[1:3].copy(), heading=float(m[3]), counter=0.0) for m in z[pre + "movers_now"]]
    return WorldSpec(waypoints=np.array([[0.0, 1000.0], [0.0, 0.0]]), vessel_init=z[pre + "pose"].copy(),
                     circles=z[pre + "circles"], polygons=polys, movers=movers, name=str(z["names"][i]))


def scene_order(z, i):
    """index in the built world's obstacle list for each reference obstacle of scene i."""
    pre = "s%d_" % i
    order = z[pre + "order"]
    nc = len(z[pre + "circles"])
    npoly = len(z[pre + "poly_off"]) - 1
    base = {0: 0, 1: nc, 2: nc + npoly}
    return np.array([base[int(k)] + int(j) for k, j in order], dtype=np.int64)
