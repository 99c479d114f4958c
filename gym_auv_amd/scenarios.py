"""Scenario generators (reset-time, host): produce WorldSpecs.

`moving_obstacles_world` follows the draw order of the reference's
MovingObstacles._generate (/root/reference/gym_auv/envs/movingobstacles.py:28-95) and
helpers.generate_obstacle (utils/helpers.py:5-35) exactly, including which draws come from
the env-local RandomState (`rng`) and which from NumPy's *global* generator (`grng` here:
an explicit RandomState seeded like `np.random.seed(seed)`), so that
    env.seed(k); np.random.seed(k); env.reset()
in the reference and `moving_obstacles_world(k)` here describe the same world
(tests/test_world.py checks this against golden worlds captured from the reference).

The remaining generators build the BASELINE.json workload shapes the reference has no
generator for: N static circles, N filled star-convex polygons, and a mix.
"""
import math
from typing import Optional

import numpy as np

from .path import Path, RandomCurveThroughOrigin
from .seeding import np_random
from .world import advance_mover
from .worldspec import MoverSpec, WorldSpec

VESSEL_WIDTH = 1.255


def _princip(a):
    return ((a + np.pi) % (2 * np.pi)) - np.pi


def _place_obstacle(rng, grng, path: Path, vessel_pose, vessel_width, displacement_dist_std=150.0,
                    obst_radius_mean=30.0):
    """helpers.generate_obstacle: rejection-sample a position beside the path that is clear of
    the vessel and of the goal."""
    vx, vy, vpsi = vessel_pose
    c, s = math.cos(-vpsi), math.sin(-vpsi)
    goal = path(path.length)
    while True:
        disp = grng.normal(0, displacement_dist_std)
        arclen = (0.1 + 0.8 * rng.rand()) * path.length
        pos = path(arclen)
        ang = _princip(path.get_direction(arclen) - np.pi / 2)
        pos = pos + disp * np.array([np.cos(ang), np.sin(ang)])
        radius = max(1, grng.poisson(obst_radius_mean))
        dx, dy = pos[0] - vx, pos[1] - vy
        # body-frame vector as in the reference (rotation by -psi); only its norm is used
        vessel_dist = np.linalg.norm(np.array([c * dx - s * dy, s * dx + c * dy, 0.0])) - vessel_width - radius
        goal_dist = np.linalg.norm(pos - goal) - radius
        if min(vessel_dist, goal_dist) > 0:
            return pos, radius


def _start_pose(rng, path: Path):
    """movingobstacles.py:34-43: start near path(0), heading uniformly random."""
    p = path(0)
    ang = path.get_direction(0)
    x = p[0] + 50 * (rng.rand() - 0.5)
    y = p[1] + 50 * (rng.rand() - 0.5)
    psi = _princip(ang + 2 * np.pi * (rng.rand() - 0.5))
    return np.array([x, y, psi])


def _linear_mover(pos, radius, direction, speed, dt, n_ticks=10000) -> MoverSpec:
    """VesselObstacle on a straight unit-time trajectory (movingobstacles.py:67-79), with
    the constructor's update(0.1) (obstacles.py:192-193) and the scenario's trailing
    _update() (movingobstacles.py:95) already applied to its reset-time state."""
    p0 = np.array([pos[0] + 0 * speed * np.cos(direction), pos[1] + 0 * speed * np.sin(direction)])
    p1 = np.array([pos[0] + 1 * speed * np.cos(direction), pos[1] + 1 * speed * np.sin(direction)])
    vel = (p1 - p0).reshape(1, 2)
    param = (float(radius), p0[0], p0[1], n_ticks - 1)
    st = (p0[0], p0[1], np.pi / 2, 0.0)
    st = advance_mover(param, vel, st, 0.1)
    st = advance_mover(param, vel, st, dt)
    return MoverSpec(width=float(radius), pos0=p0, vel=vel, n_vel=n_ticks - 1,
                     pos=np.array(st[:2]), heading=float(st[2]), counter=float(st[3]))


def moving_obstacles_world(seed: int, n_moving: int = 17, n_static: int = 11, dt: float = 0.5,
                           vessel_width: float = VESSEL_WIDTH, grng: Optional[np.random.RandomState] = None,
                           rng=None) -> WorldSpec:
    """MovingObstaclesNoRules-v0 world (17 movers + 11 circles by default);
    n_moving = n_static = 0 gives PathFollowNoObstacles-v0."""
    if rng is None:
        rng, _ = np_random(seed)
    if grng is None:
        grng = np.random.RandomState(seed)
    nwaypoints = int(np.floor(4 * rng.rand() + 2))
    path = RandomCurveThroughOrigin(rng, nwaypoints, length=800)
    pose = _start_pose(rng, path)
    movers = []
    for _ in range(n_moving):
        pos, radius = _place_obstacle(rng, grng, path, pose, vessel_width, displacement_dist_std=500,
                                      obst_radius_mean=10)
        direction = rng.rand() * 2 * np.pi
        speed = grng.uniform(low=1, high=3)
        movers.append(_linear_mover(pos, radius, direction, speed, dt))
    circles = []
    for _ in range(n_static):
        pos, radius = _place_obstacle(rng, grng, path, pose, vessel_width, displacement_dist_std=250)
        circles.append([pos[0], pos[1], radius])
    return WorldSpec(waypoints=path.init_waypoints, vessel_init=pose,
                     circles=np.asarray(circles, dtype=np.float64).reshape(-1, 3),
                     movers=movers, name="moving_obstacles_seed%d" % seed)


def static_circles_world(seed: int, n_circles: int = 20) -> WorldSpec:
    """BASELINE config 1: n static CircularObstacles placed like MovingObstacles' static ones."""
    return moving_obstacles_world(seed, n_moving=0, n_static=n_circles)


def _star_polygon(grng, centre, circumradius, k):
    ang = np.sort(grng.uniform(0, 2 * np.pi, k))
    rr = circumradius * grng.uniform(0.45, 1.0, k)
    return np.stack([centre[0] + rr * np.cos(ang), centre[1] + rr * np.sin(ang)], axis=1)


def polygon_world(seed: int, n_polygons: int = 50, n_circles: int = 0, n_moving: int = 0,
                  dt: float = 0.5, vessel_width: float = VESSEL_WIDTH) -> WorldSpec:
    """BASELINE configs 2 and 4: filled star-convex PolygonObstacles (K~U{6..16} vertices,
    circumradius max(3, Poisson(30))) with centres placed by the reference's obstacle
    placement rule; optionally mixed with circles and movers."""
    rng, _ = np_random(seed)
    grng = np.random.RandomState(seed)
    base = moving_obstacles_world(seed, n_moving=n_moving, n_static=n_circles, dt=dt,
                                  vessel_width=vessel_width, rng=rng, grng=grng)
    path = Path(base.waypoints)
    polys = []
    for _ in range(n_polygons):
        pos, _r = _place_obstacle(rng, grng, path, base.vessel_init, vessel_width, displacement_dist_std=250)
        rad = max(3, grng.poisson(30))
        # keep the start pose and the goal outside the polygon's circumcircle
        goal = path(path.length)
        while (np.linalg.norm(pos - base.vessel_init[:2]) - vessel_width - rad <= 0
               or np.linalg.norm(pos - goal) - rad <= 0):
            rad = max(3, rad // 2)
        polys.append(_star_polygon(grng, pos, rad, int(grng.randint(6, 17))))
    base.polygons = polys
    base.name = "polygon_world_seed%d" % seed
    return base
