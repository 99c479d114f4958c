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


# ------------------------------------------------------------------------------------------
# The reference's fixed test scenarios (/root/reference/gym_auv/envs/testscenario.py:20-360)
def _start_on_path(path: Path) -> np.ndarray:
    p = path(0)
    return np.array([p[0], p[1], path.get_direction(0)])


def _mover_from_table(width, waypoints: np.ndarray, dt: float, trailing_update: bool) -> MoverSpec:
    """VesselObstacle from unit-time waypoints [n, 2]: velocities = successive differences
    (obstacles.py:160-173); reset-time state = constructor's update(0.1) (+ the scenario's
    trailing _update() where it has one)."""
    vel = np.diff(waypoints, axis=0)
    n_vel = len(vel)
    if np.abs(vel - vel[0]).max() < 1e-9:
        vel = vel[:1]
    param = (float(width), waypoints[0, 0], waypoints[0, 1], n_vel)
    st = (waypoints[0, 0], waypoints[0, 1], np.pi / 2, 0.0)
    st = advance_mover(param, vel, st, 0.1)
    if trailing_update:
        st = advance_mover(param, vel, st, dt)
    return MoverSpec(width=float(width), pos0=waypoints[0].copy(), vel=vel, n_vel=n_vel,
                     pos=np.array(st[:2]), heading=float(st[2]), counter=float(st[3]))


def test_scenario1() -> WorldSpec:
    wp = np.array([[0.0, 1100.0], [0.0, 1100.0]])
    path = Path(wp)
    circles, arclen = [], 30.0
    for o in range(20):
        r = 10 + 10 * o ** 1.5
        arclen += r * 2 + 30
        p = path(arclen)
        circles.append([p[0], p[1], r])
    return WorldSpec(waypoints=wp, vessel_init=_start_on_path(path), circles=np.array(circles), name="TestScenario1")


def test_scenario2() -> WorldSpec:
    wp = np.vstack([[t * np.cos(t / 100), 2 * t] for t in range(500)]).T
    path = Path(wp)
    circles, arclen, r = [], 30.0, 5
    while True:
        arclen += 2 * r
        if arclen >= path.length:
            break
        disp = 140 - 120 / (1 + np.exp(-0.005 * arclen))
        p = path(arclen)
        ang = path.get_direction(arclen) - np.pi / 2
        off = disp * np.array([np.cos(ang), np.sin(ang)])
        circles.append([p[0] + off[0], p[1] + off[1], r])
        circles.append([p[0] - off[0], p[1] - off[1], r])
    return WorldSpec(waypoints=wp, vessel_init=_start_on_path(path), circles=np.array(circles), name="TestScenario2")


def _ring_of_circles(keep) -> WorldSpec:
    wp = np.vstack([[0, 0], [0, 500]]).T.astype(np.float64)
    path = Path(wp)
    circles = []
    for n in range(21):
        ang = keep(n)
        if ang is None:
            continue
        circles.append([np.cos(ang) * 100, np.sin(ang) * 100, 25])
    return WorldSpec(waypoints=wp, vessel_init=_start_on_path(path), circles=np.array(circles, dtype=np.float64))


def test_scenario3() -> WorldSpec:
    w = _ring_of_circles(lambda n: np.pi / 4 + n / 20 * np.pi / 2)
    w.name = "TestScenario3"
    return w


def test_scenario4() -> WorldSpec:
    # testscenario.py:125 compares a boolean (`abs(angle < 3/2*pi) < pi/12`): it skips exactly
    # the obstacles with angle >= 3/2 pi -- reproduced as is
    def keep(n):
        ang = n / 20 * 2 * np.pi
        return None if abs(ang < 3 / 2 * np.pi) < np.pi / 12 else ang
    w = _ring_of_circles(keep)
    w.name = "TestScenario4"
    return w


def _single_mover_world(path_wp, start_angle, radius, direction_vec, speed, dt, name) -> WorldSpec:
    path = Path(np.asarray(path_wp, dtype=np.float64))
    init = _start_on_path(path)
    start = np.array([init[0] + radius * np.sin(start_angle), init[1] + radius * np.cos(start_angle)])
    i = np.arange(5000)[:, None]
    pts = start[None, :] + speed * np.asarray(direction_vec)[None, :] * i
    pts[0] = start
    return WorldSpec(waypoints=np.asarray(path_wp, dtype=np.float64), vessel_init=init,
                     movers=[_mover_from_table(30, pts, dt, trailing_update=True)], name=name)


def test_head_on(seed: int = 0, dt: float = 0.5) -> WorldSpec:
    import random as _random
    a = _random.Random(seed).uniform(-5 * np.pi / 180, 5 * np.pi / 180)   # testscenario.py:147
    return _single_mover_world(np.vstack([[0, 0], [0, 250]]).T, a, 150, (-np.sin(a), -np.cos(a)), 0.5, dt, "TestHeadOn")


def test_crossing(dt: float = 0.5) -> WorldSpec:
    sh = 90 * np.pi / 180
    return _single_mover_world(np.vstack([[0, 0], [0, 500]]).T, -45 * np.pi / 180, 200, (np.sin(sh), np.cos(sh)), 0.5, dt,
                               "TestCrossing")


def test_crossing1(dt: float = 0.5) -> WorldSpec:
    sh = -50 * np.pi / 180
    return _single_mover_world(np.vstack([[0, 0], [0, 500]]).T, 70 * np.pi / 180, 200, (np.sin(sh), np.cos(sh)), 0.5, dt,
                               "TestCrossing1")


def empty_scenario() -> WorldSpec:
    wp = np.vstack([[25, 10], [25, 200]]).T.astype(np.float64)
    return WorldSpec(waypoints=wp, vessel_init=_start_on_path(Path(wp)), name="EmptyScenario")


def debug_scenario(seed: int = 0, dt: float = 0.5) -> WorldSpec:
    """testscenario.py:291-360: 5 movers on circles + 5 on straight lines, parameters from the
    env-local stream; no trailing _update()."""
    rng, _ = np_random(seed)
    wp = np.vstack([[250, 100], [250, 200]]).T.astype(np.float64)
    path = Path(wp)
    i = np.arange(10000)
    movers = []
    for idx in range(5):
        shift = rng.rand() * 2 * np.pi
        radius = rng.rand() * 40 + 30
        speed = rng.rand() * 0.003 + 0.003
        pts = np.stack([250 + radius * np.cos(speed * i + shift), 150 + 70 * idx + radius * np.sin(speed * i + shift)], axis=1)
        movers.append(_mover_from_table(6, pts, dt, trailing_update=False))
    for idx in range(5):
        start = rng.rand() * 200 + 150
        speed = rng.rand() * 0.03 + 0.03
        shift = 10 * rng.rand()
        pts = np.stack([np.full(len(i), 245 + 2.5 * idx + shift), start - 10 * speed * i], axis=1)
        movers.append(_mover_from_table(6, pts, dt, trailing_update=False))
    return WorldSpec(waypoints=wp, vessel_init=_start_on_path(path), movers=movers, name="DebugScenario")
