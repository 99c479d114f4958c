"""On-device scenario generation: MovingObstacles-type worlds are built on
the GPU from device-resident random draws, straight into fixed-capacity world slots in HBM, so a
fresh world per episode no longer needs the host.

The algorithm is the reference's `MovingObstacles._generate`
(/root/reference/gym_auv/envs/movingobstacles.py:28-95), `RandomCurveThroughOrigin` / `Path`
(objects/path.py:19-40, :96-120: three PCHIP re-parameterisations + dense polyline) and
`helpers.generate_obstacle` (utils/helpers.py:5-35), with one documented difference: the
rejection loop of generate_obstacle draws from a pool of CAND pre-drawn candidates per obstacle
(first accepted wins); should the whole pool be rejected (probability ~1e-13 per obstacle) up to EXTRA_CAND
further candidates come from a counter-based generator keyed by the pool itself, identically on host and
device, so that -- like helpers.generate_obstacle -- no obstacle is ever placed on the vessel or the goal.

`world_from_draws` is the host mirror consuming the SAME draws; tests compare the device-built
tables with `build_world(world_from_draws(...))`.  Stream parity with NumPy's generators is a
property of `scenarios.moving_obstacles_world`, not of this module (the draws come from the
torch generator on the device).
"""
import functools
import math
from dataclasses import dataclass
from typing import Tuple

import numpy as np

from .obstacles import QUADRANT_SEGMENTS, SIMPLIFY_TOLERANCE, circle_ring, douglas_peucker_keep
from .path import Path
from .scenarios import VESSEL_WIDTH, _linear_mover, _princip
from .worldspec import WorldSpec

CAND = 8                 # candidate placements per obstacle
R_TABLE = 256            # circle radii are max(1, Poisson) integers; table of segment counts by radius


@dataclass(frozen=True)
class GeneratedWorlds:
    """`worlds=` argument of BatchedAuvEnv asking for a bank built on the device: n_worlds
    MovingObstacles-type scenarios (defaults of envs/movingobstacles.py:45-48: 17 movers, 11
    circles) from the torch generator seeded with `seed` on the env's GPU."""
    n_worlds: int
    n_moving: int = 17
    n_static: int = 11
    seed: int = 0


@dataclass(frozen=True)
class FreshWorlds:
    """`worlds=` argument of BatchedAuvEnv asking for A FRESH WORLD ON EVERY RESET (auv_fresh_worlds_create): what the
    reference does -- reset() -> _generate() builds a new scenario whenever an episode ends (environment.py:176-218,
    envs/movingobstacles.py:28-95) -- with the generation on the device and off the step path.  `depth` bank slots per
    environment (>= 2); the world of environment e's k-th episode is the world of (seed, env_index_base + e, k) whatever the
    timing, the sub-batch chains or the sharding over GPUs.  A refill pass of up to `batch_cap` worlds is enqueued on a side
    stream every `period` step calls; `BatchedAuvEnv.fresh_stats()["reused"]` counts episodes that had to start in the world
    they had just finished because the pass fell behind (0 when depth / period / batch_cap fit the turn-over rate)."""
    depth: int = 3           # (measured on moving28 with i.i.d. random actions, where a quarter of the episodes end within a few
    n_moving: int = 17       #  dozen steps of their reset: depth 2 re-used 25-80 of 11 800 worlds, depth 3 none -- profiles/r05)
    n_static: int = 11
    seed: int = 0
    env_index_base: int = 0
    batch_cap: int = 64
    period: int = 16         # (a pass takes ~0.3 ms whatever it holds: fewer, fuller passes cost the chains less)


def n_draws(n_moving: int, n_static: int) -> int:
    return 11 + n_moving * (3 * CAND + 2) + n_static * 3 * CAND


def sample_draws(n_worlds: int, n_moving: int, n_static: int, seed: int = 0, device="cpu"):
    """[W, ND] float64 draws: U[0,1) everywhere except the candidate triples (z ~ N(0,1), u, Poisson)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    nd = n_draws(n_moving, n_static)
    d = torch.rand((n_worlds, nd), generator=g, device=device, dtype=torch.float64)
    col = 11
    for mean, count, extra in ((10.0, n_moving, 2), (30.0, n_static, 0)):
        for _ in range(count):
            for _c in range(CAND):
                d[:, col] = torch.randn((n_worlds,), generator=g, device=device, dtype=torch.float64)
                d[:, col + 2] = torch.poisson(torch.full((n_worlds,), mean, device=device, dtype=torch.float64), generator=g)
                col += 3
            col += extra
    assert col == nd
    return d


@functools.lru_cache(maxsize=1)
def ring_tables() -> Tuple[np.ndarray, np.ndarray]:
    """(unit ring [65, 2] = cos/sin of GEOS's accumulated vertex angles, nseg_by_radius [R_TABLE])
    -- the Douglas-Peucker outcome on a circle is a uniform power-of-two subsample (checked here
    for every integer radius), so a radius -> segment-count table reproduces the host builder."""
    unit = circle_ring(0.0, 0.0, 1.0)
    nseg = np.zeros(R_TABLE, dtype=np.int32)
    for r in range(1, R_TABLE):
        keep = douglas_peucker_keep(circle_ring(0.0, 0.0, float(r)), SIMPLIFY_TOLERANCE)
        n = len(keep) - 1
        stride = (4 * QUADRANT_SEGMENTS) // n
        assert n in (4, 8, 16, 32, 64) and np.array_equal(keep, np.arange(0, 4 * QUADRANT_SEGMENTS + 1, stride)), r
        nseg[r] = n
    nseg[0] = nseg[1]
    return unit, nseg


EXTRA_CAND = 56          # further candidates drawn on the spot if the whole pool is rejected (never seen in practice)
_M64 = (1 << 64) - 1


def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def extra_candidate(row, base: int, k: int, mean: float):
    """Candidate k >= CAND of the obstacle whose pool starts at row[base]: (z, u, poisson) from a counter-based
    generator keyed by the bit patterns of the pool's first two draws -- the same integers on the host and on the
    device (k5_generate.hip), so the two stay in step.  z by Box-Muller, the Poisson variate by inversion."""
    key = int(np.float64(row[base]).view(np.uint64)) ^ ((int(np.float64(row[base + 1]).view(np.uint64)) << 1) & _M64)
    u = [(_splitmix64((key + 4 * (k - CAND) + i) & _M64) >> 11) * 2.0 ** -53 for i in range(4)]
    z = math.sqrt(-2.0 * math.log(1.0 - u[0])) * math.cos(2.0 * math.pi * u[1])
    p = math.exp(-mean)
    cdf, n = p, 0
    while u[3] > cdf and n < 1000:
        n += 1
        p *= mean / n
        cdf += p
    return z, u[2], float(n)


def _place(row, base, path: Path, pose, width, sigma, mean=None):
    goal = path(path.length)
    c, s = math.cos(-pose[2]), math.sin(-pose[2])
    pos = radius = None
    if mean is None:
        mean = 10.0 if sigma == 500.0 else 30.0          # obst_radius_mean of the movers / the circles (movingobstacles.py:54-90)
    for k in range(CAND + EXTRA_CAND):
        if k < CAND:
            z, u, pois = row[base + 3 * k], row[base + 3 * k + 1], row[base + 3 * k + 2]
        else:
            z, u, pois = extra_candidate(row, base, k, mean)
        disp = sigma * z
        arclen = (0.1 + 0.8 * u) * path.length
        pos = path(arclen)
        ang = _princip(path.get_direction(arclen) - np.pi / 2)
        pos = pos + disp * np.array([np.cos(ang), np.sin(ang)])
        radius = max(1.0, pois)
        dx, dy = pos[0] - pose[0], pos[1] - pose[1]
        vessel_dist = math.sqrt((c * dx - s * dy) ** 2 + (s * dx + c * dy) ** 2) - width - radius
        goal_dist = math.sqrt((pos[0] - goal[0]) ** 2 + (pos[1] - goal[1]) ** 2) - radius
        if min(vessel_dist, goal_dist) > 0:
            break
    return pos, radius


def world_from_draws(row: np.ndarray, n_moving: int = 17, n_static: int = 11, dt: float = 0.5,
                     vessel_width: float = VESSEL_WIDTH) -> WorldSpec:
    """Host mirror of the device generator for one row of draws."""
    row = np.asarray(row, dtype=np.float64)
    nwaypoints = int(np.floor(4 * row[0] + 2))
    length = 800.0
    theta0 = 2 * np.pi * (row[1] - 0.5)
    start = 0.5 * length * np.array([np.cos(theta0), np.sin(theta0)])
    end = -start
    half = nwaypoints // 2
    head, tail = [], []
    for k in range(half):
        jit1 = length / (half + 1) * (row[2 + 2 * k] - 0.5)
        jit2 = length / (half + 1) * (row[3 + 2 * k] - 0.5)
        head.append((half - k) * start / (half + 1) + jit1)
        tail.insert(0, (half - k) * end / (half + 1) + jit2)
    path = Path(np.array([start] + head + [np.zeros(2)] + tail + [end]).T)
    p0 = path(0)
    pose = np.array([p0[0] + 50 * (row[8] - 0.5), p0[1] + 50 * (row[9] - 0.5),
                     _princip(path.get_direction(0) + 2 * np.pi * (row[10] - 0.5))])
    movers, circles = [], []
    col = 11
    for _ in range(n_moving):
        pos, radius = _place(row, col, path, pose, vessel_width, 500.0)
        direction = row[col + 3 * CAND] * 2 * np.pi
        speed = 1.0 + (3.0 - 1.0) * row[col + 3 * CAND + 1]
        movers.append(_linear_mover(pos, radius, direction, speed, dt))
        col += 3 * CAND + 2
    for _ in range(n_static):
        pos, radius = _place(row, col, path, pose, vessel_width, 250.0)
        circles.append([pos[0], pos[1], radius])
        col += 3 * CAND
    return WorldSpec(waypoints=path.init_waypoints, vessel_init=pose,
                     circles=np.asarray(circles, dtype=np.float64).reshape(-1, 3), movers=movers, name="devgen")


# ---- host mirror of the device's counter-based draws (k5_generate.hip: k5_draws) -------------------------------------------------
def _world_key(seed: int, env: int, serial: int) -> int:
    k0 = _splitmix64(seed & _M64)
    k1 = _splitmix64(k0 ^ (env & _M64))
    return _splitmix64(k1 ^ (serial & 0xFFFFFFFF))


def _poisson_inv(mean: float, u: float) -> float:
    p = math.exp(-mean)
    cdf, n = p, 0
    while u > cdf and n < 1000:
        n += 1
        p *= mean / n
        cdf += p
    return float(n)


def counter_draws(seed: int, env: int, serial: int, n_moving: int = 17, n_static: int = 11) -> np.ndarray:
    """The draws row of the world environment `env` (GLOBAL index) meets in its `serial`-th episode in the fresh-world mode:
    the same integers as the device (uniforms are bit-identical; z and the Poisson counts go through the host's libm, so they
    agree to the last bits / except at a comparison that falls within an ulp -- the GPU tests therefore read the device's own
    rows back, auv_fresh_worlds_draws, and use this function to pin the generator's definition)."""
    key = _world_key(seed, env, serial)
    nd = n_draws(n_moving, n_static)
    per_mover = 3 * CAND + 2
    movers_end = 11 + n_moving * per_mover
    out = np.empty(nd, dtype=np.float64)
    for j in range(nd):
        kind = 0
        if 11 <= j < movers_end:
            r = (j - 11) % per_mover
            if r < 3 * CAND:
                kind = 1 if r % 3 == 0 else (2 if r % 3 == 2 else 0)
        elif j >= movers_end:
            r = (j - movers_end) % (3 * CAND)
            kind = 1 if r % 3 == 0 else (3 if r % 3 == 2 else 0)
        u0 = (_splitmix64((key + 2 * j) & _M64) >> 11) * 2.0 ** -53
        u1 = (_splitmix64((key + 2 * j + 1) & _M64) >> 11) * 2.0 ** -53
        if kind == 1:
            out[j] = math.sqrt(-2.0 * math.log(1.0 - u0)) * math.cos(2.0 * math.pi * u1)
        elif kind == 2:
            out[j] = _poisson_inv(10.0, u0)
        elif kind == 3:
            out[j] = _poisson_inv(30.0, u0)
        else:
            out[j] = u0
    return out
