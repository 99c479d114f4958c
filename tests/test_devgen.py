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
