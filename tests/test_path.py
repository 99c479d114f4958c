"""Path build (reset-time host code) vs the reference's Path (G2 golden vectors) and SciPy."""
import numpy as np
import pytest

from gym_auv_amd.path import Path, RandomCurveThroughOrigin, hermite_coefs, pchip_slopes, ppoly_eval
from gym_auv_amd.seeding import np_random
from helpers import load


@pytest.fixture(scope="module")
def g2():
    return load("g2_path.npz")


@pytest.mark.parametrize("k", range(11))
def test_path_tables_match_reference(g2, k):
    pre = "p%d_" % k
    p = Path(g2[pre + "waypoints"])
    assert p.length == pytest.approx(float(g2[pre + "length"]), abs=1e-12)
    assert len(p.points) == int(g2[pre + "npoints"])
    np.testing.assert_allclose(p.knot_s, g2[pre + "knots_s"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(p._waypoints, g2[pre + "knots_xy"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(p.points[::61], g2[pre + "points_sub"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(p.points[-1], g2[pre + "points_last"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(p.points.sum(axis=0), g2[pre + "points_sum"], rtol=1e-13)
    ss = g2[pre + "eval_s"]   # includes extrapolation beyond both ends
    np.testing.assert_allclose(np.array([p(s) for s in ss]), g2[pre + "eval_xy"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(np.array([p.get_direction(s) for s in ss]), g2[pre + "eval_dir"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("k", range(11))
def test_projection_matches_reference(g2, k):
    pre = "p%d_" % k
    p = Path(g2[pre + "waypoints"])
    q, out = g2[pre + "nav_query"], g2[pre + "nav_out"]
    s = np.array([p.get_closest_arclength(q[i, :2]) for i in range(len(q))])
    np.testing.assert_allclose(s, out[:, 0], rtol=0, atol=1e-10)


def test_random_curve_stream_matches_reference(g2):
    for seed in range(8):
        rng, _ = np_random(seed)
        nw = int(np.floor(4 * rng.rand() + 2))
        p = RandomCurveThroughOrigin(rng, nw, length=800)
        np.testing.assert_array_equal(p.init_waypoints, g2["p%d_waypoints" % seed])
        assert p.init_waypoints.shape[1] in (5, 7)


def test_pchip_against_scipy():
    scipy_interp = pytest.importorskip("scipy.interpolate")
    rs = np.random.RandomState(0)
    for n in (2, 3, 5, 7, 50, 1000):
        x = np.cumsum(rs.uniform(0.1, 2.0, n))
        y = np.cumsum(rs.normal(0, 1, n))
        if n > 5:
            y[3:5] = y[3]          # flat run -> zero slopes
        ref = scipy_interp.PchipInterpolator(x, y)
        c = hermite_coefs(x, y, pchip_slopes(x, y))
        s = np.concatenate([rs.uniform(x[0] - 3, x[-1] + 3, 200), x])
        np.testing.assert_allclose(ppoly_eval(x, c, s), ref(s), rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(ppoly_eval(x, c, s, 1), ref.derivative()(s), rtol=1e-12, atol=1e-12)


def test_two_waypoint_path_is_straight():
    p = Path(np.array([[0.0, 0.0], [0.0, 500.0]]))
    assert p.length == pytest.approx(500.0)
    assert len(p.points) == 5000
    np.testing.assert_allclose(p(250.0), [0.0, 250.0], atol=1e-12)
    assert p.get_direction(10.0) == pytest.approx(np.pi / 2)
