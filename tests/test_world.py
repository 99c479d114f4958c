"""World construction: scenario-generation stream parity with the reference, obstacle
boundaries / cull circles, bank packing."""
import numpy as np
import pytest

from gym_auv_amd import obstacles as ob
from gym_auv_amd.scenarios import moving_obstacles_world, polygon_world, static_circles_world
from gym_auv_amd.world import advance_mover, build_world, pack_bank
from gym_auv_amd.worldspec import pack_world, unpack_world
from helpers import load


@pytest.mark.parametrize("k,seed", [(0, 11), (1, 12), (2, 13)])
def test_moving_obstacles_world_matches_reference(k, seed):
    """env.seed(k); np.random.seed(k); env.reset() in the reference == moving_obstacles_world(k)."""
    z = load("g5_rollouts.npz")
    ref = unpack_world(z, "r%d_w_" % k)
    mine = moving_obstacles_world(seed)
    np.testing.assert_array_equal(ref.waypoints, mine.waypoints)
    np.testing.assert_allclose(ref.vessel_init, mine.vessel_init, rtol=0, atol=1e-15)
    np.testing.assert_array_equal(ref.circles, mine.circles)
    assert len(ref.movers) == len(mine.movers) == 17 and len(ref.circles) == 11
    for a, b in zip(ref.movers, mine.movers):
        assert a.width == b.width and a.n_vel == b.n_vel == 9999
        np.testing.assert_allclose(a.pos, b.pos, rtol=0, atol=1e-12)
        np.testing.assert_allclose(a.vel, b.vel, rtol=0, atol=1e-12)
        assert a.heading == pytest.approx(b.heading, abs=1e-14)
        assert a.counter == pytest.approx(0.6)      # ctor update(0.1) + _update(dt=0.5)


def test_circle_ring_matches_reference_boundary():
    z = load("g3_lidar.npz")
    for r, n in zip(z["ring_radii"], z["ring_ncoords"]):
        assert len(ob.circle_segments(3.0, -7.0, float(r))) == int(n) - 1, r
    seg = ob.circle_segments(3.0, -7.0, 30.0)
    ring = z["ring30_coords"]
    np.testing.assert_allclose(seg[:, :2], ring[:-1], rtol=0, atol=1e-12)
    np.testing.assert_allclose(seg[:, 2:], ring[1:], rtol=0, atol=1e-12)
    # clockwise, starts at (cx + r, cy)
    np.testing.assert_allclose(seg[0, :2], [33.0, -7.0])
    assert seg[0, 3] < -7.0


def test_segment_count_thresholds():
    # 4 / 8 / 16 / 32 / 64 segments by radius
    for r, n in [(0.5, 4), (1.0, 4), (1.5, 8), (3.9, 8), (4.0, 16), (15.0, 16), (16.0, 32), (30.0, 32),
                 (62.0, 32), (63.0, 64)]:
        assert len(ob.circle_segments(0, 0, r)) == n, r


def test_mover_cull_circle_closed_form_vs_general_mrr():
    rs = np.random.RandomState(3)
    for _ in range(200):
        w = float(max(1, rs.poisson(10)))
        px, py, h = rs.uniform(-500, 500), rs.uniform(-500, 500), rs.uniform(-np.pi, np.pi)
        pts = ob.mover_world_points(w, px, py, h)
        cx, cy, rad = ob.enclosing_circle_of_points(pts)
        ex, ey, er = ob.mover_cull_circle(w, px, py, h)
        assert (cx, cy, rad) == pytest.approx((ex, ey, er), abs=1e-9)
        assert er == pytest.approx(w * np.sqrt(5) / 2)


def test_reference_cull_circles_in_golden_scenes():
    """enclosing circles the reference computed (MRR path) vs ours, all obstacle kinds."""
    from helpers import scene_order, scene_world
    z = load("g3_lidar.npz")
    for i in range(len(z["names"])):
        w = build_world(scene_world(z, i))
        order = scene_order(z, i)
        cull = z["s%d_cull" % i]
        for ref_k, k in enumerate(order):
            if w.obs_meta[k, 0] == 2:
                m = w.spec.movers[w.obs_meta[k, 3]]
                mine = ob.mover_cull_circle(m.width, m.pos[0], m.pos[1], m.heading)
            else:
                mine = tuple(w.obs_cull[k])
            np.testing.assert_allclose(mine, cull[ref_k], rtol=0, atol=1e-9)


def test_advance_mover_wraps():
    param = (4.0, 10.0, 20.0, 5)       # n_vel = 5 -> wrap when floor(counter) >= 4
    vt = np.array([[1.0, 0.0]])
    st = (10.0, 20.0, np.pi / 2, 3.6)
    st = advance_mover(param, vt, st, 0.5)     # counter 4.1 -> wrap
    assert st[3] == 0.0 and st[0] == pytest.approx(10.5) and st[1] == 20.0
    assert st[2] == pytest.approx(0.0)


def test_pack_bank_layout_and_roundtrip(tmp_path):
    specs = [moving_obstacles_world(1), static_circles_world(2, 20), polygon_world(3, 10)]
    built = [build_world(s) for s in specs]
    bank = pack_bank(built)
    assert int(bank["n_worlds"]) == 3
    assert bank["poly_off"][-1] == len(bank["poly_xy"]) == len(bank["poly_cum"])
    assert bank["knot_off"][-1] == len(bank["knot_s"]) == 3000
    assert bank["obs_off"].tolist() == [0, 28, 48, 58]
    assert bank["mv_off"].tolist() == [0, 17, 17, 17]
    assert bank["k_max"] == 28 and bank["m_max"] == 17
    # absolute segment offsets stay inside the segment table and are contiguous per world
    meta = bank["obs_meta"]
    static = meta[meta[:, 0] != 2]
    assert (static[:, 1] + static[:, 2]).max() == len(bank["seg"])
    assert (np.diff(bank["poly_cum"][:int(bank["poly_off"][1])]) > 0).all()
    # npz round trip of the spec
    d = pack_world("w_", specs[2])
    np.savez(tmp_path / "w.npz", **d)
    back = unpack_world(np.load(tmp_path / "w.npz"), "w_")
    assert len(back.polygons) == 10
    np.testing.assert_array_equal(back.polygons[3], specs[2].polygons[3])


def test_fixed_test_scenarios_match_reference_worlds():
    """TestScenario1 / TestHeadOn / DebugScenario generators vs worlds captured from the
    reference (first-episode semantics: the reference's TestScenario* append their obstacles
    again on every reset, so the captured second-episode list holds each circle twice)."""
    from gym_auv_amd import scenarios as sc
    z = load("g5_rollouts.npz")
    names = [str(n) for n in z["names"]]
    for name, mine in (("testscenario1", sc.test_scenario1()), ("headon", sc.test_head_on(51)),
                       ("debug", sc.debug_scenario(61))):
        ref = unpack_world(z, "r%d_w_" % names.index(name))
        np.testing.assert_array_equal(ref.waypoints, mine.waypoints)
        np.testing.assert_allclose(ref.vessel_init, mine.vessel_init, rtol=0, atol=1e-15)
        nc = len(mine.circles)
        if nc:
            np.testing.assert_array_equal(ref.circles[:nc], mine.circles)
            assert len(ref.circles) in (nc, 2 * nc)
        assert len(ref.movers) == len(mine.movers)
        for a, b in zip(ref.movers, mine.movers):
            assert a.width == b.width and a.n_vel == b.n_vel and a.counter == pytest.approx(b.counter)
            np.testing.assert_allclose(a.pos, b.pos, rtol=0, atol=1e-12)
            np.testing.assert_allclose(a.vel, b.vel, rtol=0, atol=1e-12)
    assert len(sc.test_scenario2().circles) == 278
    assert len(sc.test_scenario3().circles) == 21
    assert len(sc.test_scenario4().circles) == 15      # the boolean-compare quirk (testscenario.py:125)
    assert len(sc.test_crossing().movers) == len(sc.test_crossing1().movers) == 1
    assert len(sc.empty_scenario().circles) == 0


def test_bank_order_independent_of_worker_count():
    """World i of a bank built in parallel is seeds[i] whatever the number of worker processes
    (chunks used to be strided and the permutation was never undone)."""
    from gym_auv_amd.world import build_bank_parallel
    seeds = range(1000, 1013)
    one = build_bank_parallel("static_circles_world", seeds, procs=1, n_circles=3)
    for procs in (2, 3):
        par = build_bank_parallel("static_circles_world", seeds, procs=procs, n_circles=3)
        assert set(par) == set(one)
        for k in one:
            np.testing.assert_array_equal(np.asarray(par[k]), np.asarray(one[k]), err_msg="%s procs=%d" % (k, procs))
