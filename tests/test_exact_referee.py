"""Plain-fp64 predicates (HIP kernel = C oracle = harness shim) against an exact-arithmetic referee
(tests/referee.py, what a robust GEOS predicate answers): how many (ray, boundary segment) decisions
differ and how far the resulting ranges move.  VERDICT r1 "next" #1(b); the counts asserted here are the
ones DESIGN.md section 5 quotes.  Reference call site: objects/vessel/sensor.py:140-159.

`python tests/test_exact_referee.py` prints the full statistics."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from gym_auv_amd.obstacles import mover_world_points, polygon_segments   # noqa: E402
from gym_auv_amd.world import OBS_FILLED, OBS_MOVER, build_world          # noqa: E402

try:
    from tests import referee as Rf
    from tests.helpers import cfg_from_scalars, load, scene_world
except ImportError:                                                        # run as a script
    import referee as Rf
    from helpers import cfg_from_scalars, load, scene_world


def _scene_segments(z, i):
    """All boundary segments of golden LiDAR scene i: (seg[G,4], list of (kind, lo, hi))."""
    bw = build_world(scene_world(z, i))
    segs, obs = [bw.seg], []
    for kind, off, nseg, _ in bw.obs_meta:
        if kind != OBS_MOVER:
            obs.append((int(kind), int(off), int(off + nseg)))
    g = len(bw.seg)
    for par, st in zip(bw.mv_param, bw.mv_init):
        s = polygon_segments(mover_world_points(par[0], st[0], st[1], st[2]))
        segs.append(s)
        obs.append((OBS_FILLED, g, g + len(s)))
        g += len(s)
    return np.concatenate(segs), obs


def scene_statistics():
    z = load("g3_lidar.npz")
    tot = dict(scenes=0, rays=0, pairs=0, uncertain=0, differ=0, rays_moved=0, max_dd=0.0, pip=0, pip_differ=0)
    for i in range(len(z["names"])):
        pre = "s%d_" % i
        cfg = cfg_from_scalars(z["cfg_keys"], z[pre + "cfg"])
        S, R = cfg.vessel.n_sensors, cfg.vessel.sensor_range
        px, py, psi = (float(v) for v in z[pre + "pose"])
        seg, obs = _scene_segments(z, i)
        tot["scenes"] += 1
        for kind, lo, hi in obs:
            if kind == OBS_FILLED:
                tot["pip"] += 1
                tot["pip_differ"] += int(Rf.fp64_point_in_polygon(px, py, seg[lo:hi]) != Rf.exact_point_in_polygon(px, py, seg[lo:hi]))
        if len(seg) == 0:
            continue
        for r in range(S):
            e = Rf.ray_end(px, py, psi, r, S, R)
            st = Rf.referee_ray((px, py), e, seg, R)
            tot["rays"] += 1
            tot["pairs"] += st["n_pairs"]
            tot["uncertain"] += st["n_uncertain"]
            tot["differ"] += st["n_differ"]
            dd = abs(st["d_fp64"] - st["d_exact"])
            tot["max_dd"] = max(tot["max_dd"], dd)
            tot["rays_moved"] += int(dd > 1e-9)
    return tot


def test_golden_scenes_fp64_equals_exact():
    """All (ray, segment) pairs of the 121 golden LiDAR scenes (every beam against every boundary segment,
    a superset of what the reference's windows test): plain fp64 and exact arithmetic take the same
    decision on every pair, so no range moves."""
    t = scene_statistics()
    assert t["scenes"] == 121 and t["pairs"] > 2_000_000
    assert t["differ"] == 0, t
    assert t["pip_differ"] == 0, t
    assert t["rays_moved"] == 0 and t["max_dd"] < 1e-9, t


# ------------------------------------------------------------------------ adversarial grazing set
def _dy(rs, lo, hi, q=1024):
    """random dyadic rational in [lo, hi) with denominator q: sums / differences of a few stay exact"""
    return math.floor(rs.uniform(lo, hi) * q) / q


def _config(rs, kind):
    """One grazing configuration: (p0, e, seg[G,4] of a small closed polygon, R).  kinds:
    0 ray exactly through a vertex (dyadic coordinates: the three points are exactly collinear)
    1 ray through a vertex up to rounding (generic doubles, direction from atan2 -> cos/sin)
    2 polygon edge exactly collinear with the ray
    3 p0 exactly on an edge, ray leaves along / across it
    4 ray END point exactly on an edge (t == 1)"""
    R = 150.0
    if kind == 1:
        p0 = (rs.uniform(-500, 500), rs.uniform(-500, 500))
        dist, ang = rs.uniform(2, 140), rs.uniform(-math.pi, math.pi)
        v = (p0[0] + dist * math.cos(ang), p0[1] + dist * math.sin(ang))
        th = math.atan2(v[1] - p0[1], v[0] - p0[0])
        e = (p0[0] + math.cos(th) * R, p0[1] + math.sin(th) * R)
        nrm = (-math.sin(ang), math.cos(ang))
        s1, s2, dp = rs.uniform(1, 20), rs.uniform(1, 20), rs.uniform(1, 20)
        u = (v[0] + s1 * nrm[0] + dp * math.cos(ang) * rs.uniform(0.0, 1.0), v[1] + s1 * nrm[1] + dp * math.sin(ang) * rs.uniform(0.0, 1.0))
        w = (v[0] - s2 * nrm[0] + dp * math.cos(ang) * rs.uniform(0.0, 1.0), v[1] - s2 * nrm[1] + dp * math.sin(ang) * rs.uniform(0.0, 1.0))
        off = rs.uniform(-0.5, 0.5) * (s1 + s2)                 # the exit on the far side is a clean crossing
        far = (v[0] + (dp + 25) * math.cos(ang) + off * nrm[0], v[1] + (dp + 25) * math.sin(ang) + off * nrm[1])
        pts = [u, v, w, far]
    else:
        p0 = (_dy(rs, -500, 500), _dy(rs, -500, 500))
        d = (_dy(rs, -1, 1, 64), _dy(rs, -1, 1, 64))
        while d == (0.0, 0.0):
            d = (_dy(rs, -1, 1, 64), _dy(rs, -1, 1, 64))
        n = (-d[1], d[0])
        k1 = float(rs.randint(4, 60))
        v = (p0[0] + k1 * d[0], p0[1] + k1 * d[1])
        e = (p0[0] + 128.0 * d[0], p0[1] + 128.0 * d[1])
        a1, a2 = float(rs.randint(1, 12)), float(rs.randint(1, 12))
        b1, b2 = float(rs.randint(0, 8)), float(rs.randint(0, 8))
        u = (v[0] + a1 * n[0] + b1 * d[0], v[1] + a1 * n[1] + b1 * d[1])
        w = (v[0] - a2 * n[0] + b2 * d[0], v[1] - a2 * n[1] + b2 * d[1])
        far = (v[0] + 40.0 * d[0], v[1] + 40.0 * d[1])
        if kind == 0:
            pts = [u, v, w, far]
        elif kind == 2:
            k2 = float(rs.randint(2, 30))
            v2 = (v[0] + k2 * d[0], v[1] + k2 * d[1])           # edge v -> v2 lies on the ray's line
            pts = [u, v, v2, (v2[0] - a2 * n[0], v2[1] - a2 * n[1]), w]
        elif kind == 3:
            # p0 is the midpoint of edge (q1, q2) along n; polygon extends to +d
            q1 = (p0[0] + a1 * n[0], p0[1] + a1 * n[1])
            q2 = (p0[0] - a1 * n[0], p0[1] - a1 * n[1])
            pts = [q1, (q1[0] + 30.0 * d[0], q1[1] + 30.0 * d[1]), (q2[0] + 30.0 * d[0], q2[1] + 30.0 * d[1]), q2]
            if rs.rand() < 0.5:                                   # ray along the edge instead of across
                e = (p0[0] + 16.0 * n[0], p0[1] + 16.0 * n[1])
        else:
            # an edge through the ray's end point, perpendicular to it
            q1 = (e[0] + a1 * n[0], e[1] + a1 * n[1])
            q2 = (e[0] - a2 * n[0], e[1] - a2 * n[1])
            pts = [q1, q2, (q2[0] + 9.0 * d[0], q2[1] + 9.0 * d[1]), (q1[0] + 9.0 * d[0], q1[1] + 9.0 * d[1])]
            R = math.hypot(e[0] - p0[0], e[1] - p0[1])
    return p0, e, polygon_segments(np.asarray(pts, dtype=np.float64)), R


KINDS = ["vertex, exactly collinear", "vertex, up to rounding", "collinear edge", "p0 on an edge", "ray end on an edge"]


def grazing_statistics(n_per_kind=20000, seed=7):
    rs = np.random.RandomState(seed)
    out = []
    for kind in range(5):
        st = dict(kind=KINDS[kind], configs=0, pairs=0, uncertain=0, differ=0, rays_moved=0, max_dd=0.0, pip_differ=0)
        for _ in range(n_per_kind):
            p0, e, seg, R = _config(rs, kind)
            r = Rf.referee_ray(p0, e, seg, R)
            st["configs"] += 1
            st["pairs"] += r["n_pairs"]
            st["uncertain"] += r["n_uncertain"]
            st["differ"] += r["n_differ"]
            d_fp, d_ex = r["d_fp64"], r["d_exact"]
            if kind == 3:
                # a FILLED obstacle containing p0 (boundary included) gives range 0 on its rays whatever the
                # pair tests say (sensor.py:145-152: the clipped ray piece starts at p0)
                pf, pe = Rf.fp64_point_in_polygon(p0[0], p0[1], seg), Rf.exact_point_in_polygon(p0[0], p0[1], seg)
                st["pip_differ"] += int(pf != pe)
                d_fp, d_ex = (0.0 if pf else d_fp), (0.0 if pe else d_ex)
            dd = abs(d_fp - d_ex)
            st["max_dd"] = max(st["max_dd"], dd)
            st["rays_moved"] += int(dd > 1e-9)
        out.append(st)
    return out


def test_grazing_rays_outlier_count():
    """1e5 adversarial configurations (20 000 per class).  Plain fp64 and exact arithmetic may only
    disagree where the configuration is degenerate in EXACT arithmetic or within rounding of it, and
    the count of rays whose range moves is asserted, not hidden."""
    stats = grazing_statistics()
    by = {s["kind"]: s for s in stats}
    assert sum(s["configs"] for s in stats) >= 100000
    # (1) exactly representable degeneracies (dyadic coordinates): every product in the kernel's
    #     cross products is exact there, so fp64 decides like exact arithmetic -- except that the kernel
    #     skips PARALLEL pairs (den == 0) where GEOS reports the collinear overlap.  The neighbours of a
    #     collinear edge are met at its end points, and a filled obstacle with p0 on its boundary gives
    #     range 0 through the point-in-polygon rule, so no range moves.
    for k in ("vertex, exactly collinear", "ray end on an edge"):
        assert by[k]["differ"] == 0 and by[k]["rays_moved"] == 0, by[k]
    ce = by["collinear edge"]
    assert ce["differ"] == ce["configs"] and ce["rays_moved"] == 0 and ce["max_dd"] < 1e-9, ce
    oe = by["p0 on an edge"]
    assert oe["pip_differ"] == 0 and oe["rays_moved"] == 0 and 0 < oe["differ"] < 0.6 * oe["configs"], oe
    # (2) rays aimed at a vertex through atan2 -> cos/sin (they pass within ~1e-14 m of it): the one place
    #     where plain fp64 can take the other side.  Measured (seed 7): 3 868 of 80 000 pair decisions
    #     differ, mostly harmlessly (both neighbours reported at the vertex instead of one); 199 of the
    #     20 000 rays (1.0 %) slip between the two neighbours and report the far side of the obstacle or
    #     no return at all (largest |delta d| 147.8 m).  A ray of a real sweep comes that close to a
    #     vertex with probability ~1e-14 / (beam spacing x distance): none in the 8.4 M golden pairs.
    vr = by["vertex, up to rounding"]
    assert vr["differ"] <= 0.25 * vr["pairs"] / 4, vr
    assert 1 <= vr["rays_moved"] <= 0.015 * vr["configs"], vr


if __name__ == "__main__":
    import json
    print(json.dumps(scene_statistics(), indent=1))
    for s in grazing_statistics():
        print(json.dumps(s))
