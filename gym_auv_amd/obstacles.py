"""Obstacle boundaries as flat segment lists + enclosing ("cull") circles, host side.

Restates what /root/reference/gym_auv/objects/obstacles.py builds through Shapely/GEOS:
  * CircularObstacle._calculate_boundary (:101-106)
        Point.buffer(r).boundary.simplify(0.3, preserve_topology=False)
        = GEOS circle (16 segments per quadrant, first vertex (cx+r, cy), clockwise),
          kept as an UNFILLED ring, thinned by Douglas-Peucker (tolerance 0.3 m).
    enclosing circle = (centre, r) exactly (:108-113).
  * PolygonObstacle (:116-127): FILLED polygon; enclosing circle from the minimum rotated
    rectangle (:235-262), Shapely 1.7's min-area search over convex-hull edges.
  * VesselObstacle (:175-181, :217-233): pentagon rotated about its centroid, translated;
    enclosing circle recomputed from the rotated shape each step.
GEOS is a third-party dependency absent from /root/reference (shapely==1.7.0,
requirements.txt:5); vertex placement / Douglas-Peucker output are restated from the
published GEOS algorithms and are "parity unpinned" against real GEOS.
"""
import functools
import math
from typing import Tuple

import numpy as np

QUADRANT_SEGMENTS = 16          # Shapely default buffer resolution
SIMPLIFY_TOLERANCE = 0.3        # obstacles.py:105


def _pt_seg_dist(px, py, ax, ay, bx, by):
    if ax == bx and ay == by:
        return math.hypot(px - ax, py - ay)
    len2 = (bx - ax) * (bx - ax) + (by - ay) * (by - ay)
    r = ((px - ax) * (bx - ax) + (py - ay) * (by - ay)) / len2
    if r <= 0.0:
        return math.sqrt((px - ax) ** 2 + (py - ay) ** 2)
    if r >= 1.0:
        return math.sqrt((px - bx) ** 2 + (py - by) ** 2)
    s = ((ay - py) * (bx - ax) - (ax - px) * (by - ay)) / len2
    return abs(s) * math.sqrt(len2)


def circle_ring(cx: float, cy: float, r: float) -> np.ndarray:
    """GEOS buffer of a point: closed clockwise ring of 4*QUADRANT_SEGMENTS+1 vertices."""
    n = 4 * QUADRANT_SEGMENTS
    inc = 2.0 * math.pi / n
    pts = np.empty((n + 1, 2))
    pts[0] = (cx + r, cy)
    ang = 0.0
    for k in range(1, n):
        ang += inc                      # GEOS accumulates the angle by repeated addition
        pts[k] = (cx + r * math.cos(-ang), cy + r * math.sin(-ang))
    pts[n] = pts[0]
    return pts


def douglas_peucker_keep(pts: np.ndarray, tol: float) -> np.ndarray:
    """Indices kept by GEOS DouglasPeuckerLineSimplifier (first strict maximum splits;
    a section collapses when its maximum distance <= tol)."""
    n = len(pts)
    keep = np.ones(n, dtype=bool)
    stack = [(0, n - 1)]
    while stack:
        i, j = stack.pop()
        if i + 1 >= j:
            continue
        best, bk = -1.0, i
        for k in range(i + 1, j):
            d = _pt_seg_dist(pts[k, 0], pts[k, 1], pts[i, 0], pts[i, 1], pts[j, 0], pts[j, 1])
            if d > best:
                best, bk = d, k
        if best <= tol:
            keep[i + 1:j] = False
        else:
            stack.append((i, bk))
            stack.append((bk, j))
    return np.nonzero(keep)[0]


@functools.lru_cache(maxsize=4096)
def _ring_keep_for_radius(r: float) -> Tuple[int, ...]:
    # Douglas-Peucker on a circle depends on the radius only (translation invariant up to
    # rounding far below the 0.3 m tolerance), so cache the kept vertex indices per radius.
    return tuple(int(k) for k in douglas_peucker_keep(circle_ring(0.0, 0.0, r), SIMPLIFY_TOLERANCE))


def circle_segments(cx: float, cy: float, r: float) -> np.ndarray:
    """Boundary segments [n, 4] (ax, ay, bx, by) of a CircularObstacle."""
    ring = circle_ring(cx, cy, r)[list(_ring_keep_for_radius(float(r)))]
    return np.concatenate([ring[:-1], ring[1:]], axis=1)


def polygon_segments(points: np.ndarray) -> np.ndarray:
    p = np.asarray(points, dtype=np.float64)
    if not np.array_equal(p[0], p[-1]):
        p = np.vstack([p, p[:1]])
    return np.concatenate([p[:-1], p[1:]], axis=1)


def convex_hull(points: np.ndarray) -> np.ndarray:
    pts = sorted(set(map(tuple, np.asarray(points, dtype=np.float64))))
    if len(pts) <= 2:
        return np.array(pts)

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower, upper = [], []
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    for p in reversed(pts):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    return np.array(lower[:-1] + upper[:-1])


def enclosing_circle_of_points(points: np.ndarray) -> Tuple[float, float, float]:
    """obstacles.py:235-262: centre of the minimum rotated rectangle, radius = distance to
    its farthest corner.  MRR = min-area box over hull-edge directions (first min wins)."""
    hull = convex_hull(points)
    if len(hull) < 3:
        c = hull.mean(axis=0)
        return float(c[0]), float(c[1]), 1.0
    best_area, best = None, None
    m = len(hull)
    for i in range(m):
        e = hull[(i + 1) % m] - hull[i]
        ln = math.hypot(e[0], e[1])
        ux, uy = e[0] / ln, e[1] / ln
        a = hull[:, 0] * ux + hull[:, 1] * uy
        b = -hull[:, 0] * uy + hull[:, 1] * ux
        area = (a.max() - a.min()) * (b.max() - b.min())
        if best_area is None or area < best_area:
            best_area, best = area, (ux, uy, a.min(), a.max(), b.min(), b.max())
    ux, uy, a0, a1, b0, b1 = best
    ca, cb = 0.5 * (a0 + a1), 0.5 * (b0 + b1)
    cx, cy = ca * ux - cb * uy, ca * uy + cb * ux
    rad = 0.5 * math.hypot(a1 - a0, b1 - b0)
    return float(cx), float(cy), float(rad)


# ---- VesselObstacle (moving pentagon) -------------------------------------------------
def mover_body_points(w: float) -> np.ndarray:
    """obstacles.py:175-181."""
    return np.array([(-w / 2, -w / 2), (-w / 2, w / 2), (w / 2, w / 2), (3 / 2 * w, 0.0), (w / 2, -w / 2)])


def mover_world_points(w: float, px: float, py: float, heading: float) -> np.ndarray:
    """obstacles.py:217-228: rotate about the polygon centroid (= (5w/18, 0) in body axes),
    then translate by the position.  shapely.affinity.rotate snaps |cos|,|sin| < 2.5e-16 to 0."""
    c, s = math.cos(heading), math.sin(heading)
    if abs(c) < 2.5e-16:
        c = 0.0
    if abs(s) < 2.5e-16:
        s = 0.0
    x0 = 5.0 * w / 18.0
    b = mover_body_points(w)
    xo = x0 - x0 * c
    yo = -x0 * s
    x = c * b[:, 0] - s * b[:, 1] + xo + px
    y = s * b[:, 0] + c * b[:, 1] + yo + py
    return np.stack([x, y], axis=1)


def mover_cull_circle(w: float, px: float, py: float, heading: float) -> Tuple[float, float, float]:
    """Closed form of enclosing_circle for the pentagon: the MRR is the body-axis box
    [-w/2, 3w/2] x [-w/2, w/2]; centre = body (w/2, 0), radius = w*sqrt(5)/2."""
    c, s = math.cos(heading), math.sin(heading)
    x0 = 5.0 * w / 18.0
    dx = w / 2.0 - x0
    return px + x0 + c * dx, py + s * dx, w * math.sqrt(5.0) / 2.0
