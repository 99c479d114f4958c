"""Minimal planar-geometry stand-in for the slice of Shapely 1.7 / GEOS 3.8 that
gym-auv's step() path calls.  TEST TOOLING ONLY (oracle harness): it exists so
the reference's own Python modules can execute in a container without
Shapely/GEOS, in order to emit golden vectors.  Nothing in the product imports it.

The algorithms below restate *published* GEOS/JTS/Shapely behaviour:
  * Point.buffer(r)            GEOS OffsetCurveBuilder circle: 16 segs/quadrant,
                               first vertex (x+r, y), clockwise.
  * simplify(preserve_topology=False)
                               GEOS DouglasPeuckerLineSimplifier (first strict max
                               splits; section dropped when max distance <= tol).
  * LineString.project         GEOS LengthIndexOfPoint::indexOfFromStart (first
                               strict minimum of point-segment distance wins).
  * minimum_rotated_rectangle  Shapely 1.7's pure-Python search over hull edges.
  * line n polygon             overlay result keeps the input line's direction, so
                               coords[0] of a clipped piece is the entry point.
GEOS itself is not available here, so numeric parity with real GEOS is UNPINNED
(see DESIGN.md); control flow of the reference is what these fixtures pin.
"""
import math
import numpy as np


def _xy(c):
    if isinstance(c, Point):
        return (c.x, c.y)
    c = np.asarray(c, dtype=np.float64).reshape(-1)
    return (float(c[0]), float(c[1]))


class _CoordSeq(list):
    """list of (x, y) tuples; behaves like shapely's coords for indexing/iteration."""
    pass


class BaseGeometry:
    is_valid = True

    @property
    def is_empty(self):
        return False


class GeometryCollection(BaseGeometry):
    def __init__(self, geoms=()):
        self.geoms = list(geoms)

    @property
    def is_empty(self):
        return len(self.geoms) == 0


class MultiPoint(GeometryCollection):
    pass


class MultiLineString(GeometryCollection):
    pass


class Point(BaseGeometry):
    def __init__(self, *args):
        if len(args) == 1:
            self.x, self.y = _xy(args[0])
        else:
            self.x, self.y = float(args[0]), float(args[1])

    @property
    def coords(self):
        return _CoordSeq([(self.x, self.y)])

    def __array__(self, dtype=None, copy=None):
        return np.array([self.x, self.y], dtype=np.float64)

    def __iter__(self):
        # shapely 1.7 exposes the array interface; np.array(Point) gives [x, y]
        return iter((self.x, self.y))

    def __len__(self):
        return 2

    def buffer(self, distance, resolution=16):
        # GEOS OffsetCurveBuilder::createCircle + addFillet(p, 0, 2pi, CLOCKWISE, r)
        n_segs = int(2.0 * math.pi / (math.pi / 2.0 / resolution) + 0.5)
        inc = 2.0 * math.pi / n_segs
        pts = [(self.x + distance, self.y)]
        curr = 0.0
        total = 2.0 * math.pi
        k = 0
        while curr < total and k < n_segs:
            ang = -curr
            p = (self.x + distance * math.cos(ang), self.y + distance * math.sin(ang))
            if k > 0:
                pts.append(p)
            curr += inc
            k += 1
        pts.append(pts[0])
        return Polygon(pts)

    def distance(self, other):
        if isinstance(other, Point):
            return math.sqrt((self.x - other.x) ** 2 + (self.y - other.y) ** 2)
        if isinstance(other, Polygon):
            if other.contains_xy(self.x, self.y):
                return 0.0
            return self.distance(other.boundary)
        if isinstance(other, LineString):
            return other._dist_to_xy(self.x, self.y)
        if isinstance(other, GeometryCollection):
            return min(self.distance(g) for g in other.geoms)
        raise TypeError(type(other))

    @property
    def centroid(self):
        return self


def _pt_seg_dist(px, py, ax, ay, bx, by):
    # JTS/GEOS Distance::pointToSegment
    if ax == bx and ay == by:
        return math.sqrt((px - ax) ** 2 + (py - ay) ** 2)
    len2 = (bx - ax) * (bx - ax) + (by - ay) * (by - ay)
    r = ((px - ax) * (bx - ax) + (py - ay) * (by - ay)) / len2
    if r <= 0.0:
        return math.sqrt((px - ax) ** 2 + (py - ay) ** 2)
    if r >= 1.0:
        return math.sqrt((px - bx) ** 2 + (py - by) ** 2)
    s = ((ay - py) * (bx - ax) - (ax - px) * (by - ay)) / len2
    return abs(s) * math.sqrt(len2)


class LineString(BaseGeometry):
    def __init__(self, coords):
        self._c = [_xy(c) for c in coords]

    @property
    def coords(self):
        return _CoordSeq(self._c)

    @property
    def is_empty(self):
        return len(self._c) == 0

    @property
    def length(self):
        c = self._c
        return sum(math.sqrt((c[i + 1][0] - c[i][0]) ** 2 + (c[i + 1][1] - c[i][1]) ** 2)
                   for i in range(len(c) - 1))

    def _dist_to_xy(self, px, py):
        c = self._c
        return min(_pt_seg_dist(px, py, c[i][0], c[i][1], c[i + 1][0], c[i + 1][1])
                   for i in range(len(c) - 1))

    def distance(self, other):
        if isinstance(other, Point):
            return self._dist_to_xy(other.x, other.y)
        raise TypeError(type(other))

    def project(self, point):
        # GEOS LengthIndexOfPoint::indexOfFromStart(pt, -1): vectorised, first strict min.
        a = np.asarray(self._c, dtype=np.float64)
        px, py = point.x, point.y
        ax, ay = a[:-1, 0], a[:-1, 1]
        bx, by = a[1:, 0], a[1:, 1]
        dx, dy = bx - ax, by - ay
        len2 = dx * dx + dy * dy
        seglen = np.sqrt(len2)
        with np.errstate(divide="ignore", invalid="ignore"):
            r = ((px - ax) * dx + (py - ay) * dy) / len2
            s = ((ay - py) * dx - (ax - px) * dy) / len2
        da = np.sqrt((px - ax) ** 2 + (py - ay) ** 2)
        db = np.sqrt((px - bx) ** 2 + (py - by) ** 2)
        dist = np.where(r <= 0.0, da, np.where(r >= 1.0, db, np.abs(s) * seglen))
        dist = np.where(len2 == 0.0, da, dist)
        j = int(np.argmin(dist))  # numpy argmin returns the FIRST minimum
        start = np.concatenate([[0.0], np.cumsum(seglen)])  # sequential sum, like GEOS
        # LineSegment::projectionFactor + segmentNearestMeasure
        if len2[j] == 0.0:
            pf = 0.0
        else:
            pf = ((px - ax[j]) * dx[j] + (py - ay[j]) * dy[j]) / len2[j]
        if pf <= 0.0:
            return float(start[j])
        if pf <= 1.0:
            return float(start[j] + pf * seglen[j])
        return float(start[j] + seglen[j])

    def simplify(self, tolerance, preserve_topology=True):
        assert not preserve_topology, "only the Douglas-Peucker path is restated"
        c = self._c
        n = len(c)
        keep = [True] * n

        def section(i, j):
            if i + 1 == j:
                return
            max_d, max_k = -1.0, i
            for k in range(i + 1, j):
                d = _pt_seg_dist(c[k][0], c[k][1], c[i][0], c[i][1], c[j][0], c[j][1])
                if d > max_d:
                    max_d, max_k = d, k
            if max_d <= tolerance:
                for k in range(i + 1, j):
                    keep[k] = False
            else:
                section(i, max_k)
                section(max_k, j)

        section(0, n - 1)
        return LineString([c[k] for k in range(n) if keep[k]])

    # ---- ray/segment n other geometry -------------------------------------------------
    def intersection(self, other):
        assert len(self._c) == 2, "harness only intersects 2-point sensor rays"
        (x0, y0), (x1, y1) = self._c
        dx, dy = x1 - x0, y1 - y0
        if isinstance(other, Polygon):
            ring = other._ring
            us = _crossings(x0, y0, dx, dy, ring)
            cuts = sorted(set([0.0, 1.0] + [u for u in us if 0.0 <= u <= 1.0]))
            pieces = []
            for a, b in zip(cuts[:-1], cuts[1:]):
                um = 0.5 * (a + b)
                if other.contains_xy(x0 + um * dx, y0 + um * dy):
                    if pieces and pieces[-1][1] == a:
                        pieces[-1][1] = b
                    else:
                        pieces.append([a, b])
            geoms = [LineString([(x0 + a * dx, y0 + a * dy), (x0 + b * dx, y0 + b * dy)])
                     for a, b in pieces]
            # isolated touches (a crossing parameter not on any kept piece)
            for u in cuts[1:-1]:
                if not any(a <= u <= b for a, b in pieces):
                    geoms.append(Point(x0 + u * dx, y0 + u * dy))
            if not geoms:
                return GeometryCollection()
            if len(geoms) == 1:
                return geoms[0]
            if all(isinstance(g, LineString) for g in geoms):
                return MultiLineString(geoms)
            return GeometryCollection(geoms)
        if isinstance(other, LineString):
            us = sorted(set(u for u in _crossings(x0, y0, dx, dy, other._c) if 0.0 <= u <= 1.0))
            pts = [Point(x0 + u * dx, y0 + u * dy) for u in us]
            if not pts:
                return GeometryCollection()
            if len(pts) == 1:
                return pts[0]
            return MultiPoint(pts)
        raise TypeError(type(other))

    @property
    def centroid(self):
        c = np.asarray(self._c)
        seg = np.sqrt(((c[1:] - c[:-1]) ** 2).sum(axis=1))
        mid = 0.5 * (c[1:] + c[:-1])
        w = (mid * seg[:, None]).sum(axis=0) / seg.sum()
        return Point(w[0], w[1])


def _crossings(x0, y0, dx, dy, chain):
    """Parameters u in ray coordinates (point = p0 + u*(dx,dy)) where the ray's supporting
    line meets each segment of `chain` within the segment (0<=t<=1). Collinear overlaps are
    ignored (measure zero for the fixtures generated here)."""
    out = []
    for i in range(len(chain) - 1):
        ax, ay = chain[i]
        bx, by = chain[i + 1]
        ex, ey = bx - ax, by - ay
        den = dx * ey - dy * ex
        if den == 0.0:
            continue
        wx, wy = ax - x0, ay - y0
        u = (wx * ey - wy * ex) / den
        t = (wx * dy - wy * dx) / den
        if 0.0 <= t <= 1.0:
            out.append(u)
    return out


class _Ring:
    def __init__(self, coords):
        self.coords = _CoordSeq(coords)


class Polygon(BaseGeometry):
    def __init__(self, shell):
        pts = [_xy(c) for c in shell]
        if pts[0] != pts[-1]:
            pts.append(pts[0])
        self._ring = pts

    @property
    def exterior(self):
        return _Ring(self._ring)

    @property
    def boundary(self):
        return LineString(self._ring)

    @property
    def area(self):
        r = self._ring
        return abs(0.5 * sum(r[i][0] * r[i + 1][1] - r[i + 1][0] * r[i][1]
                             for i in range(len(r) - 1)))

    @property
    def centroid(self):
        r = self._ring
        a2 = cx = cy = 0.0
        for i in range(len(r) - 1):
            cr = r[i][0] * r[i + 1][1] - r[i + 1][0] * r[i][1]
            a2 += cr
            cx += (r[i][0] + r[i + 1][0]) * cr
            cy += (r[i][1] + r[i + 1][1]) * cr
        return Point(cx / (3.0 * a2), cy / (3.0 * a2))

    def contains_xy(self, px, py):
        # even-odd ray crossing; boundary points count as inside (closed set, like GEOS
        # intersects/distance==0 semantics)
        r = self._ring
        inside = False
        for i in range(len(r) - 1):
            ax, ay = r[i]
            bx, by = r[i + 1]
            if _pt_seg_dist(px, py, ax, ay, bx, by) == 0.0:
                return True
            if (ay > py) != (by > py):
                xint = ax + (py - ay) * (bx - ax) / (by - ay)
                if px < xint:
                    inside = not inside
        return inside

    def distance(self, other):
        if isinstance(other, Point):
            return other.distance(self)
        raise TypeError(type(other))

    def buffer(self, distance, resolution=16):
        if distance == 0:
            return self
        raise NotImplementedError("polygon buffering is outside the harness' slice")

    @property
    def convex_hull(self):
        pts = sorted(set(self._ring))
        if len(pts) <= 2:
            return LineString(pts)

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
        return Polygon(lower[:-1] + upper[:-1])

    @property
    def envelope(self):
        xs = [p[0] for p in self._ring]
        ys = [p[1] for p in self._ring]
        x0, x1, y0, y1 = min(xs), max(xs), min(ys), max(ys)
        return Polygon([(x0, y0), (x1, y0), (x1, y1), (x0, y1)])

    @property
    def minimum_rotated_rectangle(self):
        # Shapely 1.7 (pure Python): min-area axis-aligned box over hull-edge frames,
        # first minimum wins.
        from .affinity import affine_transform
        hull = self.convex_hull
        if not isinstance(hull, Polygon):
            return hull
        coords = hull.exterior.coords
        best = None
        for p1, p2 in zip(coords[:-1], coords[1:]):
            ex, ey = p2[0] - p1[0], p2[1] - p1[1]
            length = math.sqrt(ex * ex + ey * ey)
            ux, uy = ex / length, ey / length
            vx, vy = -uy, ux
            rect = affine_transform(hull, (ux, uy, vx, vy, 0, 0)).envelope
            if best is None or rect.area < best[0].area:
                best = (rect, (ux, vx, uy, vy, 0, 0))
        return affine_transform(best[0], best[1])
