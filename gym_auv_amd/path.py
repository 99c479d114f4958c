"""Path construction (reset-time, host side): PCHIP spline re-parameterised three times and
the dense polyline used for nearest-point projection.

Follows /root/reference/gym_auv/objects/path.py:19-40 (`Path.__init__`) and :96-120
(`RandomCurveThroughOrigin`).  The PCHIP itself is SciPy's (`scipy.interpolate.pchip`,
unpinned in the reference's requirements); its published algorithm (Fritsch-Butland
slopes with the three-point end rule, cubic Hermite pieces in the local power basis,
extrapolation from the end intervals) is restated here in NumPy so that the same tables
can be uploaded to HBM and later rebuilt on-device.  tests/test_path.py checks this file
against scipy 1.15.3 and against golden vectors emitted by the reference's own Path.
"""
import numpy as np

N_RESAMPLE = 1000      # path.py:29-31


def _edge_slope(h0, h1, m0, m1):
    d = ((2.0 * h0 + h1) * m0 - h0 * m1) / (h0 + h1)
    if np.sign(d) != np.sign(m0):
        return 0.0
    if np.sign(m0) != np.sign(m1) and abs(d) > 3.0 * abs(m0):
        return 3.0 * m0
    return d


def pchip_slopes(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """Derivatives d_k of the shape-preserving cubic through (x_k, y_k)."""
    h = x[1:] - x[:-1]
    m = (y[1:] - y[:-1]) / h
    if len(x) == 2:
        return np.array([m[0], m[0]])
    sm = np.sign(m)
    flat = (sm[1:] != sm[:-1]) | (m[1:] == 0) | (m[:-1] == 0)
    w1 = 2.0 * h[1:] + h[:-1]
    w2 = h[1:] + 2.0 * h[:-1]
    with np.errstate(divide="ignore", invalid="ignore"):
        whmean = (w1 / m[:-1] + w2 / m[1:]) / (w1 + w2)
    d = np.zeros_like(y)
    inner = np.zeros(len(x) - 2)
    inner[~flat] = 1.0 / whmean[~flat]
    d[1:-1] = inner
    d[0] = _edge_slope(h[0], h[1], m[0], m[1])
    d[-1] = _edge_slope(h[-1], h[-2], m[-1], m[-2])
    return d


def hermite_coefs(x: np.ndarray, y: np.ndarray, d: np.ndarray) -> np.ndarray:
    """Per-interval power-basis coefficients c[0..3] (highest power first) of
    p_i(s) = c0 (s-x_i)^3 + c1 (s-x_i)^2 + c2 (s-x_i) + c3."""
    h = x[1:] - x[:-1]
    slope = (y[1:] - y[:-1]) / h
    t = (d[:-1] + d[1:] - 2.0 * slope) / h
    c = np.empty((4, len(x) - 1))
    c[0] = t / h
    c[1] = (slope - d[:-1]) / h - t
    c[2] = d[:-1]
    c[3] = y[:-1]
    return c


def find_interval(x: np.ndarray, s):
    """Index i with x[i] <= s < x[i+1]; below/above the range -> first/last interval."""
    i = np.searchsorted(x, s, side="right") - 1
    return np.clip(i, 0, len(x) - 2)


def ppoly_eval(x: np.ndarray, c: np.ndarray, s, nu: int = 0):
    """Evaluate (nu=0) or differentiate once (nu=1), term order as in SciPy's evaluator:
    lowest power first with z accumulated by repeated multiplication."""
    s = np.asarray(s, dtype=np.float64)
    i = find_interval(x, s)
    z = s - x[i]
    if nu == 0:
        res = c[3, i] + c[2, i] * z
        z2 = z * z
        res = res + c[1, i] * z2
        res = res + c[0, i] * (z2 * z)
        return res
    res = c[2, i] + (2.0 * c[1, i]) * z
    res = res + (3.0 * c[0, i]) * (z * z)
    return res


class Path:
    """Same public surface as the reference Path (length, start, end, points, __call__,
    get_direction, get_closest_arclength) plus the raw tables the device needs."""

    def __init__(self, waypoints) -> None:
        wp = np.asarray(waypoints, dtype=np.float64)
        self.init_waypoints = wp.copy()
        for _ in range(3):
            diff = np.diff(wp, axis=1)
            s = np.concatenate([[0.0], np.cumsum(np.sqrt(np.sum(diff ** 2, axis=0)))])
            cx = hermite_coefs(s, wp[0], pchip_slopes(s, wp[0]))
            cy = hermite_coefs(s, wp[1], pchip_slopes(s, wp[1]))
            q = np.linspace(s[0], s[-1], N_RESAMPLE)
            wp = np.vstack([ppoly_eval(s, cx, q), ppoly_eval(s, cy, q)])
        self.knot_s, self.cx, self.cy = s, cx, cy
        self._waypoints = wp
        S = np.linspace(0, self.length, int(10 * self.length))
        self._points = np.stack([ppoly_eval(s, cx, S), ppoly_eval(s, cy, S)], axis=1)
        seg = self._points[1:] - self._points[:-1]
        # sequential sum of segment lengths == the measure GEOS walks in project()
        self._cum = np.concatenate([[0.0], np.cumsum(np.sqrt(seg[:, 0] * seg[:, 0] + seg[:, 1] * seg[:, 1]))])

    @property
    def length(self) -> float:
        return float(self.knot_s[-1])

    @property
    def start(self) -> np.ndarray:
        return self(0.0)

    @property
    def end(self) -> np.ndarray:
        return self(self.length)

    @property
    def points(self) -> np.ndarray:
        return self._points

    def __call__(self, arclength) -> np.ndarray:
        return np.array([ppoly_eval(self.knot_s, self.cx, arclength),
                         ppoly_eval(self.knot_s, self.cy, arclength)])

    def get_direction(self, arclength) -> float:
        return np.arctan2(ppoly_eval(self.knot_s, self.cy, arclength, 1),
                          ppoly_eval(self.knot_s, self.cx, arclength, 1))

    def get_closest_arclength(self, position) -> float:
        """LineString(points).project(position): first strict minimum of the point-segment
        distance, measure = cumulative length + clamped projection (host-side convenience;
        the per-step version lives in the HIP kernel)."""
        p = self._points
        px, py = float(position[0]), float(position[1])
        ax, ay, bx, by = p[:-1, 0], p[:-1, 1], p[1:, 0], p[1:, 1]
        dx, dy = bx - ax, by - ay
        len2 = dx * dx + dy * dy
        with np.errstate(divide="ignore", invalid="ignore"):
            r = ((px - ax) * dx + (py - ay) * dy) / len2
            sc = ((ay - py) * dx - (ax - px) * dy) / len2
        da = np.sqrt((px - ax) ** 2 + (py - ay) ** 2)
        db = np.sqrt((px - bx) ** 2 + (py - by) ** 2)
        dist = np.where(r <= 0.0, da, np.where(r >= 1.0, db, np.abs(sc) * np.sqrt(len2)))
        dist = np.where(len2 == 0.0, da, dist)
        j = int(np.argmin(dist))
        seglen = np.sqrt(len2[j])
        pf = 0.0 if len2[j] == 0.0 else ((px - ax[j]) * dx[j] + (py - ay[j]) * dy[j]) / len2[j]
        if pf <= 0.0:
            return float(self._cum[j])
        if pf <= 1.0:
            return float(self._cum[j] + pf * seglen)
        return float(self._cum[j] + seglen)


class RandomCurveThroughOrigin(Path):
    """path.py:96-120: 5 or 7 waypoints symmetric about the origin, drawn from `rng`
    (a RandomState-like object with .rand())."""

    def __init__(self, rng, nwaypoints, length=400):
        theta0 = 2 * np.pi * (rng.rand() - 0.5)
        start = 0.5 * length * np.array([np.cos(theta0), np.sin(theta0)])
        end = -start
        half = nwaypoints // 2
        head, tail = [], []
        # draw order matters for stream parity: (towards-start, towards-end) per ring k;
        # the scalar jitter is added to BOTH coordinates (path.py:107-111)
        for k in range(half):
            jit1 = length / (half + 1) * (rng.rand() - 0.5)
            jit2 = length / (half + 1) * (rng.rand() - 0.5)
            head.append((half - k) * start / (half + 1) + jit1)   # same op order as the reference
            tail.insert(0, (half - k) * end / (half + 1) + jit2)
        pts = [start] + head + [np.zeros(2)] + tail + [end]
        super().__init__(np.array(pts).T)
