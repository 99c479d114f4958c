/* auv_oracle.c — CPU restatement (plain C, fp64) of gym-auv's step() hot path.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker / the timed CPU baseline.  The product path
 * (gym_auv_amd + libauv_hip.so) never links, imports or falls back to it.
 *
 * It follows the reference's algorithm function by function (paths relative to
 * /root/reference/gym_auv/), one environment at a time, in the reference's order of
 * operations; third-party arithmetic that is absent from /root/reference (Shapely 1.7 / GEOS
 * 3.8: intersection, distance, project, minimum_rotated_rectangle, affinity.rotate; SciPy
 * PPoly evaluation) is restated from the published algorithms.
 *
 * Pinning: checked in tests/test_oracle_*.py against golden vectors emitted by the
 * reference's own code (oracle/ref_harness/make_golden.py).  The GEOS primitives in those
 * runs come from a builder-written shim, so GEOS *numerics* are "parity unpinned"; the
 * reference's control flow (culling indices, ordering, reward, done logic) is pinned.
 *
 * Data layout = include/auv_hip.h (same config / world-bank structs, host pointers here).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/auv_hip.h"

#define PI 3.141592653589793

typedef struct oracle {
  auv_config_t cfg;
  int32_t n;
  auv_world_bank_t bank; /* borrowed host pointers (caller keeps them alive) */
  int32_t k_max, m_max;
  /* per-env state */
  double* state;      /* [6][N] */
  int32_t* world_idx; /* [N] */
  int32_t* counters;  /* [N][4] t_step, step_counter, episodes, pad */
  double* lidar_d;    /* [N][S] */
  double* obs64;      /* [N][6+S] */
  double* reward64;   /* [N] */
  double* info64;     /* [N][8] */
  double* nav64;      /* [N][8] unclipped navigation features */
  double* mover;      /* [N][Mmax][4] */
  uint8_t* nearby;    /* [N][Kmax] */
  double* episode;    /* [N][4] */
  int32_t* limits;    /* [N][Kmax][2] */
  uint8_t* collision; /* [N] */
  double* step_info;  /* [N][4] */
} oracle_t;

/* ---------------------------------------------------------------- utils/geomutils.py:4-5 */
static double princip(double a) {
  double m = fmod(a + PI, 2.0 * PI); /* Python float %: result takes the divisor's sign */
  if (m < 0.0) m += 2.0 * PI;
  return m - PI;
}

/* ------------------------------------------------- utils/constants.py:4-43, 63-72 (A.1) */
static const double m_ = 23.8, x_g = 0.046, I_z = 1.760, X_udot = -2.0, Y_vdot = -10.0,
                    Y_rdot = 0.0, N_rdot = -1.0, N_vdot = 0.0, X_u = -2.0, Y_v = -7.0, Y_r = -0.1,
                    N_v = -0.1, N_r = -0.5;

/* objects/vessel/vessel.py:561-570 `_state_dot` */
static void state_dot(const double y[6], double tau_u, double tau_r, double out[6]) {
  double psi = princip(y[2]);
  double c = cos(psi), s = sin(psi);
  double u = y[3], v = y[4], r = y[5];
  /* eta_dot = Rz(psi).dot(nu)   geomutils.py:37-43 */
  out[0] = c * u + -s * v + 0.0 * r;
  out[1] = s * u + c * v + -0.0 * r;
  out[2] = 0.0 * u + 0.0 * v + 1.0 * r;
  /* M, M_inv (constants.py:33-37) */
  double m11 = m_ - X_udot, m22 = m_ - Y_vdot, m23 = m_ * x_g - Y_rdot, m32 = m_ * x_g - N_vdot,
         m33 = I_z - N_rdot;
  double det = m22 * m33 - m23 * m32;
  double i11 = 1.0 / m11, i22 = m33 / det, i23 = -m23 / det, i32 = -m32 / det, i33 = m22 / det;
  /* D.dot(nu)  (constants.py:39-43) */
  double d0 = 2.0 * u + 0.0 * v + 0.0 * r;
  double d1 = 0.0 * u + 7.0 * v + -2.5425 * r;
  double d2 = 0.0 * u + -2.5425 * v + 1.422 * r;
  /* N(nu).dot(nu)  (constants.py:63-72) */
  double n0 = -X_u * u + 0.0 * v + 0.0 * r;
  double n1 = 0.0 * u + -Y_v * v + (m_ * u - Y_r) * r;
  double n2 = 0.0 * u + -N_v * v + (m_ * x_g * u - N_r) * r;
  double r0 = tau_u - d0 - n0, r1 = 0.0 - d1 - n1, r2 = tau_r - d2 - n2;
  out[3] = i11 * r0 + 0.0 * r1 + 0.0 * r2;
  out[4] = 0.0 * r0 + i22 * r1 + i23 * r2;
  out[5] = 0.0 * r0 + i32 * r1 + i33 * r2;
}

/* objects/vessel/odesolver.py:2-47 (returns q, the combination Vessel.step keeps) */
static void odesolver45_q(const double y[6], double h, double tu, double tr, double q[6]) {
  double s1[6], s2[6], s3[6], s4[6], s5[6], s6[6], t[6];
  int i;
  state_dot(y, tu, tr, s1);
  for (i = 0; i < 6; i++) t[i] = y[i] + h * s1[i] / 4.0;
  state_dot(t, tu, tr, s2);
  for (i = 0; i < 6; i++) t[i] = y[i] + 3.0 * h * s1[i] / 32.0 + 9.0 * h * s2[i] / 32.0;
  state_dot(t, tu, tr, s3);
  for (i = 0; i < 6; i++)
    t[i] = y[i] + 1932.0 * h * s1[i] / 2197.0 - 7200.0 * h * s2[i] / 2197.0 + 7296.0 * h * s3[i] / 2197.0;
  state_dot(t, tu, tr, s4);
  for (i = 0; i < 6; i++)
    t[i] = y[i] + 439.0 * h * s1[i] / 216.0 - 8.0 * h * s2[i] + 3680.0 * h * s3[i] / 513.0 -
           845.0 * h * s4[i] / 4104.0;
  state_dot(t, tu, tr, s5);
  for (i = 0; i < 6; i++)
    t[i] = y[i] - 8.0 * h * s1[i] / 27.0 + 2 * h * s2[i] - 3544.0 * h * s3[i] / 2565 +
           1859.0 * h * s4[i] / 4104.0 - 11.0 * h * s5[i] / 40.0;
  state_dot(t, tu, tr, s6);
  for (i = 0; i < 6; i++)
    q[i] = y[i] + h * (16.0 * s1[i] / 135.0 + 6656.0 * s3[i] / 12825.0 + 28561.0 * s4[i] / 56430.0 -
                       9.0 * s5[i] / 50.0 + 2.0 * s6[i] / 55.0);
}

static double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* environment.py:314-315 (NaN guard) + vessel.py:226-247, :572-578 */
static void vessel_step(oracle_t* o, int e, double a0, double a1) {
  int n = o->n, i;
  if (isnan(a0) || isnan(a1)) a0 = a1 = 0.0;
  double tu = clipd(a0, 0.0, 1.0) * o->cfg.thrust_max;
  double tr = clipd(a1, -1.0, 1.0) * o->cfg.moment_max;
  double y[6], q[6];
  for (i = 0; i < 6; i++) y[i] = o->state[(size_t)i * n + e];
  odesolver45_q(y, o->cfg.dt, tu, tr, q);
  q[2] = princip(q[2]);
  for (i = 0; i < 6; i++) o->state[(size_t)i * n + e] = q[i];
  o->counters[4 * e + 1] += 1; /* Vessel._step_counter */
}

/* ---------------------------------------------------------------- geometry primitives */
/* JTS/GEOS Distance::pointToSegment */
static double pt_seg_dist(double px, double py, double ax, double ay, double bx, double by) {
  if (ax == bx && ay == by) return sqrt((px - ax) * (px - ax) + (py - ay) * (py - ay));
  double len2 = (bx - ax) * (bx - ax) + (by - ay) * (by - ay);
  double r = ((px - ax) * (bx - ax) + (py - ay) * (by - ay)) / len2;
  if (r <= 0.0) return sqrt((px - ax) * (px - ax) + (py - ay) * (py - ay));
  if (r >= 1.0) return sqrt((px - bx) * (px - bx) + (py - by) * (py - by));
  double s = ((ay - py) * (bx - ax) - (ax - px) * (by - ay)) / len2;
  return fabs(s) * sqrt(len2);
}

/* boundary segments of obstacle k of the env's world; movers are rebuilt from their state:
 * obstacles.py:217-228 (rotate about centroid, shapely.affinity snaps tiny cos/sin to 0). */
static int obstacle_segments(const oracle_t* o, int e, int64_t kglob, const double** seg_out, double tmp[20]) {
  const int32_t* meta = o->bank.obs_meta + 4 * kglob;
  if (meta[0] != AUV_OBS_MOVER) {
    *seg_out = o->bank.seg + 4 * (int64_t)meta[1];
    return meta[2];
  }
  int w_ = o->world_idx[e];
  int64_t mg = o->bank.mv_off[w_] + meta[3];
  double w = o->bank.mv_param[4 * mg + 0];
  const double* st = o->mover + ((size_t)e * o->m_max + meta[3]) * 4;
  double c = cos(st[2]), s = sin(st[2]);
  if (fabs(c) < 2.5e-16) c = 0.0;
  if (fabs(s) < 2.5e-16) s = 0.0;
  double x0 = 5.0 * w / 18.0; /* centroid of the pentagon, body axes */
  double bx[5] = {-w / 2, -w / 2, w / 2, 3.0 / 2 * w, w / 2};
  double by[5] = {-w / 2, w / 2, w / 2, 0.0, -w / 2};
  double xo = x0 - x0 * c + 0.0 * s, yo = 0.0 - x0 * s - 0.0 * c;
  double vx[5], vy[5];
  for (int i = 0; i < 5; i++) {
    vx[i] = (c * bx[i] + -s * by[i] + xo) + st[0];
    vy[i] = (s * bx[i] + c * by[i] + yo) + st[1];
  }
  for (int i = 0; i < 5; i++) {
    int j = (i + 1) % 5;
    tmp[4 * i + 0] = vx[i];
    tmp[4 * i + 1] = vy[i];
    tmp[4 * i + 2] = vx[j];
    tmp[4 * i + 3] = vy[j];
  }
  *seg_out = tmp;
  return 5;
}

/* obstacles.py:235-262 for a convex polygon given as closed segment list: centre of the
 * min-area rectangle over edge directions (Shapely 1.7 minimum_rotated_rectangle), radius =
 * distance to its farthest corner. */
static void enclosing_circle_convex(const double* seg, int nseg, double out[3]) {
  double best = -1.0, bu = 1, bv = 0, a0 = 0, a1 = 0, b0 = 0, b1 = 0;
  for (int i = 0; i < nseg; i++) {
    double ex = seg[4 * i + 2] - seg[4 * i], ey = seg[4 * i + 3] - seg[4 * i + 1];
    double ln = sqrt(ex * ex + ey * ey);
    double ux = ex / ln, uy = ey / ln;
    double amin = 1e300, amax = -1e300, bmin = 1e300, bmax = -1e300;
    for (int j = 0; j < nseg; j++) {
      double a = seg[4 * j] * ux + seg[4 * j + 1] * uy;
      double b = -seg[4 * j] * uy + seg[4 * j + 1] * ux;
      if (a < amin) amin = a;
      if (a > amax) amax = a;
      if (b < bmin) bmin = b;
      if (b > bmax) bmax = b;
    }
    double area = (amax - amin) * (bmax - bmin);
    if (best < 0.0 || area < best) {
      best = area, bu = ux, bv = uy, a0 = amin, a1 = amax, b0 = bmin, b1 = bmax;
    }
  }
  double ca = 0.5 * (a0 + a1), cb = 0.5 * (b0 + b1);
  out[0] = ca * bu - cb * bv;
  out[1] = ca * bv + cb * bu;
  out[2] = 0.5 * sqrt((a1 - a0) * (a1 - a0) + (b1 - b0) * (b1 - b0));
}

/* point in closed polygon (boundary counts as inside: GEOS distance()==0 there) */
static int point_in_polygon(double px, double py, const double* seg, int nseg) {
  int inside = 0;
  for (int i = 0; i < nseg; i++) {
    double ax = seg[4 * i], ay = seg[4 * i + 1], bx = seg[4 * i + 2], by = seg[4 * i + 3];
    if (pt_seg_dist(px, py, ax, ay, bx, by) == 0.0) return 1;
    if ((ay > py) != (by > py)) {
      double xint = ax + (py - ay) * (bx - ax) / (by - ay);
      if (px < xint) inside = !inside;
    }
  }
  return inside;
}

/* Point.distance(obst.boundary): ring -> distance to the ring; filled polygon -> 0 inside */
static double point_obstacle_distance(double px, double py, int kind, const double* seg, int nseg) {
  if (kind != AUV_OBS_RING && point_in_polygon(px, py, seg, nseg)) return 0.0;
  double d = 1e300;
  for (int i = 0; i < nseg; i++) {
    double t = pt_seg_dist(px, py, seg[4 * i], seg[4 * i + 1], seg[4 * i + 2], seg[4 * i + 3]);
    if (t < d) d = t;
  }
  return d;
}

/* sensor.py:140-159 `simulate_sensor` for ONE obstacle: sector_ray.intersection(boundary) ->
 * distances from p0 to the intersection geometry; returns min distance or +inf.
 * Ring: crossing points.  Filled polygon: first coordinate of the clipped piece(s) = entry
 * point, or p0 itself (distance 0) when p0 is inside. `inside` is precomputed per obstacle. */
static double ray_obstacle(double px, double py, double ex, double ey, int kind, int inside,
                           const double* seg, int nseg) {
  if (kind != AUV_OBS_RING && inside) return 0.0;
  double best = INFINITY;
  double rx = ex - px, ry = ey - py;
  for (int i = 0; i < nseg; i++) {
    double ax = seg[4 * i], ay = seg[4 * i + 1];
    double sx = seg[4 * i + 2] - ax, sy = seg[4 * i + 3] - ay;
    double den = rx * sy - ry * sx;
    if (den == 0.0) continue; /* parallel / collinear: measure zero, ignored */
    double wx = ax - px, wy = ay - py;
    double t = (wx * sy - wy * sx) / den; /* along the ray, 0..1 */
    double u = (wx * ry - wy * rx) / den; /* along the boundary segment, 0..1 */
    if (t >= 0.0 && t <= 1.0 && u >= 0.0 && u <= 1.0) {
      double X = px + t * rx, Y = py + t * ry;
      double d = sqrt((X - px) * (X - px) + (Y - py) * (Y - py));
      if (d < best) best = d;
    }
  }
  return best;
}

static int64_t floordiv_i64(int64_t a, int64_t b) { /* Python // */
  int64_t q = a / b, r = a % b;
  return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}
static int64_t pymod_i64(int64_t a, int64_t b) { return a - floordiv_i64(a, b) * b; }

/* environment.py:386-392 + obstacles.py:195-215 */
static void update_movers(oracle_t* o, int e) {
  int w = o->world_idx[e];
  int64_t m0 = o->bank.mv_off[w], m1 = o->bank.mv_off[w + 1];
  double dt = o->cfg.dt;
  for (int64_t mg = m0; mg < m1; mg++) {
    double* st = o->mover + ((size_t)e * o->m_max + (mg - m0)) * 4;
    const double* par = o->bank.mv_param + 4 * mg;
    int64_t voff = o->bank.mv_vtab_off[mg], vlen = o->bank.mv_vtab_off[mg + 1] - voff;
    st[3] += dt;
    int64_t idx = (int64_t)floor(st[3]);
    if (idx >= (int64_t)par[3] - 1) {
      st[3] = 0.0;
      idx = 0;
      st[0] = par[1];
      st[1] = par[2];
    }
    if (idx > vlen - 1) idx = vlen - 1; /* constant-velocity tables are stored with length 1 */
    double dx = dt * o->bank.mv_vtab[2 * (voff + idx)], dy = dt * o->bank.mv_vtab[2 * (voff + idx) + 1];
    st[2] = atan2(dy, dx);
    st[0] = st[0] + dx;
    st[1] = st[1] + dy;
  }
}

/* vessel.py:249-368 `perceive` (+ :370-428, sensor.py:22-97) */
static void perceive(oracle_t* o, int e) {
  const auv_config_t* c = &o->cfg;
  int S = c->n_sensors, n = o->n;
  double R = c->sensor_range, W = c->vessel_width;
  double px = o->state[0 * (size_t)n + e], py = o->state[1 * (size_t)n + e], psi = o->state[2 * (size_t)n + e];
  int w = o->world_idx[e];
  int64_t k0 = o->bank.obs_off[w], k1 = o->bank.obs_off[w + 1];
  int K = (int)(k1 - k0);
  double* d = o->lidar_d + (size_t)e * S;
  uint8_t* near = o->nearby + (size_t)e * o->k_max;
  int32_t* lim = o->limits + (size_t)e * o->k_max * 2;
  double tmp[20];
  const double* seg;
  double dangle = 2 * PI / S; /* vessel.py:63-65 */

  /* vessel.py:266-273: refresh the nearby list every sensor_interval_load_obstacles steps */
  if (o->counters[4 * e + 1] % c->sensor_interval_load_obstacles == 0) {
    for (int k = 0; k < K; k++) {
      int kind = o->bank.obs_meta[4 * (k0 + k)];
      int nseg = obstacle_segments(o, e, k0 + k, &seg, tmp);
      near[k] = (point_obstacle_distance(px, py, kind, seg, nseg) - W < R) ? 1 : 0;
    }
  }
  for (int k = 0; k < o->k_max; k++) lim[2 * k] = lim[2 * k + 1] = INT32_MIN;
  int any = 0;
  for (int k = 0; k < K; k++) any |= near[k];
  for (int i = 0; i < S; i++) d[i] = R; /* sensor.py:156 default / vessel.py:277-279 */
  if (!any) {
    o->collision[e] = 0;
    return;
  }
  /* sensor.py:74-97: per obstacle, the rays it is appended to */
  for (int k = 0; k < K; k++) {
    if (!near[k]) continue;
    int kind = o->bank.obs_meta[4 * (k0 + k)];
    int nseg = obstacle_segments(o, e, k0 + k, &seg, tmp);
    double cc[3];
    if (kind == AUV_OBS_MOVER) {
      enclosing_circle_convex(seg, nseg, cc); /* obstacles.py:230-233, recomputed each call */
    } else {
      cc[0] = o->bank.obs_cull[3 * (k0 + k)], cc[1] = o->bank.obs_cull[3 * (k0 + k) + 1],
      cc[2] = o->bank.obs_cull[3 * (k0 + k) + 2];
    }
    int64_t start, stop;
    if (c->cull_mode == AUV_CULL_EXACT) {
      start = 0, stop = S;
    } else {
      /* sensor.py:41-71 `_find_limit_angle_rays` */
      double relx = cc[0] - px, rely = cc[1] - py;
      double bearing = atan2(rely, relx) - psi; /* not wrapped */
      double dist = sqrt(relx * relx + rely * rely);
      double safe = dist > 1e-8 ? dist : 1e-8;
      double q = cc[2] / safe;
      double f = (q > 1.0 || q < -1.0 || isnan(q)) ? PI : asin(q); /* arcsin -> NaN -> pi */
      int64_t imin = (int64_t)floor((PI + (bearing - f)) / dangle);
      int64_t imax = (int64_t)ceil((PI + (bearing + f)) / dangle);
      lim[2 * k] = (int32_t)imin, lim[2 * k + 1] = (int32_t)imax;
      start = imin - 1;
      stop = pymod_i64(imax, S); /* sensor.py:93: range(idx_min_ray - 1, idx_max_ray % n_rays) */
    }
    int inside = (kind != AUV_OBS_RING) ? point_in_polygon(px, py, seg, nseg) : 0;
    for (int64_t i = start; i < stop; i++) {
      /* list indexing with a negative index wraps once; the reference raises IndexError
       * below -S (b ~ -2pi while inside the circle) -- we wrap fully instead of raising. */
      int ii = (int)pymod_i64(i, S);
      double ang = (-PI + (ii + 1) * dangle) + psi; /* vessel.py:66-68, :317 */
      double ex = px + cos(ang) * R, ey = py + sin(ang) * R;
      double di = ray_obstacle(px, py, ex, ey, kind, inside, seg, nseg);
      if (di < d[ii]) d[ii] = di;
    }
  }
  int col = 0;
  for (int i = 0; i < S; i++) col |= (d[i] < W); /* vessel.py:359 */
  o->collision[e] = (uint8_t)col;
}

/* SciPy PPoly evaluation of the final PCHIP (path.py:26-27, :61-82) */
static int64_t find_interval(const double* x, int64_t nk, double s) {
  if (!(s >= x[0])) return 0;
  if (s >= x[nk - 1]) return nk - 2;
  int64_t lo = 0, hi = nk - 1;
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) / 2;
    if (s >= x[mid]) lo = mid; else hi = mid;
  }
  return lo;
}
static void path_eval(const oracle_t* o, int w, double s, double xy[2], double dxy[2]) {
  int64_t k0 = o->bank.knot_off[w], nk = o->bank.knot_off[w + 1] - k0;
  const double* x = o->bank.knot_s + k0;
  int64_t i = find_interval(x, nk, s);
  const double* c = o->bank.knot_coef + 8 * (k0 + i);
  double z = s - x[i], z2 = z * z;
  for (int a = 0; a < 2; a++) {
    const double* ca = c + 4 * a;
    xy[a] = ((ca[3] + ca[2] * z) + ca[1] * z2) + ca[0] * (z2 * z);
    dxy[a] = (ca[2] + (2.0 * ca[1]) * z) + (3.0 * ca[0]) * z2;
  }
}

/* path.py:84-93: LineString.project == GEOS LengthIndexOfPoint::indexOfFromStart */
static double project_on_path(const oracle_t* o, int w, double px, double py) {
  int64_t p0 = o->bank.poly_off[w], P = o->bank.poly_off[w + 1] - p0;
  const double* xy = o->bank.poly_xy + 2 * p0;
  const double* cum = o->bank.poly_cum + p0;
  double best = 1.7976931348623157e308;
  int64_t bj = 0;
  for (int64_t j = 0; j + 1 < P; j++) {
    double dd = pt_seg_dist(px, py, xy[2 * j], xy[2 * j + 1], xy[2 * j + 2], xy[2 * j + 3]);
    if (dd < best) best = dd, bj = j;
  }
  double ax = xy[2 * bj], ay = xy[2 * bj + 1], bx = xy[2 * bj + 2], by = xy[2 * bj + 3];
  double dx = bx - ax, dy = by - ay, len2 = dx * dx + dy * dy;
  double seglen = sqrt(len2);
  double pf = (len2 == 0.0) ? 0.0 : ((px - ax) * dx + (py - ay) * dy) / len2;
  if (pf <= 0.0) return cum[bj];
  if (pf <= 1.0) return cum[bj] + pf * seglen;
  return cum[bj] + seglen;
}

/* vessel.py:461-541 `navigate`; writes nav[6] and info64 */
static void navigate(oracle_t* o, int e, double nav[6]) {
  const auv_config_t* c = &o->cfg;
  int n = o->n, w = o->world_idx[e];
  double px = o->state[0 * (size_t)n + e], py = o->state[1 * (size_t)n + e], psi = o->state[2 * (size_t)n + e];
  const double* ws = o->bank.world_scalar + 8 * (size_t)w;
  double L = ws[0];
  double s = project_on_path(o, w, px, py);
  double p[2], dp[2];
  path_eval(o, w, s, p, dp);
  double chi = atan2(dp[1], dp[0]);
  /* Rzyx(0,0,-chi).dot([dx, dy, 0])[1]   (geomutils.py:8-34) */
  double ddx = p[0] - px, ddy = p[1] - py;
  double cte = sin(-chi) * ddx + cos(-chi) * ddy + 0.0;
  double s_t = s + c->look_ahead_distance;
  if (L < s_t) s_t = L; /* min(path.length, ...) */
  double pt[2], dpt[2];
  path_eval(o, w, s_t, pt, dpt);
  double la = princip(atan2(dpt[1], dpt[0]) - psi);
  double he = princip(atan2(pt[1] - py, pt[0] - px) - psi);
  double progress = s / L;
  double* inf = o->info64 + 8 * (size_t)e;
  double maxp = inf[5];
  if (progress > maxp) maxp = progress;
  double gx = ws[1] - px, gy = ws[2] - py;
  double goal = sqrt(gx * gx + gy * gy);
  int reached = (goal <= c->min_goal_distance) || (progress >= c->min_path_progress);
  nav[0] = o->state[3 * (size_t)n + e];
  nav[1] = o->state[4 * (size_t)n + e];
  nav[2] = o->state[5 * (size_t)n + e];
  nav[3] = la;
  nav[4] = he;
  nav[5] = cte / 100;
  inf[1] = reached;
  inf[2] = goal;
  inf[3] = progress;
  inf[5] = maxp;
  inf[6] = s;   /* (inf[7]: running sum of |cross-track error| of the episode, kept by finish_step) */
  double* nv = o->nav64 + 8 * (size_t)e;
  for (int i = 0; i < 6; i++) nv[i] = nav[i];
  nv[6] = chi;
  nv[7] = s_t;
}

/* environment.py:247-290 `observe` */
static void observe(oracle_t* o, int e) {
  const auv_config_t* c = &o->cfg;
  int S = c->n_sensors;
  int D = 6 + (c->use_lidar ? S : 0);
  double nav[6];
  navigate(o, e, nav);
  if (c->use_lidar) perceive(o, e);
  double* ob = o->obs64 + (size_t)e * (6 + S);
  for (int i = 0; i < 6; i++) ob[i] = clipd(nav[i], -1.0, 1.0);
  if (c->use_lidar) {
    const double* d = o->lidar_d + (size_t)e * S;
    double R = c->sensor_range;
    for (int i = 0; i < S; i++) {
      /* vessel.py:88-95 */
      double cl = c->sensor_log_transform ? 1 - clipd(log(1 + d[i]) / log(1 + R), 0, 1)
                                          : 1 - clipd(d[i] / R, 0, 1);
      ob[6 + i] = clipd(cl, -1.0, 1.0);
    }
  }
  (void)D;
  o->info64[8 * (size_t)e + 0] = o->collision[e];
}

/* rewarder.py:167-241 (ColavRewarder.calculate) and :78-140 (PathFollowRewarder.calculate) */
static double reward_calc(const oracle_t* o, int e) {
  const auv_config_t* c = &o->cfg;
  int S = c->n_sensors;
  const double* inf = o->info64 + 8 * (size_t)e;
  const double* nv = o->nav64 + 8 * (size_t)e;
  const double lambda = 0.5, eta = 0.0, gamma_theta = 10.0, gamma_x = 0.1, gamma_v_y = 1.0,
               gamma_y_e = 5.0, penalty_yawrate = 10.0, neutral_speed = 0.05, max_speed = 2.0,
               collision_reward = -10000.0, negative_multiplier = 2.0;
  if (o->collision[e]) return collision_reward * (1 - lambda);
  double u = nv[0], v = nv[1], yaw_rate = nv[2], heading_error = nv[4], cross_track_error = nv[5];
  double speed = sqrt(u * u + v * v); /* linalg.norm(velocity), vessel.py:143-145 */
  double ctp = exp(-gamma_y_e * fabs(cross_track_error));
  double path_reward = (1 + cos(heading_error) * speed / max_speed) * (1 + ctp) - 1;
  double living_penalty = lambda * (2 * neutral_speed + 1) + eta * neutral_speed;
  if (c->rewarder == AUV_REWARD_PATHFOLLOW) {
    double slow_penalty = (speed < 0.1) ? -2 : 0; /* cruise_speed, rewarder.py:119-121 */
    return path_reward - living_penalty + eta * speed / max_speed - penalty_yawrate * fabs(yaw_rate) +
           slow_penalty;
  }
  /* Colav: weighted closeness penalty over all sensors (rewarder.py:192-213); the velocity
   * channel is identically zero (sensor.py:159), so max(0, speed_vec[1]) == 0 */
  double num = 0, den = 0, closeness_reward = 0;
  if (S > 0) {
    const double* d = o->lidar_d + (size_t)e * S;
    double dangle = 2 * PI / S;
    for (int i = 0; i < S; i++) {
      double angle = -PI + (i + 1) * dangle;
      double weight = 1 / (1 + fabs(gamma_theta * angle));
      double raw = c->sensor_range * exp(-gamma_x * d[i] + gamma_v_y * fmax(0.0, 0.0));
      num += weight * raw;
      den += weight;
    }
    closeness_reward = -num / den;
  }
  if (inf[3] < inf[5]) path_reward = fmin(path_reward, 0.0); /* progress < max_progress */
  double slow_penalty = (speed < 0.04) ? -2 : 0;             /* slow_speed / penalty_slow */
  double reward = lambda * path_reward + (1 - lambda) * closeness_reward - living_penalty +
                  eta * speed / max_speed - penalty_yawrate * fabs(yaw_rate) + slow_penalty;
  if (reward < 0) reward *= negative_multiplier;
  return reward;
}

/* environment.py:176-245 `reset` for one env bound to world w */
static void reset_env(oracle_t* o, int e, int w) {
  int n = o->n, S = o->cfg.n_sensors;
  const double* ws = o->bank.world_scalar + 8 * (size_t)w;
  o->world_idx[e] = w;
  o->state[0 * (size_t)n + e] = ws[3];
  o->state[1 * (size_t)n + e] = ws[4];
  o->state[2 * (size_t)n + e] = ws[5];
  for (int i = 3; i < 6; i++) o->state[(size_t)i * n + e] = 0.0; /* vessel.py:199-202 */
  o->counters[4 * e + 0] = 0;                                    /* t_step */
  o->counters[4 * e + 1] = 0;                                    /* Vessel._step_counter */
  double* inf = o->info64 + 8 * (size_t)e;
  for (int i = 0; i < 8; i++) inf[i] = 0.0;
  o->collision[e] = 0;
  for (int i = 0; i < S; i++) o->lidar_d[(size_t)e * S + i] = o->cfg.sensor_range; /* vessel.py:206-208 */
  int64_t m0 = o->bank.mv_off[w], m1 = o->bank.mv_off[w + 1];
  for (int64_t mg = m0; mg < m1; mg++)
    memcpy(o->mover + ((size_t)e * o->m_max + (mg - m0)) * 4, o->bank.mv_init + 4 * mg, 4 * sizeof(double));
  memset(o->nearby + (size_t)e * o->k_max, 0, (size_t)o->k_max);
  observe(o, e);
}

/* ------------------------------------------------------------------------------ API */
static void* zalloc(size_t n) { return calloc(n ? n : 1, 1); }

int oracle_create(const auv_config_t* cfg, int32_t n_envs, oracle_t** out) {
  oracle_t* o = (oracle_t*)zalloc(sizeof(oracle_t));
  o->cfg = *cfg;
  o->n = n_envs;
  *out = o;
  return 0;
}

static void free_env_buffers(oracle_t* o) {
  free(o->state), free(o->world_idx), free(o->counters), free(o->lidar_d), free(o->obs64);
  free(o->reward64), free(o->info64), free(o->nav64), free(o->mover), free(o->nearby);
  free(o->episode), free(o->limits), free(o->collision), free(o->step_info);
}

int oracle_destroy(oracle_t* o) {
  if (!o) return 0;
  free_env_buffers(o);
  free(o);
  return 0;
}

/* The bank's host arrays are borrowed: the caller keeps them alive while the oracle lives. */
int oracle_load_worlds(oracle_t* o, const auv_world_bank_t* bank) {
  int n = o->n, S = o->cfg.n_sensors;
  free_env_buffers(o);
  o->bank = *bank;
  int W = bank->n_worlds;
  o->k_max = 1, o->m_max = 1;
  for (int w = 0; w < W; w++) {
    int K = (int)(bank->obs_off[w + 1] - bank->obs_off[w]), M = (int)(bank->mv_off[w + 1] - bank->mv_off[w]);
    if (K > o->k_max) o->k_max = K;
    if (M > o->m_max) o->m_max = M;
  }
  o->state = (double*)zalloc(sizeof(double) * 6 * n);
  o->world_idx = (int32_t*)zalloc(sizeof(int32_t) * n);
  o->counters = (int32_t*)zalloc(sizeof(int32_t) * 4 * n);
  o->lidar_d = (double*)zalloc(sizeof(double) * (size_t)n * S);
  o->obs64 = (double*)zalloc(sizeof(double) * (size_t)n * (6 + S));
  o->reward64 = (double*)zalloc(sizeof(double) * n);
  o->info64 = (double*)zalloc(sizeof(double) * 8 * n);
  o->nav64 = (double*)zalloc(sizeof(double) * 8 * n);
  o->mover = (double*)zalloc(sizeof(double) * (size_t)n * o->m_max * 4);
  o->nearby = (uint8_t*)zalloc((size_t)n * o->k_max);
  o->episode = (double*)zalloc(sizeof(double) * 4 * n);
  o->limits = (int32_t*)zalloc(sizeof(int32_t) * (size_t)n * o->k_max * 2);
  o->collision = (uint8_t*)zalloc(n);
  o->step_info = (double*)zalloc(sizeof(double) * 4 * n);
  for (int e = 0; e < n; e++) o->world_idx[e] = e % W;
  return 0;
}

int oracle_reset(oracle_t* o, const uint8_t* mask, const int32_t* world_idx) {
  if (!o->state) return AUV_ESTATE;
#pragma omp parallel for schedule(dynamic, 8)
  for (int e = 0; e < o->n; e++) {
    if (mask && !mask[e]) continue;
    reset_env(o, e, world_idx ? world_idx[e] : o->world_idx[e]);
  }
  return 0;
}

int oracle_step_dynamics(oracle_t* o, const double* actions) {
#pragma omp parallel for schedule(static)
  for (int e = 0; e < o->n; e++) vessel_step(o, e, actions[2 * e], actions[2 * e + 1]);
  return 0;
}

int oracle_lidar(oracle_t* o, int32_t advance_movers) {
#pragma omp parallel for schedule(dynamic, 8)
  for (int e = 0; e < o->n; e++) {
    if (advance_movers) update_movers(o, e);
    if (o->cfg.use_lidar) perceive(o, e);
  }
  return 0;
}

/* environment.py:325-347: flags, reward, cumulative reward, done, t_step; VecEnv auto-reset */
static void finish_step(oracle_t* o, int e, uint8_t* done_out) {
  const auv_config_t* c = &o->cfg;
  double* inf = o->info64 + 8 * (size_t)e;
  double reward = reward_calc(o, e);
  o->reward64[e] = reward;
  inf[4] += reward; /* cumulative_reward */
  /* environment.py:345, :460-464 `_save_latest_step`: abs(cross_track_error) * 100 of every step is kept for the
   * episode's mean (save_latest_episode, :466-489); here as a running sum */
  inf[7] += fabs(o->nav64[8 * (size_t)e + 5]) * 100;
  int t_step = o->counters[4 * e];
  int done = o->collision[e] || (inf[1] != 0.0) || (t_step >= c->max_timesteps - 1 && !c->test_mode) ||
             (inf[4] < c->min_cumulative_reward && !c->test_mode); /* environment.py:375-384 */
  o->counters[4 * e] = t_step + 1;
  done_out[e] = (uint8_t)done;
  { /* info dict of this step, environment.py:336-340 */
    double* si = o->step_info + 4 * (size_t)e;
    si[0] = o->collision[e], si[1] = inf[1], si[2] = inf[2], si[3] = inf[3];
  }
  if (done) {
    double* ep = o->episode + 4 * (size_t)e;
    ep[0] = inf[4], ep[1] = t_step + 1, ep[2] = o->collision[e], ep[3] = inf[1];
    o->counters[4 * e + 2] += 1;
    if (c->auto_reset) {
      int W = o->bank.n_worlds;
      reset_env(o, e, (int)(((int64_t)o->world_idx[e] + o->n) % W));
    }
  }
}

/* navigation + observation (+ reward/done); lidar_d must be current.
 * mode 0: full; 1: observe only (reset path); 2: reward/done only from the buffers as they
 * stand (test hook: NAV64 / INFO64 / LIDAR_D / COLLISION injected by the caller). */
int oracle_nav_reward(oracle_t* o, int32_t mode, uint8_t* done_out) {
  int observe_only = (mode == 1);
#pragma omp parallel for schedule(dynamic, 8)
  for (int e = 0; e < o->n; e++) {
    if (mode == 2) {
      finish_step(o, e, done_out);
      continue;
    }
    /* observe() with the LiDAR part already done by oracle_lidar */
    const auv_config_t* c = &o->cfg;
    int S = c->n_sensors;
    double nav[6];
    navigate(o, e, nav);
    double* ob = o->obs64 + (size_t)e * (6 + S);
    for (int i = 0; i < 6; i++) ob[i] = clipd(nav[i], -1.0, 1.0);
    if (c->use_lidar) {
      const double* d = o->lidar_d + (size_t)e * S;
      for (int i = 0; i < S; i++) {
        double cl = c->sensor_log_transform ? 1 - clipd(log(1 + d[i]) / log(1 + c->sensor_range), 0, 1)
                                            : 1 - clipd(d[i] / c->sensor_range, 0, 1);
        ob[6 + i] = clipd(cl, -1.0, 1.0);
      }
    }
    o->info64[8 * (size_t)e] = o->collision[e];
    if (!observe_only) finish_step(o, e, done_out);
  }
  return 0;
}

/* environment.py:292-366 `step` for the whole batch, reference order per env */
int oracle_step(oracle_t* o, const double* actions, uint8_t* done_out) {
  if (!o->state) return AUV_ESTATE;
#pragma omp parallel for schedule(dynamic, 8)
  for (int e = 0; e < o->n; e++) {
    update_movers(o, e);                                   /* environment.py:318 */
    vessel_step(o, e, actions[2 * e], actions[2 * e + 1]); /* :321 */
    observe(o, e);                                         /* :324 */
    finish_step(o, e, done_out);                           /* :325-347 */
  }
  return 0;
}

static void* field_ptr(oracle_t* o, int field, size_t* bytes) {
  size_t n = o->n, S = o->cfg.n_sensors;
  switch (field) {
    case AUV_FIELD_STATE: *bytes = 8 * 6 * n; return o->state;
    case AUV_FIELD_LIDAR_D: *bytes = 8 * n * S; return o->lidar_d;
    case AUV_FIELD_OBS64: *bytes = 8 * n * (6 + S); return o->obs64;
    case AUV_FIELD_REWARD64: *bytes = 8 * n; return o->reward64;
    case AUV_FIELD_INFO64: *bytes = 8 * 8 * n; return o->info64;
    case AUV_FIELD_WORLD_IDX: *bytes = 4 * n; return o->world_idx;
    case AUV_FIELD_COUNTERS: *bytes = 4 * 4 * n; return o->counters;
    case AUV_FIELD_MOVER_STATE: *bytes = 8 * n * o->m_max * 4; return o->mover;
    case AUV_FIELD_NEARBY: *bytes = n * o->k_max; return o->nearby;
    case AUV_FIELD_EPISODE: *bytes = 8 * 4 * n; return o->episode;
    case AUV_FIELD_CULL_LIMITS: *bytes = 4 * n * o->k_max * 2; return o->limits;
    case AUV_FIELD_NAV64: *bytes = 8 * 8 * n; return o->nav64;
    case AUV_FIELD_COLLISION: *bytes = n; return o->collision;
    case AUV_FIELD_STEP_INFO: *bytes = 8 * 4 * n; return o->step_info;
  }
  *bytes = 0;
  return NULL;
}

size_t oracle_field_bytes(oracle_t* o, int32_t field) {
  size_t b;
  field_ptr(o, field, &b);
  return b;
}
int oracle_read(oracle_t* o, int32_t field, void* dst, size_t bytes) {
  size_t b;
  void* p = field_ptr(o, field, &b);
  if (!p || bytes != b) return AUV_EINVAL;
  memcpy(dst, p, b);
  return 0;
}
int oracle_write(oracle_t* o, int32_t field, const void* src, size_t bytes) {
  size_t b;
  void* p = field_ptr(o, field, &b);
  if (!p || bytes != b) return AUV_EINVAL;
  memcpy(p, src, b);
  return 0;
}
int32_t oracle_kmax(const oracle_t* o) { return o->k_max; }
int32_t oracle_mmax(const oracle_t* o) { return o->m_max; }

/* cpu_baseline support: number of OpenMP threads used by the batched loops */
int oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}

/* SURVEY 8(f) F3 -- sensor.py:251-296 `_feasibility_pooling` for one sector, as written:
 * ascending order of range (insertion sort of indices), first range without an opening. */
static double feasibility_pooling_sector(const double* m, int N, double width, double theta) {
  int idx[4096];
  for (int i = 0; i < N; i++) {
    int j = i;
    while (j > 0 && m[idx[j - 1]] > m[i]) idx[j] = idx[j - 1], j--;
    idx[j] = i;
  }
  double mmax = 0.0;
  for (int i = 0; i < N; i++)
    if (m[i] > mmax) mmax = m[i];
  for (int q = 0; q < N; q++) {
    const int id = idx[q];
    const double d = m[id] * theta;
    double opening_width = 0, opening_span = 0, opening_start = -theta * (N - 1) / 2;
    int found_opening = 0;
    for (int isensor = 0; isensor < N; isensor++) {
      const int survives = m[isensor] > m[id] + width;
      if (survives) {
        opening_width += d;
        opening_span += theta;
        if (opening_width > width) {
          const double opening_center = opening_start + opening_span / 2;
          if (fabs(opening_center) < theta * (N - 1) / 4) found_opening = 1;
        }
      } else {
        opening_width += 0.5 * d;
        opening_span += 0.5 * theta;
        if (opening_width > width) {
          const double opening_center = opening_start + opening_span / 2;
          if (fabs(opening_center) < theta * (N - 1) / 4) found_opening = 1;
        }
        opening_width = 0;
        opening_span = 0;
        opening_start = -theta * (N - 1) / 2 + isensor * theta;
      }
    }
    if (!found_opening) return m[id] > 0 ? m[id] : 0;
  }
  return mmax > 0 ? mmax : 0;
}

/* sensor.py:215-238 `preprocess` (distances only): out[N][n_sectors] */
int oracle_feasibility_pooling(oracle_t* o, const int32_t* sector_start, int32_t n_sectors, double width, double* out) {
  int S = o->cfg.n_sensors;
  if (S > 4096) return AUV_EINVAL;
  double theta = 2 * PI / S;
  for (int e = 0; e < o->n; e++)
    for (int k = 0; k < n_sectors; k++)
      out[(size_t)e * n_sectors + k] = feasibility_pooling_sector(o->lidar_d + (size_t)e * S + sector_start[k],
                                                                  sector_start[k + 1] - sector_start[k], width, theta);
  return 0;
}
