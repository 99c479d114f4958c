// K5 — on-device scenario generation: one 256-thread workgroup builds one
// MovingObstacles-type world from a row of random draws, straight into its fixed-capacity slot
// of the world bank in HBM.  Reset-time work, not part of step().
//
// Reference: MovingObstacles._generate         gym_auv/envs/movingobstacles.py:28-95
//            RandomCurveThroughOrigin, Path     objects/path.py:19-40, :96-120
//                (three PCHIP re-parameterisations through 1000 resampled points, then the
//                 dense polyline of int(10 L) vertices; SciPy pchip = Fritsch-Butland slopes,
//                 cubic Hermite pieces in the local power basis)
//            helpers.generate_obstacle          utils/helpers.py:5-35 (candidate pool instead of
//                                               the unbounded rejection loop, gym_auv_amd/devgen.py)
//            VesselObstacle / CircularObstacle  objects/obstacles.py:90-113, :144-215
// Host mirror consuming the same draws: gym_auv_amd.devgen.world_from_draws (tests compare the
// tables built here with build_world() of that).
#include "auv_device.h"

#include "auv_generate.h"

#define GEN_EXTRA_CAND 56   // devgen.EXTRA_CAND

namespace {

__device__ __forceinline__ unsigned long long gen_splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}


__device__ __forceinline__ double edge_slope(double h0, double h1, double m0, double m1) {
  double d = ((2.0 * h0 + h1) * m0 - h0 * m1) / (h0 + h1);
  const double sd = (d > 0) - (d < 0), s0 = (m0 > 0) - (m0 < 0), s1 = (m1 > 0) - (m1 < 0);
  if (sd != s0) return 0.0;
  if (s0 != s1 && fabs(d) > 3.0 * fabs(m0)) return 3.0 * m0;
  return d;
}

// PCHIP through (x[i], y[i]), i < n (n >= 3): slope at knot i
__device__ __forceinline__ double pchip_slope(const double* x, const double* y, int n, int i) {
  if (i == 0) return edge_slope(x[1] - x[0], x[2] - x[1], (y[1] - y[0]) / (x[1] - x[0]), (y[2] - y[1]) / (x[2] - x[1]));
  if (i == n - 1)
    return edge_slope(x[n - 1] - x[n - 2], x[n - 2] - x[n - 3], (y[n - 1] - y[n - 2]) / (x[n - 1] - x[n - 2]),
                      (y[n - 2] - y[n - 3]) / (x[n - 2] - x[n - 3]));
  const double hm = x[i] - x[i - 1], hp = x[i + 1] - x[i];
  const double mm = (y[i] - y[i - 1]) / hm, mp = (y[i + 1] - y[i]) / hp;
  const double sm = (mm > 0) - (mm < 0), sp = (mp > 0) - (mp < 0);
  if (sm != sp || mp == 0.0 || mm == 0.0) return 0.0;
  const double w1 = 2.0 * hp + hm, w2 = hp + 2.0 * hm;
  return 1.0 / ((w1 / mm + w2 / mp) / (w1 + w2));
}

// power-basis coefficients of interval i (highest power first), CubicHermiteSpline layout
__device__ __forceinline__ void hermite(const double* x, const double* y, double d0, double d1, int i, double c[4]) {
  const double h = x[i + 1] - x[i];
  const double slope = (y[i + 1] - y[i]) / h;
  const double t = (d0 + d1 - 2.0 * slope) / h;
  c[0] = t / h;
  c[1] = (slope - d0) / h - t;
  c[2] = d0;
  c[3] = y[i];
}

__device__ __forceinline__ int find_interval(const double* x, int nk, double s) {
  if (!(s >= x[0])) return 0;
  if (s >= x[nk - 1]) return nk - 2;
  int lo = 0, hi = nk - 1;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (s >= x[mid]) lo = mid; else hi = mid;
  }
  return lo;
}

// SciPy PPoly evaluation of (x(s), y(s)) and derivatives; coef rows are [8] = x:c0..c3, y:c0..c3
__device__ __forceinline__ void eval_xy(const double* ks, const double* coef, int nk, double s, double xy[2], double dxy[2]) {
  const int i = find_interval(ks, nk, s);
  const double* c = coef + 8 * (size_t)i;
  const double z = s - ks[i], z2 = z * z;
#pragma unroll
  for (int a = 0; a < 2; a++) {
    const double* ca = c + 4 * a;
    xy[a] = ((ca[3] + ca[2] * z) + ca[1] * z2) + ca[0] * (z2 * z);
    dxy[a] = (ca[2] + (2.0 * ca[1]) * z) + (3.0 * ca[0]) * z2;
  }
}

__device__ __forceinline__ double linspace_at(double a, double b, int n, int k) {
  if (k == n - 1) return b;
  return k * ((b - a) / (n - 1)) + a;
}

// out[0] = 0, out[i+1] = out[i] + len(i) for i < n (block-wide; chunked per thread).  256 threads: the 256 chunk sums are
// turned into their exclusive prefixes by ONE wave -- four consecutive sums per lane, a DPP scan over the 64 lane totals --
// instead of one thread walking all 256 through LDS (25 us of every world's ~265, three times per world).
template <typename LenFn>
__device__ __forceinline__ void block_cumsum(double* out, int n, LenFn len, double* s_part) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int chunk = (n + nt - 1) / nt;
  const int i0 = tid * chunk, i1 = min(n, i0 + chunk);
  double acc = 0.0;
  for (int i = i0; i < i1; i++) acc += len(i);
  s_part[tid] = acc;
  __syncthreads();
  if (tid < AUV_WAVE) {
    const double a0 = s_part[4 * tid], a1 = s_part[4 * tid + 1], a2 = s_part[4 * tid + 2], a3 = s_part[4 * tid + 3];
    const double p0 = a0, p1 = p0 + a1, p2 = p1 + a2, p3 = p2 + a3;
    const double incl = auv_wave_scan_incl_f64(p3);
    double excl = __shfl_up(incl, 1, AUV_WAVE);
    if (tid == 0) excl = 0.0;
    s_part[4 * tid] = excl, s_part[4 * tid + 1] = excl + p0, s_part[4 * tid + 2] = excl + p1, s_part[4 * tid + 3] = excl + p2;
  }
  __syncthreads();
  double run = s_part[tid];
  if (tid == 0) out[0] = 0.0;
  for (int i = i0; i < i1; i++) {
    run += len(i);
    out[i + 1] = run;
  }
  __syncthreads();
}

// one PCHIP re-parameterisation pass (path.py:24-31): from GEN_NK points (wx, wy) build knots ks (chord arclengths) and
// coefficient rows coef; optionally resample to (wx, wy) in place.  ks, wx, wy, dsx, dsy live in LDS (the interval search of
// every evaluation walks the knots: ten dependent reads that were ten trips to L2 while they sat in global scratch -- the
// 10 000 evaluations of the dense polyline were 3/4 of a world's build time); the coefficient rows stay in global memory.
__device__ __forceinline__ void pchip_pass(double* wx, double* wy, double* ks, double* coef, double* dsx, double* dsy, const bool resample,
                                           double* s_part) {
  const int tid = threadIdx.x, nt = blockDim.x;
  block_cumsum(ks, GEN_NK - 1, [&](int i) {
    const double dx = wx[i + 1] - wx[i], dy = wy[i + 1] - wy[i];
    return sqrt(dx * dx + dy * dy);
  }, s_part);
  for (int i = tid; i < GEN_NK; i += nt) {
    dsx[i] = pchip_slope(ks, wx, GEN_NK, i);
    dsy[i] = pchip_slope(ks, wy, GEN_NK, i);
  }
  __syncthreads();
  for (int i = tid; i < GEN_NK; i += nt) {
    double* c = coef + 8 * (size_t)i;
    if (i < GEN_NK - 1) {
      hermite(ks, wx, dsx[i], dsx[i + 1], i, c);
      hermite(ks, wy, dsy[i], dsy[i + 1], i, c + 4);
    } else {
      for (int a = 0; a < 8; a++) c[a] = 0.0;
    }
  }
  __syncthreads();
  if (resample) {
    // (in place: the evaluation reads knots and coefficient rows only)
    for (int k = tid; k < GEN_NK; k += nt) {
      double xy[2], dxy[2];
      eval_xy(ks, coef, GEN_NK, linspace_at(ks[0], ks[GEN_NK - 1], GEN_NK, k), xy, dxy);
      wx[k] = xy[0], wy[k] = xy[1];
    }
    __syncthreads();
  }
}

// `slots` / `count_dev` (both or neither): build the worlds of rows 0 .. *count_dev - 1 of `draws` into the bank slots
// slots[row] -- the refill pass of the fresh-world mode, which learns on the device which slots are stale.  Otherwise rows
// 0 .. n_worlds - 1 go to the slots w_first + row (whole-bank generation).
__global__ void __launch_bounds__(256) k5_generate(GenOut g, const double* __restrict__ draws, int w_first, int n_worlds,
                                                   const int32_t* __restrict__ slots, const int32_t* __restrict__ count_dev) {
  __shared__ double s_part[256];
  __shared__ double s_ks[GEN_NK], s_wx[GEN_NK], s_wy[GEN_NK], s_dx[GEN_NK], s_dy[GEN_NK];   // 40 KB: knots, points, slopes of the pass in hand
  __shared__ double s_wp[2][8];      // raw waypoints
  __shared__ double s_s1[8], s_d1[2][8], s_c1[2][8][4];
  __shared__ double s_pose[3], s_goal[2], s_L;
  __shared__ int s_n1, s_P;
  const int tid = threadIdx.x, nt = blockDim.x;
  const double PI = AUV_PI;
  double* cfA = g.scratch + (size_t)blockIdx.x * GEN_SCRATCH;   // [NK][8] coefficient rows of the second pass
  double* const wx = s_wx;
  double* const wy = s_wy;
  const int K = g.n_moving + g.n_static;
  if (count_dev) n_worlds = *count_dev;
  for (int wi = blockIdx.x; wi < n_worlds; wi += gridDim.x) {
    const int w = slots ? slots[wi] : w_first + wi;
    const double* row = draws + (size_t)wi * g.n_draws;
    // ---- waypoints: RandomCurveThroughOrigin (path.py:96-120) ----
    if (tid == 0) {
      const int nwp = (int)floor(4 * row[0] + 2);
      const double length = 800.0;
      const double theta0 = 2 * PI * (row[1] - 0.5);
      const double sx = 0.5 * length * cos(theta0), sy = 0.5 * length * sin(theta0);
      const int half = nwp / 2;
      const int n1 = 2 * half + 3;
      s_wp[0][0] = sx, s_wp[1][0] = sy;
      s_wp[0][n1 - 1] = -sx, s_wp[1][n1 - 1] = -sy;
      s_wp[0][half + 1] = 0.0, s_wp[1][half + 1] = 0.0;
      for (int k = 0; k < half; k++) {
        const double j1 = length / (half + 1) * (row[2 + 2 * k] - 0.5), j2 = length / (half + 1) * (row[3 + 2 * k] - 0.5);
        s_wp[0][1 + k] = (half - k) * sx / (half + 1) + j1;
        s_wp[1][1 + k] = (half - k) * sy / (half + 1) + j1;
        s_wp[0][n1 - 2 - k] = (half - k) * (-sx) / (half + 1) + j2;
        s_wp[1][n1 - 2 - k] = (half - k) * (-sy) / (half + 1) + j2;
      }
      // pass 1 on the raw waypoints (serial: <= 7 points)
      s_s1[0] = 0.0;
      for (int i = 0; i + 1 < n1; i++) {
        const double dx = s_wp[0][i + 1] - s_wp[0][i], dy = s_wp[1][i + 1] - s_wp[1][i];
        s_s1[i + 1] = s_s1[i] + sqrt(dx * dx + dy * dy);
      }
      for (int a = 0; a < 2; a++) {
        for (int i = 0; i < n1; i++) s_d1[a][i] = pchip_slope(s_s1, s_wp[a], n1, i);
        for (int i = 0; i + 1 < n1; i++) hermite(s_s1, s_wp[a], s_d1[a][i], s_d1[a][i + 1], i, s_c1[a][i]);
      }
      s_n1 = n1;
    }
    __syncthreads();
    {
      const int n1 = s_n1;
      for (int k = tid; k < GEN_NK; k += nt) {
        const double q = linspace_at(s_s1[0], s_s1[n1 - 1], GEN_NK, k);
        const int i = find_interval(s_s1, n1, q);
        const double z = q - s_s1[i], z2 = z * z;
        wx[k] = ((s_c1[0][i][3] + s_c1[0][i][2] * z) + s_c1[0][i][1] * z2) + s_c1[0][i][0] * (z2 * z);
        wy[k] = ((s_c1[1][i][3] + s_c1[1][i][2] * z) + s_c1[1][i][1] * z2) + s_c1[1][i][0] * (z2 * z);
      }
    }
    __syncthreads();
    // ---- passes 2 and 3 (path.py:24-31); the third interpolant is the path ----
    pchip_pass(wx, wy, s_ks, cfA, s_dx, s_dy, true, s_part);          // pass 2 (resample in place after the pass)
    double* cf = g.knot_coef + (size_t)w * GEN_NK * 8;
    pchip_pass(wx, wy, s_ks, cf, s_dx, s_dy, false, s_part);          // pass 3 -> bank (coefficient rows) ...
    {
      double* ks_bank = g.knot_s + (size_t)w * GEN_NK;                // ... and its knots; every evaluation below walks the LDS copy
      for (int i = tid; i < GEN_NK; i += nt) ks_bank[i] = s_ks[i];
    }
    const double* const ks = s_ks;
    if (tid == 0) {
      const double L = ks[GEN_NK - 1];
      int P = (int)(10.0 * L);
      if (P > g.p_cap) P = g.p_cap;        // cannot happen for length-800 curves (L <= ~1450 m)
      s_L = L, s_P = P;
      double xy[2], dxy[2];
      eval_xy(ks, cf, GEN_NK, L, xy, dxy);
      s_goal[0] = xy[0], s_goal[1] = xy[1];
      // vessel start (movingobstacles.py:34-43)
      eval_xy(ks, cf, GEN_NK, 0.0, xy, dxy);
      s_pose[0] = xy[0] + 50 * (row[8] - 0.5);
      s_pose[1] = xy[1] + 50 * (row[9] - 0.5);
      s_pose[2] = auv_princip(atan2(dxy[1], dxy[0]) + 2 * PI * (row[10] - 0.5));
      double* ws = g.world_scalar + 8 * (size_t)w;
      ws[0] = L, ws[1] = s_goal[0], ws[2] = s_goal[1], ws[3] = s_pose[0], ws[4] = s_pose[1], ws[5] = s_pose[2];
      ws[6] = 0.0, ws[7] = 0.0;
      g.poly_cnt[w] = P;
      g.chunk_cnt[w] = (P - 1 + AUV_CHUNK - 1) / AUV_CHUNK;
      g.knot_cnt[w] = GEN_NK;
      g.obs_cnt[w] = K;
      g.mv_cnt[w] = g.n_moving;
    }
    __syncthreads();
    const double L = s_L;
    const int P = s_P;
    // ---- dense polyline + cumulative arclength (path.py:38-40) ----
    double2* pxy = g.poly_xy + (size_t)w * g.p_cap;
    double* pcum = g.poly_cum + (size_t)w * g.p_cap;
    for (int k = tid; k < P; k += nt) {
      double xy[2], dxy[2];
      eval_xy(ks, cf, GEN_NK, linspace_at(0.0, L, P, k), xy, dxy);
      pxy[k] = make_double2(xy[0], xy[1]);
    }
    __syncthreads();
    block_cumsum(pcum, P - 1, [&](int i) {
      const double dx = pxy[i + 1].x - pxy[i].x, dy = pxy[i + 1].y - pxy[i].y;
      return sqrt(dx * dx + dy * dy);
    }, s_part);
    // ---- bounding circle of every run of AUV_CHUNK segments (K3's pruning table) ----
    {
      const int nch = (P - 1 + AUV_CHUNK - 1) / AUV_CHUNK;
      double4* cb = g.chunk_bound + (size_t)w * (g.p_cap / AUV_CHUNK);
      for (int c = tid; c < nch; c += nt) {
        const int v0 = c * AUV_CHUNK, v1 = min(v0 + AUV_CHUNK, P - 1);
        double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
        for (int v = v0; v <= v1; v++) {
          const double2 p = pxy[v];
          x0 = fmin(x0, p.x), x1 = fmax(x1, p.x), y0 = fmin(y0, p.y), y1 = fmax(y1, p.y);
        }
        const double cx = 0.5 * (x0 + x1), cy = 0.5 * (y0 + y1);
        double r2 = 0.0;
        for (int v = v0; v <= v1; v++) {
          const double dx = pxy[v].x - cx, dy = pxy[v].y - cy;
          r2 = fmax(r2, dx * dx + dy * dy);
        }
        cb[c] = make_double4(cx, cy, sqrt(r2) * (1.0 + 1e-9) + 1e-9, 0.0);
      }
    }
    // ---- obstacles: threads <-> obstacles (helpers.py:5-35 with a candidate pool) ----
    for (int j = tid; j < K; j += nt) {
      const bool mover = j < g.n_moving;
      const int base = mover ? 11 + j * (3 * GEN_CAND + 2) : 11 + g.n_moving * (3 * GEN_CAND + 2) + (j - g.n_moving) * 3 * GEN_CAND;
      const double sigma = mover ? 500.0 : 250.0;
      const double cps = cos(-s_pose[2]), sps = sin(-s_pose[2]);
      double px = 0, py = 0, radius = 1;
      for (int k = 0; k < GEN_CAND + GEN_EXTRA_CAND; k++) {
        double z, u, pois;
        if (k < GEN_CAND) {
          z = row[base + 3 * k], u = row[base + 3 * k + 1], pois = row[base + 3 * k + 2];
        } else {
          // the whole pool was rejected (~1e-13 per obstacle): further candidates from a counter-based
          // generator keyed by the pool's first two draws -- devgen.extra_candidate is the host mirror
          const unsigned long long key = (unsigned long long)__double_as_longlong(row[base]) ^
                                         ((unsigned long long)__double_as_longlong(row[base + 1]) << 1);
          double uu[4];
          for (int i = 0; i < 4; i++) uu[i] = (double)(gen_splitmix64(key + 4ull * (unsigned long long)(k - GEN_CAND) + i) >> 11) * 0x1.0p-53;
          z = sqrt(-2.0 * log(1.0 - uu[0])) * cos(2.0 * PI * uu[1]);
          u = uu[2];
          const double mean = mover ? 10.0 : 30.0;            // obst_radius_mean (movingobstacles.py:54-90)
          double pr = exp(-mean), cdf = pr;
          int nn = 0;
          while (uu[3] > cdf && nn < 1000) nn++, pr *= mean / nn, cdf += pr;
          pois = (double)nn;
        }
        const double disp = sigma * z;
        const double arclen = (0.1 + 0.8 * u) * L;
        double xy[2], dxy[2];
        eval_xy(ks, cf, GEN_NK, arclen, xy, dxy);
        const double ang = auv_princip(atan2(dxy[1], dxy[0]) - PI / 2);
        px = xy[0] + disp * cos(ang), py = xy[1] + disp * sin(ang);
        radius = pois > 1.0 ? pois : 1.0;
        const double dx = px - s_pose[0], dy = py - s_pose[1];
        const double rx = cps * dx - sps * dy, ry = sps * dx + cps * dy;
        const double vessel_dist = sqrt(rx * rx + ry * ry) - g.vessel_width - radius;
        const double gx = px - s_goal[0], gy = py - s_goal[1];
        const double goal_dist = sqrt(gx * gx + gy * gy) - radius;
        if (fmin(vessel_dist, goal_dist) > 0) break;
      }
      // obstacle order of world.build_world: circles first, movers last
      const size_t ko = (size_t)w * K + (mover ? g.n_static + j : j - g.n_moving);
      if (mover) {
        // VesselObstacle on a straight unit-time trajectory (movingobstacles.py:61-79)
        const double direction = row[base + 3 * GEN_CAND] * 2 * PI;
        const double speed = 1.0 + (3.0 - 1.0) * row[base + 3 * GEN_CAND + 1];
        const double p1x = px + 1 * speed * cos(direction), p1y = py + 1 * speed * sin(direction);
        const double vx = p1x - px, vy = p1y - py;
        // constructor's update(0.1), then the scenario's trailing _update(dt) (obstacles.py:192-215)
        double cx = px, cy = py, heading = PI / 2, counter = 0.0;
        for (int s = 0; s < 2; s++) {
          const double dts = s == 0 ? 0.1 : g.dt;
          counter += dts;
          const double ddx = dts * vx, ddy = dts * vy;
          heading = atan2(ddy, ddx);
          cx = cx + ddx, cy = cy + ddy;
        }
        const size_t mo = (size_t)w * g.n_moving + j;
        g.mv_param[mo] = make_double4(radius, px, py, 9999.0);
        g.mv_init[mo] = make_double4(cx, cy, heading, counter);
        g.mv_vtab[mo] = make_double2(vx, vy);
        g.mv_vtab_len[mo] = 1;
        g.obs_meta[ko] = make_int4(AUV_OBS_MOVER, 0, AUV_MOVER_NSEG, j);
        g.obs_cull[3 * ko] = 0.0, g.obs_cull[3 * ko + 1] = 0.0, g.obs_cull[3 * ko + 2] = 0.0;
      } else {
        const int c = j - g.n_moving;
        int ri = (int)radius;
        if (ri >= g.n_radius) ri = g.n_radius - 1;
        const int nseg = g.nseg_by_radius[ri];
        const size_t so = (size_t)w * g.g_cap + (size_t)c * 64;
        g.obs_meta[ko] = make_int4(AUV_OBS_RING, (int)so, nseg, -3);   // -3: simple ring, clockwise (GEOS buffer order)
        g.obs_cull[3 * ko] = px, g.obs_cull[3 * ko + 1] = py, g.obs_cull[3 * ko + 2] = radius;
        // ring of the GEOS buffer, thinned to nseg segments (obstacles.py:101-106)
        const int stride = 64 / nseg;
        for (int sI = 0; sI < nseg; sI++) {
          const int a = sI * stride, b = (sI + 1) * stride;
          const double ax = a == 0 || a == 64 ? px + radius : px + radius * g.ring_unit[2 * a];
          const double ay = a == 0 || a == 64 ? py : py + radius * g.ring_unit[2 * a + 1];
          const double bx = b == 64 ? px + radius : px + radius * g.ring_unit[2 * b];
          const double by = b == 64 ? py : py + radius * g.ring_unit[2 * b + 1];
          g.seg[so + sI] = make_double4(ax, ay, bx, by);
        }
      }
    }
    __syncthreads();
  }
}

// ---- draws from a counter-based generator (fresh-world mode): the world of (seed, environment, serial) -------------------
// The draws of auv_generate_worlds come from the caller (torch's generator); a refill pass on the device has no host to ask,
// and the world an environment meets in its k-th episode should not depend on when the pass ran or on how the batch is
// sharded over GPUs.  So row = f(seed, GLOBAL environment index, serial): uniform u = splitmix64(key + counter) >> 11 * 2^-53,
// z by Box-Muller from two of them, the Poisson variates by inversion -- the layout of devgen.sample_draws (devgen.py:
// counter_draws is the host mirror of this kernel).
__device__ __forceinline__ double gen_u01(const unsigned long long key, const unsigned long long ctr) {
  return (double)(gen_splitmix64(key + ctr) >> 11) * 0x1.0p-53;
}
__device__ __forceinline__ unsigned long long gen_world_key(const unsigned long long seed, const long long env, const int serial) {
  const unsigned long long k0 = gen_splitmix64(seed);
  const unsigned long long k1 = gen_splitmix64(k0 ^ (unsigned long long)env);
  return gen_splitmix64(k1 ^ (unsigned long long)(unsigned)serial);
}
__device__ __forceinline__ double gen_poisson(const double mean, const double u) {
  double pr = exp(-mean), cdf = pr;
  int nn = 0;
  while (u > cdf && nn < 1000) nn++, pr *= mean / nn, cdf += pr;
  return (double)nn;
}

// rows 0 .. *count_dev - 1 (count_dev NULL: n_rows): row i = the draws of environment env_base + envs[i], serial serials[i]
__global__ void __launch_bounds__(256) k5_draws(double* __restrict__ draws, int n_draws, int n_moving, int n_static, unsigned long long seed,
                                                long long env_base, const int32_t* __restrict__ envs, const int32_t* __restrict__ serials,
                                                int n_rows, const int32_t* __restrict__ count_dev) {
  if (count_dev) n_rows = *count_dev;
  const double PI = AUV_PI;
  for (int row = blockIdx.x; row < n_rows; row += gridDim.x) {
    const unsigned long long key = gen_world_key(seed, env_base + envs[row], serials[row]);
    double* out = draws + (size_t)row * n_draws;
    const int per_mover = 3 * GEN_CAND + 2, movers_end = 11 + n_moving * per_mover;
    for (int j = threadIdx.x; j < n_draws; j += blockDim.x) {
      int kind = 0;                                         // 0 uniform, 1 normal, 2 Poisson(10), 3 Poisson(30)
      if (j >= 11 && j < movers_end) {
        const int r = (j - 11) % per_mover;
        if (r < 3 * GEN_CAND) kind = r % 3 == 0 ? 1 : (r % 3 == 2 ? 2 : 0);
      } else if (j >= movers_end) {
        const int r = (j - movers_end) % (3 * GEN_CAND);
        kind = r % 3 == 0 ? 1 : (r % 3 == 2 ? 3 : 0);
      }
      const double u0 = gen_u01(key, 2ull * (unsigned long long)j), u1 = gen_u01(key, 2ull * (unsigned long long)j + 1ull);
      double v = u0;
      if (kind == 1) v = sqrt(-2.0 * log(1.0 - u0)) * cos(2.0 * PI * u1);
      else if (kind == 2) v = gen_poisson(10.0, u0);
      else if (kind == 3) v = gen_poisson(30.0, u0);
      out[j] = v;
    }
  }
}

// ---- the refill pass's bookkeeping kernels (auv_capi.hip: fw_refill) ----
// Pop up to `cap` stale slots off the queue, in the order they were left: batch row i <- slot, its environment, and the serial
// of the world it is about to receive (an environment's worlds are numbered in the order it needs them).  One wave, lane 0.
__global__ void k_fw_bind(FwBatch b, int32_t* __restrict__ queue, unsigned int* __restrict__ ctl, int q_cap, int n_envs, int cap,
                          int32_t* __restrict__ env_next_serial, int32_t* __restrict__ shadow_world_idx, int32_t* __restrict__ shadow_fresh_count) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const unsigned int head = ctl[1];
  const unsigned int tail = __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned int avail = tail - head;
  if (avail > (unsigned int)cap) avail = (unsigned int)cap;
  int count = 0;
  for (unsigned int i = 0; i < avail; i++) {
    int32_t* q = queue + ((head + i) % (unsigned int)q_cap);
    const int s = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (s < 0) break;                                       // reserved by a finish wave, not written yet: next pass
    const int e = s % n_envs;
    b.slot[count] = s, b.env[count] = e, b.serial[count] = env_next_serial[e]++;
    shadow_world_idx[count] = s;
    __hip_atomic_store(q, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    count++;
  }
  ctl[1] = head + (unsigned int)count;
  *b.count = count;
  *shadow_fresh_count = 0;
}

// The pass's LAST kernel: the slots it has rebuilt become READY.  Every kernel of the pass before this one has ended (stream
// order), i.e. the slots' tables and reset rows are complete in memory; the serial and the state go out as agent-scope stores,
// the state last.  A finish wave that reads READY (agent-scope load) reads the slot's tables with agent-scope loads in the
// same launch (restore_env<true>) and the plain way, behind a kernel boundary, ever after.
__global__ void k_fw_ready(FwBatch b, int32_t* __restrict__ state, int32_t* __restrict__ serial, unsigned int* __restrict__ ctl) {
  const int count = *b.count;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const int s = b.slot[i];
    __hip_atomic_store(serial + s, b.serial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0x0F70);                       // (vmcnt(0): the serial is out before the state)
    __hip_atomic_store(state + s, AUV_FW_READY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    __hip_atomic_fetch_add((unsigned long long*)(ctl + 4), (unsigned long long)count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // worlds rebuilt so far
    __hip_atomic_fetch_add((unsigned long long*)(ctl + 6), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                        // passes completed
  }
}

}  // namespace

size_t auv_gen_scratch_doubles(void) { return GEN_SCRATCH; }

void auv_launch_generate(const GenOut& g, const double* draws, int w_first, int n_worlds, int grid, hipStream_t st, const int32_t* slots,
                         const int32_t* count_dev) {
  hipLaunchKernelGGL(k5_generate, dim3(grid), dim3(256), 0, st, g, draws, w_first, n_worlds, slots, count_dev);
}

void auv_launch_draws(double* draws, int n_draws, int n_moving, int n_static, unsigned long long seed, long long env_base, const int32_t* envs,
                      const int32_t* serials, int n_rows, const int32_t* count_dev, int grid, hipStream_t st) {
  hipLaunchKernelGGL(k5_draws, dim3(grid), dim3(256), 0, st, draws, n_draws, n_moving, n_static, seed, env_base, envs, serials, n_rows, count_dev);
}

void auv_launch_fw_bind(const FwBatch& b, int32_t* queue, unsigned int* ctl, int q_cap, int n_envs, int cap, int32_t* env_next_serial,
                        int32_t* shadow_world_idx, int32_t* shadow_fresh_count, hipStream_t st) {
  hipLaunchKernelGGL(k_fw_bind, dim3(1), dim3(AUV_WAVE), 0, st, b, queue, ctl, q_cap, n_envs, cap, env_next_serial, shadow_world_idx, shadow_fresh_count);
}

void auv_launch_fw_ready(const FwBatch& b, int32_t* state, int32_t* serial, unsigned int* ctl, int cap, hipStream_t st) {
  hipLaunchKernelGGL(k_fw_ready, dim3((cap + 255) / 256), dim3(256), 0, st, b, state, serial, ctl);
}
