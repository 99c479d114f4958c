// K2 — moving-obstacle update + LiDAR sweep: one 256-thread workgroup per environment.
//
// Reference: BaseEnvironment._update              gym_auv/environment.py:386-392
//            VesselObstacle._update / boundary     objects/obstacles.py:195-233
//            Vessel.perceive                       objects/vessel/vessel.py:249-368
//            find_rays_to_simulate_for_obstacles   objects/vessel/sensor.py:74-97
//            _find_limit_angle_rays                sensor.py:41-71
//            simulate_sensor                       sensor.py:140-159
//
// Work decomposition (wave64):
//   phase A  threads <-> movers: advance kinematics, rebuild the 5 pentagon segments + cull
//            circle in LDS.
//   phase B  threads <-> obstacles: (every 25th vessel step) nearby test; cull window
//            [i_min-1, i_max % S) with the reference's Python-range / negative-index
//            semantics; point-in-polygon for filled obstacles.
//   phase C  threads <-> rays: ray end points (one sincos per ray) into LDS.
//   phase D  each wave takes obstacles round-robin; its 64 lanes enumerate the
//            (ray-in-window x boundary-segment) pairs of that obstacle, so lanes stay busy
//            however narrow the window is; a hit does an LDS atomic-min on the ray's range
//            (non-negative fp64 ordered as uint64).
//   phase E  threads <-> rays: write d, closeness (fused), block-OR the collision flag.
// Only pairs inside the reference's cull windows are evaluated (~10-15 % of S x G), which is
// what makes this kernel traffic-bound rather than VALU-bound.
// Roofline: HBM.  Algorithmic bytes per env-step (fp64 layout): 32*G (segments, G per env)
// + 24*K (cull circles) + 16*K (meta) + 24 (pose) + 16*S (d + closeness out) + K (nearby).
#include "auv_device.h"

namespace {

struct ObsLds {        // per-obstacle scratch in LDS
  int kind;
  int seg_off;         // absolute index into seg[] (static) or mover slot*5 (mover)
  int nseg;
  int start;           // first ray index of the window (may be negative)
  int count;           // number of rays in the window (0 = culled / not nearby)
  int inside;          // p0 inside a filled obstacle
};

__device__ __forceinline__ unsigned long long d2u(double x) { return (unsigned long long)__double_as_longlong(x); }
__device__ __forceinline__ double u2d(unsigned long long x) { return __longlong_as_double((long long)x); }

// sensor.py:140-159 for one (ray, boundary segment) pair: returns distance or -1
__device__ __forceinline__ double ray_seg(double px, double py, double rx, double ry, double ax, double ay,
                                          double bx, double by) {
  double sx = bx - ax, sy = by - ay;
  double den = rx * sy - ry * sx;
  if (den == 0.0) return -1.0;
  double wx = ax - px, wy = ay - py;
  double tn = wx * sy - wy * sx;   // t = tn/den along the ray
  double un = wx * ry - wy * rx;   // u = un/den along the boundary segment
  // 0 <= tn/den <= 1 and 0 <= un/den <= 1, decided without dividing (exactly equivalent for
  // correctly rounded IEEE division)
  bool pos = den > 0.0;
  bool hit = pos ? (tn >= 0.0 && tn <= den && un >= 0.0 && un <= den)
                 : (tn <= 0.0 && tn >= den && un <= 0.0 && un >= den);
  if (!hit) return -1.0;
  double t = tn / den;
  double X = px + t * rx, Y = py + t * ry;
  double dx = X - px, dy = Y - py;
  return sqrt(dx * dx + dy * dy);
}

__device__ __forceinline__ bool point_in_polygon(double px, double py, const double4* seg, int nseg) {
  bool inside = false;
  for (int i = 0; i < nseg; i++) {
    double4 s = seg[i];
    if (auv_pt_seg_dist(px, py, s.x, s.y, s.z, s.w) == 0.0) return true;
    if ((s.y > py) != (s.w > py)) {
      double xint = s.x + (py - s.y) * (s.z - s.x) / (s.w - s.y);
      if (px < xint) inside = !inside;
    }
  }
  return inside;
}

__device__ __forceinline__ double point_boundary_distance(double px, double py, const double4* seg, int nseg) {
  double best = 1.0e300;
  for (int i = 0; i < nseg; i++) {
    double4 s = seg[i];
    double t = auv_pt_seg_dist(px, py, s.x, s.y, s.z, s.w);
    if (t < best) best = t;
  }
  return best;
}

// LDS layout (dynamic): [Mmax*5] double4 mover segs | [Mmax] double4 mover cull (cx, cy, rho, -) |
//                       [S] double2 ray vectors | [S] u64 d-bits | [Kmax] ObsLds | int any-flag
__global__ void __launch_bounds__(AUV_BLOCK) k2_lidar(AuvDev d, int advance_movers, int only_fresh) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int e = blockIdx.x;
  const int tid = threadIdx.x;
  const int S = d.cfg.n_sensors;
  const size_t n = (size_t)d.n;
  int4 cnt = d.counters[e];
  if (only_fresh && cnt.w == 0) return;   // reset pass: untouched envs leave immediately

  // carve the dynamic LDS by decreasing alignment (no static __shared__ ahead of it)
  double4* mvseg = (double4*)smem;
  double4* mvcull = mvseg + d.m_max * AUV_MOVER_NSEG;
  double2* rayv = (double2*)(mvcull + d.m_max);
  unsigned long long* dbits = (unsigned long long*)(rayv + S);
  ObsLds* obs = (ObsLds*)(dbits + S);
  int& s_any = *(int*)(obs + d.k_max);

  const double px = d.state[0 * n + e], py = d.state[1 * n + e], psi = d.state[2 * n + e];
  const int w = d.world_idx[e];
  const long long k0 = d.obs_off[w];
  const int K = (int)(d.obs_off[w + 1] - k0);
  const long long m0 = d.mv_off[w];
  const int M = (int)(d.mv_off[w + 1] - m0);
  const double R = d.cfg.sensor_range, W = d.cfg.vessel_width;
  const double dangle = 2 * AUV_PI / S;
  if (tid == 0) s_any = 0;

  // ---- phase A: movers (obstacles.py:195-233) ---------------------------------------
  for (int m = tid; m < M; m += AUV_BLOCK) {
    double4 st = d.mover[(size_t)e * d.m_max + m];
    const double4 par = d.mv_param[m0 + m];
    if (advance_movers) {
      const double dt = d.cfg.dt;
      const long long voff = d.mv_vtab_off[m0 + m];
      const long long vlen = d.mv_vtab_off[m0 + m + 1] - voff;
      st.w += dt;
      long long idx = (long long)floor(st.w);
      if (idx >= (long long)par.w - 1) {
        st.w = 0.0;
        idx = 0;
        st.x = par.y;
        st.y = par.z;
      }
      if (idx > vlen - 1) idx = vlen - 1;
      double2 v = d.mv_vtab[voff + idx];
      double dx = dt * v.x, dy = dt * v.y;
      st.z = atan2(dy, dx);
      st.x = st.x + dx;
      st.y = st.y + dy;
      d.mover[(size_t)e * d.m_max + m] = st;
    }
    const double wd = par.x;
    double s, c;
    sincos(st.z, &s, &c);
    // closed form of enclosing_circle for the pentagon (MRR = body box): tests/test_world.py
    const double x0 = 5.0 * wd / 18.0, dxc = wd / 2.0 - x0;
    mvcull[m] = make_double4(st.x + x0 + c * dxc, st.y + s * dxc, wd * sqrt(5.0) / 2.0, 0.0);
    if (fabs(c) < 2.5e-16) c = 0.0;   // shapely.affinity.rotate snaps tiny cos/sin
    if (fabs(s) < 2.5e-16) s = 0.0;
    const double bx[5] = {-wd / 2, -wd / 2, wd / 2, 3.0 / 2 * wd, wd / 2};
    const double by[5] = {-wd / 2, wd / 2, wd / 2, 0.0, -wd / 2};
    const double xo = x0 - x0 * c, yo = 0.0 - x0 * s;
    double vx[5], vy[5];
#pragma unroll
    for (int i = 0; i < 5; i++) {
      vx[i] = (c * bx[i] + -s * by[i] + xo) + st.x;
      vy[i] = (s * bx[i] + c * by[i] + yo) + st.y;
    }
#pragma unroll
    for (int i = 0; i < 5; i++) {
      int j = (i + 1) % 5;
      mvseg[m * AUV_MOVER_NSEG + i] = make_double4(vx[i], vy[i], vx[j], vy[j]);
    }
  }
  // ---- phase C (independent of A): ray vectors, vessel.py:66-68, :317 ----------------
  for (int i = tid; i < S; i += AUV_BLOCK) {
    double ang = (-AUV_PI + (i + 1) * dangle) + psi;
    double s, c;
    sincos(ang, &s, &c);
    // end point minus origin, formed exactly as the reference forms the end point
    double ex = px + c * R, ey = py + s * R;
    rayv[i] = make_double2(ex - px, ey - py);
    dbits[i] = d2u(R);
  }
  __syncthreads();
  if (!d.cfg.use_lidar) return;

  // ---- phase B: nearby list + cull windows -------------------------------------------
  const bool refresh = (cnt.y % d.cfg.sensor_interval_load_obstacles) == 0;   // vessel.py:266
  for (int k = tid; k < K; k += AUV_BLOCK) {
    const int4 meta = d.obs_meta[k0 + k];
    const bool mover = meta.x == AUV_OBS_MOVER;
    const double4* seg = mover ? (mvseg + meta.w * AUV_MOVER_NSEG) : (d.seg + meta.y);
    ObsLds o;
    o.kind = meta.x;
    o.seg_off = mover ? meta.w * AUV_MOVER_NSEG : meta.y;
    o.nseg = meta.z;
    o.start = 0;
    o.count = 0;
    o.inside = 0;
    int inside_known = 0;
    uint8_t near;
    if (refresh) {
      double dist;
      if (meta.x != AUV_OBS_RING && point_in_polygon(px, py, seg, meta.z)) {
        dist = 0.0;
        o.inside = 1;
      } else {
        dist = point_boundary_distance(px, py, seg, meta.z);
      }
      inside_known = 1;
      near = (dist - W < R) ? 1 : 0;
      d.nearby[(size_t)e * d.k_max + k] = near;
    } else {
      near = d.nearby[(size_t)e * d.k_max + k];
    }
    int2 lim = make_int2(INT32_MIN, INT32_MIN);
    if (near) {
      long long start, stop;
      if (d.cfg.cull_mode == AUV_CULL_EXACT) {
        start = 0;
        stop = S;
      } else {
        double cx, cy, rho;
        if (mover) {
          double4 c4 = mvcull[meta.w];
          cx = c4.x, cy = c4.y, rho = c4.z;
        } else {
          cx = d.obs_cull[3 * (k0 + k)], cy = d.obs_cull[3 * (k0 + k) + 1], rho = d.obs_cull[3 * (k0 + k) + 2];
        }
        double relx = cx - px, rely = cy - py;
        double bearing = atan2(rely, relx) - psi;          // not wrapped (sensor.py:54)
        double dist = sqrt(relx * relx + rely * rely);
        double safe = dist > 1e-8 ? dist : 1e-8;
        double q = rho / safe;
        double f = (q > 1.0 || q < -1.0 || isnan(q)) ? AUV_PI : asin(q);   // NaN -> pi (sensor.py:34-36)
        long long imin = (long long)floor((AUV_PI + (bearing - f)) / dangle);
        long long imax = (long long)ceil((AUV_PI + (bearing + f)) / dangle);
        lim = make_int2((int)imin, (int)imax);
        start = imin - 1;
        stop = auv_pymod(imax, S);                           // range(i_min - 1, i_max % S)
      }
      if (stop > start) {
        o.start = (int)start;
        o.count = (int)(stop - start);
        if (meta.x != AUV_OBS_RING && !inside_known) o.inside = point_in_polygon(px, py, seg, meta.z) ? 1 : 0;
        s_any = 1;
      }
    }
    d.limits[(size_t)e * d.k_max + k] = lim;
    obs[k] = o;
  }
  for (int k = K + tid; k < d.k_max; k += AUV_BLOCK) d.limits[(size_t)e * d.k_max + k] = make_int2(INT32_MIN, INT32_MIN);
  __syncthreads();

  // ---- phase D: (ray, segment) pairs, one obstacle per wave at a time ------------------
  if (s_any) {
    const int wave = tid / AUV_WAVE, lane = tid % AUV_WAVE;
    for (int k = wave; k < K; k += AUV_BLOCK / AUV_WAVE) {
      const ObsLds o = obs[k];
      if (o.count == 0) continue;
      if (o.kind != AUV_OBS_RING && o.inside) {
        // p0 inside a filled polygon: the clipped ray starts at p0 -> distance 0 on every ray
        for (int q = lane; q < o.count; q += AUV_WAVE) atomicMin(&dbits[auv_pymod((long long)o.start + q, S)], 0ull);
        continue;
      }
      const double4* seg = (o.kind == AUV_OBS_MOVER) ? (mvseg + o.seg_off) : (d.seg + o.seg_off);
      const int total = o.count * o.nseg;
      const float inv = 1.0f / (float)o.nseg;
      for (int q = lane; q < total; q += AUV_WAVE) {
        int ro = (total < 32768) ? (int)(((float)q + 0.5f) * inv) : q / o.nseg;
        int si = q - ro * o.nseg;
        int i = auv_pymod((long long)o.start + ro, S);
        double4 s = seg[si];
        double2 r = rayv[i];
        double dist = ray_seg(px, py, r.x, r.y, s.x, s.y, s.z, s.w);
        if (dist >= 0.0) atomicMin(&dbits[i], d2u(dist));
      }
    }
  }
  __syncthreads();

  // ---- phase E: outputs (vessel.py:88-95, :356-359) ---------------------------------------
  int col = 0;
  const double logR = log(1 + R);
  for (int i = tid; i < S; i += AUV_BLOCK) {
    double di = u2d(dbits[i]);
    d.lidar_d[(size_t)e * S + i] = di;
    double cl = d.cfg.sensor_log_transform ? 1 - auv_clip(log(1 + di) / logR, 0.0, 1.0)
                                           : 1 - auv_clip(di / R, 0.0, 1.0);
    d.obs64[(size_t)e * (6 + S) + 6 + i] = auv_clip(cl, -1.0, 1.0);
    col |= (di < W);
  }
  col = __syncthreads_or(col);
  if (tid == 0) d.collision[e] = (uint8_t)(col != 0);
}

}  // namespace

size_t auv_k2_lds_bytes(const AuvDev& d) {
  size_t S = d.cfg.n_sensors;
  return S * 8 + S * 16 + (size_t)d.m_max * AUV_MOVER_NSEG * 32 + (size_t)d.m_max * 32 +
         (size_t)d.k_max * sizeof(ObsLds) + 16;
}

void auv_launch_k2(const AuvDev& d, int advance_movers, int only_fresh, hipStream_t st) {
  hipLaunchKernelGGL(k2_lidar, dim3(d.n), dim3(AUV_BLOCK), auv_k2_lds_bytes(d), st, d, advance_movers,
                     only_fresh);
}
