// K3 — path navigation + reward + episode termination + observation assembly:
// ONE WAVE PER ENVIRONMENT, wave-synchronous.  The step path runs the navigation as the second role
// of the one-wave workgroups of k23_lidar_nav (k_step_fused.hip) and the reward as k3_reward; the
// other kernels here serve the per-kernel API, the reset-row pass and reset().
//
// Reference: Vessel.navigate                  gym_auv/objects/vessel/vessel.py:461-541
//            Path.get_closest_arclength       objects/path.py:84-93  (GEOS LineString.project:
//                                             first strict minimum of point-segment distance)
//            Path.__call__/get_direction      objects/path.py:61-82  (SciPy PPoly evaluation)
//            ColavRewarder.calculate          objects/rewarder.py:167-241
//            PathFollowRewarder.calculate     objects/rewarder.py:78-140
//            BaseEnvironment.observe/step/_isdone  environment.py:247-290, 325-347, 375-384
//
// Nearest point on the dense polyline (P ~ 10 L vertices), EXACT but pruned: the polyline is
// cut into chunks of 64 segments, each with a bounding circle (built at load time).
//   pass 1  lanes <-> chunks: U = min over chunks of (|p - c| + rad)  -- an upper bound on the
//           distance to the path;
//   pass 2  a chunk can hold the (first) minimum only if |p - c| - rad <= U: the survivors
//           (typically 1-3 of ~160) are listed in LDS in ascending order (ballot + popcount);
//   pass 3  lanes <-> the 64 segments of each surviving chunk (two chunks per trip to memory),
//           GEOS point-segment distance, (distance, first index) min-reduced with wave shuffles;
//           the winning lane hands its segment over by shuffle.
// Every segment that could win or tie survives, so the result (and the reference's "first
// minimum wins" tie-break) is identical to the brute-force scan.  Then lane 0 evaluates the spline
// at s, lanes 1 and 2 at s + look-ahead (knot window + coefficient rows in one trip), one atan2
// serves the three angles, lane 0 finishes the navigation features and the path term of the reward.
// The reward kernel combines that term with the LiDAR term K2 left, does cumulative reward / done /
// episode bookkeeping and copies the reset rows of environments that ended.
// Roofline: HBM.  Algorithmic bytes per env-step: 32*ceil((P-1)/64) (chunk circles) + ~3 KiB
// (surviving chunks' vertices) + ~600 (knot rows, scalars, info/nav/counters); reward ~300.
#ifndef AUV_DEVICE_FUNCS_ONLY
#include <hip/hip_ext.h>
#endif

#include "auv_device.h"

namespace {

struct MinIdx {
  double d;
  int j;
};

// The tables and settings the navigation's tail and the reward phase touch, under the names AuvDev gives them: those
// two are written against "a descriptor D" and run on either.  The finish role of the one-launch step fills a StepTabs
// from the descriptor's device-side copy with scalar loads before its waits (through a generic reference to that copy
// every table pointer would be a vector load at the point of use, on the launch's critical tail).
struct StepTabs {
  struct {
    double min_cumulative_reward, min_goal_distance, min_path_progress, look_ahead_distance;
    int32_t max_timesteps, rewarder, test_mode, auto_reset, n_sensors, use_lidar, obs_channels;
  } cfg;
  const double *knot_s, *knot_coef;
  double *info64, *nav64, *obs64, *rew_path, *reward64, *step_info, *episode, *ep_log;
  unsigned long long* ep_log_count;
  int32_t ep_log_cap;
  int32_t* world_idx;
  int4* counters;
  int32_t ring_slots, ring_slot_host;
  int32_t* ring_pos;
  const int32_t* fw_serial;   // (fresh worlds: the slot's serial, for the episode log's world column)
  int32_t n;                  // environments of the handle (AuvDev::n)
};

// SciPy PPoly: interval search + power-basis eval.  The knots are near-uniform, so the interval
// is guessed from s / L and the knots and coefficient rows around the guess are fetched together
// (one trip to memory); only a guess that is off by more than one interval walks and re-fetches.
// KnotWin = such a two-interval window: knots [g, g+2] and the coefficient rows g, g+1.
struct KnotWin {
  double xa, xb, xc;       // knots g, g+1, g+2
  double ca[8], cb[8];     // coefficient rows g, g+1
  bool have;
};

template <class D>
__device__ __forceinline__ KnotWin knot_window(const D& d, long long k0, int nk, double x0, double xl, double s,
                                               double L) {
  KnotWin w;
  w.have = false;
  w.xa = w.xb = w.xc = 0.0;
#pragma unroll
  for (int a = 0; a < 8; a++) w.ca[a] = w.cb[a] = 0.0;
  if (!(s >= x0) || s >= xl || nk < 4) return w;
  const double* x = d.knot_s + k0;
  const double* cf = d.knot_coef + 8 * k0;
  const double fi = s / L * (nk - 1);
  int i = (int)fi;
  i = i < 0 ? 0 : (i > nk - 2 ? nk - 2 : i);
  // two adjacent intervals around the guess: the guessed one and the neighbour on the side the
  // fractional position leans to
  int g = (fi - (double)i < 0.5) ? i - 1 : i;
  g = g < 0 ? 0 : (g > nk - 3 ? nk - 3 : g);
  w.xa = x[g], w.xb = x[g + 1], w.xc = x[g + 2];
  // coefficient rows are 64-byte records: four 16-byte loads each instead of eight 8-byte ones
  // (the vector memory pipe takes a wave's load in the same time whatever its width)
  const double2* r2 = (const double2*)(cf + 8 * (size_t)g);
#pragma unroll
  for (int a = 0; a < 4; a++) {
    const double2 va = r2[a], vb = r2[4 + a];
    w.ca[2 * a] = va.x, w.ca[2 * a + 1] = va.y, w.cb[2 * a] = vb.x, w.cb[2 * a + 1] = vb.y;
  }
  w.have = true;
  return w;
}

// x0 / xl: the first / last knot, fetched by the caller ahead of time.
template <class D>
__device__ __forceinline__ void path_eval(const D& d, long long k0, int nk, double x0, double xl, double s,
                                          double L, double xy[2], double dxy[2]) {
  const double* x = d.knot_s + k0;
  const double* cf = d.knot_coef + 8 * k0;
  int i;
  double xi;
  double c[8];
  bool have = false;
  if (!(s >= x0)) {
    i = 0;
  } else if (s >= xl) {
    i = nk - 2;
  } else {
    const KnotWin w = knot_window(d, k0, nk, x0, xl, s, L);
    if (w.have && s >= w.xa && s < w.xc) {
      const bool second = s >= w.xb;
      xi = second ? w.xb : w.xa;
#pragma unroll
      for (int a = 0; a < 8; a++) c[a] = second ? w.cb[a] : w.ca[a];
      have = true;
    }
    if (!have) {
      const double fi = s / L * (nk - 1);
      i = (int)fi;
      i = i < 0 ? 0 : (i > nk - 2 ? nk - 2 : i);
      while (i > 0 && s < x[i]) i--;
      while (i < nk - 2 && s >= x[i + 1]) i++;
    }
  }
  if (!have) {
    xi = x[i];
    const double2* r2 = (const double2*)(cf + 8 * (size_t)i);
#pragma unroll
    for (int a = 0; a < 4; a++) {
      const double2 v2 = r2[a];
      c[2 * a] = v2.x, c[2 * a + 1] = v2.y;
    }
  }
  const double z = s - xi, z2 = z * z;
#pragma unroll
  for (int a = 0; a < 2; a++) {
    const double* q = c + 4 * a;
    xy[a] = ((q[3] + q[2] * z) + q[1] * z2) + q[0] * (z2 * z);
    dxy[a] = (q[2] + (2.0 * q[1]) * z) + (3.0 * q[0]) * z2;
  }
}

// path-following term of the reward (rewarder.py:110-118, :196-203)
template <class D>
__device__ __forceinline__ double reward_path_term(const D& d, double u, double v, double heading_error,
                                                   double cross_track_error, double progress, double max_progress) {
  const double gamma_y_e = 5.0, max_speed = 2.0;
  const double speed = sqrt(u * u + v * v);
  const double ctp = exp(-gamma_y_e * fabs(cross_track_error));
  double path_reward = (1 + cos(heading_error) * speed / max_speed) * (1 + ctp) - 1;
  if (d.cfg.rewarder == AUV_REWARD_COLAV && progress < max_progress) path_reward = fmin(path_reward, 0.0);
  return path_reward;
}

// the same with the speed sqrt(u u + v v) and cos(heading_error) handed in (the navigation forms them beside its angles)
template <class D>
__device__ __forceinline__ double reward_path_term_cos(const D& d, double speed, double cos_heading_error,
                                                       double cross_track_error, double progress, double max_progress) {
  const double gamma_y_e = 5.0, max_speed = 2.0;
  const double ctp = exp(-gamma_y_e * fabs(cross_track_error));
  double path_reward = (1 + cos_heading_error * speed / max_speed) * (1 + ctp) - 1;
  if (d.cfg.rewarder == AUV_REWARD_COLAV && progress < max_progress) path_reward = fmin(path_reward, 0.0);
  return path_reward;
}

// LiDAR term of the Colav reward from the ranges in HBM (rewarder.py:205-222), whole wave; the
// same arithmetic, lane assignment and summation order as the fused form at the end of K2
__device__ __forceinline__ double reward_lidar_term_wave(const AuvDev& d, const double* __restrict__ dd, int lane) {
  const int S = d.cfg.n_sensors;
  const double R = d.cfg.sensor_range;
  const double raw_free = d.derived[1];
  double num = 0.0;
  for (int i0 = 0; i0 < S; i0 += AUV_WAVE) {
    const int i = i0 + lane;
    const double di = (i < S) ? dd[i] : R;
    double raw = raw_free;
    if (__any(di != R)) raw = (di != R) ? R * exp(-0.1 * di) : raw_free;   // gamma_x; velocity channel == 0 (sensor.py:159)
    if (i < S) num += d.beam_w[i] * raw;                // gamma_theta
  }
  num = auv_wave_sum(num);
  return (S > 0) ? -num / d.derived[2] : 0.0;
}

#ifndef AUV_DEVICE_FUNCS_ONLY
// per-config constants, formed once on the device with the very functions the step would use:
// gamma_theta of every beam (vessel.py:66-68, rewarder.py:205-222), their sum in the wave-reduction
// order, log(1 + R), R exp(-0.1 R) and the all-free LiDAR term of the reward
__global__ void k_derive(AuvDev d) {
  const int lane = threadIdx.x;
  const int S = d.cfg.n_sensors;
  const double dangle = 2 * AUV_PI / S;
  const double R = d.cfg.sensor_range;
  const double raw_free = R * exp(-0.1 * R);
  double den = 0.0, num_free = 0.0;
  for (int i = lane; i < S; i += AUV_WAVE) {
    const double angle = -AUV_PI + (i + 1) * dangle;
    const double weight = 1 / (1 + fabs(10.0 * angle));
    d.beam_w[i] = weight;
    den += weight;
    num_free += weight * raw_free;                        // the sweep's own summation order (k2_back)
  }
  den = auv_wave_sum(den);
  num_free = auv_wave_sum(num_free);
  // [3]: the LiDAR term of the Colav reward when no beam has a return
  if (lane == 0) d.derived[0] = log(1 + R), d.derived[1] = raw_free, d.derived[2] = den, d.derived[3] = (S > 0) ? -num_free / den : 0.0;
  // [4], [5]: the beam spacing and its reciprocal, as the cull windows use them (k2_front)
  if (lane == 0) d.derived[4] = 2 * AUV_PI / S, d.derived[5] = (double)S / (2 * AUV_PI);
}
#endif

// restore reset-time state of env e bound to world w2 (environment.py:203-245, vessel.py:189-224).
// The reset observation (navigate + perceive at the initial pose) is a per-world constant that
// was computed once at load time; here it is copied.  While those rows are being computed
// (w_ready == 0) the env is put on the fresh list instead.
// COH (fresh-world mode): the world's tables and reset rows are read with agent-scope (sc1) loads.  A refill pass on a side
// stream may have rebuilt slot w2 and flipped it to READY while THIS launch was already running -- kernel-boundary visibility
// does not cover that, and this XCD's L2 / this CU's L1 may still hold lines of the slot's previous world (or lines shared
// with a neighbouring slot's small records).  sc1 loads are served coherently across the XCDs; the pass's stores were complete
// (its kernels had ended) before its last kernel made the slot READY, and READY was read with an agent-scope load too.  The
// environment's NEXT step is another launch: its waves start behind a kernel boundary and read the tables the plain way.
// WTM: the mover and nearby rows go out write-through -- inside a launch of several steps (k_step_multi) the environment's sweep
// wave of the NEXT step, on another CU, reads them with agent-scope loads without a kernel boundary in between
template <bool COH, bool WTM = false>
__device__ __forceinline__ void restore_env(const AuvDev& d, int e, int w2, int lane, int episodes,
                                            float* __restrict__ obs_out) {
  const int S = d.cfg.n_sensors;
  const size_t n = (size_t)d.n;
  const double* ws2 = d.world_scalar + 8 * (size_t)w2;
  const int D = 6 + (d.cfg.use_lidar ? S * (d.cfg.obs_channels == 3 ? 3 : 1) : 0);   // row stride of obs_out
  const int DL = 6 + (d.cfg.use_lidar ? S : 0);                                        // columns this path writes
  if (d.w_ready && 6 + S <= 4 * AUV_WAVE && d.k_max <= AUV_WAVE) {
    // The usual shapes: every row fits four wave passes.  The auto-reset sits on the reward kernel's
    // critical path and the compiler must assume that the rows alias, so the copy is written as:
    // every load that depends only on the world index (one trip to memory), the mover states (their
    // offset comes with the first trip), then all stores.
    const EnvDesc nd = auv_make_desc<COH>(d, w2);
    const double ix = auv_ld<COH>(ws2 + 3), iy = auv_ld<COH>(ws2 + 4), ipsi = auv_ld<COH>(ws2 + 5);
    const uint8_t wcol = auv_ld<COH>(d.w_collision + w2);
    double lv[4], ov[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int i = q * AUV_WAVE + lane;
      lv[q] = (i < S) ? auv_ld<COH>(d.w_lidar + (size_t)w2 * S + i) : 0.0;
      ov[q] = (i < 6 + S) ? auv_ld<COH>(d.w_obs64 + (size_t)w2 * (6 + S) + i) : 0.0;
    }
    double row = 0.0;
    if (lane < 8) row = auv_ld<COH>(d.w_info + 8 * (size_t)w2 + lane);
    else if (lane < 16) row = auv_ld<COH>(d.w_nav + 8 * (size_t)w2 + lane - 8);
    uint8_t nb = 0;
    int2 lm = make_int2(0, 0);
    if (lane < d.k_max) nb = auv_ld<COH>(d.w_nearby + (size_t)w2 * d.k_max + lane), lm = auv_ld2i<COH>(d.w_limits + (size_t)w2 * d.k_max + lane);
    double4 mv0 = make_double4(0.0, 0.0, 0.0, 0.0);
    if (lane < nd.M) mv0 = auv_ld4<COH>(d.mv_init + nd.m0 + lane);
    if (lane == 0) {
      d.world_idx[e] = w2;
      d.env_desc[e] = nd;
      d.state[0 * n + e] = ix, d.state[1 * n + e] = iy, d.state[2 * n + e] = ipsi;
      d.state[3 * n + e] = 0.0, d.state[4 * n + e] = 0.0, d.state[5 * n + e] = 0.0;
      d.counters[e] = make_int4(0, 0, episodes, 0);
      d.collision[e] = wcol;
    }
    if (lane < nd.M) auv_st<WTM>(&d.mover[(size_t)e * d.m_max + lane], mv0);
    for (int m = AUV_WAVE + lane; m < nd.M; m += AUV_WAVE) auv_st<WTM>(&d.mover[(size_t)e * d.m_max + m], auv_ld4<COH>(d.mv_init + nd.m0 + m));
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int i = q * AUV_WAVE + lane;
      if (i < S) d.lidar_d[(size_t)e * S + i] = lv[q];
      if (i < 6 + S) {
        d.obs64[(size_t)e * (6 + S) + i] = ov[q];
        if (obs_out && i < DL) obs_out[(size_t)e * D + i] = (float)ov[q];
      }
    }
    if (lane < 8) d.info64[8 * (size_t)e + lane] = row;
    else if (lane < 16) d.nav64[8 * (size_t)e + lane - 8] = row;
    if (lane < d.k_max) auv_st<WTM>(&d.nearby[(size_t)e * d.k_max + lane], nb), d.limits[(size_t)e * d.k_max + lane] = lm;
    return;
  }
  if (lane == 0) {
    d.world_idx[e] = w2;
    d.env_desc[e] = auv_make_desc<COH>(d, w2);
    d.state[0 * n + e] = auv_ld<COH>(ws2 + 3), d.state[1 * n + e] = auv_ld<COH>(ws2 + 4), d.state[2 * n + e] = auv_ld<COH>(ws2 + 5);
    d.state[3 * n + e] = 0.0, d.state[4 * n + e] = 0.0, d.state[5 * n + e] = 0.0;
    d.counters[e] = make_int4(0, 0, episodes, 0);
    if (!d.w_ready) {
      double* inf = d.info64 + 8 * (size_t)e;
#pragma unroll
      for (int i = 0; i < 8; i++) inf[i] = 0.0;
      d.collision[e] = 0;
      d.fresh_list[atomicAdd(d.fresh_count, 1)] = e;
    } else {
      d.collision[e] = auv_ld<COH>(d.w_collision + w2);
    }
  }
  const long long m0 = d.mv_off[w2];
  const int M = auv_ld<COH>(d.mv_cnt + w2);
  for (int m = lane; m < M; m += AUV_WAVE) auv_st<WTM>(&d.mover[(size_t)e * d.m_max + m], auv_ld4<COH>(d.mv_init + m0 + m));
  if (!d.w_ready) {
    for (int i = lane; i < S; i += AUV_WAVE) d.lidar_d[(size_t)e * S + i] = d.cfg.sensor_range;
    for (int k = lane; k < d.k_max; k += AUV_WAVE) d.nearby[(size_t)e * d.k_max + k] = 0;
    return;
  }
  for (int i = lane; i < S; i += AUV_WAVE) d.lidar_d[(size_t)e * S + i] = auv_ld<COH>(d.w_lidar + (size_t)w2 * S + i);
  for (int i = lane; i < 6 + S; i += AUV_WAVE) {
    const double v = auv_ld<COH>(d.w_obs64 + (size_t)w2 * (6 + S) + i);
    d.obs64[(size_t)e * (6 + S) + i] = v;
    if (obs_out && i < DL) obs_out[(size_t)e * D + i] = (float)v;
  }
  if (lane < 8) d.info64[8 * (size_t)e + lane] = auv_ld<COH>(d.w_info + 8 * (size_t)w2 + lane);
  else if (lane < 16) d.nav64[8 * (size_t)e + lane - 8] = auv_ld<COH>(d.w_nav + 8 * (size_t)w2 + lane - 8);
  for (int k = lane; k < d.k_max; k += AUV_WAVE) {
    auv_st<WTM>(&d.nearby[(size_t)e * d.k_max + k], auv_ld<COH>(d.w_nearby + (size_t)w2 * d.k_max + k));
    d.limits[(size_t)e * d.k_max + k] = auv_ld2i<COH>(d.w_limits + (size_t)w2 * d.k_max + k);
  }
}
// the plain form, or the coherent one where a refill pass may be rebuilding slots beside the launch
__device__ __forceinline__ void restore_env(const AuvDev& d, int e, int w2, int lane, int episodes, float* __restrict__ obs_out) {
  if (d.fw_state) restore_env<true>(d, e, w2, lane, episodes, obs_out);
  else restore_env<false>(d, e, w2, lane, episodes, obs_out);
}

// ---- navigation part (Vessel.navigate + the six navigation observations) ----------------------
// Independent of the LiDAR sweep, so the step path runs it concurrently with K2 or right behind K1.
//
// The nearest-point search in two parts:
//   nav_bounds(p)   chunk circles against the pose p, U = min over circles of (|p - c| + rad) and over the 64 segments
//                   of the chunk that held the nearest point LAST step (exact distances: within centimetres of the
//                   truth), survivors |p - c| - rad <= U listed in LDS, and -- if at most NAV_SPEC survive -- their 64
//                   segments each fetched into registers;
//   nav_finish(p)   exact distances to the listed segments, (distance, first index) minimum = the brute-force result
//                   incl. its tie-break, then the scalar tail.
// (Round 2's two-kernel shape ran the first part for the pose BEFORE the step while the dynamics integrated and
// re-validated it afterwards; that shape was measured slower and removed in round 3, and with it the guess logic.)
constexpr int NAV_CPL = 4;           // chunk circles per lane kept in registers (paths up to 16 k vertices)
constexpr int NAV_SPEC = 2;          // speculative chunks whose segments are held in registers

struct NavSpec {
  EnvDesc ed;
  double L, goal_x, goal_y, knot_first, knot_last, maxp_in;
  int n_list;                        // surviving chunks listed in LDS (ascending)
  bool in_regs;                      // n_list <= NAV_SPEC: their segments are in A / B / cum below
  double2 A[NAV_SPEC], B[NAV_SPEC];  // this lane's segment of speculative chunk q
  double cum[NAV_SPEC];              // cumulative arclength at its first vertex
  double dist[NAV_SPEC];             // exact distance of p to that segment where the bound has formed it already (else < 0)
};

__device__ __forceinline__ NavSpec nav_bounds(const AuvDev& d, const int e, const int lane, int* list, const double qx,
                                              const double qy, const EnvDesc* ed_pre = nullptr) {
  NavSpec sp;
  sp.ed = ed_pre ? *ed_pre : d.env_desc[e];
  const EnvDesc& ed = sp.ed;
  const double* ws = d.world_scalar + 8 * (size_t)ed.w;
  sp.L = ws[0], sp.goal_x = ws[1], sp.goal_y = ws[2];
  sp.knot_first = d.knot_s[ed.kn0], sp.knot_last = d.knot_s[ed.kn0 + ed.nk - 1];
  const double2 prog = ((const double2*)(d.info64 + 8 * (size_t)e))[3 - 0];   // [6] arclength of last step, [7] spare
  sp.maxp_in = d.info64[8 * (size_t)e + 5];
  const int nch = ed.nch, P = ed.P;
  const double4* cb = d.chunk_bound + ed.c0;
  const double2* xy = d.poly_xy + ed.p0;
  double U = 1.7976931348623157e308;
  int n_act = 0;
  int cstar = -1;                                  // the chunk whose segments sA / sB / scum hold (register-resident route)
  double2 sA = make_double2(0.0, 0.0), sB = sA;
  double scum = 0.0, sdist = -1.0;
  if (nch <= NAV_CPL * AUV_WAVE) {
    double4 b[NAV_CPL];
    double cdist[NAV_CPL];
    // hint: the chunk that held the nearest point LAST step (its arclength is in INFO64; the dense polyline
    // is uniform in the spline parameter, so arclength / L locates the segment to within a few).  Its 64
    // segments are requested together with the chunk circles: the exact distance to them bounds the
    // distance to the path within centimetres, without a second trip to memory
    {
      // (only a hint -- whichever chunk it names, the survivors and the minimum are the same: a hardware reciprocal will do)
      int jh = (int)(prog.x * __builtin_amdgcn_rcp(sp.L) * (double)(P - 1));
      jh = jh < 0 ? 0 : (jh > P - 2 ? P - 2 : jh);
      cstar = jh / AUV_CHUNK;
    }
#pragma unroll
    for (int i = 0; i < NAV_CPL; i++) {
      const int c = i * AUV_WAVE + lane;
      b[i] = cb[c < nch ? c : nch - 1];
    }
    const int jhl = cstar * AUV_CHUNK + lane;
    {
      const int jj = jhl < P - 1 ? jhl : 0;
      sA = xy[jj], sB = xy[jj + 1], scum = d.poly_cum[ed.p0 + jj];   // kept if this chunk survives
    }
#ifdef AUV_NAV_CIRCLE_SQRT
#pragma unroll
    for (int i = 0; i < NAV_CPL; i++) {
      const double dx = qx - b[i].x, dy = qy - b[i].y;
      cdist[i] = sqrt(dx * dx + dy * dy);
      if (i * AUV_WAVE + lane < nch) U = fmin(U, cdist[i] + b[i].z);   // from the circles: min of |q - c| + rad
    }
#else
    // Round 4: no square roots here.  The upper bound U is the exact distance to the hint chunk's segments alone (any
    // distance to any segment of the path bounds the minimum from above; the circles' own bound |q - c| + rad was only ever
    // the tighter one when the hint was off by whole chunks), and a chunk survives iff |q - c| <= U + rad, decided on the
    // squares: (U + rad)^2 is formed with the radius that was inflated by 1e-9 (relative and absolute) at load time, six
    // orders of magnitude more than the roundings of the two squares, so every chunk that can hold the minimum survives --
    // the survivors' exact distances and the (distance, first index) minimum are what they were.  Four fp64 square roots per
    // lane less (~100 of this phase's 325 instructions, profiles/r03/valu_budget_polygons50_four_roles.json), and the
    // search record leaves earlier.  (Round 3 tried these distances in fp32 with widened comparisons: 325 -> 347
    // instructions, the conversions and margins cost what the square roots did.)
#pragma unroll
    for (int i = 0; i < NAV_CPL; i++) {
      const double dx = qx - b[i].x, dy = qy - b[i].y;
      cdist[i] = dx * dx + dy * dy;                                    // |q - c|^2
    }
#endif
    {
      double dd = 1.7976931348623157e308;
      if (jhl < P - 1) dd = auv_pt_seg_dist(qx, qy, sA.x, sA.y, sB.x, sB.y);
      sdist = dd;
      U = auv_wave_min(fmin(U, dd));
    }
#pragma unroll
    for (int i = 0; i < NAV_CPL; i++) {
      const int c = i * AUV_WAVE + lane;
#ifdef AUV_NAV_CIRCLE_SQRT
      const bool act = (c < nch) && (cdist[i] - b[i].z <= U);
#else
      const double reach = U + b[i].z;
      const bool act = (c < nch) && (cdist[i] <= reach * reach);
#endif
      const unsigned long long mask = __ballot(act);
      if (act) list[n_act + __popcll(mask & ((1ull << lane) - 1ull))] = c;
      n_act += __popcll(mask);
    }
  } else {
    for (int c = lane; c < nch; c += AUV_WAVE) {
      const double4 b = cb[c];
      const double dx = qx - b.x, dy = qy - b.y;
      U = fmin(U, sqrt(dx * dx + dy * dy) + b.z);
    }
    U = auv_wave_min(U);
    for (int cbase = 0; cbase < nch; cbase += AUV_WAVE) {
      const int c = cbase + lane;
      bool act = false;
      if (c < nch) {
        const double4 b = cb[c];
        const double dx = qx - b.x, dy = qy - b.y;
        act = (sqrt(dx * dx + dy * dy) - b.z) <= U;
      }
      const unsigned long long mask = __ballot(act);
      if (act) list[n_act + __popcll(mask & ((1ull << lane) - 1ull))] = c;
      n_act += __popcll(mask);
    }
  }
  auv_wave_lds_sync();
  sp.n_list = n_act;
  sp.in_regs = n_act <= NAV_SPEC;
#pragma unroll
  for (int q = 0; q < NAV_SPEC; q++) {
    sp.A[q] = sp.B[q] = make_double2(0.0, 0.0);
    sp.cum[q] = 0.0;
    sp.dist[q] = -1.0;
    if (sp.in_regs && q < n_act) {
      if (list[q] == cstar) {                      // (uniform) already fetched for the upper bound
        sp.A[q] = sA, sp.B[q] = sB, sp.cum[q] = scum, sp.dist[q] = sdist;
      } else {
        const int j = list[q] * AUV_CHUNK + lane;
        const int jj = j < P - 1 ? j : 0;
        sp.A[q] = xy[jj], sp.B[q] = xy[jj + 1], sp.cum[q] = d.poly_cum[ed.p0 + jj];
      }
    }
  }
  return sp;
}

// what the navigation leaves for the reward phase (lane 0; also stored to REW_PATH / INFO64)
struct NavOut {
  double rew_path, reached, goal, progress;
  double u, v, r;                   // the velocities it was computed with (STATE rows 3..5)
  double cte100;                    // cross-track error / 100 (NAV64 [5])
  double maxp;                      // INFO64 [5] after this step (the episode's maximum progress so far)
};

// the nearest segment of the path: its end points and the cumulative arclength at its first vertex (wave-uniform)
struct NavNear {
  double2 A, B;
  double cum;
};

// exact distances to the listed segments, (distance, first index) minimum over the wave
__device__ __forceinline__ NavNear nav_nearest(const AuvDev& d, const int e, const int lane, const int* list, const double px,
                                               const double py, const NavSpec& sp) {
  const EnvDesc& ed = sp.ed;
  const long long p0 = ed.p0;
  const int P = ed.P;
  const double2* xy = d.poly_xy + p0;
  // ---- exact distances to the listed segments (path.py:84-93): (distance, first index) minimum ----
  MinIdx best;
  best.d = 1.7976931348623157e308;
  best.j = 0x7fffffff;
  double2 bA = make_double2(0.0, 0.0), bB = bA;        // end points of this lane's best segment
  double my_cum = 0.0;
  const int n_act = sp.n_list;
  if (sp.in_regs) {
#pragma unroll
    for (int q = 0; q < NAV_SPEC; q++) {
      if (q < n_act) {
        const int j = list[q] * AUV_CHUNK + lane;
        if (j < P - 1) {
          // (the very same distance was formed for the upper bound already)
          const double dd = (sp.dist[q] >= 0.0) ? sp.dist[q]
                                                             : auv_pt_seg_dist(px, py, sp.A[q].x, sp.A[q].y, sp.B[q].x, sp.B[q].y);
          if (dd < best.d) best.d = dd, best.j = j, bA = sp.A[q], bB = sp.B[q], my_cum = sp.cum[q];   // ascending j: strict '<' keeps the first minimum
        }
      }
    }
  } else {
    for (int a = 0; a < n_act; a += 2) {                 // two surviving chunks per trip to memory
      const bool two = a + 1 < n_act;
      const int ja = list[a] * AUV_CHUNK + lane, jb = two ? list[a + 1] * AUV_CHUNK + lane : ja;
      const bool va = ja < P - 1, vb = two && jb < P - 1;
      const double2 A0 = xy[va ? ja : 0], B0 = xy[va ? ja + 1 : 0], A1 = xy[vb ? jb : 0], B1 = xy[vb ? jb + 1 : 0];
      if (va) {
        const double dd = auv_pt_seg_dist(px, py, A0.x, A0.y, B0.x, B0.y);
        if (dd < best.d) best.d = dd, best.j = ja, bA = A0, bB = B0;
      }
      if (vb) {
        const double dd = auv_pt_seg_dist(px, py, A1.x, A1.y, B1.x, B1.y);
        if (dd < best.d) best.d = dd, best.j = jb, bA = A1, bB = B1;
      }
    }
    // cumulative length at this lane's candidate, requested while the reduction runs
    my_cum = d.poly_cum[p0 + (best.j < P - 1 ? best.j : 0)];
  }
  // (distance, first index) minimum over the wave: the smallest distance first; among the lanes that hold it
  // (almost always one; a shared vertex gives two) the smallest segment index -- "first minimum wins"
  const double dmin = auv_wave_min(best.d);
  unsigned long long wmask = __ballot(best.d == dmin);
  if (__popcll(wmask) > 1) {
    int jm = (best.d == dmin) ? best.j : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int t = __shfl_xor(jm, o, AUV_WAVE);
      jm = t < jm ? t : jm;
    }
    wmask = __ballot(best.d == dmin && best.j == jm);
  }
  // the winning lane hands over its segment (no second trip to memory)
  // (the lane's number is wave-uniform: scalar reads of its registers, no trip through the LDS crossbar)
  const int src = auv_uniform(wmask ? __ffsll((long long)wmask) - 1 : 0);
  NavNear nr;
  nr.A.x = auv_readlane_f64(bA.x, src), nr.A.y = auv_readlane_f64(bA.y, src);
  nr.B.x = auv_readlane_f64(bB.x, src), nr.B.y = auv_readlane_f64(bB.y, src);
  nr.cum = auv_readlane_f64(my_cum, src);
  return nr;
}

// ---- the navigation's scalar tail (vessel.py:471-515, environment.py:276-280, rewarder.py:110-118) ----
// What the tail needs of one environment: the new state, the nearest segment, the per-world constants.
struct TailIn {
  double px, py, psi, u, v, r;
  NavNear nr;
  long long kn0;                     // EnvDesc: first knot, number of knots
  int nk;
  double L, goal_x, goal_y, knot_first, knot_last, maxp_in;
};

// A GROUP of GW lanes (GW = 64: the whole wave, one environment per wave; GW = 8: eight environments per wave, the
// finish role of the one-launch step) evaluates the tail of environment e, c = lane % GW: lane 0 of the group the
// spline at s, lanes 1 and 2 at the look-ahead arclength, one atan2 serving all three angles; lanes 3 / 4 / 5 take the
// tail's other square roots and its quotient along in the same instructions.  `e` and `t` are uniform over the group
// (over the wave for GW = 64: the compiler keeps them in scalar registers then).  The same source for both widths, so
// the same operations on the same operands: bit-identical whichever shape ran.  Group lane 0 stores the rows and
// returns the NavOut.
template <int GW, class Desc>
__device__ __forceinline__ NavOut nav_tail(const Desc& d, const int e, const int c, const bool live, const TailIn& t,
                                           float* __restrict__ obs_out) {
  const int S = d.cfg.n_sensors;
  const double px = t.px, py = t.py, psi = t.psi, u = t.u, v = t.v, r = t.r;
  const double L = t.L;
  double* inf = d.info64 + 8 * (size_t)e;
  double* nv = d.nav64 + 8 * (size_t)e;
  double* ob = d.obs64 + (size_t)e * (6 + S);
  const int D = 6 + (d.cfg.use_lidar ? S * (d.cfg.obs_channels == 3 ? 3 : 1) : 0);   // row stride of obs_out
  const double2 A = t.nr.A, B = t.nr.B;
  const double cum = t.nr.cum;
  NavOut out;
  out.rew_path = out.reached = out.goal = out.progress = out.u = out.v = out.r = out.cte100 = out.maxp = 0.0;
  // cos / sin of the new heading for the reward's cos(heading error) below
  double sin_psi, cos_psi;
  sincos(psi, &sin_psi, &cos_psi);
  // measure along the polyline: LengthIndexOfPoint::segmentNearestMeasure (all lanes of the group alike)
  double dx = B.x - A.x, dy = B.y - A.y, len2 = dx * dx + dy * dy;
  double seglen = sqrt(len2);
  double pf = (len2 == 0.0) ? 0.0 : ((px - A.x) * dx + (py - A.y) * dy) / len2;
  const double s = pf <= 0.0 ? cum : (pf <= 1.0 ? cum + pf * seglen : cum + seglen);
  double s_t = s + d.cfg.look_ahead_distance;
  if (L < s_t) s_t = L;
  // vessel.py:471-515 -- lane 0 evaluates the spline at s, lanes 1 and 2 at s_t (same instructions);
  // one atan2 then serves all three angles: lane 0 chi, lane 1 the look-ahead direction, lane 2
  // the heading towards the look-ahead point
  double p[2], dp[2];
  path_eval(d, t.kn0, t.nk, t.knot_first, t.knot_last, c == 0 ? s : s_t, L, p, dp);
  // The wave issues the square root and the division below once whatever the number of busy lanes, so the lanes behind
  // the three angle lanes take the tail's other square roots and its quotient along in the same instructions (VERDICT r2
  // #2: the tail is 19 % of the step's VALU issue cycles): lane 3 the distance to the goal, |goal - p|; lane 4 the speed,
  // |(u, v)|; lane 5 the progress s / L.  Same operands, same operations, so the same bits as when lane 0 formed them.
  double ang_y = (c == 2) ? p[1] - py : dp[1], ang_x = (c == 2) ? p[0] - px : dp[0];
  if (c == 3) ang_x = t.goal_x - px, ang_y = t.goal_y - py;
  if (c == 4) ang_x = u, ang_y = v;
  if (c == 5) ang_x = s, ang_y = 0.0;
  const double dir = atan2(ang_y, ang_x);
  // unit vector of (ang_x, ang_y): lane 0 cos / sin of chi, lane 2 of the direction to the look-ahead
  // point -- formed beside the atan2, not from it, so that the cross-track error and the reward's
  // cos(heading error) do not wait for the angle (sin(atan2(y, x)) = y / |(x, y)| to rounding)
  const double root = sqrt(ang_x * ang_x + ang_y * ang_y);
  const double hyp = (c == 5) ? L : root;
  const double ux = (hyp > 0.0 || c == 5) ? ang_x / hyp : 1.0, uy = hyp > 0.0 ? ang_y / hyp : 0.0;   // atan2(0, 0) = 0
  const double la_dir = __shfl(dir, 1, GW), tgt1 = __shfl(dir, 2, GW);
  const double tx = __shfl(ux, 2, GW), ty = __shfl(uy, 2, GW);
  const double goal_l = __shfl(root, 3, GW), speed_l = __shfl(root, 4, GW), progress_l = __shfl(ux, 5, GW);
  if (c == 0 && live) {
    const double chi = dir;
    double ddx = p[0] - px, ddy = p[1] - py;
    double cte = -uy * ddx + ux * ddy;                         // vessel.py:481-483: -sin(chi) dx + cos(chi) dy
    double la = auv_princip(la_dir - psi);
    double he = auv_princip(tgt1 - psi);
    const double cos_he = tx * cos_psi + ty * sin_psi;        // cos(target direction - psi)
    const double progress = progress_l;                        // s / L
    double maxp = t.maxp_in;
    if (progress > maxp) maxp = progress;
    const double goal = goal_l;                                // sqrt(gx gx + gy gy), g = goal - p
    int reached = (goal <= d.cfg.min_goal_distance) || (progress >= d.cfg.min_path_progress);
    // rows are written with 16-byte stores (NAV64 / INFO64 rows are 64-byte records)
    const double cte100 = cte / 100;
    double2* nv2 = (double2*)nv;
    nv2[0] = make_double2(u, v), nv2[1] = make_double2(r, la), nv2[2] = make_double2(he, cte100), nv2[3] = make_double2(chi, s_t);
    double2* inf2 = (double2*)inf;
    inf[1] = (double)reached, inf2[1] = make_double2(goal, progress), inf[5] = maxp,
    inf[6] = s;                                                // ([7]: the episode's running sum of |cross-track error|, kept by the reward phase)
    // path-following term of the reward: everything it needs is at hand here, so the
    // transcendentals stay out of the reward phase
    const double rew_path = reward_path_term_cos(d, speed_l, cos_he, cte100, progress, maxp);
    d.rew_path[e] = rew_path;
    out.rew_path = rew_path, out.reached = reached, out.goal = goal, out.progress = progress, out.u = u, out.v = v, out.r = r, out.cte100 = cte100;
    out.maxp = maxp;
    // environment.py:276-280; lane 0 also emits the float32 copies of its own six values.  OBS64 rows
    // start 16-byte aligned when 6 + S is even, float32 rows 8-byte aligned when their stride is even.
    const double c0 = auv_clip(u, -1.0, 1.0), c1 = auv_clip(v, -1.0, 1.0), c2 = auv_clip(r, -1.0, 1.0),
                 c3 = auv_clip(la, -1.0, 1.0), c4 = auv_clip(he, -1.0, 1.0), c5 = auv_clip(cte100, -1.0, 1.0);
    if (((6 + S) & 1) == 0) {
      double2* ob2 = (double2*)ob;
      ob2[0] = make_double2(c0, c1), ob2[1] = make_double2(c2, c3), ob2[2] = make_double2(c4, c5);
    } else {
      ob[0] = c0, ob[1] = c1, ob[2] = c2, ob[3] = c3, ob[4] = c4, ob[5] = c5;
    }
    if (obs_out) {
      float* oo = obs_out + (size_t)e * D;
      if ((D & 1) == 0) {
        float2* oo2 = (float2*)oo;
        oo2[0] = make_float2((float)c0, (float)c1), oo2[1] = make_float2((float)c2, (float)c3), oo2[2] = make_float2((float)c4, (float)c5);
      } else {
        oo[0] = (float)c0, oo[1] = (float)c1, oo[2] = (float)c2, oo[3] = (float)c3, oo[4] = (float)c4, oo[5] = (float)c5;
      }
    }
  }
  return out;
}

// nearest segment, then the tail by the whole wave.  `out`: lane 0's NavOut, or nullptr.
__device__ __forceinline__ void nav_finish(const AuvDev& d, const int e, const int lane, int* list,
                                           float* __restrict__ obs_out, const EnvPre* pre, const NavSpec& sp, NavOut* out = nullptr) {
  const size_t n = (size_t)d.n;
  TailIn t;
  t.px = pre ? pre->s[0] : d.state[0 * n + e], t.py = pre ? pre->s[1] : d.state[1 * n + e], t.psi = pre ? pre->s[2] : d.state[2 * n + e];
  t.u = pre ? pre->s[3] : d.state[3 * n + e], t.v = pre ? pre->s[4] : d.state[4 * n + e], t.r = pre ? pre->s[5] : d.state[5 * n + e];
  AUV_STAMP_DECL
  AUV_STAMP()
  AUV_STAMP()
  t.nr = nav_nearest(d, e, lane, list, t.px, t.py, sp);
  AUV_STAMP()
  if (!AUV_RUN_N(d, 3)) return;
  t.kn0 = sp.ed.kn0, t.nk = sp.ed.nk;
  t.L = sp.L, t.goal_x = sp.goal_x, t.goal_y = sp.goal_y, t.knot_first = sp.knot_first, t.knot_last = sp.knot_last, t.maxp_in = sp.maxp_in;
  const NavOut no = nav_tail<AUV_WAVE>(d, e, lane, true, t, obs_out);
  if (out && lane == 0) *out = no;
  AUV_STAMP()
  AUV_STAMP_FLUSH(e, 8)   // 8:bounds 9:list 10:scan 11:nav
}

// the navigation of one environment.  `scratch`: NAV_SCRATCH_BYTES(nch_max) of LDS, 16-byte aligned: the list of
// surviving chunks.
#define NAV_SCRATCH_BYTES(nch_max) (((size_t)(nch_max) * sizeof(int) + 15) & ~(size_t)15)
__device__ __forceinline__ void k3_nav_env(const AuvDev& d, const int e, const int lane, unsigned char* scratch,
                                           float* __restrict__ obs_out, const EnvPre* pre = nullptr, NavOut* out = nullptr) {
  int* list = (int*)scratch;
  const size_t n = (size_t)d.n;
  const NavSpec sp = nav_bounds(d, e, lane, list, pre ? pre->s[0] : d.state[0 * n + e], pre ? pre->s[1] : d.state[1 * n + e],
                                pre ? pre->ed : nullptr);
  if (!AUV_RUN_N(d, 2)) return;
  nav_finish(d, e, lane, list, obs_out, pre, sp, out);
}

// ---- reward + done + bookkeeping part; needs K2's ranges/collision and the nav part's outputs ----
// full = false: only publish the collision flag and the float32 LiDAR observations (reset path)
// from_buffers: form the two reward terms here from NAV64 / INFO64 / LIDAR_D as they stand (the
//               per-function test hook) instead of taking what K2 / the navigation phase left
// lidar_obs:    emit the float32 closeness columns (false when K2 has already written them)
// One environment's reward / done / bookkeeping (rewarder.py:78-140, :167-241; environment.py:333-347, :375-384),
// scalar: by lane 0 of the environment's wave, or by the environment's lane of k3_reward_lanes.  The LiDAR term
// was formed by K2 (rew_lidar), the path-following term by the navigation phase (rew_path); here they are only
// combined.  `cnt` in/out; returns whether the environment is to be auto-reset.
// the values the block works on, wherever they came from
struct RewardIn {
  double u, v, yaw_rate;            // vessel velocities after the step
  double path_reward, closeness_reward;
  double reached, goal, progress;   // INFO64 [1] [2] [3] of this step
  double cum;                       // cumulative reward before this step
  double cte100, cte_sum;           // NAV64 [5] of this step; INFO64 [7] before it (sum of |cross-track error| so far)
};

// `D`: AuvDev or StepTabs (see there)
template <class D>
// `carry_out` (nullable): [0] the cumulative reward, [1] the sum of |cross-track error| after this step (k_step_multi)
__device__ __forceinline__ int reward_apply(const D& d, const int e, const int collision, int4& cnt, const RewardIn in,
                                            float* __restrict__ reward_out, uint8_t* __restrict__ done_out,
                                            const bool advance_ring, double* carry_out = nullptr) {
  double* inf = d.info64 + 8 * (size_t)e;
  const bool colav = d.cfg.rewarder == AUV_REWARD_COLAV;
  const double lambda = 0.5, eta = 0.0, penalty_yawrate = 10.0, neutral_speed = 0.05, max_speed = 2.0;
  const double u = in.u, v = in.v, yaw_rate = in.yaw_rate;
  const double cum_in = in.cum, reached_in = in.reached, goal_in = in.goal, progress_in = in.progress;
  double reward;
  if (collision) {
    reward = -10000.0 * (1 - lambda);
  } else {
    const double speed = sqrt(u * u + v * v);
    const double path_reward = in.path_reward;
    const double living_penalty = lambda * (2 * neutral_speed + 1) + eta * neutral_speed;
    if (!colav) {
      double slow_penalty = (speed < 0.1) ? -2 : 0;
      reward = path_reward - living_penalty + eta * speed / max_speed - penalty_yawrate * fabs(yaw_rate) +
               slow_penalty;
    } else {
      const double closeness_reward = in.closeness_reward;
      double slow_penalty = (speed < 0.04) ? -2 : 0;
      reward = lambda * path_reward + (1 - lambda) * closeness_reward - living_penalty +
               eta * speed / max_speed - penalty_yawrate * fabs(yaw_rate) + slow_penalty;
      if (reward < 0) reward *= 2.0;
    }
  }
  // ---- environment.py:333-347, :375-384 ----
  d.reward64[e] = reward;
  double cum = cum_in + reward;
  inf[4] = cum;
  // environment.py:345, :460-464 (_save_latest_step): |cross-track error| in metres of every step, for the episode's mean
  const double cte_sum = in.cte_sum + fabs(in.cte100) * 100;
  inf[7] = cte_sum;
  if (carry_out) carry_out[0] = cum, carry_out[1] = cte_sum;
  const int t_step = cnt.x;
  const int done = collision || (reached_in != 0.0) || (t_step >= d.cfg.max_timesteps - 1 && !d.cfg.test_mode) ||
                   (cum < d.cfg.min_cumulative_reward && !d.cfg.test_mode);
  cnt.x = t_step + 1;
  {
    double2* si = (double2*)(d.step_info + 4 * (size_t)e);   // environment.py:336-340
    si[0] = make_double2(collision, reached_in), si[1] = make_double2(goal_in, progress_in);
  }
  if (reward_out) reward_out[e] = (float)reward;
  if (done_out) done_out[e] = (uint8_t)done;
  if (done) {
    double2* ep = (double2*)(d.episode + 4 * (size_t)e);
    ep[0] = make_double2(cum, t_step + 1), ep[1] = make_double2(collision, reached_in);
    cnt.z += 1;
    // episode log (environment.py:466-489 save_latest_episode: what the reference appends to env.history), a ring of
    // 64-byte records filled in completion order: env, return, timesteps, collision, reached_goal, progress,
    // mean |cross-track error|, world
    const unsigned long long slot = atomicAdd(d.ep_log_count, 1ull);        // (64-bit: never wraps; cap is a power of two)
    double2* row = (double2*)(d.ep_log + 8 * (size_t)(slot & (unsigned long long)(d.ep_log_cap - 1)));
    row[0] = make_double2((double)e, cum), row[1] = make_double2(t_step + 1, collision), row[2] = make_double2(reached_in, progress_in);
    // the world column: the bank's index -- with a fresh world per reset the index the world WOULD have in a bank that never
    // repeats: environment + N * serial (serial: the how-manieth world of this environment), so that no two episodes of a run
    // may ever log the same number
    const int wl = d.world_idx[e];
    row[3] = make_double2(cte_sum / (double)(t_step + 1), d.fw_serial ? (double)e + (double)d.n * (double)d.fw_serial[wl] : (double)wl);
  }
  const int do_reset = done && d.cfg.auto_reset;
  if (!do_reset) d.counters[e] = cnt;
  // next action slot of a captured graph's ring
  // (ring_slot_host: >= 0 the host names the slot; -1 the device position, advanced here; -2 the device position,
  // advanced by another kernel of the step -- the side-by-side shape does it in k23_lidar_nav, so that a reward phase
  // fused with the NEXT step's dynamics never races with its own readers)
  if (e == 0 && d.ring_slots > 1 && d.ring_slot_host == -1 && advance_ring) *d.ring_pos = (*d.ring_pos + 1) % d.ring_slots;
  return do_reset;
}

__device__ __forceinline__ int reward_block(const AuvDev& d, const int e, const int collision, int4& cnt,
                                            const bool from_buffers, const double lidar_term, const double* rew_lidar_pre,
                                            float* __restrict__ reward_out, uint8_t* __restrict__ done_out,
                                            const bool advance_ring) {
  const double* inf = d.info64 + 8 * (size_t)e;
  const double* nv = d.nav64 + 8 * (size_t)e;
  // everything the block reads is requested up front (one trip to memory, whatever branch follows)
  const double2 uv = ((const double2*)nv)[0];           // rows are 64-byte records: 16-byte accesses
  RewardIn in;
  in.u = uv.x, in.v = uv.y, in.yaw_rate = nv[2];
  // (the LiDAR term arrives in a register when the sweep ran in this very wave)
  const double rew_path_in = d.rew_path[e], rew_lidar_in = rew_lidar_pre ? *rew_lidar_pre : d.rew_lidar[e];
  const double2 gp = ((const double2*)inf)[1];
  in.cum = inf[4], in.reached = inf[1], in.goal = gp.x, in.progress = gp.y;
  in.cte100 = nv[5], in.cte_sum = inf[7];
  in.path_reward = rew_path_in, in.closeness_reward = rew_lidar_in;
  if (!collision) {
    if (from_buffers) in.path_reward = reward_path_term(d, in.u, in.v, nv[4], nv[5], inf[3], inf[5]);
    if (from_buffers || !d.cfg.use_lidar) in.closeness_reward = lidar_term;
  }
  return reward_apply(d, e, collision, cnt, in, reward_out, done_out, advance_ring);
}

__device__ void k3_reward_env(const AuvDev& d, const int e, const int lane, const bool full,
                              float* __restrict__ obs_out, float* __restrict__ reward_out,
                              uint8_t* __restrict__ done_out, const EnvPre* pre = nullptr, const int collision_pre = -1,
                              const bool from_buffers = false, const bool lidar_obs = true,
                              const double* rew_lidar_pre = nullptr) {
  const int S = d.cfg.n_sensors;
  int4 cnt = pre ? pre->cnt : d.counters[e];
  const int w = d.world_idx[e];
  double* inf = d.info64 + 8 * (size_t)e;
  const double* ob = d.obs64 + (size_t)e * (6 + S);
  const int D = 6 + (d.cfg.use_lidar ? S * (d.cfg.obs_channels == 3 ? 3 : 1) : 0);   // row stride of obs_out
  const int DL = 6 + (d.cfg.use_lidar ? S : 0);                                        // columns this path writes
  const int collision = collision_pre >= 0 ? collision_pre : d.collision[e];
  if (lane == 0) inf[0] = collision;
  if (full) {
    const bool colav = d.cfg.rewarder == AUV_REWARD_COLAV;
    // without a LiDAR sweep the ranges rest at sensor_range and K2 leaves no term: form it here
    double lidar_term = 0.0;
    if (colav && (from_buffers || !d.cfg.use_lidar)) lidar_term = reward_lidar_term_wave(d, d.lidar_d + (size_t)e * S, lane);
    int do_reset = 0;
    if (lane == 0) do_reset = reward_block(d, e, collision, cnt, from_buffers, lidar_term, rew_lidar_pre, reward_out, done_out, pre == nullptr);
    do_reset = __shfl(do_reset, 0, AUV_WAVE);
    if (do_reset) {
      // VecEnv auto-reset: rebind to the next world of the bank and copy its reset rows
      restore_env(d, e, auv_next_world_wave(d, auv_uniform(w), lane), lane, __shfl(cnt.z, 0, AUV_WAVE), obs_out);
      return;
    }
  }
  // ---- LiDAR part of the float32 observation row (closeness written in fp64 by K2) ----
  if (obs_out && lidar_obs)
    for (int i = 6 + lane; i < DL; i += AUV_WAVE) obs_out[(size_t)e * D + i] = (float)ob[i];
}

#ifndef AUV_DEVICE_FUNCS_ONLY
// mode 0: navigate + observe + reward + done (+ auto-reset bookkeeping)
// mode 1: navigate + observe only (reset path)
// mode 2: reward + done only, from the buffers as they stand (test hook)
__global__ void __launch_bounds__(AUV_BLOCK) k3_nav_reward(AuvDev d, int mode, float* __restrict__ obs_out,
                                                           float* __restrict__ reward_out,
                                                           uint8_t* __restrict__ done_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int wave = threadIdx.x / AUV_WAVE, lane = threadIdx.x % AUV_WAVE;
  const int e = auv_uniform(blockIdx.x * AUV_ENVS_PER_BLOCK + wave);
  if (e >= d.n) return;
  if (mode != 2) k3_nav_env(d, e, lane, smem + (size_t)wave * NAV_SCRATCH_BYTES(d.nch_max), obs_out);
  k3_reward_env(d, e, lane, mode != 1, mode == 2 ? nullptr : obs_out, reward_out, done_out, nullptr, -1, mode == 2);
}

// the two halves as separate kernels for the step path (k3_nav overlaps with K2)
__global__ void __launch_bounds__(AUV_BLOCK) k3_nav(AuvDev d, float* __restrict__ obs_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int wave = threadIdx.x / AUV_WAVE, lane = threadIdx.x % AUV_WAVE;
  const int e = auv_uniform(blockIdx.x * AUV_ENVS_PER_BLOCK + wave);
  if (e >= d.n) return;
  k3_nav_env(d, e, lane, smem + (size_t)wave * NAV_SCRATCH_BYTES(d.nch_max), obs_out);
}

__global__ void __launch_bounds__(AUV_BLOCK) k3_reward(AuvDev d, float* __restrict__ obs_out,
                                                       float* __restrict__ reward_out,
                                                       uint8_t* __restrict__ done_out, int lidar_obs) {
  const int wave = threadIdx.x / AUV_WAVE, lane = threadIdx.x % AUV_WAVE;
  const int el = blockIdx.x * AUV_ENVS_PER_BLOCK + wave;
  if (el >= d.ne) return;
  const int e = auv_uniform(d.e0 + el);
  k3_reward_env(d, e, lane, true, obs_out, reward_out, done_out, nullptr, -1, false, lidar_obs != 0);
}

// The same for the step path when the LiDAR launch has already written the float32 closeness columns: lanes <->
// environments (64 per wave instead of one), so the launch is 64x fewer waves -- at 8192 environments the
// one-wave-per-environment form no longer fits the chip in one round.  Environments that ended are then
// restored by the whole wave, one after the other (a handful per step in the whole batch).
__global__ void __launch_bounds__(AUV_WAVE) k3_reward_lanes(AuvDev d, float* __restrict__ obs_out,
                                                            float* __restrict__ reward_out,
                                                            uint8_t* __restrict__ done_out) {
  const int lane = threadIdx.x;
  const int el = blockIdx.x * AUV_WAVE + lane;
  const int e = d.e0 + el;
  int do_reset = 0, w = 0;
  int4 cnt = make_int4(0, 0, 0, 0);
  if (el < d.ne) {
    cnt = d.counters[e];
    w = d.world_idx[e];
    const int collision = d.collision[e];
    d.info64[8 * (size_t)e] = collision;
    do_reset = reward_block(d, e, collision, cnt, false, 0.0, nullptr, reward_out, done_out, true);
  }
  unsigned long long m = __ballot(do_reset);
  while (m) {
    const int src = __ffsll((long long)m) - 1;
    m &= m - 1;
    const int er = auv_uniform(__shfl(e, src, AUV_WAVE));
    const int wr = auv_uniform(__shfl(w, src, AUV_WAVE)), ep = __shfl(cnt.z, src, AUV_WAVE);
    restore_env(d, er, auv_next_world_wave(d, wr, lane), lane, ep, obs_out);
  }
}

// reset pass: first observation of the environments on the fresh list
__global__ void __launch_bounds__(AUV_BLOCK) k3_observe_fresh(AuvDev d, float* __restrict__ obs_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int wave = threadIdx.x / AUV_WAVE, lane = threadIdx.x % AUV_WAVE;
  const int nf = *d.fresh_count;
  unsigned char* list = smem + (size_t)wave * NAV_SCRATCH_BYTES(d.nch_max);
  for (int i = blockIdx.x * AUV_ENVS_PER_BLOCK + wave; i < nf; i += gridDim.x * AUV_ENVS_PER_BLOCK) {
    const int e = auv_uniform(d.fresh_list[i]);
    k3_nav_env(d, e, lane, list, obs_out);
    k3_reward_env(d, e, lane, false, obs_out, nullptr, nullptr);
    auv_wave_lds_sync();
  }
}

// reset(): restore reset-time state (and the precomputed first observation) for masked envs
__global__ void __launch_bounds__(AUV_BLOCK) k_reset(AuvDev d, const uint8_t* __restrict__ mask,
                                                     const int32_t* __restrict__ world_idx,
                                                     float* __restrict__ obs_out) {
  const int wave = threadIdx.x / AUV_WAVE, lane = threadIdx.x % AUV_WAVE;
  const int e = auv_uniform(blockIdx.x * AUV_ENVS_PER_BLOCK + wave);
  if (e >= d.n) return;
  if (mask && !mask[e]) return;
  int w = world_idx ? world_idx[e] : d.world_idx[e];
  if (w < 0 || w >= d.n_worlds) w = d.world_idx[e];        // an out-of-range request keeps the current binding
  const int4 cnt = d.counters[e];
  // fresh worlds: reset() of an environment that has stepped in its world moves it on to its next, unseen one (the
  // reference's reset() always builds a new scenario, environment.py:176-218); one that has not stepped keeps the world it
  // has -- nobody has seen it.  (auv_reset refuses an explicit world_idx in that mode.)
  if (d.fw_state && cnt.x > 0) w = auv_next_world_wave(d, auv_uniform(w), lane);
  restore_env(d, e, w, lane, cnt.z, obs_out);
}

// refill pass of the fresh-world mode: the first *count_dev SHADOW environments (a handful of environments nobody steps,
// `d` is their descriptor: w_ready == 0) are put into the reset state of the slots the bind kernel has just bound them to --
// onto the fresh list, from which the fresh-list kernels of K2 / K3 compute their first observation
__global__ void __launch_bounds__(AUV_BLOCK) k_fw_shadow_reset(AuvDev d, const int32_t* __restrict__ count_dev) {
  const int wave = threadIdx.x / AUV_WAVE, lane = threadIdx.x % AUV_WAVE;
  const int e = auv_uniform(blockIdx.x * AUV_ENVS_PER_BLOCK + wave);
  if (e >= *count_dev || e >= d.n) return;
  restore_env(d, e, d.world_idx[e], lane, 0, nullptr);
}

// load-time pass: after K2/K3 produced the reset observation of the worlds currently bound to
// the first `count` env slots, keep those rows per world
__global__ void __launch_bounds__(AUV_BLOCK) k_harvest(AuvDev d, int count, const int32_t* __restrict__ count_dev) {
  const int wave = threadIdx.x / AUV_WAVE, lane = threadIdx.x % AUV_WAVE;
  const int e = auv_uniform(blockIdx.x * AUV_ENVS_PER_BLOCK + wave);
  if (count_dev) count = *count_dev;                       // (the refill pass of the fresh-world mode: known on the device only)
  if (e >= count) return;
  const int S = d.cfg.n_sensors;
  const int w = d.world_idx[e];
  for (int i = lane; i < S; i += AUV_WAVE) d.w_lidar[(size_t)w * S + i] = d.lidar_d[(size_t)e * S + i];
  for (int i = lane; i < 6 + S; i += AUV_WAVE) d.w_obs64[(size_t)w * (6 + S) + i] = d.obs64[(size_t)e * (6 + S) + i];
  if (lane < 8) d.w_info[8 * (size_t)w + lane] = d.info64[8 * (size_t)e + lane];
  else if (lane < 16) d.w_nav[8 * (size_t)w + lane - 8] = d.nav64[8 * (size_t)e + lane - 8];
  for (int k = lane; k < d.k_max; k += AUV_WAVE) {
    d.w_nearby[(size_t)w * d.k_max + k] = d.nearby[(size_t)e * d.k_max + k];
    d.w_limits[(size_t)w * d.k_max + k] = d.limits[(size_t)e * d.k_max + k];
  }
  if (lane == 0) d.w_collision[w] = d.collision[e];
}

#endif

}  // namespace

#ifndef AUV_DEVICE_FUNCS_ONLY
static size_t k3_lds_bytes(const AuvDev& d) { return (size_t)AUV_ENVS_PER_BLOCK * NAV_SCRATCH_BYTES(d.nch_max); }
static int env_grid(const AuvDev& d) { return (d.n + AUV_ENVS_PER_BLOCK - 1) / AUV_ENVS_PER_BLOCK; }

void auv_launch_k3(const AuvDev& d, int mode, float* obs, float* reward, uint8_t* done, hipStream_t st) {
  hipLaunchKernelGGL(k3_nav_reward, dim3(env_grid(d)), dim3(AUV_BLOCK), k3_lds_bytes(d), st, d, mode, obs, reward, done);
}

void auv_launch_k3_nav(const AuvDev& d, float* obs, hipStream_t st) {
  hipLaunchKernelGGL(k3_nav, dim3(env_grid(d)), dim3(AUV_BLOCK), k3_lds_bytes(d), st, d, obs);
}

// lidar_obs = 0: the LiDAR launch before it has written the float32 closeness columns itself
void auv_launch_k3_reward(const AuvDev& d, float* obs, float* reward, uint8_t* done, int lidar_obs, hipStream_t st,
                          hipEvent_t ev0, hipEvent_t ev1) {
  if (d.cfg.use_lidar && !lidar_obs)   // the step path behind a LiDAR launch that wrote the float32 closeness itself
    hipExtLaunchKernelGGL(k3_reward_lanes, dim3((d.ne + AUV_WAVE - 1) / AUV_WAVE), dim3(AUV_WAVE), 0, st, ev0, ev1, 0, d, obs, reward, done);
  else
    hipExtLaunchKernelGGL(k3_reward, dim3((d.ne + AUV_ENVS_PER_BLOCK - 1) / AUV_ENVS_PER_BLOCK), dim3(AUV_BLOCK), 0, st, ev0, ev1, 0, d, obs, reward, done, lidar_obs);
}

void auv_launch_k3_fresh(const AuvDev& d, float* obs, hipStream_t st) {
  int grid = env_grid(d) < AUV_FRESH_GRID ? env_grid(d) : AUV_FRESH_GRID;
  hipLaunchKernelGGL(k3_observe_fresh, dim3(grid), dim3(AUV_BLOCK), k3_lds_bytes(d), st, d, obs);
}

void auv_launch_reset(const AuvDev& d, const uint8_t* mask, const int32_t* world_idx, float* obs, hipStream_t st) {
  hipLaunchKernelGGL(k_reset, dim3(env_grid(d)), dim3(AUV_BLOCK), 0, st, d, mask, world_idx, obs);
}

namespace {
// after a caller has overwritten WORLD_IDX directly: bring the descriptors back in step
__global__ void k_refresh_desc(AuvDev d) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= d.n) return;
  int w = d.world_idx[e];
  if (w < 0 || w >= d.n_worlds) d.world_idx[e] = w = d.env_desc[e].w;
  d.env_desc[e] = auv_make_desc(d, w);
}

__global__ void k_ring_advance(AuvDev d) {
  if (threadIdx.x == 0 && d.ring_slots > 1) *d.ring_pos = (*d.ring_pos + 1) % d.ring_slots;
}
}  // namespace

void auv_launch_derive(const AuvDev& d, hipStream_t st) { hipLaunchKernelGGL(k_derive, dim3(1), dim3(AUV_WAVE), 0, st, d); }

void auv_launch_refresh_desc(const AuvDev& d, hipStream_t st) {
  hipLaunchKernelGGL(k_refresh_desc, dim3((d.n + 255) / 256), dim3(256), 0, st, d);
}

void auv_launch_ring_advance(const AuvDev& d, hipStream_t st) { hipLaunchKernelGGL(k_ring_advance, dim3(1), dim3(64), 0, st, d); }

void auv_launch_harvest(const AuvDev& d, int count, hipStream_t st, const int32_t* count_dev) {
  hipLaunchKernelGGL(k_harvest, dim3((count + AUV_ENVS_PER_BLOCK - 1) / AUV_ENVS_PER_BLOCK), dim3(AUV_BLOCK), 0, st, d, count, count_dev);
}

void auv_launch_fw_shadow_reset(const AuvDev& d, const int32_t* count_dev, hipStream_t st) {
  hipLaunchKernelGGL(k_fw_shadow_reset, dim3(env_grid(d)), dim3(AUV_BLOCK), 0, st, d, count_dev);
}
#endif
