// K3 — path navigation + reward + episode termination + observation assembly:
// one 256-thread workgroup per environment.
//
// Reference: Vessel.navigate                  gym_auv/objects/vessel/vessel.py:461-541
//            Path.get_closest_arclength       objects/path.py:84-93  (GEOS LineString.project:
//                                             first strict minimum of point-segment distance)
//            Path.__call__/get_direction      objects/path.py:61-82  (SciPy PPoly evaluation)
//            ColavRewarder.calculate          objects/rewarder.py:167-241
//            PathFollowRewarder.calculate     objects/rewarder.py:78-140
//            BaseEnvironment.observe/step/_isdone  environment.py:247-290, 325-347, 375-384
//
// The projection streams the environment's dense polyline (P ~ 10 L vertices, fp64 x,y
// interleaved = one 16-B load per lane, 1 KiB per wave-instruction) and min-reduces
// (distance, first index) with wave shuffles, then one lane evaluates the spline and the
// scalar logic.  The Colav closeness term is a block reduction over the S beams.
// Roofline: HBM.  Algorithmic bytes per env-step: 16*P (polyline) + 8*S (d in) + 8*(6+S)
// (obs64 r/w) + 4*(6+S) (obs f32 out) + ~400 (knot rows, scalars, info/nav/counters).
#include "auv_device.h"

namespace {

struct MinIdx {
  double d;
  int j;
};

__device__ __forceinline__ MinIdx min_first(MinIdx a, MinIdx b) {
  // strict '<' with ascending traversal == "first minimum wins"; ties -> smaller index
  if (b.d < a.d || (b.d == a.d && b.j < a.j)) return b;
  return a;
}

__device__ __forceinline__ MinIdx wave_min_first(MinIdx v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    MinIdx t;
    t.d = __shfl_xor(v.d, o, AUV_WAVE);
    t.j = __shfl_xor(v.j, o, AUV_WAVE);
    v = min_first(v, t);
  }
  return v;
}

// SciPy PPoly: interval search (knots are near-uniform: guess then walk) + power-basis eval
__device__ __forceinline__ void path_eval(const AuvDev& d, int w, double s, double L, double xy[2], double dxy[2]) {
  const long long k0 = d.knot_off[w];
  const int nk = (int)(d.knot_off[w + 1] - k0);
  const double* x = d.knot_s + k0;
  int i;
  if (!(s >= x[0])) {
    i = 0;
  } else if (s >= x[nk - 1]) {
    i = nk - 2;
  } else {
    i = (int)(s / L * (nk - 1));
    i = i < 0 ? 0 : (i > nk - 2 ? nk - 2 : i);
    while (i > 0 && s < x[i]) i--;
    while (i < nk - 2 && s >= x[i + 1]) i++;
  }
  const double* c = d.knot_coef + 8 * (k0 + i);
  const double z = s - x[i], z2 = z * z;
#pragma unroll
  for (int a = 0; a < 2; a++) {
    const double* ca = c + 4 * a;
    xy[a] = ((ca[3] + ca[2] * z) + ca[1] * z2) + ca[0] * (z2 * z);
    dxy[a] = (ca[2] + (2.0 * ca[1]) * z) + (3.0 * ca[0]) * z2;
  }
}

// mode 0: navigate + observe + reward + done (+ auto-reset bookkeeping)
// mode 1: navigate + observe only (reset path)
// mode 2: reward + done only, from the buffers as they stand (test hook)
__global__ void __launch_bounds__(AUV_BLOCK) k3_nav_reward(AuvDev d, int mode, int only_fresh,
                                                           float* __restrict__ obs_out,
                                                           float* __restrict__ reward_out,
                                                           uint8_t* __restrict__ done_out) {
  __shared__ MinIdx s_min[AUV_BLOCK / AUV_WAVE];
  __shared__ double s_sum[2][AUV_BLOCK / AUV_WAVE];
  __shared__ int s_reset;
  const int e = blockIdx.x, tid = threadIdx.x;
  const int wave = tid / AUV_WAVE, lane = tid % AUV_WAVE;
  const int S = d.cfg.n_sensors;
  const size_t n = (size_t)d.n;
  int4 cnt = d.counters[e];
  if (only_fresh && cnt.w == 0) return;
  const int w = d.world_idx[e];
  const double* ws = d.world_scalar + 8 * (size_t)w;
  const double L = ws[0];
  const double px = d.state[0 * n + e], py = d.state[1 * n + e], psi = d.state[2 * n + e];
  double* inf = d.info64 + 8 * (size_t)e;
  double* nv = d.nav64 + 8 * (size_t)e;
  double* ob = d.obs64 + (size_t)e * (6 + S);
  const int D = 6 + (d.cfg.use_lidar ? S : 0);

  if (mode != 2) {
    // ---- nearest point on the dense polyline (path.py:84-93) ----
    const long long p0 = d.poly_off[w];
    const int P = (int)(d.poly_off[w + 1] - p0);
    const double2* xy = d.poly_xy + p0;
    MinIdx best;
    best.d = 1.7976931348623157e308;
    best.j = 0x7fffffff;
    for (int j = tid; j < P - 1; j += AUV_BLOCK) {
      double2 a = xy[j], b = xy[j + 1];
      double dd = auv_pt_seg_dist(px, py, a.x, a.y, b.x, b.y);
      if (dd < best.d) best.d = dd, best.j = j;
    }
    best = wave_min_first(best);
    if (lane == 0) s_min[wave] = best;
    __syncthreads();
    if (tid == 0) {
      MinIdx b = s_min[0];
#pragma unroll
      for (int i = 1; i < AUV_BLOCK / AUV_WAVE; i++) b = min_first(b, s_min[i]);
      const int bj = b.j;
      // measure along the polyline: LengthIndexOfPoint::segmentNearestMeasure
      double2 A = xy[bj], B = xy[bj + 1];
      double dx = B.x - A.x, dy = B.y - A.y, len2 = dx * dx + dy * dy;
      double seglen = sqrt(len2);
      double pf = (len2 == 0.0) ? 0.0 : ((px - A.x) * dx + (py - A.y) * dy) / len2;
      double cum = d.poly_cum[p0 + bj];
      double s = pf <= 0.0 ? cum : (pf <= 1.0 ? cum + pf * seglen : cum + seglen);
      // vessel.py:471-515
      double p[2], dp[2], pt[2], dpt[2];
      path_eval(d, w, s, L, p, dp);
      double chi = atan2(dp[1], dp[0]);
      double ddx = p[0] - px, ddy = p[1] - py;
      double cte = sin(-chi) * ddx + cos(-chi) * ddy;
      double s_t = s + d.cfg.look_ahead_distance;
      if (L < s_t) s_t = L;
      path_eval(d, w, s_t, L, pt, dpt);
      double la = auv_princip(atan2(dpt[1], dpt[0]) - psi);
      double he = auv_princip(atan2(pt[1] - py, pt[0] - px) - psi);
      double progress = s / L;
      double maxp = inf[5];
      if (progress > maxp) maxp = progress;
      double gx = ws[1] - px, gy = ws[2] - py;
      double goal = sqrt(gx * gx + gy * gy);
      int reached = (goal <= d.cfg.min_goal_distance) || (progress >= d.cfg.min_path_progress);
      double u = d.state[3 * n + e], v = d.state[4 * n + e], r = d.state[5 * n + e];
      nv[0] = u, nv[1] = v, nv[2] = r, nv[3] = la, nv[4] = he, nv[5] = cte / 100, nv[6] = chi, nv[7] = s_t;
      inf[0] = d.collision[e];
      inf[1] = reached, inf[2] = goal, inf[3] = progress, inf[5] = maxp, inf[6] = s, inf[7] = 0.0;
#pragma unroll
      for (int i = 0; i < 6; i++) ob[i] = auv_clip(nv[i], -1.0, 1.0);   // environment.py:276-280
    }
    __syncthreads();
  }

  int done = 0;
  if (mode != 1) {
    // ---- reward (rewarder.py) ----
    double num = 0.0, den = 0.0;
    const bool colav = d.cfg.rewarder == AUV_REWARD_COLAV;
    if (colav) {
      const double* dd = d.lidar_d + (size_t)e * S;
      const double dangle = 2 * AUV_PI / S;
      for (int i = tid; i < S; i += AUV_BLOCK) {
        double angle = -AUV_PI + (i + 1) * dangle;          // body-frame beam angle (vessel.py:66-68)
        double weight = 1 / (1 + fabs(10.0 * angle));        // gamma_theta
        double raw = d.cfg.sensor_range * exp(-0.1 * dd[i]); // gamma_x; velocity channel == 0 (sensor.py:159)
        num += weight * raw;
        den += weight;
      }
      num = auv_wave_sum(num);
      den = auv_wave_sum(den);
      if (lane == 0) s_sum[0][wave] = num, s_sum[1][wave] = den;
    }
    __syncthreads();
    if (tid == 0) {
      const double lambda = 0.5, eta = 0.0, gamma_y_e = 5.0, penalty_yawrate = 10.0, neutral_speed = 0.05,
                   max_speed = 2.0;
      const int collision = d.collision[e];
      double reward;
      if (collision) {
        reward = -10000.0 * (1 - lambda);
      } else {
        double u = nv[0], v = nv[1], yaw_rate = nv[2], heading_error = nv[4], cross_track_error = nv[5];
        double speed = sqrt(u * u + v * v);
        double ctp = exp(-gamma_y_e * fabs(cross_track_error));
        double path_reward = (1 + cos(heading_error) * speed / max_speed) * (1 + ctp) - 1;
        double living_penalty = lambda * (2 * neutral_speed + 1) + eta * neutral_speed;
        if (!colav) {
          double slow_penalty = (speed < 0.1) ? -2 : 0;
          reward = path_reward - living_penalty + eta * speed / max_speed - penalty_yawrate * fabs(yaw_rate) +
                   slow_penalty;
        } else {
          double closeness_reward = 0.0;
          if (S > 0) {
            double tn = 0.0, td = 0.0;
#pragma unroll
            for (int i = 0; i < AUV_BLOCK / AUV_WAVE; i++) tn += s_sum[0][i], td += s_sum[1][i];
            closeness_reward = -tn / td;
          }
          if (inf[3] < inf[5]) path_reward = fmin(path_reward, 0.0);
          double slow_penalty = (speed < 0.04) ? -2 : 0;
          reward = lambda * path_reward + (1 - lambda) * closeness_reward - living_penalty +
                   eta * speed / max_speed - penalty_yawrate * fabs(yaw_rate) + slow_penalty;
          if (reward < 0) reward *= 2.0;
        }
      }
      // ---- environment.py:333-347, :375-384 ----
      d.reward64[e] = reward;
      double cum = inf[4] + reward;
      inf[4] = cum;
      const int t_step = cnt.x;
      done = collision || (inf[1] != 0.0) || (t_step >= d.cfg.max_timesteps - 1 && !d.cfg.test_mode) ||
             (cum < d.cfg.min_cumulative_reward && !d.cfg.test_mode);
      cnt.x = t_step + 1;
      if (reward_out) reward_out[e] = (float)reward;
      if (done_out) done_out[e] = (uint8_t)done;
      if (done) {
        double* ep = d.episode + 4 * (size_t)e;
        ep[0] = cum, ep[1] = t_step + 1, ep[2] = collision, ep[3] = inf[1];
        cnt.z += 1;
      }
      s_reset = done && d.cfg.auto_reset;
      if (!s_reset) {
        cnt.w = 0;
        d.counters[e] = cnt;
      }
    }
    __syncthreads();
    if (s_reset) {
      // VecEnv auto-reset: rebind to the next world of the bank and restore reset-time state
      // (environment.py:203-213, vessel.py:189-224); the reset observation is produced by the
      // follow-up "fresh" pass (K2 + K3 mode 1).
      const int w2 = (int)(((long long)w + d.n) % d.n_worlds);
      const double* ws2 = d.world_scalar + 8 * (size_t)w2;
      if (tid == 0) {
        d.world_idx[e] = w2;
        d.state[0 * n + e] = ws2[3], d.state[1 * n + e] = ws2[4], d.state[2 * n + e] = ws2[5];
        d.state[3 * n + e] = 0.0, d.state[4 * n + e] = 0.0, d.state[5 * n + e] = 0.0;
        for (int i = 0; i < 8; i++) inf[i] = 0.0;
        d.collision[e] = 0;
        d.counters[e] = make_int4(0, 0, cnt.z, 1);   // fresh
      }
      for (int i = tid; i < S; i += AUV_BLOCK) d.lidar_d[(size_t)e * S + i] = d.cfg.sensor_range;
      const long long m0 = d.mv_off[w2];
      const int M = (int)(d.mv_off[w2 + 1] - m0);
      for (int m = tid; m < M; m += AUV_BLOCK) d.mover[(size_t)e * d.m_max + m] = d.mv_init[m0 + m];
      for (int k = tid; k < d.k_max; k += AUV_BLOCK) d.nearby[(size_t)e * d.k_max + k] = 0;
      return;   // obs written by the fresh pass
    }
  } else if (tid == 0) {
    cnt.w = 0;   // observed: no longer fresh
    d.counters[e] = cnt;
  }

  // ---- observation row, float32 (environment.py:139-143, :263-280) ----
  if (obs_out && mode != 2) {
    for (int i = tid; i < D; i += AUV_BLOCK) obs_out[(size_t)e * D + i] = (float)ob[i];
  }
}

// reset(): restore reset-time state for masked envs and mark them fresh
__global__ void __launch_bounds__(AUV_BLOCK) k_reset(AuvDev d, const uint8_t* __restrict__ mask,
                                                     const int32_t* __restrict__ world_idx) {
  const int e = blockIdx.x, tid = threadIdx.x;
  if (mask && !mask[e]) return;
  const int S = d.cfg.n_sensors;
  const size_t n = (size_t)d.n;
  const int w = world_idx ? world_idx[e] : d.world_idx[e];
  const double* ws = d.world_scalar + 8 * (size_t)w;
  if (tid == 0) {
    d.world_idx[e] = w;
    d.state[0 * n + e] = ws[3], d.state[1 * n + e] = ws[4], d.state[2 * n + e] = ws[5];
    d.state[3 * n + e] = 0.0, d.state[4 * n + e] = 0.0, d.state[5 * n + e] = 0.0;
    double* inf = d.info64 + 8 * (size_t)e;
    for (int i = 0; i < 8; i++) inf[i] = 0.0;
    d.collision[e] = 0;
    int4 c = d.counters[e];
    d.counters[e] = make_int4(0, 0, c.z, 1);
  }
  for (int i = tid; i < S; i += AUV_BLOCK) d.lidar_d[(size_t)e * S + i] = d.cfg.sensor_range;
  const long long m0 = d.mv_off[w];
  const int M = (int)(d.mv_off[w + 1] - m0);
  for (int m = tid; m < M; m += AUV_BLOCK) d.mover[(size_t)e * d.m_max + m] = d.mv_init[m0 + m];
  for (int k = tid; k < d.k_max; k += AUV_BLOCK) d.nearby[(size_t)e * d.k_max + k] = 0;
}

}  // namespace

void auv_launch_k3(const AuvDev& d, int mode, int only_fresh, float* obs, float* reward, uint8_t* done,
                   hipStream_t st) {
  hipLaunchKernelGGL(k3_nav_reward, dim3(d.n), dim3(AUV_BLOCK), 0, st, d, mode, only_fresh, obs, reward, done);
}

void auv_launch_reset(const AuvDev& d, const uint8_t* mask, const int32_t* world_idx, hipStream_t st) {
  hipLaunchKernelGGL(k_reset, dim3(d.n), dim3(AUV_BLOCK), 0, st, d, mask, world_idx);
}
