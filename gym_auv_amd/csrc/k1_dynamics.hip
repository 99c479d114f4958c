// K1 — vessel dynamics: eight lanes advance one environment's 6-state 3-DOF model by one
// Runge-Kutta-Fehlberg step (the 5th-order combination `q`), SoA state in HBM.
//
// Reference: Vessel.step            gym_auv/objects/vessel/vessel.py:226-247
//            _thrust_surge/_moment_steer                 vessel.py:572-578
//            _state_dot                                  vessel.py:561-570
//            odesolver45                objects/vessel/odesolver.py:2-47
//            M, D, N(nu)                utils/constants.py:33-43, 63-72
//            NaN action -> zeros        environment.py:314-315
// Roofline: HBM.  Algorithmic traffic per env-step: 6 fp64 state in + 6 out + 2 action
// values (8 or 16 B) + one int4 counter r/w = 136..144 B; ~500 flop and 12 sin/cos.
#include <hip/hip_ext.h>

#include "auv_device.h"

namespace {

struct Vec6 {
  double v[6];
};

// x / c for the Runge-Kutta tableau's integer constants, correctly rounded without the ~28-instruction
// IEEE division sequence: q = RN(x * RN(1/c)), r = x - q c exactly (one FMA), RN(q + r * RN(1/c)).
// q + r RN(1/c) differs from x / c by at most 2^-51 ulp, while a quotient by an integer c < 2^16 is
// either representable or at least ulp / (4 c) away from any rounding boundary, so the final
// rounding lands where the division's would (tests/test_k1_constant_division.py also checks 1e8 cases).
#define AUV_DIVC(x, c) auv_div_const((x), (c), 1.0 / (c))
__device__ __forceinline__ double auv_div_const(double x, double c, double rc) {
  const double q = x * rc;
  const double r = fma(-q, c, x);
  return fma(r, rc, q);
}

// Heading of a Runge-Kutta stage = heading at the start of the step + a small increment (dt times a
// combination of yaw rates): its sine / cosine follow from those of the start heading by the addition
// theorem with a short Taylor series in the increment (|delta| <= 1/8: truncation < 1e-19, rounding
// ~2 ulp) -- one full-range sincos per step instead of six on the kernel's dependent chain.  The
// reference evaluates np.sin / np.cos of every stage heading (vessel.py:566-568); the difference is
// a few 1e-16, the K1 parity bound is 1e-12.
struct Heading0 {
  double psi, s, c;      // start heading of the step, its sine and cosine
};

__device__ __forceinline__ Heading0 heading0(double psi) {
  Heading0 h;
  h.psi = psi;
  sincos(auv_princip(psi), &h.s, &h.c);
  return h;
}

__device__ __forceinline__ void stage_sincos(const Heading0& h0, const double psi, double* s, double* c) {
  const double dl = psi - h0.psi;
  if (fabs(dl) <= 0.125) {
    const double z = dl * dl;
    double ps = fma(z, 1.0 / 6227020800.0, -1.0 / 39916800.0);
    ps = fma(z, ps, 1.0 / 362880.0), ps = fma(z, ps, -1.0 / 5040.0), ps = fma(z, ps, 1.0 / 120.0);
    ps = fma(z, ps, -1.0 / 6.0);
    const double sd = fma(dl * z, ps, dl);                       // sin(delta)
    double pc = fma(z, -1.0 / 87178291200.0, 1.0 / 479001600.0);
    pc = fma(z, pc, -1.0 / 3628800.0), pc = fma(z, pc, 1.0 / 40320.0), pc = fma(z, pc, -1.0 / 720.0);
    pc = fma(z, pc, 1.0 / 24.0), pc = fma(z, pc, -0.5);
    const double cd = fma(z, pc, 1.0);                           // cos(delta)
    *s = fma(h0.s, cd, h0.c * sd);
    *c = fma(h0.c, cd, -(h0.s * sd));
  } else {
    sincos(auv_princip(psi), s, c);                              // (a yaw rate beyond anything the model reaches)
  }
}

__device__ __forceinline__ Vec6 state_dot(const Vec6& y, double tau_u, double tau_r, const Heading0& h0) {
  // constants.py:4-16
  const double m = 23.8, x_g = 0.046, I_z = 1.760, X_udot = -2.0, Y_vdot = -10.0, Y_rdot = 0.0,
               N_rdot = -1.0, N_vdot = 0.0, X_u = -2.0, Y_v = -7.0, Y_r = -0.1, N_v = -0.1, N_r = -0.5;
  const double m11 = m - X_udot, m22 = m - Y_vdot, m23 = m * x_g - Y_rdot, m32 = m * x_g - N_vdot,
               m33 = I_z - N_rdot;
  const double det = m22 * m33 - m23 * m32;
  const double i11 = 1.0 / m11, i22 = m33 / det, i23 = -m23 / det, i32 = -m32 / det, i33 = m22 / det;
  double s, c;
  stage_sincos(h0, y.v[2], &s, &c);
  double u = y.v[3], v = y.v[4], r = y.v[5];
  Vec6 o;
  o.v[0] = c * u + -s * v;   // Rz(psi).dot(nu), geomutils.py:37-43
  o.v[1] = s * u + c * v;
  o.v[2] = r;
  double d0 = 2.0 * u;                       // D.dot(nu)
  double d1 = 7.0 * v + -2.5425 * r;
  double d2 = -2.5425 * v + 1.422 * r;
  double n0 = -X_u * u;                      // N(nu).dot(nu)
  double n1 = -Y_v * v + (m * u - Y_r) * r;
  double n2 = -N_v * v + (m * x_g * u - N_r) * r;
  double r0 = tau_u - d0 - n0, r1 = 0.0 - d1 - n1, r2 = tau_r - d2 - n2;
  o.v[3] = i11 * r0;                          // M_inv.dot(...)
  o.v[4] = i22 * r1 + i23 * r2;
  o.v[5] = i32 * r1 + i33 * r2;
  return o;
}

// The action (thrust, rudder) of environment e: `actions` is [N][2] of float or double (d.act_f64; one kernel serves
// both -- the branch is uniform over the launch), inside a captured graph the current slot of the action ring.
// `slot`: >= 0 names the ring slot (a launch of several steps reads slot (first + t) % slots for its step t)
__device__ __forceinline__ void k1_action(const AuvDev& d, const void* __restrict__ actions, const int e, double* a0, double* a1, const int slot = -1) {
  size_t i = 2 * (size_t)e;
  if (slot >= 0) i += (size_t)slot * 2 * (size_t)d.n;
  else if (d.ring_slots > 1)   // action ring: slot of this step
    i += (size_t)(d.ring_slot_host >= 0 ? d.ring_slot_host : *d.ring_pos) * 2 * (size_t)d.n;
  if (d.act_f64) {
    const double2 a = *(const double2*)((const double*)actions + i);
    *a0 = a.x, *a1 = a.y;
  } else {
    const float2 a = *(const float2*)((const float*)actions + i);
    *a0 = (double)a.x, *a1 = (double)a.y;
  }
}

// Vessel.step for environment e.  Every calling lane computes the same thing; `store` selects
// who writes the new state / step counter back.  Returns them for the phases that follow in
// the same kernel.
__device__ __forceinline__ EnvPre k1_env(const AuvDev& d, const int e, const void* __restrict__ actions, const bool store) {
  const size_t n = (size_t)d.n;
  double a0, a1;
  k1_action(d, actions, e, &a0, &a1);
  if (isnan(a0) || isnan(a1)) a0 = a1 = 0.0;
  const double tu = auv_clip(a0, 0.0, 1.0) * d.cfg.thrust_max;
  const double tr = auv_clip(a1, -1.0, 1.0) * d.cfg.moment_max;
  const double h = d.cfg.dt;
  Vec6 y, t;
#pragma unroll
  for (int i = 0; i < 6; i++) y.v[i] = d.state[i * n + e];

  const Heading0 h0 = heading0(y.v[2]);
  Vec6 s1 = state_dot(y, tu, tr, h0);
#pragma unroll
  for (int i = 0; i < 6; i++) t.v[i] = y.v[i] + h * s1.v[i] / 4.0;
  Vec6 s2 = state_dot(t, tu, tr, h0);
#pragma unroll
  for (int i = 0; i < 6; i++) t.v[i] = y.v[i] + 3.0 * h * s1.v[i] / 32.0 + 9.0 * h * s2.v[i] / 32.0;
  Vec6 s3 = state_dot(t, tu, tr, h0);
#pragma unroll
  for (int i = 0; i < 6; i++)
    t.v[i] = y.v[i] + AUV_DIVC(1932.0 * h * s1.v[i], 2197.0) - AUV_DIVC(7200.0 * h * s2.v[i], 2197.0) +
             AUV_DIVC(7296.0 * h * s3.v[i], 2197.0);
  Vec6 s4 = state_dot(t, tu, tr, h0);
#pragma unroll
  for (int i = 0; i < 6; i++)
    t.v[i] = y.v[i] + AUV_DIVC(439.0 * h * s1.v[i], 216.0) - 8.0 * h * s2.v[i] + AUV_DIVC(3680.0 * h * s3.v[i], 513.0) -
             AUV_DIVC(845.0 * h * s4.v[i], 4104.0);
  Vec6 s5 = state_dot(t, tu, tr, h0);
#pragma unroll
  for (int i = 0; i < 6; i++)
    t.v[i] = y.v[i] - AUV_DIVC(8.0 * h * s1.v[i], 27.0) + 2 * h * s2.v[i] - AUV_DIVC(3544.0 * h * s3.v[i], 2565.0) +
             AUV_DIVC(1859.0 * h * s4.v[i], 4104.0) - AUV_DIVC(11.0 * h * s5.v[i], 40.0);
  Vec6 s6 = state_dot(t, tu, tr, h0);
#pragma unroll
  for (int i = 0; i < 6; i++)
    t.v[i] = y.v[i] + h * (AUV_DIVC(16.0 * s1.v[i], 135.0) + AUV_DIVC(6656.0 * s3.v[i], 12825.0) +
                           AUV_DIVC(28561.0 * s4.v[i], 56430.0) - AUV_DIVC(9.0 * s5.v[i], 50.0) + AUV_DIVC(2.0 * s6.v[i], 55.0));
  t.v[2] = auv_princip(t.v[2]);
  EnvPre pre;
  pre.cnt = d.counters[e];
  pre.cnt.y += 1;   // Vessel._step_counter (vessel.py:247)
#pragma unroll
  for (int i = 0; i < 6; i++) pre.s[i] = t.v[i];
  if (store) {
#pragma unroll
    for (int i = 0; i < 6; i++) d.state[i * n + e] = t.v[i];
    d.counters[e].y = pre.cnt.y;
  }
  return pre;
}

// Eight lanes advance one environment: lane c < 6 owns state component c and forms its
// Runge-Kutta combinations (15 of the 90 fp64 divisions by tableau constants), every lane evaluates
// _state_dot of the stage vector it gathers from its group by DPP moves (k1_group_bcast) and keeps component c.  The
// same operations in the same order per component as k1_env, so the results are bit-identical; the
// wave retires ~2x fewer instructions per environment step than with one lane doing all six
// components.  Returns component c of the new state (lanes c >= 6: unspecified).  Lanes whose group
// is idle (`e` clamped by the caller) compute along.
#define K1_GROUP 8
// component K (2..5) of the caller's group of eight lanes in every lane of the group, by two DPP moves per register half
// instead of a trip through the LDS crossbar: the quad's own lane K % 4 (quad_perm), then the quad that does not hold
// component K takes the other quad's copy (row_shr:4 into banks 1, 3 / row_shl:4 into banks 0, 2).  All 64 lanes are active.
template <int K> __device__ __forceinline__ int k1_group_bcast32(const int x) {
  static_assert(K >= 0 && K < 8, "lane of the group");
  constexpr int q = K % 4, qp = q | (q << 2) | (q << 4) | (q << 6);
  const int a = __builtin_amdgcn_update_dpp(0, x, qp, 0xF, 0xF, true);
  return K < 4 ? __builtin_amdgcn_update_dpp(a, a, 0x114, 0xF, 0xA, false) : __builtin_amdgcn_update_dpp(a, a, 0x104, 0xF, 0x5, false);
}
template <int K> __device__ __forceinline__ double k1_group_bcast(const double t) {
  return __hiloint2double(k1_group_bcast32<K>(__double2hiint(t)), k1_group_bcast32<K>(__double2loint(t)));
}
// `y_in` (nullable): component c of the state to start from, in the lane that owns it (else the STATE rows); `slot`: see k1_action;
// `a_in` (nullable): the action, fetched by the caller already (k1_action)
__device__ __forceinline__ double k1_group(const AuvDev& d, const void* __restrict__ actions, const int e, const int lane,
                                           const double* y_in = nullptr, const int slot = -1, const double2* a_in = nullptr) {
  const int c = lane % K1_GROUP, gbase = lane - c;
  const size_t n = (size_t)d.n;
  const bool own = c < 6;
  double a0, a1;
  if (a_in) a0 = a_in->x, a1 = a_in->y;
  else k1_action(d, actions, e, &a0, &a1, slot);
  if (isnan(a0) || isnan(a1)) a0 = a1 = 0.0;
  const double tu = auv_clip(a0, 0.0, 1.0) * d.cfg.thrust_max;
  const double tr = auv_clip(a1, -1.0, 1.0) * d.cfg.moment_max;
  const double h = d.cfg.dt;
  const double y = y_in ? *y_in : d.state[(size_t)(own ? c : 0) * n + e];
  (void)gbase;
  const Heading0 h0 = heading0(k1_group_bcast<2>(y));
  // _state_dot of the stage vector whose component c this lane holds in `t`; returns component c
  auto sdot = [&](double t) {
    Vec6 v;
    v.v[0] = 0.0, v.v[1] = 0.0;                              // x, y do not enter _state_dot
    v.v[2] = k1_group_bcast<2>(t), v.v[3] = k1_group_bcast<3>(t);
    v.v[4] = k1_group_bcast<4>(t), v.v[5] = k1_group_bcast<5>(t);
    const Vec6 o = state_dot(v, tu, tr, h0);
    double r = o.v[0];
#pragma unroll
    for (int i = 1; i < 6; i++) r = (c == i) ? o.v[i] : r;
    return r;
  };
  const double s1 = sdot(y);
  double t = y + h * s1 / 4.0;
  const double s2 = sdot(t);
  t = y + 3.0 * h * s1 / 32.0 + 9.0 * h * s2 / 32.0;
  const double s3 = sdot(t);
  t = y + AUV_DIVC(1932.0 * h * s1, 2197.0) - AUV_DIVC(7200.0 * h * s2, 2197.0) + AUV_DIVC(7296.0 * h * s3, 2197.0);
  const double s4 = sdot(t);
  t = y + AUV_DIVC(439.0 * h * s1, 216.0) - 8.0 * h * s2 + AUV_DIVC(3680.0 * h * s3, 513.0) - AUV_DIVC(845.0 * h * s4, 4104.0);
  const double s5 = sdot(t);
  t = y - AUV_DIVC(8.0 * h * s1, 27.0) + 2 * h * s2 - AUV_DIVC(3544.0 * h * s3, 2565.0) + AUV_DIVC(1859.0 * h * s4, 4104.0) -
      AUV_DIVC(11.0 * h * s5, 40.0);
  const double s6 = sdot(t);
  t = y + h * (AUV_DIVC(16.0 * s1, 135.0) + AUV_DIVC(6656.0 * s3, 12825.0) + AUV_DIVC(28561.0 * s4, 56430.0) -
               AUV_DIVC(9.0 * s5, 50.0) + AUV_DIVC(2.0 * s6, 55.0));
  if (c == 2) t = auv_princip(t);
  return t;
}

#ifndef AUV_DEVICE_FUNCS_ONLY
__global__ void __launch_bounds__(AUV_BLOCK) k1_dynamics(AuvDev d, const void* __restrict__ actions) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x % AUV_WAVE;
  const int c = lane % K1_GROUP;
  const size_t n = (size_t)d.n;
  const bool live = tid / K1_GROUP < d.ne;
  const int e = d.e0 + (live ? tid / K1_GROUP : d.ne - 1);          // idle groups compute along, store nothing
  const double t = k1_group(d, actions, e, lane);
  if (live && c < 6) d.state[(size_t)c * n + e] = t;
  if (live && c == 0) d.counters[e].y += 1;                // Vessel._step_counter (vessel.py:247)
  // (cos / sin of the new heading are NOT formed here for the launch that follows: measured, the extra sincos on
  // this kernel's dependent chain cost more (+0.5 us) than the two per-environment ones it saved in k23)
}
#endif

}  // namespace

#ifndef AUV_DEVICE_FUNCS_ONLY
// ev0 / ev1 (both or neither): the dispatch itself is stamped (hipExtLaunchKernel), so the elapsed
// time between them is the kernel's own duration without the gaps around it
void auv_launch_k1(const AuvDev& d0, const void* actions, int dtype, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  AuvDev d = d0;
  d.act_f64 = dtype == AUV_F64;
  const int per_block = AUV_BLOCK / K1_GROUP;
  dim3 grid((d.ne + per_block - 1) / per_block), block(AUV_BLOCK);
  hipExtLaunchKernelGGL(k1_dynamics, grid, block, 0, st, ev0, ev1, 0, d, actions);
}
#endif
