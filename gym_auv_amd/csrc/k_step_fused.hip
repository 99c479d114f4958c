// k_step — the whole step() of one environment in ONE kernel, one wave per environment:
//   Vessel.step (K1)  ->  Vessel.navigate (K3-nav)  ->  _update + Vessel.perceive (K2)  ->
//   reward / done / bookkeeping / auto-reset (K3-reward),
// built from the very same device functions as the individually launchable kernels
// (k1_dynamics / k2_lidar / k3_nav / k3_reward), so the per-kernel parity tests cover its
// arithmetic and tests/test_gpu_parity.py::test_fused_step_equals_kernel_sequence pins the
// composition.  At 4096 environments the step is a chain of latency-bound phases; a single launch
// removes three kernel boundaries (~2 us each on one stream, ~8 us across streams) and lets the
// tail of one environment's LiDAR sweep overlap the other phases of its neighbours.
// The new state and counters are handed from phase to phase in registers (EnvPre), so no lane
// re-reads global memory that another lane of the wave has just written.
#include <cstdlib>
#include <hip/hip_ext.h>

#define AUV_DEVICE_FUNCS_ONLY
#include "k1_dynamics.hip"
#include "k2_lidar.hip"
#include "k3_nav_reward.hip"

namespace {

template <typename AT>
__global__ void __launch_bounds__(AUV_BLOCK, 3) k_step(AuvDev d, const AT* __restrict__ actions,
                                                       float* __restrict__ obs_out, float* __restrict__ reward_out,
                                                       uint8_t* __restrict__ done_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int wave = threadIdx.x / AUV_WAVE, lane = threadIdx.x % AUV_WAVE;
  const int S = d.cfg.n_sensors;
  const int e = auv_uniform(blockIdx.x * AUV_ENVS_PER_BLOCK + wave);
  if (e >= d.n) return;
  const Slice L = carve(smem + wave * k2_slice_bytes(S, d.k_max, d.m_max), S, d.k_max, d.m_max);
  // K1: every lane advances the (same) vessel; lane 0 writes it back
  const EnvPre pre = k1_env<AT>(d, e, actions, lane == 0);
  // K3-nav: its chunk list borrows the (not yet used) segment stage of the LiDAR slice
  k3_nav_env(d, e, lane, (int*)L.stage, obs_out, &pre);
  auv_wave_lds_sync();
  // K2
  int collision = 0;
  const int n_act = k2_front(d, e, lane, L, 1, &pre);
  if (d.cfg.use_lidar) {
    k2_stage_and_pairs(d, L, lane, n_act, pre.s[2]);
    collision = k2_back(d, e, lane, L, obs_out);
  }
  // K3-reward (lidar_d / closeness rows are re-read by the lanes that wrote them)
  k3_reward_env(d, e, lane, true, obs_out, reward_out, done_out, &pre, collision, false, !d.cfg.use_lidar);
}

// K2 and K3-nav of ALL environments in one launch: workgroups [0, nb) sweep the LiDAR of their
// four environments, workgroups [nb, 2 nb) navigate theirs.  The two are independent given the
// new vessel state, so they run side by side (the LiDAR workgroups are dispatched first and fill
// the chip; navigation workgroups move in as those retire) -- the concurrency of two streams
// without the ~8 us a cross-stream event wait costs on each side.
__global__ void __launch_bounds__(AUV_BLOCK, 4) k23_lidar_nav(AuvDev d, float* __restrict__ obs_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int wave = threadIdx.x / AUV_WAVE, lane = threadIdx.x % AUV_WAVE;
  const int S = d.cfg.n_sensors;
  const int wpb = blockDim.x / AUV_WAVE;                            // waves per workgroup
  const int nb = (d.n + wpb - 1) / wpb;                             // LiDAR workgroups: one env per wave
  const bool nav_role = (int)blockIdx.x >= nb;                      // workgroup-uniform
  unsigned char* slice = smem + wave * k2_slice_bytes(S, d.k_max, d.m_max);
  if (nav_role) {
    const int e = auv_uniform(((int)blockIdx.x - nb) * wpb + wave);
    if (e >= d.n) return;
#ifdef AUV_STAMPS
    const unsigned long long t_nav0 = wall_clock64();
#endif
    k3_nav_env(d, e, lane, (int*)slice, obs_out);
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 12] = t_nav0, d.stamps[(size_t)e * 16 + 13] = wall_clock64();
#endif
  } else {
    const int e = auv_uniform((int)blockIdx.x * wpb + wave);
    if (e >= d.n) return;
    const Slice L = carve(slice, S, d.k_max, d.m_max);
    AUV_STAMP_DECL
#ifdef AUV_STAMPS
    const unsigned long long t_real0 = wall_clock64();
#endif
    const int n_act = k2_front(d, e, lane, L, 1);
    if (d.cfg.use_lidar) {
      AUV_STAMP()
#ifdef AUV_STAMPS
      unsigned long long sub[4] = {0, 0, 0, 0};
      k2_stage_and_pairs(d, L, lane, n_act, d.state[2 * (size_t)d.n + e], sub);
      if (lane == 0) d.stamps[(size_t)e * 16 + 7] = sub[0], d.stamps[(size_t)e * 16 + 14] = sub[1], d.stamps[(size_t)e * 16 + 15] = sub[2], d.stamps[(size_t)e * 16 + 6] = sub[3];
#else
      k2_stage_and_pairs(d, L, lane, n_act, d.state[2 * (size_t)d.n + e]);
#endif
      AUV_STAMP()
      k2_back(d, e, lane, L, obs_out);
      AUV_STAMP()
      AUV_STAMP_FLUSH(e, 0)   // 0: front (A, C, B, S)  1: pairs (D)  2: back (E)
#ifdef AUV_STAMPS
      if (lane == 0) d.stamps[(size_t)e * 16 + 3] = t_real0, d.stamps[(size_t)e * 16 + 4] = wall_clock64();
      if (lane == 0) d.stamps[(size_t)e * 16 + 5] = (unsigned long long)L.sbase[n_act];
#endif
    }
  }
}

}  // namespace

// the nav chunk list must fit the segment stage it borrows
bool auv_step_fused_ok(const AuvDev& d) { return (size_t)d.nch_max * sizeof(int) <= (size_t)K2_SEG_CAP * 32; }

void auv_launch_step_fused(const AuvDev& d, const void* actions, int dtype, float* obs, float* reward, uint8_t* done,
                           hipStream_t st) {
  const size_t lds = k2_slice_bytes(d.cfg.n_sensors, d.k_max, d.m_max) * AUV_ENVS_PER_BLOCK;
  const dim3 grid((d.n + AUV_ENVS_PER_BLOCK - 1) / AUV_ENVS_PER_BLOCK), block(AUV_BLOCK);
  if (dtype == AUV_F64)
    hipLaunchKernelGGL(k_step<double>, grid, block, lds, st, d, (const double*)actions, obs, reward, done);
  else
    hipLaunchKernelGGL(k_step<float>, grid, block, lds, st, d, (const float*)actions, obs, reward, done);
}

// the navigation role keeps its chunk list at the start of the wave's slice
bool auv_k23_ok(const AuvDev& d) { return (size_t)d.nch_max * sizeof(int) <= k2_slice_bytes(d.cfg.n_sensors, d.k_max, d.m_max); }

// Workgroup shape of the side-by-side launch: ONE wave per workgroup, so a wave slot is handed on
// the moment an environment's sweep ends instead of when the slowest of four does -- the
// navigation workgroups queued behind the LiDAR ones start (and end) earlier.
static int k23_wpb() {
  static int wpb = 0;
  if (!wpb) {
    const char* v = getenv("AUV_K23_WPB");
    wpb = v ? atoi(v) : 1;
    if (wpb != 1 && wpb != 2 && wpb != 4) wpb = 1;
  }
  return wpb;
}

void auv_launch_k23(const AuvDev& d, float* obs, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  const int wpb = k23_wpb();
  const size_t lds = k2_slice_bytes(d.cfg.n_sensors, d.k_max, d.m_max) * wpb;
  const int nb = (d.n + wpb - 1) / wpb;
  hipExtLaunchKernelGGL(k23_lidar_nav, dim3(2 * nb), dim3(AUV_WAVE * wpb), (uint32_t)lds, st, ev0, ev1, 0, d, obs);
}

hipError_t auv_step_fused_prepare(const AuvDev& d) {
  const size_t b = k2_slice_bytes(d.cfg.n_sensors, d.k_max, d.m_max) * AUV_ENVS_PER_BLOCK;
  if (b <= 64 * 1024) return hipSuccess;
  hipError_t e = hipFuncSetAttribute((const void*)k23_lidar_nav, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)k_step<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)k_step<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
}
