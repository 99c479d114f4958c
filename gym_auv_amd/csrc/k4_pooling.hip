// K4 — feasibility pooling (optional post-kernel): one wave per environment.
//
// Reference: LidarPreprocessor._feasibility_pooling   gym_auv/objects/vessel/sensor.py:251-296
//            LidarPreprocessor.preprocess (np.split by sector)      sensor.py:215-238
//
// The reference walks a sector's sensors in ascending order of range and returns the first
// range x for which the sector has no opening wider than `width` among the sensors whose range
// exceeds x + width (else the sector maximum).  The test depends on x only through its value,
// so the result is  min { x_i : no opening for threshold x_i }.  Lanes <-> sensors: each lane
// runs the opening scan of its own sensor over its sector (same fp64 operations, same order),
// then a per-sector minimum through LDS (non-negative fp64 ordered as uint64).
// Roofline: HBM.  Algorithmic bytes per env: 8*S in + 12*n_sectors out.
#include "auv_device.h"

namespace {

__device__ __forceinline__ unsigned long long pd2u(double x) { return (unsigned long long)__double_as_longlong(x); }
__device__ __forceinline__ double pu2d(unsigned long long x) { return __longlong_as_double((long long)x); }

// LDS per wave: [S] double ranges | [n_sectors] u64 min-bits | [n_sectors] u64 max-bits
__global__ void __launch_bounds__(AUV_BLOCK) k4_pooling(AuvDev d, const int32_t* __restrict__ sector_start,
                                                        int n_sectors, double width, double* __restrict__ out_dist,
                                                        float* __restrict__ out_clos) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int wave = threadIdx.x / AUV_WAVE, lane = threadIdx.x % AUV_WAVE;
  const int S = d.cfg.n_sensors;
  const int e = auv_uniform(blockIdx.x * AUV_ENVS_PER_BLOCK + wave);
  if (e >= d.n) return;
  double* x = (double*)(smem + (size_t)wave * ((size_t)S + 2 * (size_t)n_sectors) * 8);
  unsigned long long* mn = (unsigned long long*)(x + S);
  unsigned long long* mx = mn + n_sectors;
  const double theta = 2 * AUV_PI / S;                       // vessel.py:63-65
  for (int i = lane; i < S; i += AUV_WAVE) x[i] = d.lidar_d[(size_t)e * S + i];
  for (int k = lane; k < n_sectors; k += AUV_WAVE) mn[k] = pd2u(1.0e308), mx[k] = 0ull;
  auv_wave_lds_sync();
  for (int i = lane; i < S; i += AUV_WAVE) {
    int k = 0;
    while (k + 1 < n_sectors && i >= sector_start[k + 1]) k++;
    const int s0 = sector_start[k], N = sector_start[k + 1] - s0;
    const double xi = x[i];
    // sensor.py:268-291 for threshold x_i
    const double dd = xi * theta;
    double opening_width = 0, opening_span = 0, opening_start = -theta * (N - 1) / 2;
    bool found = false;
    for (int j = 0; j < N; j++) {
      const bool survives = x[s0 + j] > xi + width;
      if (survives) {
        opening_width += dd;
        opening_span += theta;
        if (opening_width > width) {
          const double centre = opening_start + opening_span / 2;
          if (fabs(centre) < theta * (N - 1) / 4) found = true;
        }
      } else {
        opening_width += 0.5 * dd;
        opening_span += 0.5 * theta;
        if (opening_width > width) {
          const double centre = opening_start + opening_span / 2;
          if (fabs(centre) < theta * (N - 1) / 4) found = true;
        }
        opening_width = 0;
        opening_span = 0;
        opening_start = -theta * (N - 1) / 2 + j * theta;
      }
    }
    atomicMax(&mx[k], pd2u(xi));
    if (!found) atomicMin(&mn[k], pd2u(xi));
  }
  auv_wave_lds_sync();
  const double R = d.cfg.sensor_range, logR = log(1 + R);
  for (int k = lane; k < n_sectors; k += AUV_WAVE) {
    double v = pu2d(mn[k]);
    if (v > 1.0e307) v = pu2d(mx[k]);                        // every threshold had an opening: np.max
    v = v > 0.0 ? v : 0.0;                                   // max(0, .)
    if (out_dist) out_dist[(size_t)e * n_sectors + k] = v;
    if (out_clos) {
      const double cl = d.cfg.sensor_log_transform ? 1 - auv_clip(log(1 + v) / logR, 0.0, 1.0)
                                                   : 1 - auv_clip(v / R, 0.0, 1.0);
      out_clos[(size_t)e * n_sectors + k] = (float)cl;
    }
  }
}

}  // namespace

void auv_launch_k4(const AuvDev& d, const int32_t* sector_start, int n_sectors, double width, double* out_dist,
                   float* out_closeness, hipStream_t st) {
  const size_t lds = (size_t)AUV_ENVS_PER_BLOCK * ((size_t)d.cfg.n_sensors + 2 * (size_t)n_sectors) * 8;
  hipLaunchKernelGGL(k4_pooling, dim3((d.n + AUV_ENVS_PER_BLOCK - 1) / AUV_ENVS_PER_BLOCK), dim3(AUV_BLOCK), lds, st, d,
                     sector_start, n_sectors, width, out_dist, out_closeness);
}
