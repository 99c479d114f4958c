// auv_generate.h -- arguments of the on-device scenario generator (k5_generate.hip), shared with
// the C-ABI translation unit.
#pragma once
#include "auv_device.h"

#define GEN_NK 1000          // resampled knots per PCHIP pass (path.py:29-31)
#define GEN_CAND AUV_GEN_CAND   // candidate placements per obstacle (devgen.CAND)

struct GenOut {              // writable views of the bank tables (fixed-capacity slots)
  int32_t *poly_cnt, *chunk_cnt, *knot_cnt, *obs_cnt, *mv_cnt, *mv_vtab_len;
  double2* poly_xy;
  double* poly_cum;
  double4* chunk_bound;
  double* knot_s;
  double* knot_coef;
  double* world_scalar;
  int4* obs_meta;
  double* obs_cull;
  double4* seg;
  double4* mv_param;
  double4* mv_init;
  double2* mv_vtab;
  // capacities / parameters
  int32_t p_cap, g_cap, n_moving, n_static, n_draws, n_radius;
  double dt, vessel_width;
  const double* ring_unit;        // [65][2]
  const int32_t* nseg_by_radius;  // [n_radius]
  double* scratch;                // per workgroup: GEN_SCRATCH doubles
};

#define GEN_SCRATCH ((GEN_NK + GEN_NK * 8) + 4 * GEN_NK)

// fresh-world mode: one refill pass's batch in device memory (rows filled by k_fw_bind)
struct FwBatch {
  int32_t *slot, *env, *serial;   // [cap]
  int32_t* count;                 // [1]
};
size_t auv_gen_scratch_doubles(void);
void auv_launch_generate(const GenOut& g, const double* draws, int w_first, int n_worlds, int grid, hipStream_t st,
                         const int32_t* slots = nullptr, const int32_t* count_dev = nullptr);
void auv_launch_draws(double* draws, int n_draws, int n_moving, int n_static, unsigned long long seed, long long env_base, const int32_t* envs,
                      const int32_t* serials, int n_rows, const int32_t* count_dev, int grid, hipStream_t st);
void auv_launch_fw_bind(const FwBatch& b, int32_t* queue, unsigned int* ctl, int q_cap, int n_envs, int cap, int32_t* env_next_serial,
                        int32_t* shadow_world_idx, int32_t* shadow_fresh_count, hipStream_t st);
void auv_launch_fw_ready(const FwBatch& b, int32_t* state, int32_t* serial, unsigned int* ctl, int cap, hipStream_t st);
