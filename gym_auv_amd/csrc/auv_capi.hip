// auv_capi.hip — host side of the C ABI declared in include/auv_hip.h.
// Owns device memory (environment state, world bank), launches K1/K2/K3 on the caller's
// stream, and exposes hipGraph capture and per-kernel HIP-event timing.  No exceptions cross
// the ABI; errors are reported by code + auv_last_error().
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>

#include "auv_device.h"
#include "auv_generate.h"

void auv_launch_k1(const AuvDev& d, const void* actions, int dtype, hipStream_t st, hipEvent_t ev0 = nullptr,
                   hipEvent_t ev1 = nullptr);
void auv_launch_k2(const AuvDev& d, int advance_movers, hipStream_t st);
void auv_launch_k2_fresh(const AuvDev& d, hipStream_t st);
void auv_launch_k3(const AuvDev& d, int mode, float* obs, float* reward, uint8_t* done, hipStream_t st);
void auv_launch_k3_fresh(const AuvDev& d, float* obs, hipStream_t st);
void auv_launch_k3_nav(const AuvDev& d, float* obs, hipStream_t st);
void auv_launch_k3_reward(const AuvDev& d, float* obs, float* reward, uint8_t* done, int lidar_obs, hipStream_t st,
                          hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
void auv_launch_reset(const AuvDev& d, const uint8_t* mask, const int32_t* world_idx, float* obs, hipStream_t st);
void auv_launch_harvest(const AuvDev& d, int count, hipStream_t st);
void auv_launch_ring_advance(const AuvDev& d, hipStream_t st);
void auv_launch_refresh_desc(const AuvDev& d, hipStream_t st);
void auv_launch_derive(const AuvDev& d, hipStream_t st);
size_t auv_k2_lds_bytes(const AuvDev& d);
hipError_t auv_k2_prepare(const AuvDev& d);
bool auv_step_fused_ok(const AuvDev& d);
bool auv_k23_ok(const AuvDev& d);
void auv_launch_k23(const AuvDev& d, float* obs, hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
hipError_t auv_step_fused_prepare(const AuvDev& d);
void auv_launch_step_fused(const AuvDev& d, const void* actions, int dtype, float* obs, float* reward, uint8_t* done,
                           hipStream_t st);
bool auv_two_kernel_ok(const AuvDev& d);
bool auv_paired_ok(const AuvDev& d);
bool auv_roles_ok(const AuvDev& d);
void auv_launch_step_roles(const AuvDev& d, const void* actions, int dtype, float* obs, float* reward, uint8_t* done,
                           hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
void auv_launch_k23_paired(const AuvDev& d, float* obs, float* reward, uint8_t* done, hipStream_t st, hipEvent_t ev0 = nullptr,
                           hipEvent_t ev1 = nullptr);
void auv_launch_k1n(const AuvDev& d, const void* actions, int dtype, float* obs, hipStream_t st, hipEvent_t ev0 = nullptr,
                    hipEvent_t ev1 = nullptr);
void auv_launch_k2r(const AuvDev& d, float* obs, float* reward, uint8_t* done, hipStream_t st, hipEvent_t ev0 = nullptr,
                    hipEvent_t ev1 = nullptr);
void auv_launch_k31(const AuvDev& d, const void* actions, int dtype, float* obs, float* reward, uint8_t* done, hipStream_t st);
void auv_launch_k4(const AuvDev& d, const int32_t* sector_start, int n_sectors, double width, double* out_dist,
                   float* out_closeness, hipStream_t st);

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) return fail(AUV_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                      __FILE__, __LINE__);                                    \
  } while (0)

struct auv_handle {
  AuvDev d;
  int device;
  bool worlds_loaded;
  std::vector<void*> env_allocs, bank_allocs;
  hipStream_t cap_stream;
  hipStream_t aux_stream;        // second branch of the step: K3-nav runs beside K2
  hipEvent_t ev_fork, ev_join;
  hipGraph_t graph;
  hipGraphExec_t graph_exec;
  int step_mode;                 // AUV_STEP_* (include/auv_hip.h)
  int32_t* pair_error_host;      // pinned, mapped: set by a navigation wave of the paired step that gave up polling
  hipEvent_t ev[6];
  // on-device generation (auv_generate_worlds): shape of the slot bank, 0 = packed upload
  int gen_worlds, gen_moving, gen_static, gen_grid;
  GenOut gen;
};

// paired step: a navigation wave that gave up polling for its sweep's word has left the step unfinished
#define PAIR_CHECK(h)                                                                                           \
  do {                                                                                                          \
    if ((h)->pair_error_host && *(volatile int32_t*)(h)->pair_error_host)                                       \
      return fail(AUV_ESTATE, "paired step: a navigation wave timed out waiting for its LiDAR sweep; results " \
                              "since then are incomplete (auv_set_step_mode(AUV_STEP_SIDE_BY_SIDE) avoids it)"); \
  } while (0)

template <typename T>
static int dev_alloc(std::vector<void*>& pool, T** out, size_t count) {
  void* p = nullptr;
  size_t bytes = (count ? count : 1) * sizeof(T);
  HIP_TRY(hipMalloc(&p, bytes));
  HIP_TRY(hipMemset(p, 0, bytes));
  pool.push_back(p);
  *out = (T*)p;
  return AUV_OK;
}

template <typename T, typename U>
static int dev_upload(std::vector<void*>& pool, const T** out, const U* host, size_t count_T) {
  T* p = nullptr;
  int rc = dev_alloc(pool, &p, count_T);
  if (rc) return rc;
  if (count_T) HIP_TRY(hipMemcpy(p, host, count_T * sizeof(T), hipMemcpyHostToDevice));
  *out = p;
  return AUV_OK;
}

static void free_pool(std::vector<void*>& pool) {
  for (void* p : pool) (void)hipFree(p);
  pool.clear();
}

// Winding of a closed boundary if rays from outside can only first-hit its front-facing edges
// (true for any simple ring): +1 counter-clockwise, -1 clockwise, 0 = leave every edge in
// (open, degenerate, self-touching or too long to check).  K2 then skips the back-facing half of
// an obstacle's edges; see k2_lidar.hip, phase S.
static int ring_winding(const double* seg, int nseg) {
  if (nseg < 3 || nseg > 256) return 0;
  double area2 = 0.0, scale = 0.0;
  for (int i = 0; i < nseg; i++) {
    const double* a = seg + 4 * (size_t)i;
    const double* b = seg + 4 * (size_t)((i + 1) % nseg);
    if (a[2] != b[0] || a[3] != b[1]) return 0;                   // not a closed chain
    area2 += a[0] * a[3] - a[2] * a[1];
    const double m = fabs(a[0]) + fabs(a[1]);
    if (m > scale) scale = m;
  }
  if (!(fabs(area2) > 1e-9 * (1.0 + scale * scale))) return 0;
  auto orient = [](const double* p, const double* q, const double* r) {
    return (q[0] - p[0]) * (r[1] - p[1]) - (q[1] - p[1]) * (r[0] - p[0]);
  };
  const double tol = 1e-12 * (1.0 + scale * scale);
  for (int i = 0; i < nseg; i++) {
    const double* a = seg + 4 * (size_t)i;
    if (a[0] == a[2] && a[1] == a[3]) return 0;                   // zero-length edge
    for (int j = i + 2; j < nseg; j++) {
      if (i == 0 && j == nseg - 1) continue;                       // neighbours through the closure
      const double* b = seg + 4 * (size_t)j;
      const double o1 = orient(a, a + 2, b), o2 = orient(a, a + 2, b + 2);
      const double o3 = orient(b, b + 2, a), o4 = orient(b, b + 2, a + 2);
      const bool apart = (o1 > tol && o2 > tol) || (o1 < -tol && o2 < -tol) || (o3 > tol && o4 > tol) || (o3 < -tol && o4 < -tol);
      if (!apart) return 0;                                        // crossing or touching: not simple
    }
  }
  return area2 > 0.0 ? 1 : -1;
}

// Second half of loading / generating a world bank: environment buffers sized for the bank,
// the per-world reset rows, and the initial binding env e -> world e % W.
static int finish_bank(auv_handle* h, bool alloc_env) {
  AuvDev& d = h->d;
  const int W = d.n_worlds, k_max = d.k_max, m_max = d.m_max;
  const size_t n = (size_t)d.n, S = (size_t)d.cfg.n_sensors;
  int rc = 0;
  if (alloc_env) {
  free_pool(h->env_allocs);
  auto& ep = h->env_allocs;
  rc |= dev_alloc(ep, &d.state, 6 * n);
  rc |= dev_alloc(ep, &d.world_idx, n);
  rc |= dev_alloc(ep, &d.env_desc, n);
  rc |= dev_alloc(ep, &d.counters, n);
  rc |= dev_alloc(ep, &d.lidar_d, n * S);
  rc |= dev_alloc(ep, &d.obs64, n * (6 + S));
  rc |= dev_alloc(ep, &d.reward64, n);
  rc |= dev_alloc(ep, &d.info64, n * 8);
  rc |= dev_alloc(ep, &d.nav64, n * 8);
  rc |= dev_alloc(ep, &d.mover, n * m_max);
  rc |= dev_alloc(ep, &d.nearby, n * k_max);
  rc |= dev_alloc(ep, &d.episode, n * 4);
  rc |= dev_alloc(ep, &d.limits, n * k_max);
  rc |= dev_alloc(ep, &d.collision, n);
  rc |= dev_alloc(ep, &d.step_info, n * 4);
  rc |= dev_alloc(ep, &d.pose_cs, n);
  rc |= dev_alloc(ep, &d.pair_word, n);
  rc |= dev_alloc(ep, &d.k1_pkt, 8 * n);
  rc |= dev_alloc(ep, &d.k1_done, 4);
  rc |= dev_alloc(ep, &d.fresh_count, 4);
  rc |= dev_alloc(ep, &d.fresh_list, n);
  rc |= dev_alloc(ep, &d.stamps, n * 16);
  rc |= dev_alloc(ep, &d.ring_pos, 4);
  rc |= dev_alloc(ep, &d.rew_path, n);
  rc |= dev_alloc(ep, &d.rew_lidar, n);
  {
    std::vector<double> bcs(2 * (S ? S : 1), 0.0);
    const double dangle = 2 * AUV_PI / (double)(S ? S : 1);
    for (size_t i = 0; i < S; i++) {
      const double ang = -AUV_PI + (double)(i + 1) * dangle;
      bcs[2 * i] = cos(ang), bcs[2 * i + 1] = sin(ang);
    }
    rc |= dev_upload(ep, &d.beam_cs, bcs.data(), S ? S : 1);
  }
  rc |= dev_alloc(ep, &d.beam_w, S ? S : 1);
  rc |= dev_alloc(ep, &d.derived, 4);
  rc |= dev_alloc(ep, &d.w_obs64, (size_t)W * (6 + S));
  rc |= dev_alloc(ep, &d.w_lidar, (size_t)W * S);
  rc |= dev_alloc(ep, &d.w_info, (size_t)W * 8);
  rc |= dev_alloc(ep, &d.w_nav, (size_t)W * 8);
  rc |= dev_alloc(ep, &d.w_nearby, (size_t)W * k_max);
  rc |= dev_alloc(ep, &d.w_limits, (size_t)W * k_max);
  rc |= dev_alloc(ep, &d.w_collision, (size_t)W);
  if (rc) return AUV_EHIP;
  }
  // a new bank starts with a plain action buffer (a captured graph, and with it the ring, is gone)
  d.ring_slots = 1;
  d.ring_slot_host = -1;
  HIP_TRY(hipMemset(d.ring_pos, 0, sizeof(int32_t)));
  {
    // paired step: no sweep has left a word yet
    std::vector<unsigned long long> empty(n ? n : 1, AUV_PAIR_EMPTY);
    HIP_TRY(hipMemcpy(d.pair_word, empty.data(), n * sizeof(unsigned long long), hipMemcpyHostToDevice));
    if (!h->pair_error_host) {
      HIP_TRY(hipHostMalloc((void**)&h->pair_error_host, sizeof(int32_t), hipHostMallocMapped));
      *h->pair_error_host = 0;
      HIP_TRY(hipHostGetDevicePointer((void**)&d.pair_error, h->pair_error_host, 0));
    }
    *h->pair_error_host = 0;   // (a new bank starts with a clean slate; see PAIR_CHECK)
    HIP_TRY(hipMemset(d.k1_pkt, 0, 8 * n * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(d.k1_done, 0, sizeof(int32_t)));
  }
  d.w_ready = 0;
  auv_launch_derive(d, nullptr);
  std::vector<int32_t> wi(n);
  if (auv_k2_lds_bytes(d) > 160 * 1024) return fail(AUV_EINVAL, "K2 LDS footprint %zu B exceeds the 160 KiB of a CU", auv_k2_lds_bytes(d));
  HIP_TRY(auv_k2_prepare(d));
  HIP_TRY(auv_step_fused_prepare(d));
  if ((size_t)AUV_ENVS_PER_BLOCK * (d.nch_max * 4 + 512) > 64 * 1024) return fail(AUV_EINVAL, "path too long for K3's chunk list");
  if (h->graph_exec) {
    (void)hipGraphExecDestroy(h->graph_exec);
    h->graph_exec = nullptr;
  }
  // ---- reset rows: the first observation of every world (navigate + perceive at its initial
  // pose) is a constant of the world; compute it once, N worlds at a time, with the step's own
  // kernels (reset state -> K2 -> K3 on the fresh list), and keep the rows per world.
  for (int w0 = 0; w0 < W; w0 += d.n) {
    const int count = (W - w0 < d.n) ? (W - w0) : d.n;
    for (size_t e = 0; e < n; e++) wi[e] = (int32_t)(w0 + (int)(e % (size_t)count));
    HIP_TRY(hipMemcpy(d.world_idx, wi.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d.fresh_count, 0, sizeof(int32_t)));
    auv_launch_reset(d, nullptr, nullptr, nullptr, nullptr);   // w_ready == 0: everything goes on the fresh list
    auv_launch_k2_fresh(d, nullptr);
    auv_launch_k3_fresh(d, nullptr, nullptr);
    auv_launch_harvest(d, count, nullptr);
    HIP_TRY(hipDeviceSynchronize());
  }
  d.w_ready = 1;
  {
    // the device-side copy of this struct: what the rarely taken paths of the paired step read their tables from
    if (!d.self) {
      AuvDev* p = nullptr;
      HIP_TRY(hipMalloc((void**)&p, sizeof(AuvDev)));
      d.self = p;
    }
    HIP_TRY(hipMemcpy((void*)d.self, &d, sizeof(AuvDev), hipMemcpyHostToDevice));
  }
  // initial binding e -> world e % W, reset-time state (the first reset() call is then a copy)
  for (size_t e = 0; e < n; e++) wi[e] = (int32_t)(e % (size_t)W);
  HIP_TRY(hipMemcpy(d.world_idx, wi.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(d.counters, 0, n * sizeof(int4)));
  auv_launch_reset(d, nullptr, nullptr, nullptr, nullptr);
  HIP_TRY(hipDeviceSynchronize());
  h->worlds_loaded = true;
  return AUV_OK;
}

extern "C" {

int32_t auv_abi_version(void) { return AUV_ABI_VERSION; }
const char* auv_last_error(void) { return g_err; }

int auv_create(const auv_config_t* cfg, int32_t n_envs, int32_t device_id, auv_handle_t** out) {
  if (!cfg || !out || n_envs <= 0) return fail(AUV_EINVAL, "auv_create: bad arguments");
  if (cfg->n_sensors < 0 || cfg->n_sensors > 4096) return fail(AUV_EINVAL, "n_sensors out of range");
  if (cfg->sensor_interval_load_obstacles <= 0) return fail(AUV_EINVAL, "sensor_interval_load_obstacles <= 0");
  if (cfg->obs_channels != 1 && cfg->obs_channels != 3) return fail(AUV_EINVAL, "obs_channels must be 1 or 3");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) return fail(AUV_EINVAL, "device %d not present (%d visible)", device_id, ndev);
  HIP_TRY(hipSetDevice(device_id));
  auv_handle* h = new auv_handle();
  memset(&h->d, 0, sizeof(h->d));
  h->d.cfg = *cfg;
  h->d.n = n_envs;
  h->d.e0 = 0, h->d.ne = n_envs;
  h->device = device_id;
  h->worlds_loaded = false;
  h->graph = nullptr;
  h->graph_exec = nullptr;
  h->cap_stream = nullptr;
  h->step_mode = AUV_STEP_ONE_LAUNCH;
  h->gen_worlds = 0;
  h->aux_stream = nullptr;
  h->ev_fork = h->ev_join = nullptr;
  for (auto& e : h->ev) e = nullptr;
  if (hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
    delete h;
    return fail(AUV_EHIP, "auv_create: cannot create the auxiliary stream / events");
  }
  *out = h;
  return AUV_OK;
}

int auv_destroy(auv_handle_t* h) {
  if (!h) return AUV_OK;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
  if (h->graph) (void)hipGraphDestroy(h->graph);
  if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
  if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  for (auto& e : h->ev)
    if (e) (void)hipEventDestroy(e);
  free_pool(h->env_allocs);
  free_pool(h->bank_allocs);
  if (h->pair_error_host) (void)hipHostFree(h->pair_error_host);
  if (h->d.self) (void)hipFree((void*)h->d.self);
  delete h;
  return AUV_OK;
}

int auv_load_worlds(auv_handle_t* h, const auv_world_bank_t* b) {
  if (!h || !b || b->n_worlds <= 0) return fail(AUV_EINVAL, "auv_load_worlds: bad arguments");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  free_pool(h->bank_allocs);
  AuvDev& d = h->d;
  const int W = b->n_worlds;
  // host-side validation of every offset the kernels will trust
  int k_max = 1, m_max = 1;
  const int64_t nP = b->poly_off[W], nK = b->knot_off[W], nO = b->obs_off[W], nM = b->mv_off[W];
  for (int w = 0; w < W; w++) {
    int64_t P = b->poly_off[w + 1] - b->poly_off[w], kn = b->knot_off[w + 1] - b->knot_off[w];
    int64_t K = b->obs_off[w + 1] - b->obs_off[w], M = b->mv_off[w + 1] - b->mv_off[w];
    if (P < 2 || kn < 2 || K < 0 || M < 0) return fail(AUV_EINVAL, "world %d: degenerate path/obstacle tables", w);
    if (!(b->world_scalar[8 * (size_t)w] > 0.0)) return fail(AUV_EINVAL, "world %d: path length <= 0", w);
    if (K > k_max) k_max = (int)K;
    if (M > m_max) m_max = (int)M;
    for (int64_t k = b->obs_off[w]; k < b->obs_off[w + 1]; k++) {
      const int32_t* meta = b->obs_meta + 4 * k;
      if (meta[0] == AUV_OBS_MOVER) {
        if (meta[3] < 0 || meta[3] >= M || meta[2] != AUV_MOVER_NSEG) return fail(AUV_EINVAL, "world %d: bad mover meta", w);
      } else if (meta[0] == AUV_OBS_RING || meta[0] == AUV_OBS_FILLED) {
        if (meta[1] < 0 || meta[2] < 1 || (int64_t)meta[1] + meta[2] > b->n_seg)
          return fail(AUV_EINVAL, "world %d: segment range out of bounds", w);
      } else {
        return fail(AUV_EINVAL, "world %d: unknown obstacle kind %d", w, meta[0]);
      }
    }
  }
  for (int64_t m = 0; m < nM; m++) {
    if (b->mv_vtab_off[m + 1] - b->mv_vtab_off[m] < 1) return fail(AUV_EINVAL, "mover %lld: empty velocity table", (long long)m);
    if (!(b->mv_param[4 * m + 3] >= 2.0)) return fail(AUV_EINVAL, "mover %lld: n_vel < 2", (long long)m);
  }
  d.n_worlds = W;
  d.k_max = k_max;
  d.m_max = m_max;
  int rc = 0;
  auto& bp = h->bank_allocs;
  {
    std::vector<int32_t> c_poly(W), c_knot(W), c_obs(W), c_mv(W), c_vt((size_t)(nM > 0 ? nM : 1));
    for (int w = 0; w < W; w++) {
      c_poly[w] = (int32_t)(b->poly_off[w + 1] - b->poly_off[w]);
      c_knot[w] = (int32_t)(b->knot_off[w + 1] - b->knot_off[w]);
      c_obs[w] = (int32_t)(b->obs_off[w + 1] - b->obs_off[w]);
      c_mv[w] = (int32_t)(b->mv_off[w + 1] - b->mv_off[w]);
    }
    for (int64_t m = 0; m < nM; m++) c_vt[m] = (int32_t)(b->mv_vtab_off[m + 1] - b->mv_vtab_off[m]);
    rc |= dev_upload(bp, &d.poly_cnt, c_poly.data(), (size_t)W);
    rc |= dev_upload(bp, &d.knot_cnt, c_knot.data(), (size_t)W);
    rc |= dev_upload(bp, &d.obs_cnt, c_obs.data(), (size_t)W);
    rc |= dev_upload(bp, &d.mv_cnt, c_mv.data(), (size_t)W);
    rc |= dev_upload(bp, &d.mv_vtab_len, c_vt.data(), c_vt.size());
  }
  rc |= dev_upload(bp, &d.poly_off, b->poly_off, (size_t)W + 1);
  rc |= dev_upload(bp, &d.poly_xy, b->poly_xy, (size_t)nP);
  rc |= dev_upload(bp, &d.poly_cum, b->poly_cum, (size_t)nP);
  {
    // derived data: bounding circle of every run of AUV_CHUNK polyline segments (K3's exact
    // pruning).  Radius inflated so that rounding can never exclude a chunk that matters.
    std::vector<int64_t> coff((size_t)W + 1, 0);
    std::vector<int32_t> ccnt((size_t)W, 0);
    std::vector<double> cbound;
    int nch_max = 1;
    for (int w = 0; w < W; w++) {
      const int64_t p0 = b->poly_off[w], P = b->poly_off[w + 1] - p0;
      const int64_t nch = (P - 1 + AUV_CHUNK - 1) / AUV_CHUNK;
      coff[w + 1] = coff[w] + nch;
      ccnt[w] = (int32_t)nch;
      if (nch > nch_max) nch_max = (int)nch;
      for (int64_t c = 0; c < nch; c++) {
        const int64_t v0 = c * AUV_CHUNK, v1 = (v0 + AUV_CHUNK < P - 1 ? v0 + AUV_CHUNK : P - 1);
        double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
        for (int64_t v = v0; v <= v1; v++) {
          const double x = b->poly_xy[2 * (p0 + v)], y = b->poly_xy[2 * (p0 + v) + 1];
          x0 = x < x0 ? x : x0, x1 = x > x1 ? x : x1, y0 = y < y0 ? y : y0, y1 = y > y1 ? y : y1;
        }
        const double cx = 0.5 * (x0 + x1), cy = 0.5 * (y0 + y1);
        double r2 = 0.0;
        for (int64_t v = v0; v <= v1; v++) {
          const double dx = b->poly_xy[2 * (p0 + v)] - cx, dy = b->poly_xy[2 * (p0 + v) + 1] - cy;
          const double q = dx * dx + dy * dy;
          r2 = q > r2 ? q : r2;
        }
        cbound.push_back(cx), cbound.push_back(cy);
        cbound.push_back(sqrt(r2) * (1.0 + 1e-9) + 1e-9), cbound.push_back(0.0);
      }
    }
    d.nch_max = nch_max;
    rc |= dev_upload(bp, &d.chunk_off, coff.data(), (size_t)W + 1);
    rc |= dev_upload(bp, &d.chunk_cnt, ccnt.data(), (size_t)W);
    rc |= dev_upload(bp, &d.chunk_bound, cbound.data(), cbound.size() / 4);
  }
  rc |= dev_upload(bp, &d.knot_off, b->knot_off, (size_t)W + 1);
  rc |= dev_upload(bp, &d.knot_s, b->knot_s, (size_t)nK);
  rc |= dev_upload(bp, &d.knot_coef, b->knot_coef, (size_t)nK * 8);
  rc |= dev_upload(bp, &d.world_scalar, b->world_scalar, (size_t)W * 8);
  rc |= dev_upload(bp, &d.obs_off, b->obs_off, (size_t)W + 1);
  {
    // static obstacles: note the winding of simple rings in meta.w (-1 none, -2 CCW, -3 CW); the
    // caller's meta.w is -1 for them and stays the mover index for movers
    std::vector<int32_t> meta(b->obs_meta, b->obs_meta + 4 * (size_t)nO);
    for (int64_t k = 0; k < nO; k++) {
      int32_t* m = meta.data() + 4 * k;
      if (m[0] == AUV_OBS_MOVER) continue;
      const int wind = ring_winding(b->seg + 4 * (size_t)m[1], m[2]);
      m[3] = wind > 0 ? -2 : (wind < 0 ? -3 : -1);
    }
    rc |= dev_upload(bp, &d.obs_meta, (const int4*)meta.data(), (size_t)nO);
  }
  rc |= dev_upload(bp, &d.obs_cull, b->obs_cull, (size_t)nO * 3);
  rc |= dev_upload(bp, &d.seg, b->seg, (size_t)b->n_seg);
  rc |= dev_upload(bp, &d.mv_off, b->mv_off, (size_t)W + 1);
  rc |= dev_upload(bp, &d.mv_param, b->mv_param, (size_t)nM);
  rc |= dev_upload(bp, &d.mv_init, b->mv_init, (size_t)nM);
  rc |= dev_upload(bp, &d.mv_vtab_off, b->mv_vtab_off, (size_t)nM + 1);
  rc |= dev_upload(bp, &d.mv_vtab, b->mv_vtab, (size_t)b->mv_vtab_off[nM]);
  if (rc) return AUV_EHIP;
  h->gen_worlds = 0;
  return finish_bank(h, true);
}

#define REQUIRE_READY(h)                                                          \
  do {                                                                            \
    if (!(h)) return fail(AUV_EINVAL, "null handle");                             \
    if (!(h)->worlds_loaded) return fail(AUV_ESTATE, "auv_load_worlds not called"); \
  } while (0)

int auv_generate_worlds(auv_handle_t* h, int32_t n_worlds, int32_t n_moving, int32_t n_static, const double* draws_dev,
                        int32_t n_draws, const double* ring_unit, const int32_t* nseg_by_radius, int32_t n_radius) {
  if (!h || n_worlds <= 0 || n_moving < 0 || n_static < 0 || !draws_dev)
    return fail(AUV_EINVAL, "auv_generate_worlds: bad arguments");
  if (n_moving + n_static > 256) return fail(AUV_EINVAL, "auv_generate_worlds: more than 256 obstacles per world");
  const int expect = 11 + n_moving * (3 * GEN_CAND + 2) + n_static * 3 * GEN_CAND;
  if (n_draws != expect) return fail(AUV_EINVAL, "auv_generate_worlds: %d draws per world, expected %d", n_draws, expect);
  HIP_TRY(hipSetDevice(h->device));
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, draws_dev) != hipSuccess || attr.type != hipMemoryTypeDevice) {
    (void)hipGetLastError();
    return fail(AUV_EINVAL, "auv_generate_worlds: draws must be device memory");
  }
  HIP_TRY(hipDeviceSynchronize());
  AuvDev& d = h->d;
  GenOut& g = h->gen;
  const int W = n_worlds, K = n_moving + n_static, M = n_moving;
  const bool same_shape = h->gen_worlds == W && h->gen_moving == n_moving && h->gen_static == n_static;
  if (!same_shape) {
    if (!ring_unit || !nseg_by_radius || n_radius < 2) return fail(AUV_EINVAL, "auv_generate_worlds: ring tables missing");
    for (int r = 0; r < n_radius; r++) {
      const int ns = nseg_by_radius[r];
      if (ns != 4 && ns != 8 && ns != 16 && ns != 32 && ns != 64) return fail(AUV_EINVAL, "nseg_by_radius[%d] = %d", r, ns);
    }
    h->worlds_loaded = false;
    h->gen_worlds = 0;
    free_pool(h->bank_allocs);
    auto& bp = h->bank_allocs;
    memset(&g, 0, sizeof(g));
    g.p_cap = AUV_GEN_POLY_CAP;
    g.g_cap = 64 * (n_static > 0 ? n_static : 1);
    g.n_moving = n_moving, g.n_static = n_static, g.n_draws = n_draws, g.n_radius = n_radius;
    g.dt = d.cfg.dt, g.vessel_width = d.cfg.vessel_width;
    const size_t Wz = (size_t)W, Kz = (size_t)(K > 0 ? K : 1), Mz = (size_t)(M > 0 ? M : 1);
    const size_t nch = (size_t)g.p_cap / AUV_CHUNK;
    int rc = 0;
    rc |= dev_alloc(bp, &g.poly_cnt, Wz);
    rc |= dev_alloc(bp, &g.chunk_cnt, Wz);
    rc |= dev_alloc(bp, &g.knot_cnt, Wz);
    rc |= dev_alloc(bp, &g.obs_cnt, Wz);
    rc |= dev_alloc(bp, &g.mv_cnt, Wz);
    rc |= dev_alloc(bp, &g.mv_vtab_len, Wz * Mz);
    rc |= dev_alloc(bp, &g.poly_xy, Wz * g.p_cap);
    rc |= dev_alloc(bp, &g.poly_cum, Wz * g.p_cap);
    rc |= dev_alloc(bp, &g.chunk_bound, Wz * nch);
    rc |= dev_alloc(bp, &g.knot_s, Wz * GEN_NK);
    rc |= dev_alloc(bp, &g.knot_coef, Wz * GEN_NK * 8);
    rc |= dev_alloc(bp, &g.world_scalar, Wz * 8);
    rc |= dev_alloc(bp, &g.obs_meta, Wz * Kz);
    rc |= dev_alloc(bp, &g.obs_cull, Wz * Kz * 3);
    rc |= dev_alloc(bp, &g.seg, Wz * g.g_cap);
    rc |= dev_alloc(bp, &g.mv_param, Wz * Mz);
    rc |= dev_alloc(bp, &g.mv_init, Wz * Mz);
    rc |= dev_alloc(bp, &g.mv_vtab, Wz * Mz);
    h->gen_grid = W < 1024 ? W : 1024;
    rc |= dev_alloc(bp, &g.scratch, (size_t)h->gen_grid * auv_gen_scratch_doubles());
    rc |= dev_upload(bp, &g.ring_unit, ring_unit, (size_t)65 * 2);
    rc |= dev_upload(bp, &g.nseg_by_radius, nseg_by_radius, (size_t)n_radius);
    // slot offsets: every world owns a fixed-capacity slice of each table
    std::vector<int64_t> o_poly(Wz + 1), o_chunk(Wz + 1), o_knot(Wz + 1), o_obs(Wz + 1), o_mv(Wz + 1), o_vt(Wz * Mz + 1);
    for (size_t w = 0; w <= Wz; w++) {
      o_poly[w] = (int64_t)(w * g.p_cap), o_chunk[w] = (int64_t)(w * nch), o_knot[w] = (int64_t)(w * GEN_NK);
      o_obs[w] = (int64_t)(w * K), o_mv[w] = (int64_t)(w * M);
    }
    for (size_t m = 0; m <= Wz * Mz; m++) o_vt[m] = (int64_t)m;
    rc |= dev_upload(bp, &d.poly_off, o_poly.data(), Wz + 1);
    rc |= dev_upload(bp, &d.chunk_off, o_chunk.data(), Wz + 1);
    rc |= dev_upload(bp, &d.knot_off, o_knot.data(), Wz + 1);
    rc |= dev_upload(bp, &d.obs_off, o_obs.data(), Wz + 1);
    rc |= dev_upload(bp, &d.mv_off, o_mv.data(), Wz + 1);
    rc |= dev_upload(bp, &d.mv_vtab_off, o_vt.data(), Wz * Mz + 1);
    if (rc) return AUV_EHIP;
    d.poly_cnt = g.poly_cnt, d.chunk_cnt = g.chunk_cnt, d.knot_cnt = g.knot_cnt, d.obs_cnt = g.obs_cnt;
    d.mv_cnt = g.mv_cnt, d.mv_vtab_len = g.mv_vtab_len;
    d.poly_xy = g.poly_xy, d.poly_cum = g.poly_cum, d.chunk_bound = g.chunk_bound;
    d.knot_s = g.knot_s, d.knot_coef = g.knot_coef, d.world_scalar = g.world_scalar;
    d.obs_meta = g.obs_meta, d.obs_cull = g.obs_cull, d.seg = g.seg;
    d.mv_param = g.mv_param, d.mv_init = g.mv_init, d.mv_vtab = g.mv_vtab;
    d.n_worlds = W;
    d.k_max = K > 0 ? K : 1;
    d.m_max = M > 0 ? M : 1;
    d.nch_max = (int)nch;
  }
  auv_launch_generate(g, draws_dev, 0, W, h->gen_grid, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  int rc = finish_bank(h, !same_shape);
  if (rc) return rc;
  h->gen_worlds = W, h->gen_moving = n_moving, h->gen_static = n_static;
  return AUV_OK;
}

static const void* bank_ptr(const auv_handle_t* h, int32_t table, size_t* bytes) {
  const AuvDev& d = h->d;
  // sizes are only known for a generated (slot) bank; a packed upload is already on the host
  if (!h->gen_worlds) return nullptr;
  const GenOut& g = h->gen;
  const size_t W = (size_t)h->gen_worlds, K = (size_t)d.k_max, M = (size_t)d.m_max;
  switch (table) {
    case AUV_B_POLY_CNT: *bytes = W * 4; return d.poly_cnt;
    case AUV_B_POLY_XY: *bytes = W * g.p_cap * 16; return d.poly_xy;
    case AUV_B_POLY_CUM: *bytes = W * g.p_cap * 8; return d.poly_cum;
    case AUV_B_KNOT_S: *bytes = W * GEN_NK * 8; return d.knot_s;
    case AUV_B_KNOT_COEF: *bytes = W * GEN_NK * 64; return d.knot_coef;
    case AUV_B_WORLD_SCALAR: *bytes = W * 64; return d.world_scalar;
    case AUV_B_OBS_META: *bytes = W * K * 16; return d.obs_meta;
    case AUV_B_OBS_CULL: *bytes = W * K * 24; return d.obs_cull;
    case AUV_B_SEG: *bytes = W * g.g_cap * 32; return d.seg;
    case AUV_B_MV_PARAM: *bytes = W * M * 32; return d.mv_param;
    case AUV_B_MV_INIT: *bytes = W * M * 32; return d.mv_init;
    case AUV_B_MV_VTAB: *bytes = W * M * 16; return d.mv_vtab;
    case AUV_B_CHUNK_BOUND: *bytes = W * (g.p_cap / AUV_CHUNK) * 32; return d.chunk_bound;
    default: return nullptr;
  }
}

size_t auv_bank_bytes(const auv_handle_t* h, int32_t table) {
  size_t bytes = 0;
  if (!h || !h->worlds_loaded) return 0;
  return bank_ptr(h, table, &bytes) ? bytes : 0;
}

int auv_read_bank(auv_handle_t* h, int32_t table, void* dst_dev, size_t bytes, void* stream) {
  REQUIRE_READY(h);
  size_t have = 0;
  const void* p = bank_ptr(h, table, &have);
  if (!p) return fail(AUV_EINVAL, "auv_read_bank: table %d not available (only generated banks can be read back)", table);
  if (bytes != have) return fail(AUV_EINVAL, "auv_read_bank: table %d is %zu bytes, caller passed %zu", table, have, bytes);
  HIP_TRY(hipMemcpyAsync(dst_dev, p, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return AUV_OK;
}

int auv_reset(auv_handle_t* h, const uint8_t* mask_dev, const int32_t* world_idx_dev, float* obs_dev, void* stream) {
  REQUIRE_READY(h);
  hipStream_t st = (hipStream_t)stream;
  auv_launch_reset(h->d, mask_dev, world_idx_dev, obs_dev, st);
  HIP_TRY(hipGetLastError());
  return AUV_OK;
}

// One step: K1, then K2 (LiDAR) on the caller's stream with K3-nav forked onto the auxiliary
// stream (they are independent given the new state), joined before K3-reward.  Works the same
// eagerly and under stream capture (the fork/join events become graph edges).
// `skip_k1`: the dynamics of this step were done by the previous step's fused kernel; `fuse_next`: this step's
// reward phase also runs the dynamics of the NEXT step (both only inside a captured graph of several steps)
static int enqueue_step(auv_handle_t* h, const void* actions, int32_t dtype, float* obs, float* reward, uint8_t* done,
                        hipStream_t st, bool capturing, bool skip_k1 = false, bool fuse_next = false) {
  // The action ring belongs to captured graphs only: an eager step reads `actions` as ONE plain
  // [N][2] buffer and neither reads nor advances the ring position (a caller that launches
  // eagerly can pass a different pointer every step).
  AuvDev d = h->d;
  if (!capturing) d.ring_slots = 1;
  if (h->step_mode == AUV_STEP_ONE_KERNEL && auv_step_fused_ok(d)) {
    // the whole step in one kernel (csrc/k_step_fused.hip).  A captured graph cannot change
    // arguments, so it reads the device-side ring position and advances it with a tiny follow-up node.
    auv_launch_step_fused(d, actions, dtype, obs, reward, done, st);
    if (d.ring_slots > 1) auv_launch_ring_advance(d, st);
    return AUV_OK;
  }
  if (h->step_mode == AUV_STEP_TWO_KERNELS && auv_two_kernel_ok(d)) {
    // [K1 -> K3-nav] -> [K2 -> K3-reward], two launches on one stream (csrc/k_step_fused.hip)
    auv_launch_k1n(d, actions, dtype, obs, st);
    auv_launch_k2r(d, obs, reward, done, st);
    return AUV_OK;
  }
  if (h->step_mode == AUV_STEP_ONE_LAUNCH && auv_roles_ok(d)) {
    // dynamics, LiDAR and navigation + reward as three roles of ONE launch (csrc/k_step_fused.hip: k_step_roles;
    // inside a captured graph its dynamics role advances the action ring)
    auv_launch_step_roles(d, actions, dtype, obs, reward, done, st);
    return AUV_OK;
  }
  if ((h->step_mode == AUV_STEP_PAIRED || h->step_mode == AUV_STEP_ONE_LAUNCH) && auv_paired_ok(d)) {
    // K1 -> [K2 and K3-nav side by side, the second of an environment's two waves runs K3-reward]: two launches
    auv_launch_k1(d, actions, dtype, st);
    auv_launch_k23_paired(d, obs, reward, done, st);   // (advances a captured graph's action ring)
    return AUV_OK;
  }
  if (h->step_mode != AUV_STEP_TWO_STREAMS && auv_k23_ok(d)) {
    // default: K1 -> [K2 and K3-nav side by side in one launch] -> K3-reward, one stream
    if (!skip_k1) auv_launch_k1(d, actions, dtype, st);
    auv_launch_k23(d, obs, st);                       // (advances a captured graph's action ring)
    AuvDev dr = d;
    if (d.ring_slots > 1) dr.ring_slot_host = -2;     // ... so the reward phase does not
    if (fuse_next) auv_launch_k31(dr, actions, dtype, obs, reward, done, st);
    else auv_launch_k3_reward(dr, obs, reward, done, d.cfg.use_lidar ? 0 : 1, st);
    return AUV_OK;
  }
  auv_launch_k1(d, actions, dtype, st);
  HIP_TRY(hipEventRecord(h->ev_fork, st));
  HIP_TRY(hipStreamWaitEvent(h->aux_stream, h->ev_fork, 0));
  auv_launch_k3_nav(d, obs, h->aux_stream);
  HIP_TRY(hipEventRecord(h->ev_join, h->aux_stream));
  auv_launch_k2(d, 1, st);
  HIP_TRY(hipStreamWaitEvent(st, h->ev_join, 0));
  auv_launch_k3_reward(d, obs, reward, done, 1, st);   // a done env with auto-reset copies its next world's reset rows
  return AUV_OK;
}

int auv_step(auv_handle_t* h, const void* actions_dev, int32_t action_dtype, float* obs_dev, float* reward_dev,
             uint8_t* done_dev, void* stream) {
  REQUIRE_READY(h);
  if (!actions_dev) return fail(AUV_EINVAL, "auv_step: null actions");
  if (action_dtype != AUV_F32 && action_dtype != AUV_F64) return fail(AUV_EINVAL, "auv_step: bad action dtype");
  PAIR_CHECK(h);
  int rc = enqueue_step(h, actions_dev, action_dtype, obs_dev, reward_dev, done_dev, (hipStream_t)stream, false);
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return AUV_OK;
}

int auv_step_slice(auv_handle_t* h, int32_t e0, int32_t ne, const void* actions_dev, int32_t action_dtype, float* obs_dev,
                   float* reward_dev, uint8_t* done_dev, void* stream) {
  REQUIRE_READY(h);
  if (!actions_dev) return fail(AUV_EINVAL, "auv_step_slice: null actions");
  if (action_dtype != AUV_F32 && action_dtype != AUV_F64) return fail(AUV_EINVAL, "auv_step_slice: bad action dtype");
  if (e0 < 0 || ne < 1 || (int64_t)e0 + ne > h->d.n) return fail(AUV_EINVAL, "auv_step_slice: slice [%d, %d) outside [0, %d)", e0, e0 + ne, h->d.n);
  PAIR_CHECK(h);
  if (!(h->step_mode == AUV_STEP_ONE_LAUNCH && auv_roles_ok(h->d)))
    return fail(AUV_ESTATE, "auv_step_slice: needs the one-launch step shape");
  AuvDev d = h->d;
  d.ring_slots = 1;
  d.e0 = e0, d.ne = ne;
  auv_launch_step_roles(d, actions_dev, action_dtype, obs_dev, reward_dev, done_dev, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return AUV_OK;
}

int auv_set_action_ring(auv_handle_t* h, int32_t n_slots) {
  REQUIRE_READY(h);
  if (n_slots < 1) return fail(AUV_EINVAL, "auv_set_action_ring: n_slots must be >= 1");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(h->d.ring_pos, 0, sizeof(int32_t)));
  h->d.ring_slots = n_slots;
  if (h->graph_exec) {   // a captured graph has the old value baked into its kernel arguments
    HIP_TRY(hipGraphExecDestroy(h->graph_exec));
    h->graph_exec = nullptr;
  }
  return AUV_OK;
}

int auv_set_step_mode(auv_handle_t* h, int32_t mode) {
  if (!h) return fail(AUV_EINVAL, "null handle");
  if (mode < 0 || mode > 5) return fail(AUV_EINVAL, "auv_set_step_mode: mode must be one of AUV_STEP_*");
  h->step_mode = mode;
  {
    // test hook of the paired step: AUV_PAIR_SKEW=k leaves k idle workgroups between the two roles, which puts an
    // environment's two waves on different XCDs (read here, not cached, so that a test can switch it)
    const char* v = getenv("AUV_PAIR_SKEW");
    const int k = v ? atoi(v) : 0;
    h->d.pair_skew = (k > 0 && k < 8) ? k : 0;
    const char* f = getenv("AUV_PAIR_FAULT");                // test hook: one sweep withholds its word
    h->d.pair_fault = (f && atoi(f) == 1) ? 1 : 0;
  }
  if (h->graph_exec) {
    (void)hipGraphExecDestroy(h->graph_exec);
    h->graph_exec = nullptr;
  }
  return AUV_OK;
}

int auv_step_dynamics(auv_handle_t* h, const void* actions_dev, int32_t action_dtype, void* stream) {
  REQUIRE_READY(h);
  if (!actions_dev) return fail(AUV_EINVAL, "auv_step_dynamics: null actions");
  if (action_dtype != AUV_F32 && action_dtype != AUV_F64) return fail(AUV_EINVAL, "bad action dtype");
  auv_launch_k1(h->d, actions_dev, action_dtype, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return AUV_OK;
}

int auv_lidar(auv_handle_t* h, int32_t advance_movers, void* stream) {
  REQUIRE_READY(h);
  auv_launch_k2(h->d, advance_movers ? 1 : 0, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return AUV_OK;
}

int auv_nav_reward(auv_handle_t* h, int32_t mode, float* obs_dev, float* reward_dev, uint8_t* done_dev, void* stream) {
  REQUIRE_READY(h);
  if (mode < 0 || mode > 2) return fail(AUV_EINVAL, "auv_nav_reward: mode must be 0, 1 or 2");
  auv_launch_k3(h->d, mode, obs_dev, reward_dev, done_dev, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return AUV_OK;
}

static void* field_ptr(const auv_handle_t* h, int32_t field, size_t* bytes) {
  const AuvDev& d = h->d;
  const size_t n = (size_t)d.n, S = (size_t)d.cfg.n_sensors;
  switch (field) {
    case AUV_FIELD_STATE: *bytes = 8 * 6 * n; return d.state;
    case AUV_FIELD_LIDAR_D: *bytes = 8 * n * S; return d.lidar_d;
    case AUV_FIELD_OBS64: *bytes = 8 * n * (6 + S); return d.obs64;
    case AUV_FIELD_REWARD64: *bytes = 8 * n; return d.reward64;
    case AUV_FIELD_INFO64: *bytes = 8 * 8 * n; return d.info64;
    case AUV_FIELD_WORLD_IDX: *bytes = 4 * n; return d.world_idx;
    case AUV_FIELD_COUNTERS: *bytes = 16 * n; return d.counters;
    case AUV_FIELD_MOVER_STATE: *bytes = 32 * n * d.m_max; return d.mover;
    case AUV_FIELD_NEARBY: *bytes = n * d.k_max; return d.nearby;
    case AUV_FIELD_EPISODE: *bytes = 8 * 4 * n; return d.episode;
    case AUV_FIELD_CULL_LIMITS: *bytes = 8 * n * d.k_max; return d.limits;
    case AUV_FIELD_NAV64: *bytes = 8 * 8 * n; return d.nav64;
    case AUV_FIELD_COLLISION: *bytes = n; return d.collision;
    case AUV_FIELD_STAMPS: *bytes = 8 * 16 * n; return d.stamps;
    case AUV_FIELD_STEP_INFO: *bytes = 8 * 4 * n; return d.step_info;
  }
  *bytes = 0;
  return nullptr;
}

size_t auv_field_bytes(const auv_handle_t* h, int32_t field) {
  if (!h || !h->worlds_loaded) return 0;
  size_t b = 0;
  field_ptr(h, field, &b);
  return b;
}

int auv_read(auv_handle_t* h, int32_t field, void* dst_dev, size_t bytes, void* stream) {
  REQUIRE_READY(h);
  size_t b = 0;
  void* p = field_ptr(h, field, &b);
  if (!p || !dst_dev || bytes != b) return fail(AUV_EINVAL, "auv_read: field %d expects %zu bytes, got %zu", field, b, bytes);
  HIP_TRY(hipMemcpyAsync(dst_dev, p, b, hipMemcpyDefault, (hipStream_t)stream));
  return AUV_OK;
}

int auv_write(auv_handle_t* h, int32_t field, const void* src_dev, size_t bytes, void* stream) {
  REQUIRE_READY(h);
  size_t b = 0;
  void* p = field_ptr(h, field, &b);
  if (!p || !src_dev || bytes != b) return fail(AUV_EINVAL, "auv_write: field %d expects %zu bytes, got %zu", field, b, bytes);
  HIP_TRY(hipMemcpyAsync(p, src_dev, b, hipMemcpyDefault, (hipStream_t)stream));
  if (field == AUV_FIELD_WORLD_IDX) auv_launch_refresh_desc(h->d, (hipStream_t)stream);   // out-of-range entries keep their binding
  return AUV_OK;
}

int auv_feasibility_pooling(auv_handle_t* h, const int32_t* sector_start_dev, int32_t n_sectors, double width,
                            double* out_dist_dev, float* out_closeness_dev, void* stream) {
  REQUIRE_READY(h);
  if (!sector_start_dev || n_sectors < 1 || n_sectors > h->d.cfg.n_sensors || !(width >= 0.0))
    return fail(AUV_EINVAL, "auv_feasibility_pooling: bad arguments");
  auv_launch_k4(h->d, sector_start_dev, n_sectors, width, out_dist_dev, out_closeness_dev, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return AUV_OK;
}

int auv_graph_capture(auv_handle_t* h, const void* actions_dev, int32_t action_dtype, float* obs_dev,
                      float* reward_dev, uint8_t* done_dev, void* stream) {
  return auv_graph_capture_steps(h, actions_dev, action_dtype, obs_dev, reward_dev, done_dev, 1, stream);
}

int auv_graph_capture_steps(auv_handle_t* h, const void* actions_dev, int32_t action_dtype, float* obs_dev,
                            float* reward_dev, uint8_t* done_dev, int32_t n_steps, void* stream) {
  REQUIRE_READY(h);
  (void)stream;
  if (!actions_dev) return fail(AUV_EINVAL, "auv_graph_capture: null actions");
  if (n_steps < 1 || n_steps > 4096) return fail(AUV_EINVAL, "auv_graph_capture_steps: n_steps must be in [1, 4096]");
  if (action_dtype != AUV_F32 && action_dtype != AUV_F64) return fail(AUV_EINVAL, "auv_graph_capture: bad action dtype");
  if (!h->cap_stream) HIP_TRY(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
  if (h->graph_exec) {
    HIP_TRY(hipGraphExecDestroy(h->graph_exec));
    h->graph_exec = nullptr;
  }
  if (h->graph) {
    HIP_TRY(hipGraphDestroy(h->graph));
    h->graph = nullptr;
  }
  HIP_TRY(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
  int rc = AUV_OK;
  // inside the graph, step k's reward phase and step k+1's dynamics share a launch (side-by-side shape with a LiDAR
  // sweep: the shapes whose reward kernel maps lanes to environments)
  // (also for the paired and one-launch shapes: replayed launches cost ~3 us more than eager ones on this stack, and
  // inside a graph of several steps the fused reward + dynamics launch makes up for more of that than they do --
  // 97.6 M against 95.5 M (paired) and 96.6 M (one launch) env-steps/s at 16 steps per graph, 95.3 against 90.6 M at
  // 8192 x 256 -- and the bits are the same.  A graph of ONE step keeps the handle's own shape.)
  const bool fuse = n_steps > 1 &&
                    (h->step_mode == AUV_STEP_SIDE_BY_SIDE || h->step_mode == AUV_STEP_PAIRED || h->step_mode == AUV_STEP_ONE_LAUNCH) &&
                    auv_k23_ok(h->d) && h->d.cfg.use_lidar;
  const int mode_was = h->step_mode;
  if (fuse) h->step_mode = AUV_STEP_SIDE_BY_SIDE;
  for (int32_t k = 0; k < n_steps && rc == AUV_OK; k++)
    rc = enqueue_step(h, actions_dev, action_dtype, obs_dev, reward_dev, done_dev, h->cap_stream, true,
                      fuse && k > 0, fuse && k + 1 < n_steps);
  h->step_mode = mode_was;
  hipError_t ce = hipStreamEndCapture(h->cap_stream, &h->graph);
  if (rc) return rc;
  HIP_TRY(ce);
  HIP_TRY(hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
  return AUV_OK;
}

int auv_graph_launch(auv_handle_t* h, void* stream) {
  REQUIRE_READY(h);
  if (!h->graph_exec) return fail(AUV_ESTATE, "auv_graph_launch: no captured graph");
  PAIR_CHECK(h);
  HIP_TRY(hipGraphLaunch(h->graph_exec, (hipStream_t)stream));
  return AUV_OK;
}

int auv_step_timed(auv_handle_t* h, const void* actions_dev, int32_t action_dtype, float* obs_dev, float* reward_dev,
                   uint8_t* done_dev, void* stream, float* out_ms4) {
  REQUIRE_READY(h);
  if (!actions_dev || !out_ms4) return fail(AUV_EINVAL, "auv_step_timed: null argument");
  PAIR_CHECK(h);
  hipStream_t st = (hipStream_t)stream;
  for (auto& e : h->ev)
    if (!e) HIP_TRY(hipEventCreate(&e));
  AuvDev d = h->d;
  d.ring_slots = 1;   // eager: `actions_dev` is one plain [N][2] buffer (see enqueue_step)
  // every dispatch of the step is stamped with its own start and stop event (hipExtLaunchKernel): the
  // elapsed times are the kernels' own durations, as a kernel trace reports them, without the gaps
  int nk;
  if (h->step_mode == AUV_STEP_TWO_KERNELS && auv_two_kernel_ok(d)) {
    auv_launch_k1n(d, actions_dev, action_dtype, obs_dev, st, h->ev[0], h->ev[1]);
    auv_launch_k2r(d, obs_dev, reward_dev, done_dev, st, h->ev[2], h->ev[3]);
    nk = 2;
  } else if (h->step_mode == AUV_STEP_ONE_LAUNCH && auv_roles_ok(d)) {
    auv_launch_step_roles(d, actions_dev, action_dtype, obs_dev, reward_dev, done_dev, st, h->ev[0], h->ev[1]);
    nk = 1;
  } else if ((h->step_mode == AUV_STEP_PAIRED || h->step_mode == AUV_STEP_ONE_LAUNCH) && auv_paired_ok(d)) {
    auv_launch_k1(d, actions_dev, action_dtype, st, h->ev[0], h->ev[1]);
    auv_launch_k23_paired(d, obs_dev, reward_dev, done_dev, st, h->ev[2], h->ev[3]);
    nk = 2;
  } else {
    if (!auv_k23_ok(d)) return fail(AUV_EINVAL, "auv_step_timed: path too long for the side-by-side launch");
    auv_launch_k1(d, actions_dev, action_dtype, st, h->ev[0], h->ev[1]);
    auv_launch_k23(d, obs_dev, st, h->ev[2], h->ev[3]);
    auv_launch_k3_reward(d, obs_dev, reward_dev, done_dev, d.cfg.use_lidar ? 0 : 1, st, h->ev[4], h->ev[5]);
    nk = 3;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventSynchronize(h->ev[2 * nk - 1]));
  out_ms4[1] = out_ms4[2] = 0.0f;
  for (int i = 0; i < nk; i++) HIP_TRY(hipEventElapsedTime(&out_ms4[i], h->ev[2 * i], h->ev[2 * i + 1]));
  HIP_TRY(hipEventElapsedTime(&out_ms4[3], h->ev[0], h->ev[2 * nk - 1]));   // whole step, first start to last stop
  return AUV_OK;
}

}  // extern "C"
