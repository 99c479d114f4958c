// auv_capi.hip — host side of the C ABI declared in include/auv_hip.h.
// Owns device memory (environment state, world bank), launches K1/K2/K3 on the caller's
// stream, and exposes hipGraph capture and per-kernel HIP-event timing.  No exceptions cross
// the ABI; errors are reported by code + auv_last_error().
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <chrono>
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
void auv_launch_harvest(const AuvDev& d, int count, hipStream_t st, const int32_t* count_dev = nullptr);
void auv_launch_fw_shadow_reset(const AuvDev& d, const int32_t* count_dev, hipStream_t st);
void auv_launch_refresh_desc(const AuvDev& d, hipStream_t st);
void auv_launch_derive(const AuvDev& d, hipStream_t st);
size_t auv_k2_lds_bytes(const AuvDev& d);
hipError_t auv_k2_prepare(const AuvDev& d);
bool auv_k23_ok(const AuvDev& d);
void auv_launch_k23(const AuvDev& d, float* obs, hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
hipError_t auv_step_fused_prepare(const AuvDev& d);
uint32_t auv_step_lds_bytes(const AuvDev& d);
int auv_pick_seg_cap(const AuvDev& d);
bool auv_roles_ok(const AuvDev& d);
void auv_launch_step_roles(const AuvDev& d, const void* actions, int dtype, float* obs, float* reward, uint8_t* done,
                           hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
void auv_launch_k31(const AuvDev& d, const void* actions, int dtype, float* obs, float* reward, uint8_t* done, hipStream_t st);
void auv_launch_step_multi(const AuvDev& d, const void* actions, int dtype, float* obs, float* reward, uint8_t* done, int n_steps, int first_slot,
                           int n_slots, unsigned long long seq0, int order, int lead, int lag, hipStream_t st);
void auv_launch_spin(unsigned long long ticks, hipStream_t st);
void auv_launch_rdv_publish(unsigned long long* word, unsigned long long seq, hipStream_t st);
void auv_launch_rdv_arrive(unsigned long long* word, hipStream_t st);
void auv_launch_rdv_wait(const unsigned long long* word, unsigned long long target, int32_t* err, int code, double limit_s, int32_t* abort_flag,
                         int32_t* report, hipStream_t st);
hipError_t auv_launch_probe(unsigned int* words, int np, int nc, unsigned int tag, unsigned int* failures, uint32_t lds, hipStream_t st);
size_t auv_policy_param_floats_impl(int obs_dim);
size_t auv_policy_lds_bytes(int obs_dim);
hipError_t auv_policy_prepare(int obs_dim);
void auv_launch_policy(const auv_policy_io_t& io, int e0, int ne, hipStream_t st, long long t_host = -1, long long gstep_host = -1);
void auv_launch_gae(const float* R, const float* V, const float* Dn, const float* last_v, float gamma, float lam, float* adv, float* ret,
                    int T, int N, hipStream_t st);
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
  hipGraph_t graph;
  hipGraphExec_t graph_exec;
  int step_mode;                 // AUV_STEP_* as requested (include/auv_hip.h)
  int32_t* pair_error_host;      // pinned, mapped: set by a wave of the one-launch step that gave up polling
  // in-launch hand-overs (the one-launch shape): allowed only while the load-time probe of the dispatch
  // order has passed and no poll has ever run out on this handle
  bool handover_ok;
  int probe_failures;            // of the last probe (0 = the dispatch order is what the hand-overs rely on)
  int handover_timeouts;         // polls that ran out over the life of the handle (each one disables the hand-overs)
  int last_timeout_e0, last_timeout_ne, last_reset_envs;   // the launch that reported the last time-out; environments its recovery reset
  hipEvent_t ev[6];
  std::vector<hipEvent_t> slice_ev;   // auv_step_pipelined_timed: start / stop event per sub-batch launch
  // auv_step_async / auv_step_wait: the chains of the pending step that do NOT run on the caller's stream, and how
  // they are ordered against it (0: no step pending, else 1 + AUV_RDV_*)
  int async_pending;
  std::vector<hipStream_t> async_streams;
  hipEvent_t ev_actions;              // AUV_RDV_EVENTS: recorded on the caller's stream behind the actions
  std::vector<hipEvent_t> ev_chain;   //                 recorded behind each chain's launch
  unsigned long long* rdv;            // AUV_RDV_DEVICE / _CP: words in device (signal) memory, one 128-byte line each:
                                      //   [0] actions-ready sequence, [16] chains-arrived count, [32 + 16 j] chain j's sequence
  unsigned long long rdv_seq, rdv_target;
  double rdv_limit_s;                 // how long a rendezvous kernel waits before it gives up (pair_error 4 / 5)
  bool rdv_device_ok;                 // AUV_RDV_DEVICE may be used: false once a trial or a real rendezvous has run out -- events from then on
  std::vector<hipStream_t> rdv_tried; // the remote streams the device rendezvous has been tried on (auv_step_async: rdv_trial)
  int rdv_timeouts;                   // rendezvous waits that ran out over the life of the handle (trial included)
  // captured chains (auv_graph_capture_chains): one linear graph per sub-batch, replayed on the sub-batch's stream
  std::vector<hipGraph_t> chain_graph;
  std::vector<hipGraphExec_t> chain_exec;
  std::vector<hipStream_t> fork_streams;   // side streams of the one-graph (fork / join) form
  std::vector<int32_t> chain_bounds;       // the slices of the captured chains
  int chain_steps = 1, graph_steps = 1;    // steps per replay of the captured chains / of the one graph
  unsigned long long multi_seq = 0;        // auv_step_multi: step numbers handed out so far (every mark of a step carries its number)
  int multi_order = 1, multi_lead = 16, multi_lag = 30;   // auv_set_multi_order: workgroup order of a launch of several steps
  // on-device generation (auv_generate_worlds): shape of the slot bank, 0 = packed upload
  int gen_worlds, gen_moving, gen_static, gen_grid;
  GenOut gen;
  // a fresh world on every reset (auv_fresh_worlds_create): the refill pass and its bookkeeping
  struct Fresh {
    bool on = false;
    int depth = 0, cap = 0, period = 0, n_draws = 0;
    unsigned long long seed = 0;
    long long env_base = 0;
    hipStream_t side = nullptr;               // the refill pass's stream: a plain stream of the library's, or the caller's choice
    bool side_owned = false;                  // (auv_fresh_worlds_set_stream: one that shares no hardware queue with the chains)
    std::vector<void*> allocs;                // shadow environments + queue / state words
    AuvDev shadow;                            // `cap` environments nobody steps: where a regenerated slot's reset rows are computed
    FwBatch batch;                            // device: the pass in flight on `side`
    int32_t* batch_block = nullptr;           // [1 + 3 cap]: count, slot[], env[], serial[] (one D2H copy per pass)
    double* draws = nullptr;                  // [cap][n_draws]
    int32_t* env_next_serial = nullptr;       // [N]
    static const int NEV = 64;                // pacing events (ring): a pass starts when chain 0 has reached the step it was enqueued behind
    hipEvent_t pace[NEV];
    unsigned long long issued = 0, calls = 0;
    hipGraphExec_t pass_exec = nullptr;       // the pass as ONE graph launch (captured at create)
    hipGraph_t pass_graph = nullptr;
  } fw;
};

// From how many environments per launch on the three-launch shape is used by AUV_STEP_AUTO.  With the four-role step
// the one launch is ahead at every size measured (one chain, tools/archive/auto_threshold.sh: 139.3 against 129.5 M env-steps/s
// at 8192 environments per launch, 149.1 / 141.4 M at 16384, 149.9 / 145.8 M at 32768): the margin halves with every
// doubling -- many rounds of waves per slot leave little to gain by hiding a launch boundary -- so beyond what was
// measured the fence-free shape is the default.  (With three roles the crossover was at 16384.)
#define AUV_AUTO_THREE_LAUNCHES_FROM 65536
#define AUV_MAX_CHAINS 64      // sub-batch chains per handle (BatchedAuvEnv.set_sub_batches allows up to 64)
#define AUV_RDV_WORDS (32 + 16 * AUV_MAX_CHAINS)       // rendezvous words, one 128-byte line each (auv_handle::rdv)
#define AUV_RDV_BYTES (AUV_RDV_WORDS * sizeof(unsigned long long))

// The shape a step of `ne` environments is actually launched in: the requested one, degraded to the fence-free
// three-launch shape where the in-launch hand-overs may not be used (probe failed / a poll timed out / no LiDAR).
static int effective_mode(const auv_handle* h, int ne) {
  int m = h->step_mode;
  if (m == AUV_STEP_AUTO) m = ne >= AUV_AUTO_THREE_LAUNCHES_FROM ? AUV_STEP_SIDE_BY_SIDE : AUV_STEP_ONE_LAUNCH;
  if (m != AUV_STEP_SIDE_BY_SIDE && (!h->handover_ok || !auv_roles_ok(h->d))) m = AUV_STEP_SIDE_BY_SIDE;
  return m;
}

// every captured graph of the handle has launch arguments of the current bank / mode / ring baked in
static void drop_graphs(auv_handle* h) {
  if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
  h->graph_exec = nullptr;
  for (auto& g : h->chain_exec) (void)hipGraphExecDestroy(g);
  for (auto& g : h->chain_graph) (void)hipGraphDestroy(g);
  h->chain_exec.clear(), h->chain_graph.clear();
}

static int recover_from_timeout(auv_handle* h, float* obs_now);
static int check_slices(const auv_handle* h, int32_t n_slices, const int32_t* bounds, const void* streams, const char* who);
static int probe_dispatch_order(auv_handle* h);

// A wave of the one-launch step that gave up polling has left its environment's step unfinished.  The
// next call on the handle notices (mapped host word), repairs the handle -- hand-over words cleared, EVERY
// environment put back into its reset state, three-launch shape from now on -- and reports AUV_ESTATE once.
#define PAIR_CHECK(h, obs) PAIR_CHECK_ON(h, nullptr, false, obs)
// `st`: the stream the call will enqueue on.  While that stream is being CAPTURED (a torch CUDAGraph around auv_step /
// auv_step_slice, examples/ppo.py) the recovery -- device synchronisation, copies, a reset launch on the null stream --
// would be illegal and would surface as a HIP capture error: the call then only reports AUV_ESTATE; the first call
// outside a capture recovers.
// `obs`: the observation buffer of THIS call (NULL if it has none): where the recovery writes the reset rows of the environments it
// resets -- never a pointer remembered from an earlier call (the caller may have freed or rotated that buffer).
#define PAIR_CHECK_ON(h, st, have_st, obs)                                       \
  do {                                                                           \
    if ((h)->pair_error_host && *(volatile int32_t*)(h)->pair_error_host) {      \
      if ((have_st) && stream_capturing((hipStream_t)(st)))                      \
        return fail(AUV_ESTATE, "a hand-over time-out is pending and this stream is being captured: nothing was enqueued; " \
                                "the next call outside a capture recovers and reports");      \
      int _rc = recover_from_timeout(h, (float*)(obs));                          \
      if (_rc) return _rc;                                                       \
    }                                                                            \
  } while (0)

static bool stream_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return cs == hipStreamCaptureStatusActive;
}

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
  d.ep_log_cap = 65536;                                        // >= max(65536, 4 n), a power of two (the kernels mask)
  while ((size_t)d.ep_log_cap < 4 * n) d.ep_log_cap *= 2;
  rc |= dev_alloc(ep, &d.ep_log, 8 * (size_t)d.ep_log_cap);
  rc |= dev_alloc(ep, &d.ep_log_count, 1);
  rc |= dev_alloc(ep, &d.pair_word, n);
  rc |= dev_alloc(ep, &d.abort_flag, 4);
  rc |= dev_alloc(ep, &d.broken, n);
  rc |= dev_alloc(ep, &d.k1_pkt, 8 * n);
  rc |= dev_alloc(ep, &d.nav_hand, 8 * n);
  rc |= dev_alloc(ep, &d.carry, 24 * n);
  rc |= dev_alloc(ep, &d.k1_done, AUV_MAX_CHAINS);      // (one per captured chain)
  rc |= dev_alloc(ep, &d.fresh_count, 4);
  rc |= dev_alloc(ep, &d.fresh_list, n);
  rc |= dev_alloc(ep, &d.stamps, n * AUV_STAMP_WORDS);
  rc |= dev_alloc(ep, &d.ring_pos, AUV_MAX_CHAINS);     // (one per captured chain)
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
  rc |= dev_alloc(ep, &d.derived, 8);
  rc |= dev_alloc(ep, &d.w_obs64, (size_t)W * (6 + S));
  rc |= dev_alloc(ep, &d.w_lidar, (size_t)W * S);
  rc |= dev_alloc(ep, &d.w_info, (size_t)W * 8);
  rc |= dev_alloc(ep, &d.w_nav, (size_t)W * 8);
  rc |= dev_alloc(ep, &d.w_nearby, (size_t)W * k_max);
  rc |= dev_alloc(ep, &d.w_limits, (size_t)W * k_max);
  rc |= dev_alloc(ep, &d.w_collision, (size_t)W);
  if (rc) return AUV_EHIP;
  }
  HIP_TRY(hipMemset(d.ep_log_count, 0, sizeof(unsigned long long)));   // the episode log restarts with every bank
  // a new bank starts with a plain action buffer (a captured graph, and with it the ring, is gone)
  d.ring_slots = 1;
  d.ring_slot_host = -1;
  HIP_TRY(hipMemset(d.ring_pos, 0, AUV_MAX_CHAINS * sizeof(int32_t)));
  {
    // one-launch step: no sweep has left a word yet
    std::vector<unsigned long long> empty(n ? n : 1, AUV_PAIR_EMPTY);
    HIP_TRY(hipMemcpy(d.pair_word, empty.data(), n * sizeof(unsigned long long), hipMemcpyHostToDevice));
    if (!h->pair_error_host) {
      HIP_TRY(hipHostMalloc((void**)&h->pair_error_host, 4 * sizeof(int32_t), hipHostMallocMapped));
      memset(h->pair_error_host, 0, 4 * sizeof(int32_t));
      HIP_TRY(hipHostGetDevicePointer((void**)&d.pair_error, h->pair_error_host, 0));
    }
    *h->pair_error_host = 0;   // (a new bank starts with a clean slate; see PAIR_CHECK)
    HIP_TRY(hipMemset(d.abort_flag, 0, sizeof(int32_t)));
    HIP_TRY(hipMemset(d.broken, 0, n));
    HIP_TRY(hipMemset(d.k1_pkt, 0, 8 * n * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(d.nav_hand, 0, 8 * n * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(d.carry, 0, 24 * n * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(d.k1_done, 0, AUV_MAX_CHAINS * sizeof(int32_t)));
    if (((uintptr_t)d.k1_pkt & 63) != 0 || ((uintptr_t)d.nav_hand & 63) != 0) return fail(AUV_EHIP, "hand-over records are not 64-byte aligned");
  }
  if (!h->rdv) {
    // the words of the step_async / step_wait rendezvous (signal memory: what hipStreamWaitValue64 / WriteValue64 accept; an
    // ordinary device allocation to the kernels) -- here, so that no step call ever allocates or synchronises
    if (hipExtMallocWithFlags((void**)&h->rdv, AUV_RDV_BYTES, hipMallocSignalMemory) != hipSuccess) {
      (void)hipGetLastError();
      h->rdv = nullptr;
      HIP_TRY(hipMalloc((void**)&h->rdv, AUV_RDV_BYTES));
    }
  }
  HIP_TRY(hipMemset(h->rdv, 0, AUV_RDV_BYTES));
  h->rdv_seq = h->rdv_target = 0;
  h->async_pending = 0;
  d.w_ready = 0;
  auv_launch_derive(d, nullptr);
  std::vector<int32_t> wi(n);
  d.seg_cap = auv_pick_seg_cap(d);
  if (auv_k2_lds_bytes(d) > 160 * 1024) return fail(AUV_EINVAL, "K2 LDS footprint %zu B exceeds the 160 KiB of a CU", auv_k2_lds_bytes(d));
  HIP_TRY(auv_k2_prepare(d));
  HIP_TRY(auv_step_fused_prepare(d));
  {
    int rc_p = probe_dispatch_order(h);
    if (rc_p) return rc_p;
  }
  if ((size_t)AUV_ENVS_PER_BLOCK * (d.nch_max * 4 + 512) > 64 * 1024) return fail(AUV_EINVAL, "path too long for K3's chunk list");
  drop_graphs(h);
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
    // the device-side copy of this struct: what the rarely taken paths of the one-launch step read their tables from
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

// One launch with the step's structure and LDS footprint and several times more one-wave workgroups than the chip has
// slots (k_step_fused.hip: k_probe_order): do consumers always find their lower-indexed producers?  Sets
// handover_ok; a failure is not an error -- the handle then simply steps in the three-launch shape.
static int probe_streams(auv_handle* h, int n_streams, hipStream_t const* streams, bool foreign) {
  const int np = 512, nc = 8192;
  const size_t per = (size_t)(np + 2 * nc + 1);
  if (n_streams < 1) return fail(AUV_EINVAL, "dispatch-order probe: no stream");
  unsigned int* words = nullptr;
  HIP_TRY(hipMalloc((void**)&words, (size_t)n_streams * per * sizeof(unsigned int)));
  HIP_TRY(hipMemset(words, 0, (size_t)n_streams * per * sizeof(unsigned int)));
  HIP_TRY(hipDeviceSynchronize());
  hipError_t e = hipSuccess;
  // what production runs: one probe launch per chain stream, all in flight together, with a foreign kernel on each
  // stream in between (a policy's kernels sit between a chain's steps) -- twice, so that the second round's launches
  // queue behind live ones
  for (int round = 0; round < 2 && e == hipSuccess; round++)
    for (int i = 0; i < n_streams && e == hipSuccess; i++) {
      unsigned int* w = words + (size_t)i * per;
      e = auv_launch_probe(w, np, nc, 0xA5A5u + (unsigned)round, w + np + 2 * nc, auv_step_lds_bytes(h->d), streams[i]);
      if (foreign && e == hipSuccess) auv_launch_spin(300, streams[i]);          // 3 us of somebody else's kernel
    }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  unsigned int nf = 0;
  for (int i = 0; i < n_streams && e == hipSuccess; i++) {
    unsigned int f = 0;
    e = hipMemcpy(&f, words + (size_t)i * per + np + 2 * nc, sizeof(f), hipMemcpyDeviceToHost);
    nf += f;
  }
  (void)hipFree(words);
  if (e != hipSuccess) return fail(AUV_EHIP, "dispatch-order probe: %s", hipGetErrorString(e));
  h->probe_failures = (int)nf;
  h->handover_ok = nf == 0 && h->handover_timeouts == 0;
  return AUV_OK;
}

static int probe_dispatch_order(auv_handle* h) {
  hipStream_t null_stream = nullptr;
  return probe_streams(h, 1, &null_stream, false);
}

static int recover_from_timeout(auv_handle* h, float* obs_now) {
  AuvDev& d = h->d;
  const int code = *(volatile int32_t*)h->pair_error_host;
  const int fe0 = ((volatile int32_t*)h->pair_error_host)[1], fne = ((volatile int32_t*)h->pair_error_host)[2];
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  drop_graphs(h);
  const size_t n = (size_t)d.n;
  h->async_pending = 0;
  if (h->rdv) HIP_TRY(hipMemset(h->rdv, 0, AUV_RDV_BYTES));
  h->rdv_seq = h->rdv_target = 0;
  const bool rendezvous = code == 4 || code == 5;
  if (rendezvous) {
    // a rendezvous kernel of auv_step_async / auv_step_wait gave up: the in-launch hand-overs have nothing to do with it and
    // stay in use; the device-word rendezvous does not -- events from now on (they cannot run out)
    h->rdv_timeouts += 1;
    h->rdv_device_ok = false;
  } else {
    h->handover_timeouts += 1;
    h->handover_ok = false;
  }
  // Which environments were left half-stepped?  Exactly those are reset; every other environment is consistent: it completed
  // the steps of the launches that ran, and launches queued behind the time-out did nothing (ABORT packets, k_step_roles --
  // also behind a chain's gate that ran out: its step was NOT taken on actions that may not have been there).  The marks of
  // the aborted launches are still up: cleared here whatever the code was (the rendezvous branch used to return
  // with abort_flag up, and every later one-launch step then did nothing, silently).
  std::vector<uint8_t> broken(n ? n : 1, 0);
  HIP_TRY(hipMemcpy(broken.data(), d.broken, n, hipMemcpyDeviceToHost));
  int n_broken = 0;
  for (size_t e = 0; e < n; e++) n_broken += broken[e] != 0;
  std::vector<unsigned long long> empty(n ? n : 1, AUV_PAIR_EMPTY);
  HIP_TRY(hipMemcpy(d.pair_word, empty.data(), n * sizeof(unsigned long long), hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(d.k1_pkt, 0, 8 * n * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(d.nav_hand, 0, 8 * n * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(d.carry, 0, 24 * n * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(d.k1_done, 0, AUV_MAX_CHAINS * sizeof(int32_t)));
  if (n_broken) auv_launch_reset(d, d.broken, nullptr, obs_now, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(d.broken, 0, n));
  HIP_TRY(hipMemset(d.abort_flag, 0, 4 * sizeof(int32_t)));
  memset(h->pair_error_host, 0, 4 * sizeof(int32_t));
  h->last_reset_envs = n_broken;
  if (rendezvous)
    return fail(AUV_ESTATE, "step_async / step_wait rendezvous timed out (%s).  %s  %d environment(s) were reset; the handle orders its chains "
                            "by HIP events from now on (AUV_RDV_EVENTS); this error is reported once",
                code == 4 ? "a chain waited in vain for the caller's stream to publish the actions" : "the caller's stream waited in vain for the chains to arrive",
                code == 4 ? "The step of that chain, and every launch queued behind it, was NOT executed (no environment stepped on actions that were not "
                            "there); chains whose gates had opened did step, so the slices may be one step apart."
                          : "The chains' steps were executed; work enqueued behind step_wait may have read results that were not ready.",
                n_broken);
  h->last_timeout_e0 = fe0, h->last_timeout_ne = fne;
  return fail(AUV_ESTATE, "in-launch hand-over timed out (%s) in the launch over environments [%d, %d): the %d environment(s) whose step "
                          "was left unfinished are back in their reset state (reset observation in OBS64%s); all "
                          "others keep their state, but steps enqueued behind the time-out were not executed.  The handle steps in "
                          "the three-launch shape from now on (no in-launch hand-over); this error is reported once",
              code == 2 ? "a sweep or search wave waited in vain for the dynamics role's state"
                        : code == 6 ? "a wave of a launch of several steps waited in vain for the previous step's finish wave"
                        : (code == 3 ? "a finish wave waited in vain for a state packet or a search record" : "a finish wave waited in vain for a sweep's word"),
              fe0, fe0 + fne, n_broken, obs_now ? " and in this call's observation buffer" : "");
}

// Do the kernels of the device-word rendezvous get to run side by side on these streams RIGHT NOW?  The pattern of one
// step_async / step_wait with nothing at stake: publish on the caller's stream, on every chain stream a gate, 20 us of a
// do-nothing wave (standing for the step) and the count-off, one wait on the caller's stream -- every wait limited to 50 ms,
// reporting code 6 (no abort, no recovery).  Where dispatches are serialised (a counter-collecting profiler executes one kernel
// at a time, in an order of its own) a waiting kernel sits in front of what it waits for: the trial then costs 50-100 ms ONCE
// and the handle orders its chains by events from then on -- round 4 lost two profiling runs to 300 s of silence instead.
// Synchronises the streams involved (once per set of streams).
static int rdv_trial(auv_handle* h, hipStream_t cs) {
  const size_t nr = h->async_streams.size();
  h->rdv_seq += 1;
  auv_launch_rdv_publish(h->rdv, h->rdv_seq, cs);
  for (size_t j = 0; j < nr; j++) {
    hipStream_t st = h->async_streams[j];
    auv_launch_rdv_wait(h->rdv, h->rdv_seq, h->d.pair_error, 6, 0.05, nullptr, h->d.abort_flag + 2, st);
    auv_launch_spin(2000, st);
    auv_launch_rdv_arrive(h->rdv + 16, st);
  }
  h->rdv_target += nr;
  auv_launch_rdv_wait(h->rdv + 16, h->rdv_target, h->d.pair_error, 6, 0.05, nullptr, h->d.abort_flag + 2, cs);
  HIP_TRY(hipGetLastError());
  for (size_t j = 0; j < nr; j++) HIP_TRY(hipStreamSynchronize(h->async_streams[j]));
  HIP_TRY(hipStreamSynchronize(cs));
  h->rdv_tried = h->async_streams;
  {
    // (a hand-over time-out of a step still in flight on some chain may have crossed the trial's report on the one host word:
    // the device-side flag is the truth)
    int32_t up = 0;
    HIP_TRY(hipMemcpy(&up, h->d.abort_flag, sizeof(up), hipMemcpyDeviceToHost));
    const int32_t seen = *(volatile int32_t*)h->pair_error_host;
    if (up && (seen == 0 || seen == 6)) {
      h->rdv_device_ok = false;
      *(volatile int32_t*)h->pair_error_host = 1;
      return AUV_OK;                                   // the next call's PAIR_CHECK recovers
    }
  }
  if (*(volatile int32_t*)h->pair_error_host == 6) {
    *(volatile int32_t*)h->pair_error_host = 0;
    HIP_TRY(hipMemset(h->d.abort_flag + 2, 0, sizeof(int32_t)));
    HIP_TRY(hipMemset(h->rdv, 0, AUV_RDV_BYTES));
    h->rdv_seq = h->rdv_target = 0;
    h->rdv_device_ok = false;
    h->rdv_timeouts += 1;
  }
  return AUV_OK;
}

// ---- a fresh world on every reset: the refill pass (host side) -------------------------------------------------------------
// The reference builds a new scenario whenever an episode ends (environment.py:176-218 reset -> _generate,
// envs/movingobstacles.py:28-95).  Here (auv_device.h: auv_next_world) a finished environment moves to its next bank slot and
// queues the slot it leaves; this pass -- one graph launch on a side stream every `period` step calls, never waited for by the
// step path -- pops up to `cap` queued slots ON THE DEVICE, draws their worlds from the counter-based generator
// (seed, global environment index, serial), rebuilds their tables (k5_generate) and their reset rows (the step's own fresh-list
// kernels on `cap` shadow environments), and its last kernel flips them to READY.  No host round trip anywhere: the host may
// be thousands of launches ahead of the GPU (a first version published READY by a kernel on the chains' own streams once the
// host had seen the pass complete -- behind an open-loop backlog that was 100+ steps too late for the episodes that end
// within a few dozen steps of their reset, a quarter of them under a random policy).  Coherence: see restore_env<COH>.
// A pass is paced in GPU time: it waits for an event recorded on the first chain's stream at the call that enqueued it.
static void fw_disable(auv_handle* h) {
  auv_handle::Fresh& f = h->fw;
  if (f.side) (void)hipStreamSynchronize(f.side);
  if (f.on || f.side) {
    for (int b = 0; b < auv_handle::Fresh::NEV; b++)
      if (f.pace[b]) (void)hipEventDestroy(f.pace[b]), f.pace[b] = nullptr;
  }
  if (f.pass_exec) (void)hipGraphExecDestroy(f.pass_exec);
  if (f.pass_graph) (void)hipGraphDestroy(f.pass_graph);
  f.pass_exec = nullptr, f.pass_graph = nullptr;
  if (f.side && f.side_owned) (void)hipStreamDestroy(f.side);
  f.side = nullptr, f.side_owned = false;
  free_pool(f.allocs);
  f.on = false;
  f.issued = f.calls = 0;
  h->d.fw_state = nullptr, h->d.fw_serial = nullptr, h->d.fw_queue = nullptr, h->d.fw_ctl = nullptr, h->d.fw_cap = 0;
}

// the kernels of one pass, in order, on `st` (captured once into a graph; also launchable one by one)
static void fw_pass_kernels(auv_handle* h, hipStream_t st) {
  auv_handle::Fresh& f = h->fw;
  const AuvDev& d = h->d;
  const int grid = f.cap < h->gen_grid ? f.cap : h->gen_grid;
  auv_launch_fw_bind(f.batch, d.fw_queue, d.fw_ctl, d.fw_cap, d.n, f.cap, f.env_next_serial, f.shadow.world_idx, f.shadow.fresh_count, st);
  auv_launch_draws(f.draws, f.n_draws, h->gen_moving, h->gen_static, f.seed, f.env_base, f.batch.env, f.batch.serial, f.cap, f.batch.count, grid, st);
  auv_launch_generate(h->gen, f.draws, 0, f.cap, grid, st, f.batch.slot, f.batch.count);
  auv_launch_fw_shadow_reset(f.shadow, f.batch.count, st);
  auv_launch_k2_fresh(f.shadow, st);
  auv_launch_k3_fresh(f.shadow, nullptr, st);
  auv_launch_harvest(f.shadow, f.cap, st, f.batch.count);
  auv_launch_fw_ready(f.batch, d.fw_state, d.fw_serial, d.fw_ctl, f.cap, st);
}

// One refill pass on the side stream.  `behind` (nullable): the pass starts when THAT stream has reached this point -- the
// host may run thousands of launches ahead of the GPU (open-loop chains, a whole rollout enqueued by one call), and a pass
// that ran at enqueue time would find an empty queue long before the episodes it is meant for have ended.
static int fw_enqueue_pass(auv_handle* h, hipStream_t behind, bool paced) {
  auv_handle::Fresh& f = h->fw;
  if (paced) {
    hipEvent_t ev = f.pace[f.issued % auv_handle::Fresh::NEV];
    HIP_TRY(hipEventRecord(ev, behind));
    HIP_TRY(hipStreamWaitEvent(f.side, ev, 0));
  }
  if (f.pass_exec) HIP_TRY(hipGraphLaunch(f.pass_exec, f.side));
  else {
    fw_pass_kernels(h, f.side);
    HIP_TRY(hipGetLastError());
  }
  f.issued += 1;
  return AUV_OK;
}

// once per step call of the whole batch (the callers name the chains' streams): every `period`-th call a pass is enqueued,
// paced behind the first chain.  Never waits; skipped while a stream is being captured.
static int fw_tick(auv_handle* h, int n_slices, const int32_t* bounds, void* const* streams, int n_steps = 1) {
  auv_handle::Fresh& f = h->fw;
  if (!f.on) return AUV_OK;
  (void)bounds;
  for (int s = 0; s < n_slices; s++)
    if (stream_capturing((hipStream_t)streams[s])) return AUV_OK;
  const unsigned long long before = f.calls / (unsigned long long)f.period;
  f.calls += (unsigned long long)n_steps;
  if (f.calls / (unsigned long long)f.period != before) return fw_enqueue_pass(h, (hipStream_t)streams[0], true);
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
  h->step_mode = AUV_STEP_AUTO;
  h->gen_worlds = 0;
  h->pair_error_host = nullptr;
  h->handover_ok = false;
  h->probe_failures = -1;
  h->handover_timeouts = 0;
  h->last_timeout_e0 = h->last_timeout_ne = -1, h->last_reset_envs = 0;
  for (auto& e : h->ev) e = nullptr;
  h->async_pending = 0;
  h->ev_actions = nullptr;
  h->rdv = nullptr;
  h->rdv_seq = h->rdv_target = 0;
  // A chain's gate waits for as long as the caller's stream is busy between publish and ... nothing: the publish kernel is enqueued by
  // auv_step_async itself, behind whatever the caller's stream still has to do.  10 s covers a PPO update enqueued in front of it; a
  // caller whose stream can be busy for longer raises the limit (auv_set_rendezvous_limit) or synchronises first.  (Round 4: 300 s --
  // two profiling runs sat silent for 420 s behind polling kernels whose streams a counter-collecting profiler had serialised.)
  h->rdv_limit_s = 10.0;
  h->rdv_device_ok = true;
  h->rdv_timeouts = 0;
  *out = h;
  return AUV_OK;
}

int auv_destroy(auv_handle_t* h) {
  if (!h) return AUV_OK;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  fw_disable(h);
  drop_graphs(h);
  if (h->graph) (void)hipGraphDestroy(h->graph);
  if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
  for (auto& e : h->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : h->slice_ev) (void)hipEventDestroy(e);
  for (auto& e : h->ev_chain) (void)hipEventDestroy(e);
  if (h->ev_actions) (void)hipEventDestroy(h->ev_actions);
  if (h->rdv) (void)hipFree(h->rdv);
  for (auto& g : h->chain_exec) (void)hipGraphExecDestroy(g);
  for (auto& g : h->chain_graph) (void)hipGraphDestroy(g);
  for (auto& st : h->fork_streams) (void)hipStreamDestroy(st);
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
  fw_disable(h);
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
  fw_disable(h);
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

int auv_fresh_worlds_create(auv_handle_t* h, int32_t depth, int32_t n_moving, int32_t n_static, uint64_t seed, int64_t env_index_base,
                            int32_t batch_cap, int32_t period, const double* ring_unit, const int32_t* nseg_by_radius, int32_t n_radius) {
  if (!h || depth < 2 || depth > 8 || batch_cap < 1 || batch_cap > 4096 || period < 1 || env_index_base < 0)
    return fail(AUV_EINVAL, "auv_fresh_worlds_create: depth in [2, 8], batch_cap in [1, 4096], period >= 1, env_index_base >= 0");
  if (!h->d.cfg.auto_reset) return fail(AUV_EINVAL, "auv_fresh_worlds_create: needs auto_reset (the turn-over is the auto-reset's)");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  fw_disable(h);
  const int n = h->d.n;
  const long long Wll = (long long)depth * n;
  if (Wll > (1ll << 30)) return fail(AUV_EINVAL, "auv_fresh_worlds_create: %lld world slots", Wll);
  const int W = (int)Wll;
  const int nd = 11 + n_moving * (3 * GEN_CAND + 2) + n_static * 3 * GEN_CAND;
  // ---- the initial bank: slot e + j N holds the world of (seed, env_index_base + e, serial j) ----
  {
    std::vector<void*> tmp;
    double* draws = nullptr;
    int32_t *envs = nullptr, *serials = nullptr;
    int rc = dev_alloc(tmp, &draws, (size_t)W * nd);
    rc |= dev_alloc(tmp, &envs, (size_t)W);
    rc |= dev_alloc(tmp, &serials, (size_t)W);
    if (rc) {
      free_pool(tmp);
      return AUV_ENOMEM;
    }
    std::vector<int32_t> he(W), hs(W);
    for (int s = 0; s < W; s++) he[s] = s % n, hs[s] = s / n;
    hipError_t e1 = hipMemcpy(envs, he.data(), (size_t)W * 4, hipMemcpyHostToDevice), e2 = hipMemcpy(serials, hs.data(), (size_t)W * 4, hipMemcpyHostToDevice);
    auv_launch_draws(draws, nd, n_moving, n_static, seed, env_index_base, envs, serials, W, nullptr, W < 1024 ? W : 1024, nullptr);
    hipError_t e3 = hipDeviceSynchronize();
    rc = (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) ? fail(AUV_EHIP, "auv_fresh_worlds_create: draws: %s", hipGetErrorString(e3))
                                                                   : auv_generate_worlds(h, W, n_moving, n_static, draws, nd, ring_unit, nseg_by_radius, n_radius);
    free_pool(tmp);
    if (rc) return rc;
  }
  // ---- the mode's own state ----
  auv_handle::Fresh& f = h->fw;
  AuvDev& d = h->d;
  f.depth = depth, f.cap = batch_cap, f.period = period, f.n_draws = nd, f.seed = seed, f.env_base = env_index_base;
  auto& ap = f.allocs;
  int rc = 0;
  rc |= dev_alloc(ap, &d.fw_state, (size_t)W);
  rc |= dev_alloc(ap, &d.fw_serial, (size_t)W);
  rc |= dev_alloc(ap, &d.fw_queue, (size_t)W);
  rc |= dev_alloc(ap, &d.fw_ctl, 8);
  rc |= dev_alloc(ap, &f.env_next_serial, (size_t)n);
  rc |= dev_alloc(ap, &f.batch_block, (size_t)(1 + 3 * batch_cap));
  for (int b = 0; b < auv_handle::Fresh::NEV; b++) f.pace[b] = nullptr;
  rc |= dev_alloc(ap, &f.draws, (size_t)batch_cap * nd);
  d.fw_cap = W;
  f.batch.count = f.batch_block, f.batch.slot = f.batch_block + 1, f.batch.env = f.batch_block + 1 + batch_cap, f.batch.serial = f.batch_block + 1 + 2 * batch_cap;
  // shadow environments: the handle's descriptor with `cap` environments of their own and w_ready = 0 (restore -> fresh list)
  AuvDev& sh = f.shadow;
  sh = d;
  const size_t c = (size_t)batch_cap, S = (size_t)d.cfg.n_sensors;
  sh.n = batch_cap, sh.e0 = 0, sh.ne = batch_cap, sh.w_ready = 0, sh.self = nullptr;
  sh.fw_state = nullptr, sh.fw_serial = nullptr, sh.fw_queue = nullptr, sh.fw_ctl = nullptr;
  rc |= dev_alloc(ap, &sh.state, 6 * c);
  rc |= dev_alloc(ap, &sh.world_idx, c);
  rc |= dev_alloc(ap, &sh.env_desc, c);
  rc |= dev_alloc(ap, &sh.counters, c);
  rc |= dev_alloc(ap, &sh.lidar_d, c * S);
  rc |= dev_alloc(ap, &sh.obs64, c * (6 + S));
  rc |= dev_alloc(ap, &sh.reward64, c);
  rc |= dev_alloc(ap, &sh.info64, c * 8);
  rc |= dev_alloc(ap, &sh.nav64, c * 8);
  rc |= dev_alloc(ap, &sh.mover, c * d.m_max);
  rc |= dev_alloc(ap, &sh.nearby, c * d.k_max);
  rc |= dev_alloc(ap, &sh.episode, c * 4);
  rc |= dev_alloc(ap, &sh.limits, c * d.k_max);
  rc |= dev_alloc(ap, &sh.collision, c);
  rc |= dev_alloc(ap, &sh.step_info, c * 4);
  rc |= dev_alloc(ap, &sh.fresh_count, 4);
  rc |= dev_alloc(ap, &sh.fresh_list, c);
  rc |= dev_alloc(ap, &sh.stamps, c * 16);
  rc |= dev_alloc(ap, &sh.rew_path, c);
  rc |= dev_alloc(ap, &sh.rew_lidar, c);
  if (rc) {
    fw_disable(h);
    return AUV_ENOMEM;
  }
  {
    std::vector<int32_t> st(W), se(W), q(W, -1), ns(n, depth);
    for (int s = 0; s < W; s++) st[s] = s < n ? AUV_FW_IN_USE : AUV_FW_READY, se[s] = s / n;
    HIP_TRY(hipMemcpy(d.fw_state, st.data(), (size_t)W * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d.fw_serial, se.data(), (size_t)W * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d.fw_queue, q.data(), (size_t)W * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(f.env_next_serial, ns.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  }
  // The pass's stream: a PLAIN stream.  (Measured, tools/side_stream_ab.sh: merely creating a stream with a non-default
  // priority -- lowest or highest -- halves the throughput of four sub-batch chains on this stack, 68 M against 139 M env-steps/s
  // with the stream idle: two of the four chains then share a hardware queue.)
  HIP_TRY(hipStreamCreateWithFlags(&f.side, hipStreamNonBlocking));
  f.side_owned = true;
  for (int b = 0; b < auv_handle::Fresh::NEV; b++) HIP_TRY(hipEventCreateWithFlags(&f.pace[b], hipEventDisableTiming));
  f.issued = f.calls = 0;
  {
    // the eight kernels of a pass as ONE graph: a pass costs the host one launch, not eight (every argument is a fixed
    // device buffer of this mode; what a pass works on it learns on the device)
    f.on = true;                                                   // (fw_disable below tears down whatever exists on failure)
    HIP_TRY(hipStreamBeginCapture(f.side, hipStreamCaptureModeThreadLocal));
    fw_pass_kernels(h, f.side);
    hipGraph_t g = nullptr;
    hipError_t ce = hipStreamEndCapture(f.side, &g);
    if (ce == hipSuccess) {
      f.pass_graph = g;
      HIP_TRY(hipGraphInstantiate(&f.pass_exec, g, nullptr, nullptr, 0));
    } else {
      (void)hipGetLastError();
      if (g) (void)hipGraphDestroy(g);
    }
  }
  HIP_TRY(hipMemcpy((void*)d.self, &d, sizeof(AuvDev), hipMemcpyHostToDevice));   // the one-launch step reads its tables through this copy
  HIP_TRY(hipDeviceSynchronize());
  f.on = true;
  return AUV_OK;
}

int auv_fresh_worlds_refill(auv_handle_t* h, int32_t n_slices, const int32_t* bounds, void* const* streams, int32_t flush) {
  REQUIRE_READY(h);
  if (!h->fw.on) return fail(AUV_ESTATE, "auv_fresh_worlds_refill: auv_fresh_worlds_create first");
  int rc = check_slices(h, n_slices, bounds, streams, "auv_fresh_worlds_refill");
  if (rc) return rc;
  HIP_TRY(hipSetDevice(h->device));
  auv_handle::Fresh& f = h->fw;
  if (!flush) return fw_enqueue_pass(h, (hipStream_t)streams[0], true);
  // flush: the chains' streams are synchronised first (their finish waves have queued what they will queue), then passes run
  // until the queue is empty -- every slot an environment has left so far is READY again when this returns
  for (int s = 0; s < n_slices; s++) HIP_TRY(hipStreamSynchronize((hipStream_t)streams[s]));
  for (int pass = 0; pass < (1 << 20); pass++) {
    HIP_TRY(hipStreamSynchronize(f.side));
    unsigned int ctl[2] = {0, 0};
    HIP_TRY(hipMemcpy(ctl, h->d.fw_ctl, sizeof(ctl), hipMemcpyDeviceToHost));
    if (ctl[0] == ctl[1]) break;
    rc = fw_enqueue_pass(h, nullptr, false);
    if (rc) return rc;
  }
  return AUV_OK;
}

int auv_fresh_worlds_set_stream(auv_handle_t* h, void* stream) {
  REQUIRE_READY(h);
  auv_handle::Fresh& f = h->fw;
  if (!f.on) return fail(AUV_ESTATE, "auv_fresh_worlds_set_stream: auv_fresh_worlds_create first");
  if (!stream) return fail(AUV_EINVAL, "auv_fresh_worlds_set_stream: a stream of the caller's (not the NULL stream)");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(f.side));
  if (f.side_owned) HIP_TRY(hipStreamDestroy(f.side));
  f.side = (hipStream_t)stream, f.side_owned = false;
  return AUV_OK;
}

int auv_fresh_worlds_stats(auv_handle_t* h, int64_t* out8) {
  if (!h || !out8) return fail(AUV_EINVAL, "auv_fresh_worlds_stats: bad arguments");
  for (int i = 0; i < 8; i++) out8[i] = 0;
  const auv_handle::Fresh& f = h->fw;
  if (!f.on) return AUV_OK;
  HIP_TRY(hipSetDevice(h->device));
  unsigned int ctl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  HIP_TRY(hipMemcpy(ctl, h->d.fw_ctl, sizeof(ctl), hipMemcpyDeviceToHost));
  unsigned long long regen = 0, done = 0;
  memcpy(&regen, ctl + 4, 8), memcpy(&done, ctl + 6, 8);
  out8[0] = 1, out8[1] = (int64_t)regen, out8[2] = (int64_t)ctl[2], out8[3] = (int64_t)(unsigned int)(ctl[0] - ctl[1]);
  out8[4] = (int64_t)f.issued, out8[5] = (int64_t)done, out8[6] = f.depth, out8[7] = f.cap;
  return AUV_OK;
}

int auv_fresh_worlds_draws(auv_handle_t* h, const int32_t* envs_host, const int32_t* serials_host, int32_t n_rows, double* dst_dev, void* stream) {
  REQUIRE_READY(h);
  if (!h->fw.on) return fail(AUV_ESTATE, "auv_fresh_worlds_draws: auv_fresh_worlds_create first");
  if (!envs_host || !serials_host || n_rows < 1 || !dst_dev) return fail(AUV_EINVAL, "auv_fresh_worlds_draws: bad arguments");
  HIP_TRY(hipSetDevice(h->device));
  const auv_handle::Fresh& f = h->fw;
  int32_t* keys = nullptr;
  HIP_TRY(hipMalloc((void**)&keys, (size_t)2 * n_rows * sizeof(int32_t)));
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemcpyAsync(keys, envs_host, (size_t)n_rows * 4, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(keys + n_rows, serials_host, (size_t)n_rows * 4, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    auv_launch_draws(dst_dev, f.n_draws, h->gen_moving, h->gen_static, f.seed, f.env_base, keys, keys + n_rows, n_rows, nullptr, n_rows < 1024 ? n_rows : 1024, st);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(keys);
  HIP_TRY(e);
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
  if (h->fw.on && world_idx_dev)
    return fail(AUV_EINVAL, "auv_reset: with a fresh world per reset the library chooses the world (an environment's next unseen slot)");
  auv_launch_reset(h->d, mask_dev, world_idx_dev, obs_dev, st);
  HIP_TRY(hipGetLastError());
  return AUV_OK;
}

// One step of the environments [e0, e0 + ne) on `st`, in the shape effective_mode() names.  Works the same eagerly
// and under stream capture.
// `skip_k1`: the dynamics of this step were done by the previous step's fused kernel; `fuse_next`: this step's
// reward phase also runs the dynamics of the NEXT step (both only inside a captured graph of several steps)
// `chain`: inside a captured chain (auv_graph_capture_chains) the launch uses the ring position / count-off word of its chain
static int enqueue_step(auv_handle_t* h, int mode, int e0, int ne, const void* actions, int32_t dtype, float* obs, float* reward,
                        uint8_t* done, hipStream_t st, bool capturing, bool skip_k1 = false, bool fuse_next = false, int chain = 0) {
  // The action ring belongs to captured graphs only: an eager step reads `actions` as ONE plain
  // [N][2] buffer and neither reads nor advances the ring position (a caller that launches
  // eagerly can pass a different pointer every step).
  AuvDev d = h->d;
  if (!capturing) d.ring_slots = 1;
  d.e0 = e0, d.ne = ne;
  d.ring_pos += chain, d.k1_done += chain;
  if (mode == AUV_STEP_ONE_LAUNCH) {
    // dynamics, LiDAR sweep, navigation search and finish as four roles of ONE launch (csrc/k_step_fused.hip: k_step_roles;
    // inside a captured graph its dynamics role advances the action ring)
    auv_launch_step_roles(d, actions, dtype, obs, reward, done, st);
    return AUV_OK;
  }
  // K1 -> [K2 and K3-nav side by side in one launch] -> K3-reward: no hand-over inside a launch
  if (!auv_k23_ok(d)) return fail(AUV_EINVAL, "path too long for the navigation's chunk list (%d chunks)", d.nch_max);
  if (!skip_k1) auv_launch_k1(d, actions, dtype, st);
  auv_launch_k23(d, obs, st);                         // (advances a captured graph's action ring)
  AuvDev dr = d;
  if (d.ring_slots > 1) dr.ring_slot_host = -2;       // ... so the reward phase does not
  if (fuse_next) auv_launch_k31(dr, actions, dtype, obs, reward, done, st);
  else auv_launch_k3_reward(dr, obs, reward, done, d.cfg.use_lidar ? 0 : 1, st);
  return AUV_OK;
}

static int check_slices(const auv_handle_t* h, int32_t n_slices, const int32_t* bounds, const void* streams, const char* who) {
  if (n_slices < 1 || n_slices > AUV_MAX_CHAINS || !bounds || !streams) return fail(AUV_EINVAL, "%s: bad arguments (1 .. %d slices)", who, AUV_MAX_CHAINS);
  if (bounds[0] != 0 || bounds[n_slices] != h->d.n) return fail(AUV_EINVAL, "%s: bounds must run from 0 to %d", who, h->d.n);
  for (int i = 0; i < n_slices; i++)
    if (bounds[i + 1] <= bounds[i]) return fail(AUV_EINVAL, "%s: empty or reversed slice %d", who, i);
  return AUV_OK;
}

static int check_actions(const void* actions_dev, int32_t action_dtype, const char* who) {
  if (!actions_dev) return fail(AUV_EINVAL, "%s: null actions", who);
  if (action_dtype != AUV_F32 && action_dtype != AUV_F64) return fail(AUV_EINVAL, "%s: bad action dtype", who);
  // the kernels fetch an environment's (thrust, rudder) pair with one load
  if ((uintptr_t)actions_dev & (action_dtype == AUV_F64 ? 15 : 7))
    return fail(AUV_EINVAL, "%s: the action buffer must be %d-byte aligned", who, action_dtype == AUV_F64 ? 16 : 8);
  return AUV_OK;
}

int auv_step(auv_handle_t* h, const void* actions_dev, int32_t action_dtype, float* obs_dev, float* reward_dev,
             uint8_t* done_dev, void* stream) {
  REQUIRE_READY(h);
  int rc = check_actions(actions_dev, action_dtype, "auv_step");
  if (rc) return rc;
  PAIR_CHECK_ON(h, stream, true, obs_dev);
  if (h->fw.on) {
    const int32_t whole[2] = {0, h->d.n};
    rc = fw_tick(h, 1, whole, &stream);
    if (rc) return rc;
  }
  rc = enqueue_step(h, effective_mode(h, h->d.n), 0, h->d.n, actions_dev, action_dtype, obs_dev, reward_dev, done_dev,
                    (hipStream_t)stream, false);
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return AUV_OK;
}

int auv_step_slice(auv_handle_t* h, int32_t e0, int32_t ne, const void* actions_dev, int32_t action_dtype, float* obs_dev,
                   float* reward_dev, uint8_t* done_dev, void* stream) {
  REQUIRE_READY(h);
  int rc = check_actions(actions_dev, action_dtype, "auv_step_slice");
  if (rc) return rc;
  if (e0 < 0 || ne < 1 || (int64_t)e0 + ne > h->d.n) return fail(AUV_EINVAL, "auv_step_slice: slice [%d, %d) outside [0, %d)", e0, e0 + ne, h->d.n);
  PAIR_CHECK_ON(h, stream, true, obs_dev);
  rc = enqueue_step(h, effective_mode(h, ne), e0, ne, actions_dev, action_dtype, obs_dev, reward_dev, done_dev, (hipStream_t)stream, false);
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return AUV_OK;
}

int auv_step_pipelined(auv_handle_t* h, int32_t n_slices, const int32_t* bounds, void* const* streams, const void* actions_dev,
                       int32_t action_dtype, float* obs_dev, float* reward_dev, uint8_t* done_dev) {
  REQUIRE_READY(h);
  int rc = check_actions(actions_dev, action_dtype, "auv_step_pipelined");
  if (rc) return rc;
  rc = check_slices(h, n_slices, bounds, streams, "auv_step_pipelined");
  if (rc) return rc;
  PAIR_CHECK(h, obs_dev);
  rc = fw_tick(h, n_slices, bounds, streams);
  for (int i = 0; i < n_slices && rc == AUV_OK; i++)
    rc = enqueue_step(h, effective_mode(h, bounds[i + 1] - bounds[i]), bounds[i], bounds[i + 1] - bounds[i], actions_dev, action_dtype,
                      obs_dev, reward_dev, done_dev, (hipStream_t)streams[i], false);
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return AUV_OK;
}

int auv_step_multi(auv_handle_t* h, int32_t n_slices, const int32_t* bounds, void* const* streams, const void* actions_dev, int32_t action_dtype,
                   int32_t n_slots, int32_t first_slot, int32_t n_steps, float* obs_dev, float* reward_dev, uint8_t* done_dev) {
  REQUIRE_READY(h);
  int rc = check_actions(actions_dev, action_dtype, "auv_step_multi");
  if (rc) return rc;
  rc = check_slices(h, n_slices, bounds, streams, "auv_step_multi");
  if (rc) return rc;
  if (n_slots < 1 || first_slot < 0 || first_slot >= n_slots || n_steps < 1 || n_steps > 1024)
    return fail(AUV_EINVAL, "auv_step_multi: n_slots >= 1, 0 <= first_slot < n_slots, 1 <= n_steps <= 1024");
  if (h->fw.on) return fail(AUV_ESTATE, "auv_step_multi: not with a fresh world per reset (a slot's tables may be rebuilt beside the launch)");
  if (h->d.k_max > AUV_WAVE) return fail(AUV_ESTATE, "auv_step_multi: more than 64 obstacles per world");
  for (int i = 0; i < n_slices; i++) {
    const int ne = bounds[i + 1] - bounds[i];
    if (effective_mode(h, ne) != AUV_STEP_ONE_LAUNCH) return fail(AUV_ESTATE, "auv_step_multi: needs the one-launch shape for every slice");
    const long long wg = (long long)n_steps * (2 * (8 * ((ne + 63) / 64)) + 2 * (8 * ((ne + 7) / 8)));
    if (wg > 0x7fffffffll) return fail(AUV_EINVAL, "auv_step_multi: %lld workgroups in one launch", wg);
  }
  PAIR_CHECK(h, obs_dev);
  const unsigned long long seq0 = h->multi_seq;
  h->multi_seq += (unsigned long long)n_steps;
  for (int i = 0; i < n_slices; i++) {
    AuvDev d = h->d;
    d.e0 = bounds[i], d.ne = bounds[i + 1] - bounds[i];
    auv_launch_step_multi(d, actions_dev, action_dtype, obs_dev, reward_dev, done_dev, n_steps, first_slot, n_slots, seq0, h->multi_order, h->multi_lead,
                          h->multi_lag, (hipStream_t)streams[i]);
  }
  HIP_TRY(hipGetLastError());
  return AUV_OK;
}

int auv_set_multi_order(auv_handle_t* h, int32_t order, int32_t lead, int32_t lag) {
  if (!h || order < 0 || order > 1 || lead < 0 || lag < 0 || lead > 4096 || lag > 4096) return fail(AUV_EINVAL, "auv_set_multi_order: order 0 / 1, lead and lag in [0, 4096]");
  h->multi_order = order, h->multi_lead = lead, h->multi_lag = lag;
  return AUV_OK;
}

int auv_lidar_stage(auv_handle_t* h, int32_t segments, int32_t* out_segments) {
  REQUIRE_READY(h);
  if (segments != 0) {
    if (segments < 32 || segments > 96 || (segments & 1)) return fail(AUV_EINVAL, "auv_lidar_stage: segments must be 0 (report) or even, 32 .. 96");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());                      // (launches in flight carry the old stage by value: let them end first)
    AuvDev d = h->d;
    d.seg_cap = segments;
    if (!auv_k23_ok(d)) return fail(AUV_EINVAL, "auv_lidar_stage: a %d-segment slice cannot hold the navigation's chunk list", segments);
    HIP_TRY(auv_k2_prepare(d));
    HIP_TRY(auv_step_fused_prepare(d));
    h->d.seg_cap = segments;
    if (h->fw.on) h->fw.shadow.seg_cap = segments;         // (the refill pass's captured graph keeps the stage it was captured with)
  }
  if (out_segments) *out_segments = h->d.seg_cap;
  return AUV_OK;
}

int auv_step_async(auv_handle_t* h, int32_t n_slices, const int32_t* bounds, void* const* streams, const void* actions_dev,
                   int32_t action_dtype, float* obs_dev, float* reward_dev, uint8_t* done_dev, void* caller_stream, int32_t rendezvous) {
  REQUIRE_READY(h);
  int rc = check_actions(actions_dev, action_dtype, "auv_step_async");
  if (rc) return rc;
  rc = check_slices(h, n_slices, bounds, streams, "auv_step_async");
  if (rc) return rc;
  if (rendezvous != AUV_RDV_EVENTS && rendezvous != AUV_RDV_DEVICE && rendezvous != AUV_RDV_CP)
    return fail(AUV_EINVAL, "auv_step_async: rendezvous must be one of AUV_RDV_*");
  if (h->async_pending) return fail(AUV_ESTATE, "auv_step_async: the previous step has not been waited for (auv_step_wait)");
  PAIR_CHECK(h, obs_dev);
  HIP_TRY(hipSetDevice(h->device));
  rc = fw_tick(h, n_slices, bounds, streams);
  if (rc) return rc;
  hipStream_t cs = (hipStream_t)caller_stream;
  h->async_streams.clear();
  for (int i = 0; i < n_slices; i++)
    if ((hipStream_t)streams[i] != cs) h->async_streams.push_back((hipStream_t)streams[i]);
  const size_t nr = h->async_streams.size();         // chains on streams of their own: the others are in stream order already
  if (nr && rendezvous != AUV_RDV_EVENTS && !h->rdv) return fail(AUV_ESTATE, "auv_step_async: rendezvous words missing (no bank loaded?)");
  if (nr && rendezvous == AUV_RDV_DEVICE) {
    // The device-word rendezvous needs (i) the one-launch shape for every slice -- a gate that runs out stops its step by
    // ABORT packets, which only that shape reads -- and (ii) streams that run side by side NOW.  (ii) is tried before the
    // first real step on a set of streams (rdv_trial: 50 ms, no step at stake); a trial or a real wait that has run out
    // demotes the handle to events for good.
    bool one_launch = true;
    for (int i = 0; i < n_slices; i++) one_launch = one_launch && effective_mode(h, bounds[i + 1] - bounds[i]) == AUV_STEP_ONE_LAUNCH;
    if (!one_launch || !h->rdv_device_ok) rendezvous = AUV_RDV_EVENTS;
    else if (h->rdv_tried != h->async_streams && !stream_capturing(cs)) {
      int rc_t = rdv_trial(h, cs);
      if (rc_t) return rc_t;
      if (!h->rdv_device_ok) rendezvous = AUV_RDV_EVENTS;
    }
  }
  if (nr && rendezvous == AUV_RDV_EVENTS) {
    if (!h->ev_actions) HIP_TRY(hipEventCreateWithFlags(&h->ev_actions, hipEventDisableTiming));
    while (h->ev_chain.size() < nr) {
      hipEvent_t e;
      HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      h->ev_chain.push_back(e);
    }
    HIP_TRY(hipEventRecord(h->ev_actions, cs));                                    // behind whatever produced the actions
  } else if (nr) {
    h->rdv_seq += 1;
    if (rendezvous == AUV_RDV_DEVICE) auv_launch_rdv_publish(h->rdv, h->rdv_seq, cs);
    else HIP_TRY(hipStreamWriteValue64(cs, h->rdv, h->rdv_seq, 0));
  }
  size_t j = 0;
  for (int i = 0; i < n_slices && rc == AUV_OK; i++) {
    hipStream_t st = (hipStream_t)streams[i];
    const bool remote = st != cs;
    if (remote) {
      if (rendezvous == AUV_RDV_EVENTS) HIP_TRY(hipStreamWaitEvent(st, h->ev_actions, 0));
      else if (rendezvous == AUV_RDV_DEVICE) auv_launch_rdv_wait(h->rdv, h->rdv_seq, h->d.pair_error, 4, h->rdv_limit_s, h->d.abort_flag, h->d.abort_flag + 1, st);
      else HIP_TRY(hipStreamWaitValue64(st, h->rdv, h->rdv_seq, hipStreamWaitValueGte, ~0ull));
    }
    rc = enqueue_step(h, effective_mode(h, bounds[i + 1] - bounds[i]), bounds[i], bounds[i + 1] - bounds[i], actions_dev, action_dtype,
                      obs_dev, reward_dev, done_dev, st, false);
    if (remote && rc == AUV_OK) {
      if (rendezvous == AUV_RDV_EVENTS) HIP_TRY(hipEventRecord(h->ev_chain[j], st));
      else if (rendezvous == AUV_RDV_DEVICE) auv_launch_rdv_arrive(h->rdv + 16, st);
      else HIP_TRY(hipStreamWriteValue64(st, h->rdv + 32 + 16 * j, h->rdv_seq, 0));
      j++;
    }
  }
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  if (rendezvous == AUV_RDV_DEVICE) h->rdv_target += nr;
  h->async_pending = 1 + rendezvous;
  return AUV_OK;
}

int auv_step_wait(auv_handle_t* h, void* caller_stream) {
  REQUIRE_READY(h);
  if (!h->async_pending) return fail(AUV_ESTATE, "auv_step_wait: no step pending (auv_step_async)");
  const int rendezvous = h->async_pending - 1;
  hipStream_t cs = (hipStream_t)caller_stream;
  const size_t nr = h->async_streams.size();
  h->async_pending = 0;
  if (nr) {
    if (rendezvous == AUV_RDV_EVENTS) {
      for (size_t j = 0; j < nr; j++) HIP_TRY(hipStreamWaitEvent(cs, h->ev_chain[j], 0));
    } else if (rendezvous == AUV_RDV_DEVICE) {
      auv_launch_rdv_wait(h->rdv + 16, h->rdv_target, h->d.pair_error, 5, h->rdv_limit_s, nullptr, h->d.abort_flag + 1, cs);   // ONE wait for all chains
      HIP_TRY(hipGetLastError());
    } else {
      for (size_t j = 0; j < nr; j++) HIP_TRY(hipStreamWaitValue64(cs, h->rdv + 32 + 16 * j, h->rdv_seq, hipStreamWaitValueGte, ~0ull));
    }
  }
  return AUV_OK;
}

int auv_set_rendezvous_limit(auv_handle_t* h, double seconds) {
  if (!h || !(seconds > 0.0) || seconds > 3600.0) return fail(AUV_EINVAL, "auv_set_rendezvous_limit: seconds must be in (0, 3600]");
  h->rdv_limit_s = seconds;
  return AUV_OK;
}

int auv_graph_capture_chains(auv_handle_t* h, int32_t n_slices, const int32_t* bounds, const void* actions_dev, int32_t action_dtype,
                             float* obs_dev, float* reward_dev, uint8_t* done_dev, int32_t n_steps, int32_t one_graph) {
  REQUIRE_READY(h);
  int rc = check_actions(actions_dev, action_dtype, "auv_graph_capture_chains");
  if (rc) return rc;
  rc = check_slices(h, n_slices, bounds, bounds, "auv_graph_capture_chains");   // (the chains' streams are named at replay)
  if (rc) return rc;
  if (n_steps < 1 || n_steps > 4096) return fail(AUV_EINVAL, "auv_graph_capture_chains: n_steps must be in [1, 4096]");
  PAIR_CHECK(h, obs_dev);
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  drop_graphs(h);
  if (h->graph) {
    HIP_TRY(hipGraphDestroy(h->graph));
    h->graph = nullptr;
  }
  if (!h->cap_stream) HIP_TRY(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
  // every chain starts at slot 0 of the ring and keeps a position of its own (advanced by its own dynamics waves)
  HIP_TRY(hipMemset(h->d.ring_pos, 0, AUV_MAX_CHAINS * sizeof(int32_t)));
  HIP_TRY(hipMemset(h->d.k1_done, 0, AUV_MAX_CHAINS * sizeof(int32_t)));
  h->chain_bounds.assign(bounds, bounds + n_slices + 1), h->chain_steps = n_steps, h->graph_steps = n_steps;
  if (!one_graph) {
    // K linear graphs, replayed on K streams of the caller's choice (auv_graph_launch_chains): which hardware queue a
    // chain runs on stays the caller's decision, as for eager chains
    for (int i = 0; i < n_slices; i++) {
      const int ne = bounds[i + 1] - bounds[i];
      HIP_TRY(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
      for (int32_t k = 0; k < n_steps && rc == AUV_OK; k++)
        rc = enqueue_step(h, effective_mode(h, ne), bounds[i], ne, actions_dev, action_dtype, obs_dev, reward_dev, done_dev, h->cap_stream,
                          true, false, false, i);
      hipGraph_t g = nullptr;
      hipError_t ce = hipStreamEndCapture(h->cap_stream, &g);
      if (rc) return rc;
      HIP_TRY(ce);
      h->chain_graph.push_back(g);
      hipGraphExec_t ge = nullptr;
      HIP_TRY(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      h->chain_exec.push_back(ge);
    }
    return AUV_OK;
  }
  // ONE graph whose K branches are the chains (fork behind the root, join at the end): replayed with auv_graph_launch,
  // the runtime picks the streams of the branches
  while ((int)h->fork_streams.size() < n_slices - 1) {
    hipStream_t st;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    h->fork_streams.push_back(st);
  }
  while ((int)h->ev_chain.size() < n_slices) {
    hipEvent_t e;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    h->ev_chain.push_back(e);
  }
  HIP_TRY(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
  hipError_t fe = hipEventRecord(h->ev_chain[0], h->cap_stream);
  for (int i = 1; i < n_slices && fe == hipSuccess; i++) fe = hipStreamWaitEvent(h->fork_streams[i - 1], h->ev_chain[0], 0);
  for (int i = 0; i < n_slices && rc == AUV_OK && fe == hipSuccess; i++) {
    const int ne = bounds[i + 1] - bounds[i];
    hipStream_t st = i == 0 ? h->cap_stream : h->fork_streams[i - 1];
    for (int32_t k = 0; k < n_steps && rc == AUV_OK; k++)
      rc = enqueue_step(h, effective_mode(h, ne), bounds[i], ne, actions_dev, action_dtype, obs_dev, reward_dev, done_dev, st, true, false,
                        false, i);
  }
  for (int i = 1; i < n_slices && fe == hipSuccess; i++) {
    fe = hipEventRecord(h->ev_chain[i], h->fork_streams[i - 1]);
    if (fe == hipSuccess) fe = hipStreamWaitEvent(h->cap_stream, h->ev_chain[i], 0);
  }
  hipError_t ce = hipStreamEndCapture(h->cap_stream, &h->graph);
  if (rc) return rc;
  HIP_TRY(fe);
  HIP_TRY(ce);
  HIP_TRY(hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
  return AUV_OK;
}

int auv_graph_launch_chains(auv_handle_t* h, int32_t n_slices, void* const* streams) {
  REQUIRE_READY(h);
  if (!streams || n_slices != (int32_t)h->chain_exec.size() || n_slices < 1)
    return fail(AUV_ESTATE, "auv_graph_launch_chains: %d streams for %d captured chains", n_slices, (int)h->chain_exec.size());
  PAIR_CHECK(h, nullptr);
  {
    int rc_t = fw_tick(h, n_slices, h->chain_bounds.data(), streams, h->chain_steps);
    if (rc_t) return rc_t;
  }
  for (int i = 0; i < n_slices; i++) HIP_TRY(hipGraphLaunch(h->chain_exec[i], (hipStream_t)streams[i]));
  return AUV_OK;
}

int auv_step_pipelined_timed(auv_handle_t* h, int32_t n_slices, const int32_t* bounds, void* const* streams, const void* actions_dev,
                             int32_t action_dtype, float* obs_dev, float* reward_dev, uint8_t* done_dev, float* out_ms) {
  REQUIRE_READY(h);
  int rc = check_actions(actions_dev, action_dtype, "auv_step_pipelined_timed");
  if (rc) return rc;
  if (n_slices < 1 || !bounds || !streams || !out_ms) return fail(AUV_EINVAL, "auv_step_pipelined_timed: bad arguments");
  if (bounds[0] != 0 || bounds[n_slices] != h->d.n) return fail(AUV_EINVAL, "auv_step_pipelined_timed: bounds must run from 0 to %d", h->d.n);
  PAIR_CHECK(h, obs_dev);
  while ((int)h->slice_ev.size() < 2 * n_slices) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    h->slice_ev.push_back(e);
  }
  for (int i = 0; i < n_slices; i++) {
    const int ne = bounds[i + 1] - bounds[i];
    if (ne < 1) return fail(AUV_EINVAL, "auv_step_pipelined_timed: empty slice %d", i);
    if (effective_mode(h, ne) != AUV_STEP_ONE_LAUNCH) return fail(AUV_ESTATE, "auv_step_pipelined_timed: needs the one-launch shape");
    AuvDev d = h->d;
    d.ring_slots = 1;
    d.e0 = bounds[i], d.ne = ne;
    auv_launch_step_roles(d, actions_dev, action_dtype, obs_dev, reward_dev, done_dev, (hipStream_t)streams[i], h->slice_ev[2 * i],
                          h->slice_ev[2 * i + 1]);
  }
  HIP_TRY(hipGetLastError());
  for (int i = 0; i < n_slices; i++) {
    HIP_TRY(hipEventSynchronize(h->slice_ev[2 * i + 1]));
    HIP_TRY(hipEventElapsedTime(&out_ms[i], h->slice_ev[2 * i], h->slice_ev[2 * i + 1]));
  }
  return AUV_OK;
}

int auv_episode_log(auv_handle_t* h, double* dst_dev, int64_t max_rows, int64_t first, int64_t* out_total, int64_t* out_first,
                    void* stream) {
  REQUIRE_READY(h);
  if (!out_total || (max_rows > 0 && !dst_dev) || max_rows < 0 || first < 0) return fail(AUV_EINVAL, "auv_episode_log: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  unsigned long long total_u = 0;
  HIP_TRY(hipMemcpyAsync(&total_u, h->d.ep_log_count, sizeof(total_u), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const int64_t total = (int64_t)total_u;
  *out_total = total;
  const int64_t cap = h->d.ep_log_cap;
  if (first > total) return fail(AUV_EINVAL, "auv_episode_log: first = %lld beyond the %lld episodes logged", (long long)first, (long long)total);
  // rows the ring has overwritten since `first` are gone: the read starts at the oldest row still held and says so
  // (out_first - first rows were dropped) instead of failing -- a reader that fell behind catches up with this call
  if (total - first > cap) first = total - cap;
  if (out_first) *out_first = first;
  int64_t nrow = total - first;
  if (nrow > max_rows) nrow = max_rows;
  // rows first .. first + nrow of the ring, in at most two pieces
  int64_t done = 0;
  while (done < nrow) {
    const int64_t pos = (first + done) & (cap - 1);
    const int64_t piece = (nrow - done < cap - pos) ? nrow - done : cap - pos;
    HIP_TRY(hipMemcpyAsync(dst_dev + 8 * done, h->d.ep_log + 8 * pos, (size_t)piece * 64, hipMemcpyDeviceToDevice, st));
    done += piece;
  }
  return AUV_OK;
}

int auv_streams_overlap(auv_handle_t* h, void* stream_a, void* stream_b, float* out_ratio) {
  if (!h || !out_ratio) return fail(AUV_EINVAL, "auv_streams_overlap: bad arguments");
  HIP_TRY(hipSetDevice(h->device));
  const double spin_us = 300.0;
  const unsigned long long ticks = (unsigned long long)(spin_us * 100.0);          // wall_clock64 runs at 100 MHz
  hipStream_t sa = (hipStream_t)stream_a, sb = (hipStream_t)stream_b;
  auv_launch_spin(100, sa), auv_launch_spin(100, sb);                               // (code object loaded, queues awake)
  HIP_TRY(hipStreamSynchronize(sa));
  HIP_TRY(hipStreamSynchronize(sb));
  const auto t0 = std::chrono::steady_clock::now();
  auv_launch_spin(ticks, sa);
  auv_launch_spin(ticks, sb);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(sa));
  HIP_TRY(hipStreamSynchronize(sb));
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  *out_ratio = (float)(us / spin_us);
  return AUV_OK;
}

int auv_health(auv_handle_t* h, int32_t* out8) {
  if (!h || !out8) return fail(AUV_EINVAL, "auv_health: bad arguments");
  out8[0] = h->handover_ok ? 1 : 0;
  out8[1] = h->probe_failures;
  out8[2] = h->handover_timeouts;
  out8[3] = (h->pair_error_host && *(volatile int32_t*)h->pair_error_host) ? 1 : 0;   // a time-out not yet recovered from
  out8[4] = h->last_timeout_e0, out8[5] = h->last_timeout_ne, out8[6] = h->last_reset_envs;
  out8[7] = h->rdv_device_ok ? 0 : (h->rdv_timeouts > 0 ? h->rdv_timeouts : 1);   // > 0: rendezvous waits that ran out (trial or real): events from then on
  return AUV_OK;
}

int auv_probe_streams(auv_handle_t* h, int32_t n_streams, void* const* streams) {
  REQUIRE_READY(h);
  if (n_streams < 1 || n_streams > AUV_MAX_CHAINS || !streams) return fail(AUV_EINVAL, "auv_probe_streams: 1 .. %d streams", AUV_MAX_CHAINS);
  HIP_TRY(hipSetDevice(h->device));
  std::vector<hipStream_t> st(n_streams);
  for (int i = 0; i < n_streams; i++) st[i] = (hipStream_t)streams[i];
  return probe_streams(h, n_streams, st.data(), true);
}

int auv_effective_step_mode(auv_handle_t* h, int32_t n_envs_per_launch) {
  if (!h) return fail(AUV_EINVAL, "null handle");
  return effective_mode(h, n_envs_per_launch > 0 ? n_envs_per_launch : h->d.n);
}

int auv_set_action_ring(auv_handle_t* h, int32_t n_slots) {
  REQUIRE_READY(h);
  if (n_slots < 1) return fail(AUV_EINVAL, "auv_set_action_ring: n_slots must be >= 1");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(h->d.ring_pos, 0, AUV_MAX_CHAINS * sizeof(int32_t)));
  h->d.ring_slots = n_slots;
  drop_graphs(h);
  return AUV_OK;
}

int auv_set_step_mode(auv_handle_t* h, int32_t mode) {
  if (!h) return fail(AUV_EINVAL, "null handle");
  if (mode != AUV_STEP_SIDE_BY_SIDE && mode != AUV_STEP_ONE_LAUNCH && mode != AUV_STEP_AUTO)
    return fail(AUV_EINVAL, "auv_set_step_mode: mode must be one of AUV_STEP_*");
  h->step_mode = mode;
  drop_graphs(h);
  return AUV_OK;
}

#ifdef AUV_CUTS
// Only in the diagnostic build (tools/build_variant.sh cuts "-DAUV_CUTS"; tools/valu_budget.py): switch off the
// LiDAR role's phases from `cut_lidar` on and the navigation role's from `cut_nav` on (0 = run everything).
int auv_diag_cuts(auv_handle_t* h, int32_t cut_lidar, int32_t cut_nav) {
  if (!h) return fail(AUV_EINVAL, "null handle");
  h->d.cut_lidar = cut_lidar, h->d.cut_nav = cut_nav;
  return AUV_OK;
}
#endif

#ifdef AUV_TEST_HOOKS
// Only in libauv_hip_hooks.so (make hooks): skew = idle workgroups between the roles of the one-launch
// shapes (an environment's waves then sit on different XCDs); fault: the first environment of every launch never gets
// its sweep's word (1), its state packet (2) or its search record (3), so the polls that wait for them run out.
int auv_test_hooks(auv_handle_t* h, int32_t skew, int32_t fault) {
  if (!h) return fail(AUV_EINVAL, "null handle");
  h->d.pair_skew = (skew > 0 && skew < 8) ? skew : 0;
  h->d.pair_fault = (fault >= 1 && fault <= 3) ? fault : 0;
  if (h->d.self) HIP_TRY(hipMemcpy((void*)h->d.self, &h->d, sizeof(AuvDev), hipMemcpyHostToDevice));
  return AUV_OK;
}
#endif

int auv_step_dynamics(auv_handle_t* h, const void* actions_dev, int32_t action_dtype, void* stream) {
  REQUIRE_READY(h);
  {
    int rc_a = check_actions(actions_dev, action_dtype, "auv_step_dynamics");
    if (rc_a) return rc_a;
  }
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
    case AUV_FIELD_BROKEN: *bytes = n; return d.broken;
    case AUV_FIELD_FW_STATE: *bytes = d.fw_state ? 4 * (size_t)d.n_worlds : 0; return d.fw_state;
    case AUV_FIELD_FW_SERIAL: *bytes = d.fw_serial ? 4 * (size_t)d.n_worlds : 0; return d.fw_serial;
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

size_t auv_policy_param_floats(int32_t obs_dim) { return obs_dim > 0 ? auv_policy_param_floats_impl(obs_dim) : 0; }

static int check_policy_io(const auv_handle_t* h, int32_t e0, int32_t ne, const auv_policy_io_t* io, const char* who) {
  if (!io) return fail(AUV_EINVAL, "%s: null io", who);
  const auv_config_t& c = h->d.cfg;
  const int D = 6 + (c.use_lidar ? c.n_sensors * (c.obs_channels == 3 ? 3 : 1) : 0);
  if (io->obs_dim != D) return fail(AUV_EINVAL, "%s: obs_dim %d, the handle's observation has %d columns", who, io->obs_dim, D);
  if (e0 < 0 || ne < 1 || (int64_t)e0 + ne > h->d.n) return fail(AUV_EINVAL, "%s: slice [%d, %d) outside [0, %d)", who, e0, e0 + ne, h->d.n);
  if (io->T < 1) return fail(AUV_EINVAL, "%s: T must be >= 1", who);
  if (io->env_base < 0 || io->env_base > e0 || (int64_t)e0 + ne - io->env_base > io->ld)
    return fail(AUV_EINVAL, "%s: slice [%d, %d) does not fit a rollout row of %d environments starting at environment %d", who, e0, e0 + ne, io->ld, io->env_base);
  if (!io->obs || !io->params || !io->ctr || !io->reward_in || !io->done_in || !io->actions_out || !io->A || !io->LP || !io->V || !io->R || !io->Dn)
    return fail(AUV_EINVAL, "%s: null buffer", who);
  if (((uintptr_t)io->params & 15) || ((uintptr_t)io->params_bf16 & 15) || ((uintptr_t)io->actions_out & 7) || ((uintptr_t)io->ctr & 7))
    return fail(AUV_EINVAL, "%s: params must be 16-byte, actions_out and ctr 8-byte aligned", who);
  if (auv_policy_lds_bytes(io->obs_dim) > 160 * 1024) return fail(AUV_EINVAL, "%s: observation too wide for the policy kernel's LDS tile", who);
  return AUV_OK;
}

int auv_policy_act(auv_handle_t* h, int32_t e0, int32_t ne, const auv_policy_io_t* io, void* stream) {
  REQUIRE_READY(h);
  int rc = check_policy_io(h, e0, ne, io, "auv_policy_act");
  if (rc) return rc;
  HIP_TRY(auv_policy_prepare(io->obs_dim));
  auv_launch_policy(*io, e0, ne, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return AUV_OK;
}

int auv_gae(auv_handle_t* h, const float* R, const float* V, const float* Dn, const float* last_v, float gamma, float lam, float* adv_out,
            float* ret_out, int32_t T, int32_t N, void* stream) {
  if (!h) return fail(AUV_EINVAL, "null handle");
  if (!R || !V || !Dn || !last_v || !adv_out || !ret_out || T < 1 || N < 1) return fail(AUV_EINVAL, "auv_gae: bad arguments");
  HIP_TRY(hipSetDevice(h->device));
  auv_launch_gae(R, V, Dn, last_v, gamma, lam, adv_out, ret_out, T, N, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return AUV_OK;
}

int auv_policy_rollout(auv_handle_t* h, int32_t n_slices, const int32_t* bounds, void* const* streams, const auv_policy_io_t* ios,
                       float* obs_dev, float* reward_dev, uint8_t* done_dev, int32_t n_steps, int32_t flush, const int64_t* t0,
                       const int64_t* gstep0) {
  REQUIRE_READY(h);
  int rc = check_slices(h, n_slices, bounds, streams, "auv_policy_rollout");
  if (rc) return rc;
  if (!ios || n_steps < 0 || (!t0) != (!gstep0)) return fail(AUV_EINVAL, "auv_policy_rollout: bad arguments");
  for (int i = 0; i < n_slices; i++) {
    rc = check_policy_io(h, bounds[i], bounds[i + 1] - bounds[i], ios + i, "auv_policy_rollout");
    if (rc) return rc;
    rc = check_actions(ios[i].actions_out, AUV_F32, "auv_policy_rollout");
    if (rc) return rc;
    if (t0 && (t0[i] < 0 || gstep0[i] < 0)) return fail(AUV_EINVAL, "auv_policy_rollout: negative counter for slice %d", i);
  }
  PAIR_CHECK(h, obs_dev);
  HIP_TRY(auv_policy_prepare(ios[0].obs_dim));
  // t0 / gstep0 (per slice; both or neither): the host names every launch's rollout position and generator step (this is a
  // plain loop of launches, the values are known here) -- the launches then neither read nor count off on io.ctr, which
  // is worth ~4 us per launch.  NULL: the device counters, as auv_policy_act.
  for (int32_t k = 0; k < n_steps; k++) {
    if (rc == AUV_OK) rc = fw_tick(h, n_slices, bounds, streams);
    for (int i = 0; i < n_slices && rc == AUV_OK; i++) {
      const int ne = bounds[i + 1] - bounds[i];
      auv_launch_policy(ios[i], bounds[i], ne, (hipStream_t)streams[i], t0 ? t0[i] + k : -1, t0 ? gstep0[i] + k : -1);
      rc = enqueue_step(h, effective_mode(h, ne), bounds[i], ne, ios[i].actions_out, AUV_F32, obs_dev, reward_dev, done_dev,
                        (hipStream_t)streams[i], false);
    }
  }
  if (rc) return rc;
  if (flush)
    for (int i = 0; i < n_slices; i++)
      auv_launch_policy(ios[i], bounds[i], bounds[i + 1] - bounds[i], (hipStream_t)streams[i], t0 ? t0[i] + n_steps : -1,
                        t0 ? gstep0[i] + n_steps : -1);
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
  if (n_steps < 1 || n_steps > 4096) return fail(AUV_EINVAL, "auv_graph_capture_steps: n_steps must be in [1, 4096]");
  {
    int rc_a = check_actions(actions_dev, action_dtype, "auv_graph_capture");
    if (rc_a) return rc_a;
  }
  PAIR_CHECK(h, obs_dev);
  if (!h->cap_stream) HIP_TRY(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
  drop_graphs(h);
  if (h->graph) {
    HIP_TRY(hipGraphDestroy(h->graph));
    h->graph = nullptr;
  }
  HIP_TRY(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
  int rc = AUV_OK;
  // inside the graph, step k's reward phase and step k+1's dynamics share a launch (side-by-side shape with a LiDAR
  // sweep: the shapes whose reward kernel maps lanes to environments)
  // (also when the handle steps in the one-launch shape: replayed launches cost ~3 us more than eager ones on this
  // stack, and inside a graph of several steps the fused reward + dynamics launch makes up for more of that than the
  // one launch does -- 97.6 M against 96.6 M env-steps/s at 16 steps per graph, 95.3 against 90.6 M at 8192 x 256 --
  // and the bits are the same.  A graph of ONE step keeps the handle's own shape.)
  h->graph_steps = n_steps;
  const bool fuse = n_steps > 1 && auv_k23_ok(h->d) && h->d.cfg.use_lidar;
  const int mode = fuse ? AUV_STEP_SIDE_BY_SIDE : effective_mode(h, h->d.n);
  for (int32_t k = 0; k < n_steps && rc == AUV_OK; k++)
    rc = enqueue_step(h, mode, 0, h->d.n, actions_dev, action_dtype, obs_dev, reward_dev, done_dev, h->cap_stream, true,
                      fuse && k > 0, fuse && k + 1 < n_steps);
  hipError_t ce = hipStreamEndCapture(h->cap_stream, &h->graph);
  if (rc) return rc;
  HIP_TRY(ce);
  HIP_TRY(hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
  return AUV_OK;
}

int auv_graph_launch(auv_handle_t* h, void* stream) {
  REQUIRE_READY(h);
  if (!h->graph_exec) return fail(AUV_ESTATE, "auv_graph_launch: no captured graph");
  PAIR_CHECK(h, nullptr);
  if (h->fw.on) {
    const int32_t whole[2] = {0, h->d.n};
    int rc_t = fw_tick(h, 1, whole, &stream, h->graph_steps);
    if (rc_t) return rc_t;
  }
  HIP_TRY(hipGraphLaunch(h->graph_exec, (hipStream_t)stream));
  return AUV_OK;
}

int auv_step_timed(auv_handle_t* h, const void* actions_dev, int32_t action_dtype, float* obs_dev, float* reward_dev,
                   uint8_t* done_dev, void* stream, float* out_ms4) {
  REQUIRE_READY(h);
  if (!out_ms4) return fail(AUV_EINVAL, "auv_step_timed: null argument");
  {
    int rc_a = check_actions(actions_dev, action_dtype, "auv_step_timed");
    if (rc_a) return rc_a;
  }
  PAIR_CHECK(h, obs_dev);
  hipStream_t st = (hipStream_t)stream;
  for (auto& e : h->ev)
    if (!e) HIP_TRY(hipEventCreate(&e));
  AuvDev d = h->d;
  d.ring_slots = 1;   // eager: `actions_dev` is one plain [N][2] buffer (see enqueue_step)
  // every dispatch of the step is stamped with its own start and stop event (hipExtLaunchKernel): the
  // elapsed times are the kernels' own durations, as a kernel trace reports them, without the gaps
  int nk;
  const int mode = effective_mode(h, d.n);
  if (mode == AUV_STEP_ONE_LAUNCH) {
    auv_launch_step_roles(d, actions_dev, action_dtype, obs_dev, reward_dev, done_dev, st, h->ev[0], h->ev[1]);
    nk = 1;
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
