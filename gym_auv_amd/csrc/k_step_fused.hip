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

#ifndef AUV_K23_MIN_WAVES
#define AUV_K23_MIN_WAVES 4   // waves per SIMD the LiDAR launches are compiled for (128 VGPRs)
#endif

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
  k3_nav_env(d, e, lane, (unsigned char*)L.stage, obs_out, &pre);
  auv_wave_lds_sync();
  // K2
  int collision = 0;
  const int n_act = k2_front(d, e, lane, L, 1, &pre);
  if (d.cfg.use_lidar) {
    k2_stage_and_pairs(d, L, lane, n_act, pre.s[2]);
    collision = k2_back(d, e, lane, L, n_act, obs_out);
  }
  // K3-reward (lidar_d / closeness rows are re-read by the lanes that wrote them)
  k3_reward_env(d, e, lane, true, obs_out, reward_out, done_out, &pre, collision, false, !d.cfg.use_lidar);
}

// ---- the paired finish: reward / done / auto-reset inside the side-by-side launch -------------------
// An environment's LiDAR wave and its navigation wave are two one-wave workgroups of the same launch; the
// navigation wave also runs the reward phase (rewarder.py:78-140, :167-241; environment.py:333-347, :375-384) --
// no third launch.  What it needs of the sweep is one 64-bit word per environment (the LiDAR term of the reward,
// or a marker for "collision"): the LiDAR wave stores it last, the navigation wave requests it when it starts
// (almost always it is there by then: navigation workgroups are dispatched behind all LiDAR workgroups and get
// their slot when a sweep retires), consumes it at its end and puts the "empty" marker back.
//   Coherence without device-scope fences (their L2 write-back per wave is what made round 1's attempt 5x slower):
//   * the word goes by relaxed agent-scope atomic store / load (sc1: written through and read past the XCD's
//     L2, which is not coherent with the other seven);
//   * every row the LiDAR wave writes is stored write-through too (WT = true in k2_front / k2_back), and the wave
//     waits for the completion of all its stores before it stores the word.  A navigation wave that has seen the
//     word therefore knows that nothing of its environment is in flight or dirty in another L2: its plain stores
//     (reward phase; restore_env, which overwrites the sweep's rows when the episode ended) are the last word
//     whichever XCDs the two ran on.  Rows of other environments share cache lines but not bytes.
//   * the launch places the two waves of an environment on the same XCD (workgroups go round-robin over the
//     eight XCDs; the navigation role starts at a multiple of 8), which keeps the word in one L2; correctness
//     does not depend on it (tests run with the roles skewed onto different XCDs).
//   * a navigation wave whose sweep is still running (a handful per launch) polls.  Its LiDAR workgroup has a
//     smaller index in the same launch, so it was dispatched earlier and finishes without needing anything from
//     anyone; the poll is bounded all the same: when it runs out the wave reports through `pair_error`
//     (auv_step fails from then on) instead of hanging the device.
#define PAIR_EMPTY AUV_PAIR_EMPTY
#define PAIR_COLLISION AUV_PAIR_COLLISION
#define PAIR_POLL_LIMIT (1 << 22)

__device__ __forceinline__ void pair_publish_lidar(const AuvDev& d, const int e, const int lane, const int collision,
                                                   const double term) {
  auv_stores_done();                                       // of every lane of this wave (one counter per wave)
  if (d.pair_fault && e == 0) return;                      // (test hook: tests/test_gpu_parity.py, the poll's time-out)
  if (lane == 0)
    __hip_atomic_store(d.pair_word + e, collision ? PAIR_COLLISION : (unsigned long long)__double_as_longlong(term),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// what the navigation wave requests before it starts working (one trip, hidden behind the search)
struct PairPre {
  unsigned long long word;
  double cum;
  int4 cnt;
  int w;
};
__device__ __forceinline__ PairPre pair_prefetch(const AuvDev& d, const int e) {
  PairPre p;
  p.word = __hip_atomic_load(d.pair_word + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // (written by earlier launches: the dynamics kernel and the previous step's reward phase)
  p.cum = d.info64[8 * (size_t)e + 4];
  p.cnt = d.counters[e];
  p.w = d.world_idx[e];
  return p;
}

__device__ __forceinline__ unsigned long long pair_uniform(const unsigned long long v) {   // (readfirstlane returns int)
  return (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v) |
         ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32);
}

__device__ __forceinline__ void pair_finish_nav(const AuvDev& dk, const int e, const int lane, PairPre p, const NavOut no,
                                                float* __restrict__ obs_out, float* __restrict__ reward_out,
                                                uint8_t* __restrict__ done_out) {
  unsigned long long word = pair_uniform(p.word);
  const int limit = dk.pair_fault ? (1 << 12) : PAIR_POLL_LIMIT;
  for (int polls = 0; word == PAIR_EMPTY; polls++) {
    if (polls == limit) {
      if (lane == 0) __hip_atomic_store(dk.pair_error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    __builtin_amdgcn_s_sleep(8);
    const unsigned long long t = __hip_atomic_load(dk.pair_word + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    word = pair_uniform(t);
  }
#ifdef AUV_STAMPS
  if (lane == 0) dk.stamps[(size_t)e * 16 + 7] = wall_clock64();   // the sweep's word is here: start of the reward phase
#endif
  int do_reset = 0;
  int4 cnt = p.cnt;
  const AuvDev& d = dk;
  if (lane == 0) {
    __hip_atomic_store(d.pair_word + e, PAIR_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next step
    RewardIn in;
    const int collision = word == PAIR_COLLISION;
    in.closeness_reward = collision ? 0.0 : __longlong_as_double((long long)word);
    in.path_reward = no.rew_path, in.reached = no.reached, in.goal = no.goal, in.progress = no.progress;
    in.u = no.u, in.v = no.v, in.yaw_rate = no.r;
    in.cum = p.cum;
    d.info64[8 * (size_t)e] = collision;
    do_reset = reward_apply(d, e, collision, cnt, in, reward_out, done_out, false);
  }
  do_reset = __builtin_amdgcn_readfirstlane(do_reset);
  if (do_reset) {
    // (rare) the ~25 tables of the copy are read through the device-side copy of `d` (AuvDev::self), addressed as
    // constant memory: scalar loads at the point of use.  As kernel arguments they would be fetched -- and, the budget
    // of scalar registers being what it is, spilled -- at the entry of every wave of BOTH roles (1.4 us of the launch).
    const AuvDev& dc = *(const AuvDev*)(const __attribute__((address_space(4))) AuvDev*)d.self;
    restore_env(dc, e, (int)(((long long)__builtin_amdgcn_readfirstlane(p.w) + d.n) % d.n_worlds), lane,
                __builtin_amdgcn_readfirstlane(cnt.z), obs_out);
  }
}

// K2 and K3-nav of ALL environments in one launch: workgroups [0, nb) sweep the LiDAR of their
// environments, workgroups [nb, 2 nb) navigate theirs.  The two are independent given the
// new vessel state, so they run side by side (the LiDAR workgroups are dispatched first and fill
// the chip; navigation workgroups move in as those retire) -- the concurrency of two streams
// without the ~8 us a cross-stream event wait costs on each side.
// PAIRED: the navigation wave also runs the reward phase (pair_finish_nav above); one-wave workgroups, nb a
// multiple of 8 (+ pair_skew idle workgroups between the roles in the coherence tests).
#ifndef AUV_PAIR_WT
#define AUV_PAIR_WT 1   // (0: plain stores in the paired launch -- a measurement build only, see pair_finish)
#endif
template <bool PAIRED>
__global__ void __launch_bounds__(AUV_BLOCK, AUV_K23_MIN_WAVES) k23_lidar_nav(AuvDev dk, float* __restrict__ obs_out,
                                                                                 float* __restrict__ reward_out,
                                                                                 uint8_t* __restrict__ done_out) {
  // (every table through the device-side copy AuvDev::self instead of the kernel arguments -- no scalar register
  // spilled any more -- was tried for the whole kernel: the loads then sit on the dependent chains, 36.3 -> 42.2 us)
  const AuvDev& d = dk;
  extern __shared__ __align__(16) unsigned char smem[];
  const int wave = threadIdx.x / AUV_WAVE, lane = threadIdx.x % AUV_WAVE;
  const int S = d.cfg.n_sensors;
  const int wpb = blockDim.x / AUV_WAVE;                            // waves per workgroup
  const int nb = PAIRED ? 8 * ((d.n + 7) / 8) : (d.n + wpb - 1) / wpb;   // LiDAR workgroups: one env per wave
  const int nav0 = PAIRED ? nb + dk.pair_skew : nb;                 // first navigation workgroup
  const bool nav_role = (int)blockIdx.x >= nb;                      // workgroup-uniform
  constexpr bool WT = PAIRED && AUV_PAIR_WT;
  unsigned char* slice = smem + wave * k2_slice_bytes(S, d.k_max, d.m_max);
  if (nav_role) {
    const int e = auv_uniform(((int)blockIdx.x - nav0) * wpb + wave);
    if (e < 0 || e >= d.n) return;
#ifdef AUV_STAMPS
    const unsigned long long t_nav0 = wall_clock64();
#endif
    NavOut no;
    no.rew_path = no.reached = no.goal = no.progress = no.u = no.v = no.r = 0.0;
    PairPre pp;
    if constexpr (PAIRED) pp = pair_prefetch(d, e);
    k3_nav_env(d, e, lane, slice, obs_out, nullptr, nullptr, PAIRED ? &no : nullptr);
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 12] = t_nav0, d.stamps[(size_t)e * 16 + 13] = wall_clock64();
#endif
    if constexpr (PAIRED) pair_finish_nav(d, e, lane, pp, no, obs_out, reward_out, done_out);
#ifdef AUV_STAMPS
    if (PAIRED && lane == 0) d.stamps[(size_t)e * 16 + 15] = wall_clock64();     // end of the navigation wave incl. its finish
#endif
  } else {
    const int e = auv_uniform((int)blockIdx.x * wpb + wave);
    if (e >= d.n) return;
    // next action slot of a captured graph's ring: the dynamics kernel of this step has read the position, the
    // one of the next step has not been launched yet
    if (e == 0 && lane == 0 && dk.ring_slots > 1 && dk.ring_slot_host == -1) *dk.ring_pos = (*dk.ring_pos + 1) % dk.ring_slots;
    const Slice L = carve(slice, S, d.k_max, d.m_max);
    AUV_STAMP_DECL
#ifdef AUV_STAMPS
    const unsigned long long t_real0 = wall_clock64();
#endif
    const int n_act = k2_front<WT>(d, e, lane, L, 1);
    if (d.cfg.use_lidar) {
      AUV_STAMP()
#ifdef AUV_STAMPS
      unsigned long long sub[4] = {0, 0, 0, 0};
      k2_stage_and_pairs(d, L, lane, n_act, d.state[2 * (size_t)d.n + e], sub);
      if (lane == 0 && !PAIRED) d.stamps[(size_t)e * 16 + 7] = sub[0], d.stamps[(size_t)e * 16 + 14] = sub[1], d.stamps[(size_t)e * 16 + 15] = sub[2], d.stamps[(size_t)e * 16 + 6] = sub[3];
#else
      k2_stage_and_pairs(d, L, lane, n_act, d.state[2 * (size_t)d.n + e]);
#endif
      AUV_STAMP()
      double term = 0.0;
      const int collision = k2_back<WT>(d, e, lane, L, n_act, obs_out, &term);
      AUV_STAMP()
      AUV_STAMP_FLUSH(e, 0)   // 0: front (A, C, B, S)  1: pairs (D)  2: back (E)
#ifdef AUV_STAMPS
      if (lane == 0) d.stamps[(size_t)e * 16 + 3] = t_real0, d.stamps[(size_t)e * 16 + 4] = wall_clock64();
      if (lane == 0) d.stamps[(size_t)e * 16 + 5] = (unsigned long long)L.sbase[n_act];
#endif
      if constexpr (PAIRED) pair_publish_lidar(d, e, lane, collision, term);
#ifdef AUV_STAMPS
      if (PAIRED && lane == 0) d.stamps[(size_t)e * 16 + 14] = wall_clock64();   // end of the LiDAR wave incl. its finish
#endif
    }
  }
}

// ---- ONE launch per step: three roles -------------------------------------------------------------
// k_step_roles   workgroups [0, nk) integrate the dynamics (Vessel.step, one wave = eight environments, eight lanes
//                each: k1_group), workgroups [nk, nk + nb) sweep the LiDAR of one environment each, the rest navigate
//                one environment each and run its reward phase (the paired finish above).  The sweep and the
//                navigation need the state the dynamics role produces in the same launch: it hands each environment
//                a 64-byte packet (x, y, psi, u, v, r, the vessel's step counter, a "ready" mark), stored
//                write-through and completed before the mark is stored; the other two roles poll for the mark (the
//                dynamics workgroups have the smallest indices, are dispatched first and wait for nobody; the poll
//                is bounded like the paired finish's).  The navigation wave takes the mark away again when it has
//                finished the environment's step (its sweep wave has read the packet long before: the navigation
//                wave has consumed the word the sweep stores last), so the next launch finds every mark down.  What
//                is saved is the dynamics kernel's launch ramp and the kernel boundary behind it.  No launch
//                argument changes from step to step, so the shape can be captured in a hipGraph; there the dynamics
//                waves count themselves off and the last one advances the action ring (every one of them has read
//                the position by then).
// (Tried and dropped: the sweep waves also running the navigation's search over the chunk circles for the old pose
// while they wait, handing the survivor list to the navigation wave in one word -- the navigation wave gets 2.5 us
// shorter, but the extra traffic and issue slots stretch the dynamics role's chain from 6 to 7.5 us and the whole
// launch waits for that: 98.4 M against 101.9 M env-steps/s.)
// Dynamics wave b takes the environments 8 (8 (b / 8) + g) + b % 8, g = 0..7: the ones whose other two waves run on
// its own XCD (all three counts are multiples of 8), so the packets stay in one L2.
__device__ __forceinline__ double pair_lane_value(const unsigned long long v, const int src) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

#define ROLES_READY 1ull
// the state the dynamics role left for environment e in this launch; false: gave up polling
__device__ __forceinline__ bool roles_wait_state(const AuvDev& d, const int e, const int lane, EnvPre& pre) {
  const unsigned long long seq = ROLES_READY;
  const unsigned long long* pk = d.k1_pkt + 8 * (size_t)e;
  // (polling harder does not pay: with two requests in flight per wave the packet is noticed sooner, but the
  // traffic of 3500 polling waves slows the dynamics role down by more -- 102.3 M against 104.7 M env-steps/s;
  // keeping the early waves quiet until the dynamics are about due moved every workload by +-2 % in no pattern)
  unsigned long long v = 0;
  for (int polls = 0;; polls++) {
    v = __hip_atomic_load(pk + (lane & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one 64-byte request per wave
    const unsigned long long got = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 7) << 32) |
                                   (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 7);
    if (got == seq) break;
    if (polls == PAIR_POLL_LIMIT) {
      if (lane == 0) __hip_atomic_store(d.pair_error, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return false;
    }
    __builtin_amdgcn_s_sleep(8);
  }
#pragma unroll
  for (int i = 0; i < 6; i++) pre.s[i] = pair_lane_value(v, i);
  pre.cnt.y = (int)__builtin_amdgcn_readlane((int)(unsigned)v, 6);   // the vessel's step counter of this launch
  return true;
}

template <typename AT>
__global__ void __launch_bounds__(AUV_WAVE, AUV_K23_MIN_WAVES) k_step_roles(AuvDev dk, const AT* __restrict__ actions,
                                                                           float* __restrict__ obs_out,
                                                                           float* __restrict__ reward_out,
                                                                           uint8_t* __restrict__ done_out) {
  const AuvDev& d = dk;
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const int S = d.cfg.n_sensors;
  const int ne = d.ne;                                              // environments of this launch: [e0, e0 + ne)
  const int nk = 8 * ((ne + 63) / 64);                              // dynamics workgroups
  const int nb = 8 * ((ne + 7) / 8);                                // LiDAR workgroups
  const int b = (int)blockIdx.x;
  if (b < nk) {
    // ---- Vessel.step of eight environments ----
    // (everybody else in the launch waits for these 512 waves: they go first wherever they share a SIMD)
    __builtin_amdgcn_s_setprio(3);
    const int g = lane / K1_GROUP, c = lane % K1_GROUP;
    const int er = 8 * (8 * (b / 8) + g) + (b % 8);
    const bool live = er < ne;
    const int eg = d.e0 + (live ? er : ne - 1);                     // idle groups compute along, store nothing
    const size_t n = (size_t)d.n;
    const int y = d.counters[eg].y + 1;                             // Vessel._step_counter (vessel.py:247); requested up front
    const double t = k1_group<AT>(d, actions, eg, lane);
    unsigned long long* pk = d.k1_pkt + 8 * (size_t)eg;
    if (live && c < 6) {
      auv_st<true>(&d.state[(size_t)c * n + eg], t);
      __hip_atomic_store(pk + c, (unsigned long long)__double_as_longlong(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (live && c == 6) {
      __hip_atomic_store(&d.counters[eg].y, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(pk + 6, (unsigned long long)(unsigned)y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    auv_stores_done();
    if (live && c == 7) __hip_atomic_store(pk + 7, ROLES_READY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // a captured graph's action ring: every dynamics wave has read the position before it counts itself off, so
    // the last one to do so may move it on
    if (d.ring_slots > 1 && d.ring_slot_host == -1 && lane == 0) {
      const int old = __hip_atomic_fetch_add(d.k1_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == nk - 1) {
        *d.ring_pos = (*d.ring_pos + 1) % d.ring_slots;
        __hip_atomic_store(d.k1_done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#ifdef AUV_STAMPS
    if (live && c == 0) d.stamps[(size_t)eg * 16 + 0] = wall_clock64();   // the state of this environment is out
#endif
    return;   // (carrying on as the LiDAR wave of an environment instead of handing the slot to a fresh workgroup was
              // tried: the merged code path costs more scalar-register spills than the later start of 512 sweeps: -2.5 %;
              // with the dynamics in a function of its own (not inlined, arguments by value) they are 1 us slower: -4.5 %.
              // So was the scalar form of the integrator, lanes <-> environments, on 64 waves instead of 512: its
              // chain is 3 us longer and everybody waits for it, 97.2 M against 104.7 M env-steps/s.)
  }
  EnvPre pre;
  EnvDesc ed;
  if (b < nk + nb) {
    // ---- _update + Vessel.perceive of one environment ----
    if (b - nk >= ne) return;
    const int e = auv_uniform(d.e0 + b - nk);
    // (The SIMD's arbiter favours its oldest wave: at equal work the sweeps dispatched last take half as long again as
    // the first, 17 against 12 us, and end the launch -- tools/phase_stamps3.py.  A priority graded by workgroup index
    // evens that out and LOSES 1-2 % either way round: the early finishers make room for navigation waves, whose
    // memory latency then hides behind the late sweeps' arithmetic.)
    // while the dynamics role integrates: everything of the sweep that does not need the vessel's new state --
    // descriptor and counters (written by earlier launches), the movers' kinematics, the obstacle records
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 3] = wall_clock64();
#endif
    ed = d.env_desc[e];
    pre.cnt = d.counters[e];                               // t_step, episodes; the step counter comes with the state
    pre.ed = &ed;
    const Slice L = carve(smem, S, d.k_max, d.m_max);
    k2_movers<true>(d, e, lane, L, ed, 1);
    const K2Pre kp = k2_prefetch(d, e, lane, ed);
    k2_stage_beams(d, lane, L);
    // (also tried here: warming the caches with the nearby obstacles' boundary segments -- it has to wait for the
    // obstacle records, and its traffic delays the dynamics role: 102.1 M against 104.5 M env-steps/s without)
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 1] = wall_clock64();
#endif
    if (!roles_wait_state(d, e, lane, pre)) return;
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 2] = wall_clock64();
#endif
    const int n_act = k2_front<true>(d, e, lane, L, 1, &pre, nullptr, 1, &kp, true);
    k2_stage_and_pairs(d, L, lane, n_act, pre.s[2]);
    double term = 0.0;
    const int collision = k2_back<true>(d, e, lane, L, n_act, obs_out, &term);
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 4] = wall_clock64();
#endif
    pair_publish_lidar(d, e, lane, collision, term);
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 14] = wall_clock64();
#endif
  } else {
    // ---- Vessel.navigate of one environment, then its reward / done / auto-reset ----
    const int el = b - nk - nb - d.pair_skew;
    if (el < 0 || el >= ne) return;
    const int e = auv_uniform(d.e0 + el);
    ed = d.env_desc[e];
    pre.cnt = d.counters[e];
    pre.ed = &ed;
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 12] = wall_clock64();
#endif
    if (!roles_wait_state(d, e, lane, pre)) return;
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 5] = wall_clock64();
#endif
    PairPre pp = pair_prefetch(d, e);
    pp.cnt = pre.cnt;
    NavOut no;
    no.rew_path = no.reached = no.goal = no.progress = no.u = no.v = no.r = 0.0;
    k3_nav_env(d, e, lane, smem, obs_out, &pre, nullptr, &no);
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 13] = wall_clock64();
#endif
    pair_finish_nav(d, e, lane, pp, no, obs_out, reward_out, done_out);
    // the environment's step is complete (and its sweep wave gone): the packet's mark comes down for the next launch
    if (lane == 0) __hip_atomic_store(d.k1_pkt + 8 * (size_t)e + 7, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 15] = wall_clock64();
#endif
  }
}

// ---- the default step: TWO launches --------------------------------------------------------------
// k1n_dyn_nav      Vessel.step for eight environments by ONE wave (eight lanes per environment), then
//                  Vessel.navigate of those eight by the workgroup's eight waves.  While the
//                  dynamics are being integrated every wave runs its navigation's nearest-point
//                  search against the pose BEFORE the step and fetches what it finds (nav_speculate);
//                  behind the barrier only arithmetic on registers is left (nav_finish).
// k2r_lidar_reward _update + Vessel.perceive of one environment per one-wave workgroup and, in the
//                  same wave, reward / done / bookkeeping / auto-reset: the navigation's results
//                  are a kernel boundary old by then, the sweep's are in registers.
// Same device functions, hence the same bits, as the other launch shapes.
#ifndef K1N_ENVS
#define K1N_ENVS 8
#endif
#define K1N_THREADS ((K1N_ENVS + 1) * AUV_WAVE)   // eight navigation waves + the dynamics wave
template <typename AT>
__global__ void __launch_bounds__(K1N_THREADS, 5) k1n_dyn_nav(AuvDev d, const AT* __restrict__ actions,
                                                            float* __restrict__ obs_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  double* sh = (double*)smem;                                       // [K1N_ENVS][8]: the new state
  double* wins = sh + K1N_ENVS * 8;                                 // [K1N_ENVS][3][20]: parked spline windows
  int* lists = (int*)(wins + K1N_ENVS * 3 * 20);                    // [K1N_ENVS][nch_max] surviving chunks
  const int wave = threadIdx.x / AUV_WAVE, lane = threadIdx.x % AUV_WAVE;
  const size_t n = (size_t)d.n;
  const int e0 = (int)blockIdx.x * K1N_ENVS;
  if (wave == K1N_ENVS) {
    // ---- the dynamics wave: Vessel.step of the workgroup's eight environments, eight lanes each.  (Its own
    // branch, so that the navigation's prefetched registers are not live across the integrator.)
    const int g = lane / K1_GROUP, c = lane % K1_GROUP;
    const bool live = g < K1N_ENVS && e0 + g < d.n;
    const int eg = live ? e0 + g : d.n - 1;                         // idle groups compute along, store nothing
    const double t = k1_group<AT>(d, actions, eg, lane);
    if (live && c < 6) d.state[(size_t)c * n + eg] = t, sh[g * 8 + c] = t;
    if (live && c == 0) d.counters[eg].y += 1;                      // Vessel._step_counter (vessel.py:247)
    // cos / sin of the new heading, for the reward's cos(heading error) and for the LiDAR sweep of this
    // step (k2_front): here they cost one sincos per eight environments instead of one per environment
    double sn, co;
    sincos(__shfl(t, lane - c + 2, AUV_WAVE), &sn, &co);
    if (live && c == 6) sh[g * 8 + 6] = co, sh[g * 8 + 7] = sn, d.pose_cs[eg] = make_double2(co, sn);
    __syncthreads();
    return;
  }
  const int e = auv_uniform(e0 + wave);
  const bool valid = e < d.n;
#ifdef AUV_STAMPS
  const unsigned long long t_wg0 = wall_clock64();
#endif
  // the navigation's search against the pose BEFORE the step, with every load it needs (chunk circles,
  // the surviving chunks' segments, spline windows): all of it overlaps the dynamics.  (A wave that
  // reads the state after the dynamics wave has already stored the new one merely gets a better guess.)
  NavSpec sp;
  if (valid) sp = nav_speculate(d, e, lane, lists + (size_t)wave * d.nch_max, d.state[0 * n + e], d.state[1 * n + e],
                                2.0 * NAV_DELTA, wins + wave * 3 * 20);
  __syncthreads();
  if (!valid) return;
  EnvPre pre;
#pragma unroll
  for (int i = 0; i < 6; i++) pre.s[i] = sh[wave * 8 + i];
  pre.cnt = make_int4(0, 0, 0, 0);                                  // (the navigation does not look at the counters)
#ifdef AUV_STAMPS
  const unsigned long long t_nav0 = wall_clock64();
#endif
  const double2 cs = make_double2(sh[wave * 8 + 6], sh[wave * 8 + 7]);
  nav_finish(d, e, lane, lists + (size_t)wave * d.nch_max, obs_out, &pre, sp, true, wins + wave * 3 * 20, &cs);
#ifdef AUV_STAMPS
  if (lane == 0) d.stamps[(size_t)e * 16 + 12] = t_nav0, d.stamps[(size_t)e * 16 + 13] = wall_clock64(), d.stamps[(size_t)e * 16 + 14] = t_wg0;
#endif
}

__global__ void __launch_bounds__(AUV_BLOCK, AUV_K23_MIN_WAVES) k2r_lidar_reward(AuvDev d, float* __restrict__ obs_out,
                                                                 float* __restrict__ reward_out,
                                                                 uint8_t* __restrict__ done_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x % AUV_WAVE;
  const int S = d.cfg.n_sensors;
  const int e = auv_uniform((int)blockIdx.x);
  if (e >= d.n) return;
  const Slice L = carve(smem, S, d.k_max, d.m_max);
  AUV_STAMP_DECL
#ifdef AUV_STAMPS
  const unsigned long long t_real0 = wall_clock64();
#endif
  int collision = -1;
  double term = 0.0;
  const int n_act = k2_front(d, e, lane, L, 1, nullptr, d.pose_cs);
  if (d.cfg.use_lidar) {
    AUV_STAMP()
    k2_stage_and_pairs(d, L, lane, n_act, d.state[2 * (size_t)d.n + e]);
    AUV_STAMP()
    collision = k2_back(d, e, lane, L, n_act, obs_out, &term);
    AUV_STAMP()
    AUV_STAMP_FLUSH(e, 0)   // 0: front (A, C, B, S)  1: pairs (D)  2: back (E)
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 3] = t_real0, d.stamps[(size_t)e * 16 + 4] = wall_clock64();
    if (lane == 0) d.stamps[(size_t)e * 16 + 5] = (unsigned long long)L.sbase[n_act];
#endif
  }
  k3_reward_env(d, e, lane, true, obs_out, reward_out, done_out, nullptr, collision, false, !d.cfg.use_lidar,
                d.cfg.use_lidar ? &term : nullptr);
#ifdef AUV_STAMPS
  if (lane == 0) d.stamps[(size_t)e * 16 + 15] = wall_clock64();   // end of the reward phase
#endif
}

// ---- inside a captured graph of several steps: reward / done / auto-reset of step t and Vessel.step of step t + 1
// in ONE launch (the actions of an open-loop stretch are in the ring already, so nothing sits between the two).
// Lanes <-> environments for both: the scalar form of the dynamics (k1_env: the same operations in the same order
// as the eight-lane kernel, bit-identical) has the same dependent chain, and 64 of them share a wave.
template <typename AT>
__global__ void __launch_bounds__(AUV_WAVE) k31_reward_dyn(AuvDev d, const AT* __restrict__ actions, float* __restrict__ obs_out,
                                                           float* __restrict__ reward_out, uint8_t* __restrict__ done_out) {
  const int lane = threadIdx.x;
  const int e = blockIdx.x * AUV_WAVE + lane;
  int do_reset = 0, w = 0;
  int4 cnt = make_int4(0, 0, 0, 0);
  if (e < d.n) {
    cnt = d.counters[e];
    w = d.world_idx[e];
    const int collision = d.collision[e];
    d.info64[8 * (size_t)e] = collision;
    do_reset = reward_block(d, e, collision, cnt, false, 0.0, nullptr, reward_out, done_out, false);
  }
  unsigned long long m = __ballot(do_reset);
  const bool any_reset = m != 0;
  while (m) {
    const int src = __ffsll((long long)m) - 1;
    m &= m - 1;
    const int er = auv_uniform(__shfl(e, src, AUV_WAVE));
    const int wr = __shfl(w, src, AUV_WAVE), ep = __shfl(cnt.z, src, AUV_WAVE);
    restore_env(d, er, (int)(((long long)wr + d.n) % d.n_worlds), lane, ep, obs_out);
  }
  // a restored environment's state was written by lane 0, its dynamics below read it from another lane
  if (any_reset) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  if (e < d.n) k1_env<AT>(d, e, actions, true);
}

}  // namespace

void auv_launch_k31(const AuvDev& d, const void* actions, int dtype, float* obs, float* reward, uint8_t* done, hipStream_t st) {
  const dim3 grid((d.n + AUV_WAVE - 1) / AUV_WAVE), block(AUV_WAVE);
  if (dtype == AUV_F64)
    hipLaunchKernelGGL(k31_reward_dyn<double>, grid, block, 0, st, d, (const double*)actions, obs, reward, done);
  else
    hipLaunchKernelGGL(k31_reward_dyn<float>, grid, block, 0, st, d, (const float*)actions, obs, reward, done);
}

// the nav chunk list must fit the segment stage it borrows
bool auv_step_fused_ok(const AuvDev& d) { return NAV_SCRATCH_BYTES(d.nch_max) <= (size_t)K2_SEG_CAP * 32; }

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
bool auv_k23_ok(const AuvDev& d) { return NAV_SCRATCH_BYTES(d.nch_max) <= k2_slice_bytes(d.cfg.n_sensors, d.k_max, d.m_max); }

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
  hipExtLaunchKernelGGL(k23_lidar_nav<false>, dim3(2 * nb), dim3(AUV_WAVE * wpb), (uint32_t)lds, st, ev0, ev1, 0, d, obs,
                        (float*)nullptr, (uint8_t*)nullptr);
}

// ... with the reward phase run by the second of an environment's two waves (the paired step; needs a LiDAR sweep)
bool auv_paired_ok(const AuvDev& d) { return auv_k23_ok(d) && d.cfg.use_lidar; }

void auv_launch_k23_paired(const AuvDev& d, float* obs, float* reward, uint8_t* done, hipStream_t st, hipEvent_t ev0,
                           hipEvent_t ev1) {
  const size_t lds = k2_slice_bytes(d.cfg.n_sensors, d.k_max, d.m_max);
  const int nb = 8 * ((d.n + 7) / 8);
  hipExtLaunchKernelGGL(k23_lidar_nav<true>, dim3(2 * nb + d.pair_skew), dim3(AUV_WAVE), (uint32_t)lds, st, ev0, ev1, 0, d, obs,
                        reward, done);
}

// ---- launches of the two-kernel step ----
static size_t k1n_lds_bytes(const AuvDev& d) {
  return K1N_ENVS * 8 * sizeof(double) + K1N_ENVS * 3 * 20 * sizeof(double) + (size_t)K1N_ENVS * d.nch_max * sizeof(int);
}
bool auv_two_kernel_ok(const AuvDev& d) { return k1n_lds_bytes(d) <= 64 * 1024; }

void auv_launch_k1n(const AuvDev& d, const void* actions, int dtype, float* obs, hipStream_t st, hipEvent_t ev0,
                    hipEvent_t ev1) {
  const dim3 grid((d.n + K1N_ENVS - 1) / K1N_ENVS), block(K1N_THREADS);
  const uint32_t lds = (uint32_t)k1n_lds_bytes(d);
  if (dtype == AUV_F64)
    hipExtLaunchKernelGGL(k1n_dyn_nav<double>, grid, block, lds, st, ev0, ev1, 0, d, (const double*)actions, obs);
  else
    hipExtLaunchKernelGGL(k1n_dyn_nav<float>, grid, block, lds, st, ev0, ev1, 0, d, (const float*)actions, obs);
}

void auv_launch_k2r(const AuvDev& d, float* obs, float* reward, uint8_t* done, hipStream_t st, hipEvent_t ev0,
                    hipEvent_t ev1) {
  const size_t lds = k2_slice_bytes(d.cfg.n_sensors, d.k_max, d.m_max);
  hipExtLaunchKernelGGL(k2r_lidar_reward, dim3(d.n), dim3(AUV_WAVE), (uint32_t)lds, st, ev0, ev1, 0, d, obs, reward, done);
}

// ---- the one-launch step ----
bool auv_roles_ok(const AuvDev& d) { return auv_paired_ok(d); }

void auv_launch_step_roles(const AuvDev& d, const void* actions, int dtype, float* obs, float* reward, uint8_t* done,
                           hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  const uint32_t lds = (uint32_t)k2_slice_bytes(d.cfg.n_sensors, d.k_max, d.m_max);
  const int nk = 8 * ((d.ne + 63) / 64), nb = 8 * ((d.ne + 7) / 8);
  const dim3 grid(nk + 2 * nb + d.pair_skew), block(AUV_WAVE);
  if (dtype == AUV_F64)
    hipExtLaunchKernelGGL(k_step_roles<double>, grid, block, lds, st, ev0, ev1, 0, d, (const double*)actions, obs, reward, done);
  else
    hipExtLaunchKernelGGL(k_step_roles<float>, grid, block, lds, st, ev0, ev1, 0, d, (const float*)actions, obs, reward, done);
}

hipError_t auv_step_fused_prepare(const AuvDev& d) {
  {
    const size_t b1 = k2_slice_bytes(d.cfg.n_sensors, d.k_max, d.m_max);
    if (b1 > 64 * 1024) {
      hipError_t e1 = hipFuncSetAttribute((const void*)k2r_lidar_reward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b1);
      if (e1 != hipSuccess) return e1;
    }
  }
  const size_t b = k2_slice_bytes(d.cfg.n_sensors, d.k_max, d.m_max) * AUV_ENVS_PER_BLOCK;
  if (b <= 64 * 1024) return hipSuccess;
  hipError_t e = hipFuncSetAttribute((const void*)k23_lidar_nav<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)k23_lidar_nav<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)k_step_roles<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)k_step_roles<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)k_step<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)k_step<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
}
