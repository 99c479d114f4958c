// k_step_fused.hip -- the launch shapes of step() that put more than one phase into a launch, built from the very
// same device functions as the individually launchable kernels (k1_dynamics / k2_lidar / k3_nav / k3_reward), so the
// per-kernel parity tests cover their arithmetic and tests/test_gpu_parity.py pins the compositions bit for bit:
//   k_step_roles            ONE launch per step: dynamics, LiDAR sweep, navigation + reward as three roles
//   k23_lidar_nav           K2 and K3-nav side by side in one launch (the fence-free three-launch shape)
//   k31_reward_dyn          inside a captured graph of several steps: reward phase of step t + dynamics of step t + 1
// Every launch covers the slice [e0, e0 + ne) of the handle's environments (AuvDev::e0 / ne; the whole batch by
// default): sub-batches of one handle stepped on different streams overlap each other's head and tail.
// (Measured and removed in round 3, numbers in DESIGN.md: the whole step as one wave per environment, 66.9 M
// env-steps/s; [K1 + navigation] -> [LiDAR + reward], 73.6 M; navigation forked onto a second stream, 64.5 M; K1 ->
// [LiDAR | navigation + reward] ("paired"), 98 M; the navigation's tail + reward with lanes <-> environments as a
// second launch or a fourth role: 20 % fewer VALU instructions but a longer step per chain, 106-112 M against 133 M
// with four sub-batch chains -- git history and profiles/r03/shapes_sweep*.log.)
#include <cstdlib>
#include <hip/hip_ext.h>

#define AUV_DEVICE_FUNCS_ONLY
#include "k1_dynamics.hip"
#include "k2_lidar.hip"
#include "k3_nav_reward.hip"

#ifndef AUV_K23_MIN_WAVES
#define AUV_K23_MIN_WAVES 4   // waves per SIMD the LiDAR launches are compiled for (128 VGPRs)
#endif

// Test hooks (tests/test_gpu_parity.py) exist only in the library built with -DAUV_TEST_HOOKS (make hooks ->
// libauv_hip_hooks.so): idle workgroups between the roles (puts an environment's waves on different XCDs) and a
// sweep that never publishes its word (the poll must run out and fail loudly).  The shipped library has neither.
#ifdef AUV_TEST_HOOKS
#define AUV_HOOK_SKEW(d) ((d).pair_skew)
#define AUV_HOOK_FAULT(d) ((d).pair_fault)
#else
#define AUV_HOOK_SKEW(d) 0
#define AUV_HOOK_FAULT(d) 0
#endif

namespace {

// ---- the in-launch finish: reward / done / auto-reset by the navigation wave of the one-launch step ----
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
  if (AUV_HOOK_FAULT(d) && e == d.e0) return;              // (test hook: the poll's time-out)
  if (lane == 0)
    __hip_atomic_store(d.pair_word + e, collision ? PAIR_COLLISION : (unsigned long long)__double_as_longlong(term),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// what the navigation wave requests before it starts working (one trip, hidden behind the search)
struct PairPre {
  unsigned long long word;
  double cum, cte_sum;
  int4 cnt;
  int w;
};
__device__ __forceinline__ PairPre pair_prefetch(const AuvDev& d, const int e) {
  PairPre p;
  p.word = __hip_atomic_load(d.pair_word + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // (written by earlier launches: the dynamics kernel and the previous step's reward phase)
  p.cum = d.info64[8 * (size_t)e + 4];
  p.cte_sum = d.info64[8 * (size_t)e + 7];
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
  const int limit = AUV_HOOK_FAULT(dk) ? (1 << 12) : PAIR_POLL_LIMIT;
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
    in.cte100 = no.cte100, in.cte_sum = p.cte_sum;
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

// K2 and K3-nav of the launch's environments in one launch of one-wave workgroups: workgroups [0, ne) sweep the LiDAR
// of their environments, workgroups [ne, 2 ne) navigate theirs.  The two are independent given the new vessel state,
// so they run side by side (the LiDAR workgroups are dispatched first and fill the chip; navigation workgroups move
// in as those retire; a wave slot is handed on the moment an environment's sweep ends) -- the concurrency of two
// streams without the ~8 us a cross-stream event wait costs on each side.  Nothing is handed over inside the launch.
__global__ void __launch_bounds__(AUV_BLOCK, AUV_K23_MIN_WAVES) k23_lidar_nav(AuvDev dk, float* __restrict__ obs_out) {
  // (every table through the device-side copy AuvDev::self instead of the kernel arguments -- no scalar register
  // spilled any more -- was tried for the whole kernel: the loads then sit on the dependent chains, 36.3 -> 42.2 us)
  const AuvDev& d = dk;
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const int S = d.cfg.n_sensors;
  const int ne = d.ne;
  if ((int)blockIdx.x >= ne) {
    const int el = (int)blockIdx.x - ne;
    if (el >= ne) return;
    const int e = auv_uniform(d.e0 + el);
#ifdef AUV_STAMPS
    const unsigned long long t_nav0 = wall_clock64();
#endif
    k3_nav_env(d, e, lane, smem, obs_out);
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 12] = t_nav0, d.stamps[(size_t)e * 16 + 13] = wall_clock64();
#endif
  } else {
    const int e = auv_uniform(d.e0 + (int)blockIdx.x);
    // next action slot of a captured graph's ring: the dynamics kernel of this step has read the position, the
    // one of the next step has not been launched yet (a captured step covers the whole batch: e0 = 0)
    if (e == 0 && lane == 0 && dk.ring_slots > 1 && dk.ring_slot_host == -1) *dk.ring_pos = (*dk.ring_pos + 1) % dk.ring_slots;
    const Slice L = carve(smem, S, d.k_max, d.m_max);
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
      double term = 0.0;
      k2_back(d, e, lane, L, n_act, obs_out, &term);
      AUV_STAMP()
      AUV_STAMP_FLUSH(e, 0)   // 0: front (A, C, B, S)  1: pairs (D)  2: back (E)
#ifdef AUV_STAMPS
      if (lane == 0) d.stamps[(size_t)e * 16 + 3] = t_real0, d.stamps[(size_t)e * 16 + 4] = wall_clock64();
      if (lane == 0) d.stamps[(size_t)e * 16 + 5] = (unsigned long long)L.sbase[n_act];
#endif
    }
  }
}

// ---- ONE launch per step: three roles -------------------------------------------------------------
// k_step_roles   workgroups [0, nk) integrate the dynamics (Vessel.step, one wave = eight environments, eight lanes
//                each: k1_group), workgroups [nk, nk + nb) sweep the LiDAR of one environment each, the rest navigate
//                one environment each and run its reward phase (the in-launch finish above).  The sweep and the
//                navigation need the state the dynamics role produces in the same launch: it hands each environment
//                a 64-byte packet (x, y, psi, u, v, r, the vessel's step counter, a "ready" mark), stored
//                write-through and completed before the mark is stored; the other two roles poll for the mark (the
//                dynamics workgroups have the smallest indices, are dispatched first and wait for nobody; the poll
//                is bounded like the in-launch finish's).  The navigation wave takes the mark away again when it has
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

// The packet's eighth word is its "ready" mark AND a checksum of the other seven: (xor of words 0..6, xor a constant)
// with bit 0 set; "down" is 0.  A reader accepts the payload only if the mark it loaded WITH it matches the payload,
// so a request served in pieces (new mark, old state) can never pass -- the hand-over does not rest on the 64-byte
// request being served atomically, and costs the reader a few scalar instructions, no second trip.
#define ROLES_MAGIC 0x9e3779b97f4a7c15ull
__device__ __forceinline__ unsigned long long roles_mark(const unsigned long long x) { return (x ^ ROLES_MAGIC) | 1ull; }

// xor of `w` over the eight lanes of the caller's group (lanes 8 g .. 8 g + 7), by DPP moves: quad_perm [1,0,3,2],
// quad_perm [2,3,0,1], row_half_mirror -- no LDS, a dozen VALU instructions at the end of the dynamics' chain
__device__ __forceinline__ unsigned long long roles_group_xor(const unsigned long long w) {
  int lo = (int)(unsigned)w, hi = (int)(unsigned)(w >> 32);
  lo ^= __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true), hi ^= __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
  lo ^= __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true), hi ^= __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true);
  lo ^= __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, true), hi ^= __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, true);
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

__device__ __forceinline__ unsigned long long roles_lane_word(const unsigned long long v, const int src) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src) << 32) |
         (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
}

// the state the dynamics role left for environment e in this launch; false: gave up polling
__device__ __forceinline__ bool roles_wait_state(const AuvDev& d, const int e, const int lane, EnvPre& pre) {
  const unsigned long long* pk = d.k1_pkt + 8 * (size_t)e;
  // (polling harder does not pay: with two requests in flight per wave the packet is noticed sooner, but the
  // traffic of 3500 polling waves slows the dynamics role down by more -- 102.3 M against 104.7 M env-steps/s;
  // keeping the early waves quiet until the dynamics are about due moved every workload by +-2 % in no pattern)
  unsigned long long v = 0;
  for (int polls = 0;; polls++) {
    v = __hip_atomic_load(pk + (lane & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one 64-byte request per wave
    const unsigned long long got = roles_lane_word(v, 7);
    if (got != 0ull) {
      unsigned long long x = roles_lane_word(v, 0);
#pragma unroll
      for (int i = 1; i < 7; i++) x ^= roles_lane_word(v, i);
      if (roles_mark(x) == got) break;                       // mark and payload belong together
    }
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

__global__ void __launch_bounds__(AUV_WAVE, AUV_K23_MIN_WAVES) k_step_roles(AuvDev dk, const void* __restrict__ actions,
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
    const double t = k1_group(d, actions, eg, lane);
    unsigned long long* pk = d.k1_pkt + 8 * (size_t)eg;
    const unsigned long long word = c < 6 ? (unsigned long long)__double_as_longlong(t) : (c == 6 ? (unsigned long long)(unsigned)y : 0ull);
    const unsigned long long mark = roles_mark(roles_group_xor(word));
    if (live && c < 6) auv_st<true>(&d.state[(size_t)c * n + eg], t);
    if (live && c < 7) __hip_atomic_store(pk + c, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (live && c == 6) __hip_atomic_store(&d.counters[eg].y, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    auv_stores_done();
    if (live && c == 7) __hip_atomic_store(pk + 7, mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    int n_act = 0;
    if (AUV_RUN_L(d, 1)) n_act = k2_front<true>(d, e, lane, L, 1, &pre, 1, &kp, true);
    if (AUV_RUN_L(d, 3)) k2_stage_and_pairs(d, L, lane, n_act, pre.s[2]);
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
    const int el = b - nk - nb - AUV_HOOK_SKEW(d);
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
    no.rew_path = no.reached = no.goal = no.progress = no.u = no.v = no.r = no.cte100 = 0.0;
    if (AUV_RUN_N(d, 1)) k3_nav_env(d, e, lane, smem, obs_out, &pre, &no);
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

// ---- inside a captured graph of several steps: reward / done / auto-reset of step t and Vessel.step of step t + 1
// in ONE launch (the actions of an open-loop stretch are in the ring already, so nothing sits between the two).
// Lanes <-> environments for both: the scalar form of the dynamics (k1_env: the same operations in the same order
// as the eight-lane kernel, bit-identical) has the same dependent chain, and 64 of them share a wave.
__global__ void __launch_bounds__(AUV_WAVE) k31_reward_dyn(AuvDev d, const void* __restrict__ actions, float* __restrict__ obs_out,
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
  if (e < d.n) k1_env(d, e, actions, true);
}

// ---- load-time probe of what the in-launch hand-overs rely on ----------------------------------------------------
// The one-launch shape lets a wave poll for a word that a workgroup with a SMALLER index of the same
// launch stores.  That terminates if workgroups are dispatched in index order (a poller's producer is resident or
// done by the time the poller gets a slot) -- what gfx950 does, but HIP does not promise it.  k_probe_order has the
// step's structure without its arithmetic: `np` producers (a short wait, then an sc1 word each), behind them
// consumers that poll their producer's word, publish a word of their own and are in turn polled by a second
// generation -- three generations like dynamics / sweep / navigation, every workgroup one wave with the step's LDS
// footprint, and several times more workgroups than the chip has slots.  Any poll that runs out (bounded: ~20 ms)
// counts a failure; the host then keeps the handle on the three-launch shape (auv_capi.hip: probe_dispatch_order).
__global__ void __launch_bounds__(AUV_WAVE) k_probe_order(unsigned int* __restrict__ words, int np, int nc, unsigned int tag,
                                                          unsigned int* __restrict__ failures) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int b = (int)blockIdx.x, lane = threadIdx.x;
  if (lane == 0) smem[0] = 1;   // (touch the allocation so that it is not optimised away)
  unsigned int* mine = words + b;
  const unsigned int* src = nullptr;
  if (b >= np + nc) src = words + np + (b - np - nc);          // third generation: its own second-generation wave
  else if (b >= np) src = words + (b - np) % np;               // second generation: one of the producers
  if (src) {
    int polls = 0;
    while (__hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) {
      if (++polls == (1 << 14)) {
        if (lane == 0) atomicAdd(failures, 1u);
        return;
      }
      __builtin_amdgcn_s_sleep(32);
    }
  } else {
    __builtin_amdgcn_s_sleep(40), __builtin_amdgcn_s_sleep(40);   // ~2 us, like the dynamics' chain
  }
  if (lane == 0 && b < np + nc) __hip_atomic_store(mine, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// a wave that does nothing for `ticks` of the 100 MHz wall clock (auv_streams_overlap: do two streams run side by side?)
__global__ void k_spin(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

}  // namespace

void auv_launch_spin(unsigned long long ticks, hipStream_t st) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(AUV_WAVE), 0, st, ticks); }

hipError_t auv_launch_probe(unsigned int* words, int np, int nc, unsigned int tag, unsigned int* failures, uint32_t lds, hipStream_t st) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k_probe_order, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_probe_order, dim3(np + 2 * nc), dim3(AUV_WAVE), lds, st, words, np, nc, tag, failures);
  return hipGetLastError();
}

void auv_launch_k31(const AuvDev& d0, const void* actions, int dtype, float* obs, float* reward, uint8_t* done, hipStream_t st) {
  AuvDev d = d0;
  d.act_f64 = dtype == AUV_F64;
  const dim3 grid((d.n + AUV_WAVE - 1) / AUV_WAVE), block(AUV_WAVE);
  hipLaunchKernelGGL(k31_reward_dyn, grid, block, 0, st, d, actions, obs, reward, done);
}

// the navigation role keeps its chunk list at the start of the wave's slice
bool auv_k23_ok(const AuvDev& d) { return NAV_SCRATCH_BYTES(d.nch_max) <= k2_slice_bytes(d.cfg.n_sensors, d.k_max, d.m_max); }

// K2 + K3-nav side by side: ONE wave per workgroup, so a wave slot is handed on the moment an environment's sweep
// ends -- the navigation workgroups queued behind the LiDAR ones start (and end) earlier.
void auv_launch_k23(const AuvDev& d, float* obs, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  const size_t lds = k2_slice_bytes(d.cfg.n_sensors, d.k_max, d.m_max);
  hipExtLaunchKernelGGL(k23_lidar_nav, dim3(2 * d.ne), dim3(AUV_WAVE), (uint32_t)lds, st, ev0, ev1, 0, d, obs);
}

// ---- the one-launch step (needs a LiDAR sweep: its word is what the navigation wave finishes the step on) ----
bool auv_roles_ok(const AuvDev& d) { return auv_k23_ok(d) && d.cfg.use_lidar; }

void auv_launch_step_roles(const AuvDev& d0, const void* actions, int dtype, float* obs, float* reward, uint8_t* done,
                           hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  AuvDev d = d0;
  d.act_f64 = dtype == AUV_F64;
  const uint32_t lds = (uint32_t)k2_slice_bytes(d.cfg.n_sensors, d.k_max, d.m_max);
  const int nk = 8 * ((d.ne + 63) / 64), nb = 8 * ((d.ne + 7) / 8);
  const dim3 grid(nk + 2 * nb + AUV_HOOK_SKEW(d)), block(AUV_WAVE);
  hipExtLaunchKernelGGL(k_step_roles, grid, block, lds, st, ev0, ev1, 0, d, actions, obs, reward, done);
}

uint32_t auv_step_lds_bytes(const AuvDev& d) { return (uint32_t)k2_slice_bytes(d.cfg.n_sensors, d.k_max, d.m_max); }

hipError_t auv_step_fused_prepare(const AuvDev& d) {
  const size_t b = k2_slice_bytes(d.cfg.n_sensors, d.k_max, d.m_max);
  if (b <= 64 * 1024) return hipSuccess;
  hipError_t e = hipFuncSetAttribute((const void*)k23_lidar_nav, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)k_step_roles, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
}
