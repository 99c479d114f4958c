// k_step_fused.hip -- the launch shapes of step() that put more than one phase into a launch, built from the very
// same device functions as the individually launchable kernels (k1_dynamics / k2_lidar / k3_nav / k3_reward), so the
// per-kernel parity tests cover their arithmetic and tests/test_gpu_parity.py pins the compositions bit for bit:
//   k_step_roles            ONE launch per step: dynamics, LiDAR sweep, navigation search, tail + reward as four roles
//   k23_lidar_nav           K2 and K3-nav side by side in one launch (the fence-free three-launch shape)
//   k31_reward_dyn          inside a captured graph of several steps: reward phase of step t + dynamics of step t + 1
// Every launch covers the slice [e0, e0 + ne) of the handle's environments (AuvDev::e0 / ne; the whole batch by
// default): sub-batches of one handle stepped on different streams overlap each other's head and tail.
// (Measured and removed in round 3, numbers in DESIGN.md: the whole step as one wave per environment, 66.9 M
// env-steps/s; [K1 + navigation] -> [LiDAR + reward], 73.6 M; navigation forked onto a second stream, 64.5 M; K1 ->
// [LiDAR | navigation + reward] ("paired"), 98 M; the navigation's tail + reward with lanes <-> environments as a
// second launch, 106-112 M against 133 M with four sub-batch chains; the one-launch step with the navigation WAVE of an
// environment also running its tail and its reward phase -- three roles -- was the shape until late in round 3: 19 %
// more VALU issue cycles than the four roles below, 130 M against 136 M in the same sweep -- git history and
// profiles/r03/shapes_sweep*.log.)
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
// hand-over that never happens for the first environment of the launch (fault 1: its sweep withholds the word; 2: the
// dynamics role withholds its state packet; 3: its search withholds the record) -- the polls must run out and fail
// loudly.  The shipped library has neither.
#ifdef AUV_TEST_HOOKS
#define AUV_HOOK_SKEW(d) ((d).pair_skew)
#define AUV_HOOK_FAULT(d) ((d).pair_fault)
#else
#define AUV_HOOK_SKEW(d) 0
#define AUV_HOOK_FAULT(d) 0
#endif

namespace {

// ---- the in-launch hand-over of the sweep's result -------------------------------------------------------------
// An environment's LiDAR wave and the finish wave that runs its reward phase (rewarder.py:78-140, :167-241;
// environment.py:333-347, :375-384) are workgroups of the same launch -- no third launch.  What the reward phase needs
// of the sweep is one 64-bit word per environment (the LiDAR term of the reward, or a marker for "collision"): the
// LiDAR wave stores it last, the finish wave polls for it, consumes it and puts the "empty" marker back.
//   Coherence without device-scope fences (their L2 write-back per wave is what made round 1's attempt 5x slower):
//   * the word goes by relaxed agent-scope atomic store / load (sc1: written through and read past the XCD's
//     L2, which is not coherent with the other seven);
//   * every row the LiDAR wave writes is stored write-through too (WT = true in k2_front / k2_back), and the wave
//     waits for the completion of all its stores before it stores the word.  A finish wave that has seen the
//     word therefore knows that nothing of that environment is in flight or dirty in another L2: its plain stores
//     (reward phase; restore_env, which overwrites the sweep's rows when the episode ended) are the last word
//     whichever XCDs the two ran on.  Rows of other environments share cache lines but not bytes.
//   * the launch places all waves of an environment on the same XCD (workgroups go round-robin over the eight XCDs;
//     every role starts at a multiple of 8 and the dynamics / finish waves take the environments of their own
//     XCD), which keeps the words in one L2; correctness does not depend on it (tests run with the roles skewed
//     onto different XCDs).
//     (Plain stores for the sweep's rows -- complete when the XCD's own L2 has them, which is enough when the
//     finish wave runs on the same XCD -- were measured: +0.6 % with four chains, +1.3 % with one; not worth making
//     correctness depend on the workgroup -> XCD mapping.)
//   * whoever polls, polls for a word of a workgroup with a SMALLER index in the same launch: that one was
//     dispatched earlier and finishes without needing anything from a later one.  The polls are bounded all the
//     same: when one runs out the wave reports through `pair_error` (the next call recovers and reports) instead
//     of hanging the device.
#define PAIR_EMPTY AUV_PAIR_EMPTY
#define PAIR_COLLISION AUV_PAIR_COLLISION
#define PAIR_POLL_LIMIT (1 << 22)
#ifndef ROLES_POLL_SLEEP
#define ROLES_POLL_SLEEP 8          // 64-cycle quanta between two looks at a hand-over word (2, 4 and 16 measured in round 4: no difference, profiles/r04/ab_poll_sleep.jsonl)
#endif
#define ROLES_ABORT_COUNTER 0xffffffffu   // step-counter word of an ABORT packet (a real Vessel._step_counter never gets there)

// A poll has run out.  The wave records which environments it leaves unfinished (`broken`: the caller does, per
// environment), raises the device-wide abort flag -- launches queued behind this one then do nothing instead of consuming
// half-finished hand-overs -- and tells the host (mapped memory: code, and the slice of the launch that reports).
__device__ __forceinline__ void roles_give_up(const AuvDev& d, const int e0, const int ne, const int code, const int lane) {
  if (lane == 0) {
    __hip_atomic_store(d.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d.abort_flag + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // "a report is out": a rendezvous kernel that runs out later keeps its code to itself (k_rdv_wait)
    __hip_atomic_store(d.pair_error + 1, e0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(d.pair_error + 2, ne, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(d.pair_error, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__device__ __forceinline__ void pair_publish_lidar(const AuvDev& d, const int e, const int lane, const int collision,
                                                   const double term) {
  auv_stores_done();                                       // of every lane of this wave (one counter per wave)
  if (AUV_HOOK_FAULT(d) == 1 && e == d.e0) return;         // (test hook: the poll's time-out)
  if (lane == 0)
    __hip_atomic_store(d.pair_word + e, collision ? PAIR_COLLISION : (unsigned long long)__double_as_longlong(term),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    // one of the next step has not been launched yet.  The first environment of the LAUNCH advances it: a captured chain
    // covers a slice [e0, e0 + ne) and keeps a ring position of its own (`e == 0` left the chains with e0 > 0
    // on slot 0 for good when a captured chain stepped in this shape)
    if (e == d.e0 && lane == 0 && dk.ring_slots > 1 && dk.ring_slot_host == -1) *dk.ring_pos = (*dk.ring_pos + 1) % dk.ring_slots;
    const Slice L = carve(smem, d);
    AUV_STAMP_DECL
#ifdef AUV_STAMPS
    const unsigned long long t_real0 = wall_clock64();
#endif
    int2 lim0 = make_int2(INT32_MIN, INT32_MIN);           // this lane's cull-limit row: stored by k2_back (see there)
    const int n_act = k2_front(d, e, lane, L, 1, nullptr, 0, nullptr, false, d.cfg.use_lidar ? &lim0 : nullptr);
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
      k2_back(d, e, lane, L, n_act, obs_out, &term, &lim0);
      AUV_STAMP()
      AUV_STAMP_FLUSH(e, 0)   // 0: front (A, C, B, S)  1: pairs (D)  2: back (E)
#ifdef AUV_STAMPS
      if (lane == 0) d.stamps[(size_t)e * 16 + 3] = t_real0, d.stamps[(size_t)e * 16 + 4] = wall_clock64();
      if (lane == 0) d.stamps[(size_t)e * 16 + 5] = (unsigned long long)L.sbase[n_act];
#endif
    }
  }
}

// ---- ONE launch per step: four roles --------------------------------------------------------------
// k_step_roles   workgroups [0, nk) integrate the dynamics (Vessel.step, one wave = eight environments, eight lanes
//                each: k1_group); workgroups [nk, nk + nb) sweep the LiDAR of one environment each; workgroups
//                [nk + nb, nk + 2 nb) search the nearest point of the path for one environment each (the wide part of
//                Vessel.navigate); the last nk workgroups -- the finish role, eight environments per wave like the
//                dynamics -- evaluate the navigation's scalar tail and run the reward phase (see there).
//                The sweep, the search and the finish need the state the dynamics role produces in the same launch:
//                it hands each environment a 64-byte packet (x, y, psi, u, v, r, the vessel's step counter, a "ready"
//                mark that is also a checksum of the rest), written through (sc1) by ONE store instruction -- and that
//                is all the dynamics role stores: the STATE rows and the vessel's step counter are written from the
//                packet by the finish wave.  The other roles poll for a mark that matches the payload (the dynamics
//                workgroups have the smallest indices, are dispatched first and wait for nobody; every poll is
//                bounded).  The search leaves its result in a record of the same kind (NAV_HAND), the sweep its word.  The finish wave takes all three marks away again when it
//                has finished the environment's step (the environment's other waves are gone by then: it has seen
//                what each of them stores last), so the next launch finds every mark down.  What is saved against
//                separate launches is the launch ramps and the kernel boundaries between them.  No launch argument
//                changes from step to step, so the shape can be captured in a hipGraph; there the dynamics waves
//                count themselves off and the last one advances the action ring (every one of them has read the
//                position by then).
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

// the state the dynamics role left for environment e in this launch.  0: here it is; 1: gave up polling (reported, the
// environment marked broken); 2: an ABORT packet -- a launch behind a time-out: the wave ends without touching anything
// `tagmix`: 0 for a launch of one step; a launch of several steps mixes the step's number into every mark (roles_tagmix), so
// that a wave of step t + 1 that starts early cannot take step t's packet -- still up -- for its own
__device__ __forceinline__ unsigned long long roles_tagmix(const unsigned long long tag) { return tag * 0xd6e8feb86659fd93ull; }
// `have_first`: the caller has requested the packet already (`v_first`: its lane's word), ahead of other work -- in a launch of
// several steps the packet is usually there when the wave starts, and a wave that asks for it only HERE holds its slot for one more
// trip to memory (~1 us of a 12 us sweep: tools/multi_stamps.py); a packet of another step fails the mark and is polled for as ever
__device__ __forceinline__ int roles_wait_state(const AuvDev& d, const int e, const int lane, EnvPre& pre, const unsigned long long tagmix = 0ull,
                                                const bool have_first = false, const unsigned long long v_first = 0ull) {
  const unsigned long long* pk = d.k1_pkt + 8 * (size_t)e;
  // (polling harder does not pay: with two requests in flight per wave the packet is noticed sooner, but the
  // traffic of 3500 polling waves slows the dynamics role down by more -- 102.3 M against 104.7 M env-steps/s;
  // keeping the early waves quiet until the dynamics are about due moved every workload by +-2 % in no pattern)
  unsigned long long v = 0;
  for (int polls = 0;; polls++) {
    if (have_first && polls == 0) v = v_first;
    else v = __hip_atomic_load(pk + (lane & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one 64-byte request per wave
    const unsigned long long got = roles_lane_word(v, 7);
    if (got != 0ull) {
      unsigned long long x = roles_lane_word(v, 0);
#pragma unroll
      for (int i = 1; i < 7; i++) x ^= roles_lane_word(v, i);
      if (roles_mark(x ^ tagmix) == got) break;              // mark and payload belong together (and to this step)
    }
    if (polls == (AUV_HOOK_FAULT(d) ? (1 << 12) : PAIR_POLL_LIMIT)) {
      if (lane == 0) auv_st<true>(d.broken + e, (uint8_t)1);
      roles_give_up(d, d.e0, d.ne, 2, lane);
      return 1;
    }
    __builtin_amdgcn_s_sleep(ROLES_POLL_SLEEP);
  }
#pragma unroll
  for (int i = 0; i < 6; i++) pre.s[i] = pair_lane_value(v, i);
  pre.cnt.y = (int)__builtin_amdgcn_readlane((int)(unsigned)v, 6);   // the vessel's step counter of this launch
  return (unsigned)pre.cnt.y == ROLES_ABORT_COUNTER ? 2 : 0;
}

// ---- the finish role: navigation tail + reward / done / auto-reset, eight environments per wave ----
// Evaluated by one wave per environment the navigation's scalar tail keeps three to six lanes busy and was 19 % of the
// step's VALU issue cycles (profiles/r03/valu_budget_polygons50.json).  So the navigation wave only SEARCHES: it leaves
// the nearest segment (A, B, cumulative arclength) in a 64-byte record of NAV_HAND, checksummed like the state packet,
// and ends, handing its slot on.  A finish wave -- workgroups behind all others, group g of eight lanes <->
// environment, the dynamics role's mapping (same XCD as the environment's other waves) -- polls for the state packets
// and the search records of its eight environments, evaluates their tails (nav_tail<8>: the very code of the
// whole-wave form, so the same bits), polls for the sweeps' words, runs the reward phase in each group's first lane
// and takes the three marks down; environments that ended are then restored by the whole wave, one after the other.
// The tails run while the sweeps are still busy: only the reward phase (0.5-0.7 us) follows the last sweep.
// (A first version of this role in round 3 waited for all three hand-overs BEFORE the tail and fetched its table
// pointers with vector loads at the point of use: 38 against 31 us per chain step.  With the tail ahead of the second
// wait and the tables in scalar registers before the waits: 136 M against 130 M env-steps/s with four chains, 107
// against 100 M with one -- profiles/r03/shapes_sweep_finish_role.log.)
__device__ __forceinline__ void roles_publish_search(const AuvDev& d, const int e, const int lane, const NavNear& nr, const unsigned long long tagmix = 0ull) {
  unsigned long long* h = d.nav_hand + 8 * (size_t)e;
  // (wave-uniform values: lane 0 stores the five words one by one -- picking "this lane's word" would make the compiler
  // build a table in scratch memory; words 5 and 6 of the record stay zero)
  const unsigned long long w0 = (unsigned long long)__double_as_longlong(nr.A.x), w1 = (unsigned long long)__double_as_longlong(nr.A.y),
                           w2 = (unsigned long long)__double_as_longlong(nr.B.x), w3 = (unsigned long long)__double_as_longlong(nr.B.y),
                           w4 = (unsigned long long)__double_as_longlong(nr.cum);
  // (the mark goes out right behind the payload, not after its completion: it is a checksum of the payload, so a reader
  // that sees it ahead of a payload word polls again -- this wave stores nothing else)
  if (AUV_HOOK_FAULT(d) == 3 && e == d.e0) return;         // (test hook: the finish wave's time-out)
  if (lane == 0) {
    __hip_atomic_store(h + 0, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_store(h + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(h + 2, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_store(h + 3, w3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(h + 4, w4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(h + 7, roles_mark(w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ tagmix), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// word `src` of the caller's group of eight lanes
__device__ __forceinline__ unsigned long long roles_group_word(const unsigned long long v, const int src) {
  const unsigned lo = (unsigned)__shfl((int)(unsigned)v, src, K1_GROUP), hi = (unsigned)__shfl((int)(unsigned)(v >> 32), src, K1_GROUP);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double roles_group_value(const unsigned long long v, const int src) {
  return __longlong_as_double((long long)roles_group_word(v, src));
}
// lane c of a group holds word c of a checksummed 64-byte record: does it hold together?
__device__ __forceinline__ bool roles_record_ok(const unsigned long long v, const int c, const unsigned long long tagmix = 0ull) {
  const unsigned long long x = roles_group_xor(c < 7 ? v : 0ull), mark = roles_group_word(v, 7);
  return mark != 0ull && roles_mark(x ^ tagmix) == mark;
}

__device__ __forceinline__ void roles_finish_wave(const AuvDev& dk, const int f, const int lane, float* __restrict__ obs_out,
                                                  float* __restrict__ reward_out, uint8_t* __restrict__ done_out) {
  // This role touches two dozen of the descriptor's tables, mostly while it waits: as kernel arguments they would be
  // fetched, and spilled, at the entry of every wave of every role.  They are read from the descriptor's device-side
  // copy (AuvDev::self) instead, through a pointer typed as constant memory -- scalar loads, issued here, before the
  // waits.  (Through a generic reference to the copy every table pointer is a vector load at its point of use.)  The
  // launch's own slice comes with the arguments.
  const __attribute__((address_space(4))) AuvDev* dc = (const __attribute__((address_space(4))) AuvDev*)dk.self;
  const AuvDev& d = *(const AuvDev*)dc;                           // (restore path, diagnostics)
  StepTabs st;                                                    // the tail's and the reward phase's tables and settings
  st.cfg.min_cumulative_reward = dc->cfg.min_cumulative_reward, st.cfg.min_goal_distance = dc->cfg.min_goal_distance;
  st.cfg.min_path_progress = dc->cfg.min_path_progress, st.cfg.look_ahead_distance = dc->cfg.look_ahead_distance;
  st.cfg.max_timesteps = dc->cfg.max_timesteps, st.cfg.rewarder = dc->cfg.rewarder, st.cfg.test_mode = dc->cfg.test_mode;
  st.cfg.auto_reset = dc->cfg.auto_reset, st.cfg.n_sensors = dc->cfg.n_sensors, st.cfg.use_lidar = dc->cfg.use_lidar;
  st.cfg.obs_channels = dc->cfg.obs_channels;
  st.knot_s = dc->knot_s, st.knot_coef = dc->knot_coef;
  st.info64 = dc->info64, st.nav64 = dc->nav64, st.obs64 = dc->obs64, st.rew_path = dc->rew_path, st.reward64 = dc->reward64;
  st.step_info = dc->step_info, st.episode = dc->episode, st.ep_log = dc->ep_log, st.ep_log_count = dc->ep_log_count;
  st.ep_log_cap = dc->ep_log_cap, st.world_idx = dc->world_idx, st.counters = dc->counters;
  st.fw_serial = dc->fw_serial, st.n = dc->n;
  st.ring_slots = 1, st.ring_slot_host = 0, st.ring_pos = nullptr;          // (the dynamics role advances a captured graph's ring)
  unsigned long long* const pair_word = dc->pair_word;
  unsigned long long* const k1_pkt = dc->k1_pkt;
  unsigned long long* const nav_hand = dc->nav_hand;
  const EnvDesc* const env_desc = dc->env_desc;
  const double* const world_scalar = dc->world_scalar;
  double* const state = dc->state;
  const int n_envs = dc->n;
  const int ne = dk.ne, e0 = dk.e0;
  const int g = lane / K1_GROUP, c = lane % K1_GROUP;
  const int er = 8 * (8 * (f / 8) + g) + (f % 8);                 // the dynamics role's mapping
  const bool live = er < ne;
  const int e = e0 + (live ? er : ne - 1);                        // idle groups look at the last environment, store nothing
  // what earlier launches left (requested before the polls)
  int4 cnt = st.counters[e];
  const int w = st.world_idx[e];
  const double cum_in = st.info64[8 * (size_t)e + 4], cte_sum_in = st.info64[8 * (size_t)e + 7];
  const EnvDesc ed = env_desc[e];
  TailIn t;
  t.kn0 = ed.kn0, t.nk = ed.nk;
  {
    const double* ws = world_scalar + 8 * (size_t)ed.w;
    t.L = ws[0], t.goal_x = ws[1], t.goal_y = ws[2];
    t.knot_first = st.knot_s[ed.kn0], t.knot_last = st.knot_s[ed.kn0 + ed.nk - 1];
    t.maxp_in = st.info64[8 * (size_t)e + 5];
  }
  const unsigned long long* pk = k1_pkt + 8 * (size_t)e;
  unsigned long long* hd = nav_hand + 8 * (size_t)e;
  const int limit = AUV_HOOK_FAULT(dk) ? (1 << 12) : PAIR_POLL_LIMIT;
#ifdef AUV_STAMPS
  if (live && c == 0) d.stamps[(size_t)e * 16 + 10] = wall_clock64();   // the finish wave has its slot
#endif
  // ---- the state packet and the search record of every group ----
  unsigned long long vp = 0ull, vh = 0ull;
  bool okp = !live, okh = !live, aborted = false;
  uint8_t* const broken = dc->broken;
  for (int polls = 0;; polls++) {
    if (!okp) vp = __hip_atomic_load(pk + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // one 64-byte request per group
    if (!okh) vh = __hip_atomic_load(hd + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    okp = !live || roles_record_ok(vp, c);
    // an ABORT packet (a launch behind a time-out): no search record, no sweep's word will follow
    aborted = live && okp && (unsigned)roles_group_word(vp, 6) == ROLES_ABORT_COUNTER;
    okh = !live || aborted || roles_record_ok(vh, c);
    if (!__any(!(okp && okh))) break;
    if (polls == limit) {
      if (live && c == 0) auv_st<true>(broken + e, (uint8_t)1);
      roles_give_up(d, e0, ne, 3, lane);
      return;
    }
    __builtin_amdgcn_s_sleep(ROLES_POLL_SLEEP);
  }
  if (__any(aborted)) {
    // the eight environments of a finish wave are the eight of ONE dynamics wave: all of them or none.  The ABORT packets
    // stay up -- this wave does not know whether the environments' sweep and search waves have seen them yet; later
    // launches are aborted too (they overwrite them with their own) and the host's recovery clears every mark
    return;
  }
  t.px = roles_group_value(vp, 0), t.py = roles_group_value(vp, 1), t.psi = roles_group_value(vp, 2);
  t.u = roles_group_value(vp, 3), t.v = roles_group_value(vp, 4), t.r = roles_group_value(vp, 5);
  cnt.y = (int)(unsigned)roles_group_word(vp, 6);                 // the vessel's step counter of this launch
  t.nr.A = make_double2(roles_group_value(vh, 0), roles_group_value(vh, 1));
  t.nr.B = make_double2(roles_group_value(vh, 2), roles_group_value(vh, 3));
  t.nr.cum = roles_group_value(vh, 4);
#ifdef AUV_STAMPS
  if (live && c == 0) d.stamps[(size_t)e * 16 + 9] = wall_clock64();    // state and search result are here
#endif
  NavOut no;
  no.rew_path = no.reached = no.goal = no.progress = no.u = no.v = no.r = no.cte100 = 0.0;
  // a first look at the sweeps' words BEFORE the tail's stores (a look behind them waits for them: with one chain the words
  // are there by the time a finish wave runs, and the poll below is then not entered at all)
  unsigned long long word = live ? __hip_atomic_load(pair_word + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  if (AUV_RUN_N(dk, 3)) no = nav_tail<K1_GROUP>(st, e, c, live, t, obs_out);
  // STATE rows of the new state (the dynamics role leaves only the packet): lane c of a group holds component c.  (Behind the
  // tail, not in front of it: the tail's loads would wait for this store -- one counter, in issue order.)
  if (live && c < 6) state[(size_t)c * (size_t)n_envs + e] = __longlong_as_double((long long)vp);
#ifdef AUV_STAMPS
  if (live && c == 0) d.stamps[(size_t)e * 16 + 13] = wall_clock64();   // the tail is done
#endif
  // ---- the sweeps' words ----
  for (int polls = 0;; polls++) {
    if (!__any(word == PAIR_EMPTY)) break;
    if (word == PAIR_EMPTY) word = __hip_atomic_load(pair_word + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!__any(word == PAIR_EMPTY)) break;
    if (polls == limit) {
      if (live && c == 0) auv_st<true>(broken + e, (uint8_t)1);
      roles_give_up(d, e0, ne, 1, lane);
      return;
    }
    __builtin_amdgcn_s_sleep(ROLES_POLL_SLEEP);
  }
#ifdef AUV_STAMPS
  if (live && c == 0) d.stamps[(size_t)e * 16 + 7] = wall_clock64();    // the sweep's word is here: start of the reward phase
#endif
  int do_reset = 0;
  if (live && c == 0) {
    // the three marks come down for the next launch (the sweep and the search waves of this environment are gone)
    __hip_atomic_store(pair_word + e, PAIR_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(k1_pkt + 8 * (size_t)e + 7, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(hd + 7, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    RewardIn in;
    const int collision = word == PAIR_COLLISION;
    in.closeness_reward = collision ? 0.0 : __longlong_as_double((long long)word);
    in.path_reward = no.rew_path, in.reached = no.reached, in.goal = no.goal, in.progress = no.progress;
    in.u = no.u, in.v = no.v, in.yaw_rate = no.r;
    in.cum = cum_in;
    in.cte100 = no.cte100, in.cte_sum = cte_sum_in;
    st.info64[8 * (size_t)e] = collision;
    do_reset = reward_apply(st, e, collision, cnt, in, reward_out, done_out, false);
  }
  unsigned long long m = __ballot(do_reset);
  if (m) {
    while (m) {
      const int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      const int er2 = auv_uniform(__shfl(e, src, AUV_WAVE));
      const int wr = auv_uniform(__shfl(w, src, AUV_WAVE)), ep = auv_uniform(__shfl(cnt.z, src, AUV_WAVE));
      const AuvDev* dp = dk.self;
      asm volatile("" : "+s"(dp));      // (the copy's tables are fetched inside the loop: hoisted out of it they are spilled, and the
                                        // spill slots keep 36 B of scratch memory enabled for every wave of the launch)
      const AuvDev& dr = *(const AuvDev*)(const __attribute__((address_space(4))) AuvDev*)dp;
      restore_env(dr, er2, auv_next_world_wave(dr, wr, lane), lane, ep, obs_out);
    }
  }
#ifdef AUV_STAMPS
  if (live && c == 0) d.stamps[(size_t)e * 16 + 15] = wall_clock64();
#endif
}

__global__ void __launch_bounds__(AUV_WAVE, AUV_K23_MIN_WAVES) k_step_roles(AuvDev dk, const void* __restrict__ actions,
                                                                           float* __restrict__ obs_out,
                                                                           float* __restrict__ reward_out,
                                                                           uint8_t* __restrict__ done_out) {
  const AuvDev& d = dk;
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const int ne = d.ne;                                              // environments of this launch: [e0, e0 + ne)
  const int nk = 8 * ((ne + 63) / 64);                              // dynamics workgroups
  const int nb = 8 * ((ne + 7) / 8);                                // LiDAR workgroups
  // (the search workgroups dispatched AHEAD of the LiDAR ones -- they are short, and with one chain they only get their
  // slots when sweeps retire -- was measured: 102.3 against 106.5 M env-steps/s with one chain, 129 against 136 M with
  // four: the sweeps are the long pole and must start first.  Blocks of eight LiDAR and eight search workgroups
  // alternating: 95.6 against 111.6 M with one chain, 130 against 142 M with four.)
  const int b = (int)blockIdx.x;
  if (b < nk) {
    // ---- Vessel.step of eight environments ----
    // (everybody else in the launch waits for these 512 waves: they go first wherever they share a SIMD)
    __builtin_amdgcn_s_setprio(3);
    const int g = lane / K1_GROUP, c = lane % K1_GROUP;
    const int er = 8 * (8 * (b / 8) + g) + (b % 8);
    const bool live = er < ne;
    const int eg = d.e0 + (live ? er : ne - 1);                     // idle groups compute along, store nothing
    const int y = d.counters[eg].y + 1;                             // Vessel._step_counter (vessel.py:247); requested up front
    // a launch queued behind a hand-over time-out (the flag is wave-uniform, a scalar load): the dynamics role hands out
    // ABORT packets instead of states, on which every other wave of the launch ends without touching its environment --
    // the whole decision hangs on THIS wave's one look at the flag, so an environment's four waves always agree
    const int aborted = auv_uniform(*d.abort_flag);
    double t = 0.0;
    if (!aborted) t = k1_group(d, actions, eg, lane);
    unsigned long long* pk = d.k1_pkt + 8 * (size_t)eg;
    const unsigned long long word = aborted ? (c == 6 ? (unsigned long long)ROLES_ABORT_COUNTER : 0ull)
                                            : (c < 6 ? (unsigned long long)__double_as_longlong(t) : (c == 6 ? (unsigned long long)(unsigned)y : 0ull));
    const unsigned long long mark = roles_mark(roles_group_xor(word));
    // The packet is ALL this role stores: payload and mark in ONE store instruction, 64 contiguous bytes per group, and
    // the wave does not wait for it.  The mark is a checksum of the payload, so a reader that sees the mark before all
    // of the payload (should the request ever be served in pieces) just polls again.  STATE and the vessel's step
    // counter are written by the finish wave from the packet (roles_finish_wave) -- the wave that may also overwrite
    // them when the episode ends, so the two stores are ordered by ITS program order, not by a wait here -- and the
    // 0.5-1 us this wave used to wait for its write-through stores before raising the mark are off the head of the
    // launch that everybody else waits for.
    if (live && !(AUV_HOOK_FAULT(d) == 2 && eg == d.e0))       // (test hook: nobody gets the first environment's state)
      __hip_atomic_store(pk + c, c < 7 ? word : mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    // memory latency then hides behind the late sweeps' arithmetic.  With four roles: a sweep that is behind schedule
    // raising its own priority -- clock read after the front and after the pair sweep, s_setprio 2 beyond 4 / 8 or
    // 5.5 / 11 us since the state arrived -- changes nothing, 135.3-135.9 against 135.6-136.4 M with four chains: the
    // stragglers are not short of issue slots.)
    // while the dynamics role integrates: everything of the sweep that does not need the vessel's new state --
    // descriptor and counters (written by earlier launches), the movers' kinematics, the obstacle records
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 3] = wall_clock64();
#endif
    // (a launch queued behind a hand-over time-out: nothing is touched.  The dynamics wave's look at the flag is the one
    // that counts -- see there -- this one only keeps the movers from being advanced by a launch that will be aborted)
    if (auv_uniform(*d.abort_flag)) return;
    ed = d.env_desc[e];
    pre.cnt = d.counters[e];                               // t_step, episodes; the step counter comes with the state
    pre.ed = &ed;
    const Slice L = carve(smem, d);
    k2_movers<true>(d, e, lane, L, ed, 1);
    const K2Pre kp = k2_prefetch(d, e, lane, ed);
    k2_stage_beams(d, lane, L);
    // (also tried here: warming the caches with the nearby obstacles' boundary segments -- it has to wait for the
    // obstacle records, and its traffic delays the dynamics role: 102.1 M against 104.5 M env-steps/s without)
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 1] = wall_clock64();
#endif
    {
      const int ws = roles_wait_state(d, e, lane, pre);
      if (ws) {
        // an ABORT packet although this wave saw the flag down a moment ago: the flag went up in between, and the movers of
        // this environment have been advanced for a step that will not happen -- the recovery resets it
        if (ws == 2 && ed.M > 0 && lane == 0) auv_st<true>(d.broken + e, (uint8_t)1);
        return;
      }
    }
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 2] = wall_clock64();
#endif
    int n_act = 0;
    int2 lim0 = make_int2(INT32_MIN, INT32_MIN);           // this lane's cull-limit row: stored by k2_back (see there)
    if (AUV_RUN_L(d, 1)) n_act = k2_front<true>(d, e, lane, L, 1, &pre, 1, &kp, true, &lim0);
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 8] = wall_clock64();      // front (B0, B, C) done
#endif
    if (AUV_RUN_L(d, 3)) k2_stage_and_pairs(d, L, lane, n_act, pre.s[2]);
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 11] = wall_clock64();     // staging + pair sweep (S, D) done
#endif
    double term = 0.0;
    const int collision = k2_back<true>(d, e, lane, L, n_act, obs_out, &term, &lim0);
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 4] = wall_clock64();
#endif
    pair_publish_lidar(d, e, lane, collision, term);
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 14] = wall_clock64();
#endif
  } else if (b < nk + 2 * nb + AUV_HOOK_SKEW(d)) {
    // ---- Vessel.navigate of one environment: the nearest-point search ----
    const int el = b - nk - nb - AUV_HOOK_SKEW(d);
    if (el < 0 || el >= ne) return;
    const int e = auv_uniform(d.e0 + el);
    ed = d.env_desc[e];
    pre.cnt = d.counters[e];
    pre.ed = &ed;
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 12] = wall_clock64();
#endif
    if (roles_wait_state(d, e, lane, pre)) return;
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 5] = wall_clock64();
#endif
    // the search only: nearest segment of the path -> NAV_HAND; the finish role takes it from there
    NavNear nr;
    nr.A = nr.B = make_double2(0.0, 0.0), nr.cum = 0.0;
    if (AUV_RUN_N(d, 1)) {
      int* list = (int*)smem;
      const NavSpec sp = nav_bounds(d, e, lane, list, pre.s[0], pre.s[1], &ed);
      if (AUV_RUN_N(d, 2)) nr = nav_nearest(d, e, lane, list, pre.s[0], pre.s[1], sp);
    }
    roles_publish_search(d, e, lane, nr);
#ifdef AUV_STAMPS
    if (lane == 0) d.stamps[(size_t)e * 16 + 6] = wall_clock64();     // the search result is out
#endif
  } else {
    // ---- navigation tail + reward / done / auto-reset of eight environments ----
    const int f = b - nk - 2 * nb - AUV_HOOK_SKEW(d);
    if (f < 0 || f >= nk) return;
    __builtin_amdgcn_s_setprio(2);
    roles_finish_wave(d, f, lane, obs_out, reward_out, done_out);
  }
}

// ---- SEVERAL steps in ONE launch (auv_step_multi) ---------------------------------------------------------
// An open-loop stretch knows its actions ahead (a ring of slots), so nothing but the launch boundary makes step t + 1 of an
// environment wait for the slowest sweep of the whole slice.  k_step_multi is k_step_roles' grid T times over, step-major:
// workgroup b plays role b % per of step b / per.  Workgroups are still dispatched in index order, so every wave's producers --
// its own step's lower roles and the previous step's finish wave -- are resident or done when it gets its slot; as step t's
// waves retire, step t + 1's take their slots, and an environment's next dynamics starts when ITS finish wave is through.
// What a kernel boundary used to do between two steps is done explicitly:
//   * the CARRY record (AuvDev::carry, 24 words per environment): what the next step's roles need of this step's outcome --
//     the state and the counters (dynamics: words 0..7, the very format of the state packet), episode count, cumulative reward,
//     maximum progress, cross-track sum and the table descriptor (sweep, search, finish).  The finish wave writes it last, by
//     agent-scope stores, with two checksum marks that carry the step's number; the next step's waves poll for marks that match
//     THEIR step.  Step 0 of a launch reads the arrays as ever (a kernel boundary lies in front of it), and the last step's finish
//     wave leaves the arrays as ever for the launch that follows.
//   * the rows one wave writes and ANOTHER step's wave reads -- mover states, the cached nearby mask -- are stored write-through
//     (they already were) and, in steps > 0, loaded with agent-scope loads (k2_movers / k2_prefetch <LD>); restore_env stores
//     them write-through too (<WTM>).  Everything else a step writes is output only (or a hint: the path search's starting
//     chunk, where a stale value costs nothing but a wider first bound).
//   * every mark of a step is mixed with the step's number (roles_tagmix), so a packet of step t still up cannot pass for
//     step t + 1's.
// Bitwise the same as T launches of k_step_roles (tests/test_gpu_multi.py).  Not with a fresh world per reset: there a slot's
// tables may be rebuilt beside the launch, and only the binding launch reads them coherently.
#define CARRY_WORDS 24

// Diagnostic build only (-DAUV_STAMPS_MULTI, tools/multi_stamps.py): wall-clock stamps of the waves of ONE step in the middle of a
// multi-step launch (the other steps stamp into a dummy half of the array), 16 words per environment:
//   sweep wave   0 role begins  1 the wave's first instruction  2 state packet here  3 all its stores acknowledged  4 pair sweep done  5 word published
//   search wave  6 role begins  7 the wave's first instruction  8 state packet here  9 record published
//   dynamics    10 has its slot 11 carry record here 12 packet stored      finish 13 has its slot 14 sweeps' words here 15 done
#ifdef AUV_STAMPS_MULTI
#define MSTAMP(e, k) do { if (lane == 0) stamp[(size_t)(e) * 16 + (k)] = wall_clock64(); } while (0)
#define MSTAMP_IF(cond, e, k) do { if (cond) stamp[(size_t)(e) * 16 + (k)] = wall_clock64(); } while (0)
#define MSTAMP_PARAM , unsigned long long* stamp
#define MSTAMP_ARG , stamp
#else
#define MSTAMP(e, k) do { } while (0)
#define MSTAMP_IF(cond, e, k) do { } while (0)
#define MSTAMP_PARAM
#define MSTAMP_ARG
#endif

__device__ __forceinline__ unsigned long long carry_ed_word(const EnvDesc& ed, const int i) {
  switch (i) {
    case 0: return (unsigned long long)ed.k0;
    case 1: return (unsigned long long)ed.m0;
    case 2: return (unsigned long long)ed.p0;
    case 3: return (unsigned long long)ed.c0;
    case 4: return (unsigned long long)ed.kn0;
    case 5: return (unsigned long long)(unsigned)ed.K | ((unsigned long long)(unsigned)ed.M << 32);
    case 6: return (unsigned long long)(unsigned)ed.P | ((unsigned long long)(unsigned)ed.nch << 32);
    default: return (unsigned long long)(unsigned)ed.nk | ((unsigned long long)(unsigned)ed.w << 32);
  }
}
__device__ __forceinline__ void carry_ed_from(EnvDesc& ed, const unsigned long long w[8]) {
  ed.k0 = (long long)w[0], ed.m0 = (long long)w[1], ed.p0 = (long long)w[2], ed.c0 = (long long)w[3], ed.kn0 = (long long)w[4];
  ed.K = (int)(unsigned)w[5], ed.M = (int)(unsigned)(w[5] >> 32), ed.P = (int)(unsigned)w[6], ed.nch = (int)(unsigned)(w[6] >> 32);
  ed.nk = (int)(unsigned)w[7], ed.w = (int)(unsigned)(w[7] >> 32);
}

// one wave, one environment (sweep, search): the carry of the previous step.  0: here it is; 1: gave up (reported, environment
// marked); 2: the abort flag went up meanwhile -- the wave ends without touching anything
__device__ __forceinline__ int carry_wait_wave(const AuvDev& d, const int e, const int lane, const unsigned long long tagmix, EnvDesc& ed, int4& cnt) {
  const unsigned long long* cw = d.carry + CARRY_WORDS * (size_t)e;
  unsigned long long v = 0ull;
  for (int polls = 0;; polls++) {
    if (lane < CARRY_WORDS) v = __hip_atomic_load(cw + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long ma = roles_lane_word(v, 7), mb = roles_lane_word(v, 20);
    if (ma != 0ull && mb != 0ull) {
      unsigned long long xa = roles_lane_word(v, 0), xb = roles_lane_word(v, 8);
#pragma unroll
      for (int i = 1; i < 7; i++) xa ^= roles_lane_word(v, i);
#pragma unroll
      for (int i = 9; i < 20; i++) xb ^= roles_lane_word(v, i);
      if (roles_mark(xa ^ tagmix) == ma && roles_mark(xb ^ tagmix) == mb) break;
    }
    if ((polls & 31) == 31 && auv_uniform(__hip_atomic_load(d.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) return 2;
    if (polls == (AUV_HOOK_FAULT(d) ? (1 << 12) : PAIR_POLL_LIMIT)) {
      if (lane == 0) auv_st<true>(d.broken + e, (uint8_t)1);
      roles_give_up(d, d.e0, d.ne, 6, lane);
      return 1;
    }
    __builtin_amdgcn_s_sleep(ROLES_POLL_SLEEP);
  }
  unsigned long long w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = roles_lane_word(v, 12 + i);
  carry_ed_from(ed, w);
  const unsigned long long c6 = roles_lane_word(v, 6), c8 = roles_lane_word(v, 8);
  cnt = make_int4((int)(unsigned)c6, (int)(unsigned)(c6 >> 32), (int)(unsigned)c8, 0);
  return 0;
}

template <bool MULTI>
__device__ __forceinline__ void roles_finish_wave_multi(const AuvDev& dk, const int f, const int lane, float* __restrict__ obs_out,
                                                        float* __restrict__ reward_out, uint8_t* __restrict__ done_out, const int step,
                                                        const bool last_step, const unsigned long long tagmix, const unsigned long long tagmix_prev MSTAMP_PARAM) {
  // (roles_finish_wave with the previous step's outcome from the carry record instead of the arrays, and its own outcome into
  // the record at the end; the arithmetic in between is the very same code)
  const __attribute__((address_space(4))) AuvDev* dc = (const __attribute__((address_space(4))) AuvDev*)dk.self;
  const AuvDev& d = *(const AuvDev*)dc;
  StepTabs st;
  st.cfg.min_cumulative_reward = dc->cfg.min_cumulative_reward, st.cfg.min_goal_distance = dc->cfg.min_goal_distance;
  st.cfg.min_path_progress = dc->cfg.min_path_progress, st.cfg.look_ahead_distance = dc->cfg.look_ahead_distance;
  st.cfg.max_timesteps = dc->cfg.max_timesteps, st.cfg.rewarder = dc->cfg.rewarder, st.cfg.test_mode = dc->cfg.test_mode;
  st.cfg.auto_reset = dc->cfg.auto_reset, st.cfg.n_sensors = dc->cfg.n_sensors, st.cfg.use_lidar = dc->cfg.use_lidar;
  st.cfg.obs_channels = dc->cfg.obs_channels;
  st.knot_s = dc->knot_s, st.knot_coef = dc->knot_coef;
  st.info64 = dc->info64, st.nav64 = dc->nav64, st.obs64 = dc->obs64, st.rew_path = dc->rew_path, st.reward64 = dc->reward64;
  st.step_info = dc->step_info, st.episode = dc->episode, st.ep_log = dc->ep_log, st.ep_log_count = dc->ep_log_count;
  st.ep_log_cap = dc->ep_log_cap, st.world_idx = dc->world_idx, st.counters = dc->counters;
  st.fw_serial = nullptr, st.n = dc->n;
  st.ring_slots = 1, st.ring_slot_host = 0, st.ring_pos = nullptr;
  unsigned long long* const pair_word = dc->pair_word;
  unsigned long long* const k1_pkt = dc->k1_pkt;
  unsigned long long* const nav_hand = dc->nav_hand;
  unsigned long long* const carry = dc->carry;
  const EnvDesc* const env_desc = dc->env_desc;
  const double* const world_scalar = dc->world_scalar;
  double* const state = dc->state;
  uint8_t* const broken = dc->broken;
  const int n_envs = dc->n;
  const int ne = dk.ne, e0 = dk.e0;
  const int g = lane / K1_GROUP, c = lane % K1_GROUP;
  const int er = 8 * (8 * (f / 8) + g) + (f % 8);
  const bool live = er < ne;
  const int e = e0 + (live ? er : ne - 1);
  const int limit = AUV_HOOK_FAULT(dk) ? (1 << 12) : PAIR_POLL_LIMIT;
  int4 cnt;
  double cum_in, cte_sum_in, maxp_in;
  EnvDesc ed;
  unsigned long long* const cw = carry + CARRY_WORDS * (size_t)e;
  if (step == 0) {
    cnt = st.counters[e];
    cum_in = st.info64[8 * (size_t)e + 4], cte_sum_in = st.info64[8 * (size_t)e + 7], maxp_in = st.info64[8 * (size_t)e + 5];
    ed = env_desc[e];
  } else {
    // ---- the previous step's carry: words c, 8 + c, 16 + c of the group's environment ----
    // (idle groups wait for nobody: the record of the environment they shadow -- the launch's last -- belongs to ITS waves, and
    // its finish wave of this step may overwrite it before a foreign reader has looked; they take that environment's table
    // descriptor from the array instead -- possibly a step old, always a valid one: their table reads stay in bounds and their
    // results are dropped)
    unsigned long long v0 = 0ull, v1 = 0ull, v2 = 0ull;
    bool ok = !live;
    for (int polls = 0;; polls++) {
      if (!ok) {
        v0 = __hip_atomic_load(cw + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v1 = __hip_atomic_load(cw + 8 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v2 = __hip_atomic_load(cw + 16 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const unsigned long long xb = roles_group_xor(v1) ^ roles_group_xor(c < 4 ? v2 : 0ull), mb = roles_group_word(v2, 4);
      ok = !live || (roles_record_ok(v0, c, tagmix_prev) && mb != 0ull && roles_mark(xb ^ tagmix_prev) == mb);
      if (!__any(!ok)) break;
      if ((polls & 31) == 31 && auv_uniform(__hip_atomic_load(dc->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) return;
      if (polls == limit) {
        if (live && c == 0) auv_st<true>(broken + e, (uint8_t)1);
        roles_give_up(d, e0, ne, 6, lane);
        return;
      }
      __builtin_amdgcn_s_sleep(ROLES_POLL_SLEEP);
    }
    const unsigned long long c6 = roles_group_word(v0, 6), c8 = roles_group_word(v1, 0);
    cnt = make_int4((int)(unsigned)c6, (int)(unsigned)(c6 >> 32), (int)(unsigned)c8, 0);
    cum_in = roles_group_value(v1, 1), maxp_in = roles_group_value(v1, 2), cte_sum_in = roles_group_value(v1, 3);
    unsigned long long w[8];
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = roles_group_word(v1, 4 + i);
#pragma unroll
    for (int i = 0; i < 4; i++) w[4 + i] = roles_group_word(v2, i);
    carry_ed_from(ed, w);
    if (!live) ed = env_desc[e];
  }
  const int w = ed.w;
  TailIn t;
  t.kn0 = ed.kn0, t.nk = ed.nk;
  {
    const double* ws = world_scalar + 8 * (size_t)ed.w;
    t.L = ws[0], t.goal_x = ws[1], t.goal_y = ws[2];
    t.knot_first = st.knot_s[ed.kn0], t.knot_last = st.knot_s[ed.kn0 + ed.nk - 1];
    t.maxp_in = maxp_in;
  }
  const unsigned long long* pk = k1_pkt + 8 * (size_t)e;
  unsigned long long* hd = nav_hand + 8 * (size_t)e;
  // ---- the state packet and the search record of every group (this step's) ----
  unsigned long long vp = 0ull, vh = 0ull;
  bool okp = !live, okh = !live, aborted = false;
  for (int polls = 0;; polls++) {
    if (!okp) vp = __hip_atomic_load(pk + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!okh) vh = __hip_atomic_load(hd + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    okp = !live || roles_record_ok(vp, c, tagmix);
    aborted = live && okp && (unsigned)roles_group_word(vp, 6) == ROLES_ABORT_COUNTER;
    okh = !live || aborted || roles_record_ok(vh, c, tagmix);
    if (!__any(!(okp && okh))) break;
    if (polls == limit) {
      if (live && c == 0) auv_st<true>(broken + e, (uint8_t)1);
      roles_give_up(d, e0, ne, 3, lane);
      return;
    }
    __builtin_amdgcn_s_sleep(ROLES_POLL_SLEEP);
  }
  if (__any(aborted)) return;
  t.px = roles_group_value(vp, 0), t.py = roles_group_value(vp, 1), t.psi = roles_group_value(vp, 2);
  t.u = roles_group_value(vp, 3), t.v = roles_group_value(vp, 4), t.r = roles_group_value(vp, 5);
  cnt.y = (int)(unsigned)roles_group_word(vp, 6);
  t.nr.A = make_double2(roles_group_value(vh, 0), roles_group_value(vh, 1));
  t.nr.B = make_double2(roles_group_value(vh, 2), roles_group_value(vh, 3));
  t.nr.cum = roles_group_value(vh, 4);
  NavOut no;
  no.rew_path = no.reached = no.goal = no.progress = no.u = no.v = no.r = no.cte100 = no.maxp = 0.0;
  unsigned long long word = live ? __hip_atomic_load(pair_word + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  no = nav_tail<K1_GROUP>(st, e, c, live, t, obs_out);
  if (live && c < 6) state[(size_t)c * (size_t)n_envs + e] = __longlong_as_double((long long)vp);
  for (int polls = 0;; polls++) {
    if (!__any(word == PAIR_EMPTY)) break;
    if (word == PAIR_EMPTY) word = __hip_atomic_load(pair_word + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!__any(word == PAIR_EMPTY)) break;
    if (polls == limit) {
      if (live && c == 0) auv_st<true>(broken + e, (uint8_t)1);
      roles_give_up(d, e0, ne, 1, lane);
      return;
    }
    __builtin_amdgcn_s_sleep(ROLES_POLL_SLEEP);
  }
  MSTAMP_IF(live && c == 0, e, 14);
  int do_reset = 0;
  double co[2] = {0.0, 0.0};
  if (live && c == 0) {
    __hip_atomic_store(pair_word + e, PAIR_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(k1_pkt + 8 * (size_t)e + 7, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(hd + 7, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    RewardIn in;
    const int collision = word == PAIR_COLLISION;
    in.closeness_reward = collision ? 0.0 : __longlong_as_double((long long)word);
    in.path_reward = no.rew_path, in.reached = no.reached, in.goal = no.goal, in.progress = no.progress;
    in.u = no.u, in.v = no.v, in.yaw_rate = no.r;
    in.cum = cum_in;
    in.cte100 = no.cte100, in.cte_sum = cte_sum_in;
    st.info64[8 * (size_t)e] = collision;
    do_reset = reward_apply(st, e, collision, cnt, in, reward_out, done_out, false, co);
  }
  unsigned long long m = __ballot(do_reset);
  const unsigned long long m_reset = m;
  while (m) {
    const int src = __ffsll((long long)m) - 1;
    m &= m - 1;
    const int er2 = auv_uniform(__shfl(e, src, AUV_WAVE));
    const int wr = auv_uniform(__shfl(w, src, AUV_WAVE)), ep = auv_uniform(__shfl(cnt.z, src, AUV_WAVE));
    const AuvDev* dp = dk.self;
    asm volatile("" : "+s"(dp));
    const AuvDev& dr = *(const AuvDev*)(const __attribute__((address_space(4))) AuvDev*)dp;
    restore_env<false, true>(dr, er2, (int)(((long long)wr + dr.n) % dr.n_worlds), lane, ep, obs_out);
  }
  if (last_step) return;                                   // (the launch that follows reads the arrays, behind a kernel boundary)
  // ---- this step's outcome into the carry record ----
  // every store the next step's waves depend on -- the marks taken down, a restored environment's mover / nearby rows -- is
  // complete before the record can be seen
  auv_stores_done();
  unsigned long long w0, w1, w2;
  {
    // group lane 0 ran the reward phase: its counters, cumulative reward, cross-track sum and the tail's maximum progress
    const int cx = __shfl(cnt.x, 0, K1_GROUP), cy = __shfl(cnt.y, 0, K1_GROUP), cz = __shfl(cnt.z, 0, K1_GROUP);
    const double cum = __shfl(co[0], 0, K1_GROUP), cte = __shfl(co[1], 0, K1_GROUP), maxp = __shfl(no.maxp, 0, K1_GROUP);
    w0 = c < 6 ? vp : (unsigned long long)(unsigned)cx | ((unsigned long long)(unsigned)cy << 32);
    w1 = c == 0 ? (unsigned long long)(unsigned)cz
                : (c == 1 ? (unsigned long long)__double_as_longlong(cum)
                          : (c == 2 ? (unsigned long long)__double_as_longlong(maxp) : (c == 3 ? (unsigned long long)__double_as_longlong(cte) : carry_ed_word(ed, c - 4))));
    w2 = c < 4 ? carry_ed_word(ed, 4 + c) : 0ull;
  }
  if (m_reset & (0xffull << (8 * g))) {
    // this group's environment was restored: what the restore has just written (this wave's own stores, complete: read past L1)
    const size_t n = (size_t)n_envs;
    const int4 rc = make_int4(0, 0, __shfl(cnt.z, 0, K1_GROUP), 0);
    const EnvDesc ne_ = auv_make_desc<true>(d, (int)(((long long)w + d.n) % d.n_worlds));
    const double sv = c < 6 ? __hip_atomic_load(state + (size_t)c * n + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    const double iv = (c >= 1 && c <= 3) ? __hip_atomic_load(st.info64 + 8 * (size_t)e + (c == 1 ? 4 : (c == 2 ? 5 : 7)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    w0 = c < 6 ? (unsigned long long)__double_as_longlong(sv) : 0ull;          // counters x = y = 0
    w1 = c == 0 ? (unsigned long long)(unsigned)rc.z : (c <= 3 ? (unsigned long long)__double_as_longlong(iv) : carry_ed_word(ne_, c - 4));
    w2 = c < 4 ? carry_ed_word(ne_, 4 + c) : 0ull;
  }
  const unsigned long long ma = roles_mark(roles_group_xor(c < 7 ? w0 : 0ull) ^ tagmix);
  const unsigned long long mb = roles_mark(roles_group_xor(w1) ^ roles_group_xor(c < 4 ? w2 : 0ull) ^ tagmix);
  if (live) {
    __hip_atomic_store(cw + c, c < 7 ? w0 : ma, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(cw + 8 + c, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(cw + 16 + c, c < 4 ? w2 : (c == 4 ? mb : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The descriptor is the kernel's FIRST argument: offset 0 of the argument segment.  Read through the segment pointer (laundered, so that
// the compiler cannot prove the loads safe to speculate) its fields are fetched where a role uses them -- not every field any role
// uses, in four or five dependent batches with scalar registers spilled in between, at the top of every wave (0.8 us of a 6 us search
// wave: tools/multi_stamps.py).  k_step_multi: 176 -> 186 M env-steps/s at 64 steps per launch, 160 -> 171 M over the driver's 20-step
// window; k_step_roles: 151 -> 150 M on four chains, so the one-step launch keeps the by-value reference
// (profiles/r05/ab_descriptor_through_segment_pointer.jsonl).
#define AUV_KERNARG_DESC(name)                                                                                                                       \
  const __attribute__((address_space(4))) void* name##_kp_ = (const __attribute__((address_space(4))) void*)__builtin_amdgcn_kernarg_segment_ptr(); \
  asm volatile("" : "+s"(name##_kp_));                                                                                                               \
  const AuvDev& name = *(const AuvDev*)(const __attribute__((address_space(4))) AuvDev*)name##_kp_

__global__ void __launch_bounds__(AUV_WAVE, AUV_K23_MIN_WAVES) k_step_multi(AuvDev dk, const void* __restrict__ actions, float* __restrict__ obs_out,
                                                                           float* __restrict__ reward_out, uint8_t* __restrict__ done_out,
                                                                           const int n_steps, const int first_slot, const int n_slots,
                                                                           const unsigned long long seq0, const int lead_dyn, const int lag_fin, const unsigned magic_c) {
  AUV_KERNARG_DESC(d);
  extern __shared__ __align__(16) unsigned char smem[];
#ifdef AUV_STAMPS_MULTI
  const unsigned long long t_entry = wall_clock64();      // the wave's first instruction
#endif
  const int lane = threadIdx.x;
  const int ne = d.ne;
  const int nk = 8 * ((ne + 63) / 64), nb = 8 * ((ne + 7) / 8);
  const int per = 2 * nk + 2 * nb;
  // role: 0 dynamics, 1 sweep, 2 search, 3 finish; bi: the wave's index within its role and step.
  // Two workgroup orders (both computed, one selected: a run-time branch around this index arithmetic makes this compiler emit a
  // vector-to-scalar copy it then rejects):
  //   step-major: all of step t's workgroups, role by role, then step t + 1's;
  //   COHORT-PIPELINED (lead_dyn >= 0).  A cohort = 64 consecutive environments = 8 dynamics + 64 sweep + 64 search + 8 finish
  //   workgroups (every run a multiple of 8: an environment's waves still share an XCD).  Cohort-steps are numbered
  //   q = step * C + cohort; position p of the grid holds the dynamics of q = p, the sweeps and searches of q = p - lead and the
  //   finish waves of q = p - lead - lag: a sweep is dispatched `lead` positions behind its dynamics -- which have finished by
  //   then -- and a finish wave `lag` positions behind its sweeps, so waves find what they need instead of holding a slot while
  //   they poll for it (the step-major order makes step t + 1's sweeps wait, resident, for a finish wave that is dispatched last
  //   of all of step t).  lead + lag < C keeps every producer ahead of its consumer in index order, also across steps: the
  //   dynamics of q + C sit at position q + C, behind the finish waves of q at q + lead + lag.
  const bool cohorts = lead_dyn >= 0;
  const int bx = (int)blockIdx.x;
  // step-major
  const int step_a = bx / per, b_a = bx - step_a * per;
  const int role_a = b_a < nk ? 0 : (b_a < nk + nb ? 1 : (b_a < nk + 2 * nb ? 2 : 3));
  const int bi_a = b_a - (role_a == 0 ? 0 : (role_a == 1 ? nk : (role_a == 2 ? nk + nb : nk + 2 * nb)));
  // cohort-pipelined
  const int C = nk / 8;
  const int p = bx / 144, r = bx - 144 * p;
  const int role_c = r < 8 ? 0 : (r < 72 ? 1 : (r < 136 ? 2 : 3));
  const int q = p - (role_c == 0 ? 0 : (role_c == 3 ? lead_dyn + lag_fin : lead_dyn));
  const bool q_ok = q >= 0 && q < n_steps * C;
  const int step_c = (int)__umulhi((unsigned)(q_ok ? q : 0), magic_c);      // q / C by the host's multiplier ceil(2^32 / C): exact for q < 2^32 / C
  const int c = (q_ok ? q : 0) - step_c * C;
  const int bi_c = (role_c == 0 ? r : (role_c == 1 ? r - 8 : (role_c == 2 ? r - 72 : r - 136))) + ((role_c == 0 || role_c == 3) ? 8 * c : 64 * c);
  int step = auv_uniform(cohorts ? (q_ok ? step_c : n_steps) : step_a);      // (a position outside the launch: ends below)
  int role = auv_uniform(cohorts ? role_c : role_a), bi = auv_uniform(cohorts ? bi_c : bi_a);
  if (step >= n_steps) return;
  const unsigned long long tagmix = roles_tagmix(seq0 + (unsigned long long)step + 1ull), tagmix_prev = roles_tagmix(seq0 + (unsigned long long)step);
#ifdef AUV_STAMPS_MULTI
  unsigned long long* const stamp = d.stamps + (step == n_steps / 2 ? (size_t)0 : (size_t)16 * (size_t)d.n);
#endif
  if (role == 0) {
    // ---- Vessel.step of eight environments ----
    if (bi >= nk) return;
    __builtin_amdgcn_s_setprio(3);
    const int b = bi;
    const int g = lane / K1_GROUP, c = lane % K1_GROUP;
    const int er = 8 * (8 * (b / 8) + g) + (b % 8);
    const bool live = er < ne;
    const int eg = d.e0 + (live ? er : ne - 1);
    int y, gave_up = 0;
    double y0 = 0.0;
    MSTAMP_IF(live && c == 0, eg, 10);
    // requested ahead of the wait for the carry record (this wave is on every environment's critical path): the flag and the action
    const int ab_early = __hip_atomic_load(d.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double act0, act1;
    k1_action(d, actions, eg, &act0, &act1, (first_slot + step) % n_slots);
    if (step == 0) {
      y = d.counters[eg].y + 1;
    } else {
      // this environment's state and counters after the previous step: the first line of its carry record
      const unsigned long long* cw = d.carry + CARRY_WORDS * (size_t)eg;
      unsigned long long v = 0ull;
      bool ok = !live;
      for (int polls = 0;; polls++) {
        if (!ok) v = __hip_atomic_load(cw + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = !live || roles_record_ok(v, c, tagmix_prev);
        if (!__any(!ok)) break;
        if ((polls & 31) == 31 && auv_uniform(__hip_atomic_load(d.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          gave_up = 1;                                                              // (ABORT packets below)
          break;
        }
        if (polls == (AUV_HOOK_FAULT(d) ? (1 << 12) : PAIR_POLL_LIMIT)) {
          if (live && c == 0) auv_st<true>(d.broken + eg, (uint8_t)1);
          roles_give_up(d, d.e0, d.ne, 6, lane);
          gave_up = 1;                                                              // (the flag is up now: ABORT packets below)
          break;
        }
        __builtin_amdgcn_s_sleep(ROLES_POLL_SLEEP);
      }
      y0 = __longlong_as_double((long long)v);
      y = (int)(unsigned)(roles_group_word(v, 6) >> 32) + 1;
    }
    const int aborted = gave_up | auv_uniform(ab_early);
    double t = 0.0;
    MSTAMP_IF(live && c == 0, eg, 11);
    const double2 act = make_double2(act0, act1);
    if (!aborted) t = k1_group(d, actions, eg, lane, step == 0 ? nullptr : &y0, (first_slot + step) % n_slots, &act);
    unsigned long long* pk = d.k1_pkt + 8 * (size_t)eg;
    const unsigned long long word = aborted ? (c == 6 ? (unsigned long long)ROLES_ABORT_COUNTER : 0ull)
                                            : (c < 6 ? (unsigned long long)__double_as_longlong(t) : (c == 6 ? (unsigned long long)(unsigned)y : 0ull));
    const unsigned long long mark = roles_mark(roles_group_xor(word) ^ tagmix);
    if (live && !(AUV_HOOK_FAULT(d) == 2 && eg == d.e0 && step == 0))
      __hip_atomic_store(pk + c, c < 7 ? word : mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    MSTAMP_IF(live && c == 0, eg, 12);
    return;
  }
  EnvPre pre;
  EnvDesc ed;
  if (role == 1) {
    // ---- _update + Vessel.perceive of one environment ----
    if (bi >= ne) return;
    const int e = auv_uniform(d.e0 + bi);
    MSTAMP(e, 0);
    // three requests in flight before the first wait: the abort flag, this step's state packet (dispatched `lead` cohorts behind
    // its dynamics, the wave usually finds it there), the carry record
    const int ab_early = __hip_atomic_load(d.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long pk_first = __hip_atomic_load(d.k1_pkt + 8 * (size_t)e + (lane & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const Slice L = carve(smem, d);
    K2Pre kp;
    if (step == 0) {
      ed = d.env_desc[e];
      pre.cnt = d.counters[e];
      pre.ed = &ed;
      if (auv_uniform(ab_early)) return;
      k2_movers<true>(d, e, lane, L, ed, 1);
      kp = k2_prefetch(d, e, lane, ed);
    } else {
      if (carry_wait_wave(d, e, lane, tagmix_prev, ed, pre.cnt)) return;
      if (auv_uniform(ab_early)) return;
      pre.ed = &ed;
      k2_movers<true, true>(d, e, lane, L, ed, 1);
      kp = k2_prefetch<true>(d, e, lane, ed);
    }
    k2_stage_beams(d, lane, L);
    {
      const int ws = roles_wait_state(d, e, lane, pre, tagmix, true, pk_first);
      if (ws) {
        if (ws == 2 && ed.M > 0 && lane == 0) auv_st<true>(d.broken + e, (uint8_t)1);
        return;
      }
    }
    MSTAMP(e, 2);
    int2 lim0 = make_int2(INT32_MIN, INT32_MIN);
    const int n_act = k2_front<true>(d, e, lane, L, 1, &pre, 1, &kp, true, &lim0);
    k2_stage_and_pairs(d, L, lane, n_act, pre.s[2]);
    MSTAMP(e, 4);
    double term = 0.0;
    const int collision = k2_back<true>(d, e, lane, L, n_act, obs_out, &term, &lim0);
    pair_publish_lidar(d, e, lane, collision, term);
    MSTAMP(e, 5);
#ifdef AUV_STAMPS_MULTI
    if (lane == 0) stamp[(size_t)e * 16 + 1] = t_entry;
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // every store of this wave has been acknowledged
    MSTAMP(e, 3);
#endif
  } else if (role == 2) {
    // ---- Vessel.navigate of one environment: the nearest-point search ----
    const int el = bi;
    if (el >= ne) return;
    const int e = auv_uniform(d.e0 + el);
    MSTAMP(e, 6);
    const unsigned long long pk_first = __hip_atomic_load(d.k1_pkt + 8 * (size_t)e + (lane & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (step == 0) {
      ed = d.env_desc[e];
      pre.cnt = d.counters[e];
    } else if (carry_wait_wave(d, e, lane, tagmix_prev, ed, pre.cnt)) {
      return;
    }
#ifdef AUV_STAMPS_MULTI
    if (lane == 0) stamp[(size_t)e * 16 + 7] = t_entry;
#endif
    pre.ed = &ed;
    if (roles_wait_state(d, e, lane, pre, tagmix, true, pk_first)) return;
    MSTAMP(e, 8);
    NavNear nr;
    int* list = (int*)smem;
    const NavSpec sp = nav_bounds(d, e, lane, list, pre.s[0], pre.s[1], &ed);
    nr = nav_nearest(d, e, lane, list, pre.s[0], pre.s[1], sp);
    roles_publish_search(d, e, lane, nr, tagmix);
    MSTAMP(e, 9);
  } else {
    // ---- navigation tail + reward / done / auto-reset of eight environments ----
    const int f = bi;
    if (f >= nk) return;
    __builtin_amdgcn_s_setprio(2);
#ifdef AUV_STAMPS_MULTI
    const int fer = 8 * (8 * (f / 8) + lane / K1_GROUP) + (f % 8);
    const bool fst = fer < ne && lane % K1_GROUP == 0;
#endif
    MSTAMP_IF(fst, d.e0 + fer, 13);
    roles_finish_wave_multi<true>(d, f, lane, obs_out, reward_out, done_out, step, step == n_steps - 1, tagmix, tagmix_prev MSTAMP_ARG);
    MSTAMP_IF(fst, d.e0 + fer, 15);
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
    const int wr = auv_uniform(__shfl(w, src, AUV_WAVE)), ep = __shfl(cnt.z, src, AUV_WAVE);
    restore_env(d, er, auv_next_world_wave(d, wr, lane), lane, ep, obs_out);
  }
  // a restored environment's state was written by lane 0, its dynamics below read it from another lane
  if (any_reset) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  if (e < d.n) k1_env(d, e, actions, true);
}

// ---- load-time probe of what the in-launch hand-overs rely on ----------------------------------------------------
// The one-launch shape lets a wave poll for a word that a workgroup with a SMALLER index of the same
// launch stores.  That terminates if workgroups are dispatched in index order (a poller's producer is resident or
// done by the time the poller gets a slot) -- what gfx950 does, but HIP does not promise it.  k_probe_order has the
// step's structure without its arithmetic: `np` producers (a short wait, then an sc1 word each), behind them `nc`
// consumers that poll their producer's word and publish a word of their own, a third generation that polls the
// second's and publishes, and nc / 8 waves of a fourth that poll eight words of each earlier generation -- like
// dynamics / sweep / search / finish, every workgroup one wave with the step's LDS footprint, and several times more
// workgroups than the chip has slots.  Any poll that runs out (bounded: ~20 ms) counts a failure; the host then keeps
// the handle on the three-launch shape (auv_capi.hip: probe_dispatch_order).  `words`: np + 2 nc of them, zeroed.
__global__ void __launch_bounds__(AUV_WAVE) k_probe_order(unsigned int* __restrict__ words, int np, int nc, unsigned int tag,
                                                          unsigned int* __restrict__ failures) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int b = (int)blockIdx.x, lane = threadIdx.x;
  if (lane == 0) smem[0] = 1;   // (touch the allocation so that it is not optimised away)
  unsigned int* mine = words + b;
  const unsigned int* src = nullptr;
  if (b >= np + 2 * nc) {                                      // fourth generation: eight of each of the others
    const int q = b - np - 2 * nc;
    if (lane < 8) src = words + np + nc + 8 * q + lane;
    else if (lane < 16) src = words + np + 8 * q + (lane - 8);
    else if (lane < 24) src = words + (8 * q + (lane - 16)) % np;
  } else if (b >= np + nc) src = words + np + (b - np - nc);   // third generation: its own second-generation wave
  else if (b >= np) src = words + (b - np) % np;               // second generation: one of the producers
  if (b >= np) {
    int polls = 0;
    for (;;) {
      const bool there = !src || __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag;
      if (!__any(!there)) break;
      if (++polls == (1 << 14)) {
        if (lane == 0) atomicAdd(failures, 1u);
        return;
      }
      __builtin_amdgcn_s_sleep(32);
    }
  } else {
    __builtin_amdgcn_s_sleep(40), __builtin_amdgcn_s_sleep(40);   // ~2 us, like the dynamics' chain
  }
  if (lane == 0 && b < np + 2 * nc) __hip_atomic_store(mine, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// a wave that does nothing for `ticks` of the 100 MHz wall clock (auv_streams_overlap: do two streams run side by side?)
__global__ void k_spin(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// ---- rendezvous of the sub-batch chains with the caller's stream, on the device (auv_step_async / auv_step_wait,
// AUV_RDV_DEVICE) ----------------------------------------------------------------------------------------------------
// VecEnv.step_async / step_wait (scripts/run.py:293-296) order K chains on K streams against the stream the actions
// come from and the results go to.  With HIP events that is a record + K waits before the step and K records + K waits
// behind it, each hop a barrier packet the command processors resolve in ~8 us.  Here the hops are one-wave kernels
// and two words in device memory:
//   caller's stream   k_rdv_publish(ready, t)   behind whatever produced the actions: "the actions of step t are there"
//   chain i's stream  k_rdv_wait(ready, t)      in front of its launch of step t (holds ONE wave slot, not a launch's)
//                     k_rdv_arrive(done)        behind it: "chain i has finished a step"
//   caller's stream   k_rdv_wait(done, target)  in step_wait: all chains have arrived
// Visibility rides on the kernel boundaries: the producer's kernel has ended (release) before the publish kernel
// starts, the waiting kernel ends before the chain's step starts (acquire); likewise behind the step.  Every waiter
// waits for something that was SUBMITTED before it, so streams sharing a hardware queue (FIFO) cannot deadlock; the
// waits are bounded by the wall clock all the same and report through pair_error (4: actions, 5: chains).
__global__ void k_rdv_publish(unsigned long long* word, unsigned long long seq) {
  if (threadIdx.x == 0) __hip_atomic_store(word, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_rdv_arrive(unsigned long long* word) {
  if (threadIdx.x == 0) __hip_atomic_fetch_add(word, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_rdv_wait(const unsigned long long* word, unsigned long long target, int32_t* err, int code, unsigned long long limit_ticks,
                           int32_t* abort_flag, int32_t* report) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = wall_clock64();                    // 100 MHz
  int polls = 0;
  while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    if (wall_clock64() - t0 > limit_ticks) {
      // A chain's gate that runs out (abort_flag given: code 4) must not let its step run on actions that may not be there:
      // the device-wide abort flag goes up FIRST, so the launch behind this kernel -- and every launch queued behind it, on
      // any stream -- hands out ABORT packets and leaves its environments untouched (k_step_roles), exactly as behind a
      // hand-over time-out.  The code is published only if no other report is pending: a hand-over time-out (codes 1-3)
      // reported by an earlier launch must reach the host as such -- its recovery is the superset.
      if (abort_flag) __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (arbitrated on a DEVICE word, `report`: no read-modify-write on the mapped host word)
      if (__hip_atomic_exchange(report, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    // eager for the first few microseconds (a step's rendezvous), then a poll per ~2 us: a chain's gate may sit here for as long
    // as the caller's stream is busy with something else -- a PPO update between two rollouts -- and should not cost anything
    if (++polls < 256) __builtin_amdgcn_s_sleep(2);
    else __builtin_amdgcn_s_sleep(127);
  }
}

}  // namespace

void auv_launch_rdv_publish(unsigned long long* word, unsigned long long seq, hipStream_t st) {
  hipLaunchKernelGGL(k_rdv_publish, dim3(1), dim3(AUV_WAVE), 0, st, word, seq);
}
void auv_launch_rdv_arrive(unsigned long long* word, hipStream_t st) { hipLaunchKernelGGL(k_rdv_arrive, dim3(1), dim3(AUV_WAVE), 0, st, word); }
void auv_launch_rdv_wait(const unsigned long long* word, unsigned long long target, int32_t* err, int code, double limit_s, int32_t* abort_flag,
                         int32_t* report, hipStream_t st) {
  hipLaunchKernelGGL(k_rdv_wait, dim3(1), dim3(AUV_WAVE), 0, st, word, target, err, code, (unsigned long long)(limit_s * 1e8), abort_flag, report);
}

void auv_launch_spin(unsigned long long ticks, hipStream_t st) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(AUV_WAVE), 0, st, ticks); }

hipError_t auv_launch_probe(unsigned int* words, int np, int nc, unsigned int tag, unsigned int* failures, uint32_t lds, hipStream_t st) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k_probe_order, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_probe_order, dim3(np + 2 * nc + nc / 8), dim3(AUV_WAVE), lds, st, words, np, nc, tag, failures);
  return hipGetLastError();
}

void auv_launch_k31(const AuvDev& d0, const void* actions, int dtype, float* obs, float* reward, uint8_t* done, hipStream_t st) {
  AuvDev d = d0;
  d.act_f64 = dtype == AUV_F64;
  const dim3 grid((d.n + AUV_WAVE - 1) / AUV_WAVE), block(AUV_WAVE);
  hipLaunchKernelGGL(k31_reward_dyn, grid, block, 0, st, d, actions, obs, reward, done);
}

// the navigation role keeps its chunk list at the start of the wave's slice
bool auv_k23_ok(const AuvDev& d) { return NAV_SCRATCH_BYTES(d.nch_max) <= k2_slice_bytes(d); }

// K2 + K3-nav side by side: ONE wave per workgroup, so a wave slot is handed on the moment an environment's sweep
// ends -- the navigation workgroups queued behind the LiDAR ones start (and end) earlier.
void auv_launch_k23(const AuvDev& d, float* obs, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  const size_t lds = k2_slice_bytes(d);
  hipExtLaunchKernelGGL(k23_lidar_nav, dim3(2 * d.ne), dim3(AUV_WAVE), (uint32_t)lds, st, ev0, ev1, 0, d, obs);
}

// ---- the one-launch step (needs a LiDAR sweep: its word is what the finish wave finishes the step on) ----
bool auv_roles_ok(const AuvDev& d) { return auv_k23_ok(d) && d.cfg.use_lidar; }

void auv_launch_step_roles(const AuvDev& d0, const void* actions, int dtype, float* obs, float* reward, uint8_t* done,
                           hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  AuvDev d = d0;
  d.act_f64 = dtype == AUV_F64;
  const uint32_t lds = (uint32_t)k2_slice_bytes(d);
  const int nk = 8 * ((d.ne + 63) / 64), nb = 8 * ((d.ne + 7) / 8);
  const dim3 grid(nk + 2 * nb + AUV_HOOK_SKEW(d) + nk), block(AUV_WAVE);
  hipExtLaunchKernelGGL(k_step_roles, grid, block, lds, st, ev0, ev1, 0, d, actions, obs, reward, done);
}

// order: 0 step-major; 1 cohort-pipelined (lead / lag chosen here: as long as the slice has cohorts for them)
void auv_launch_step_multi(const AuvDev& d0, const void* actions, int dtype, float* obs, float* reward, uint8_t* done, int n_steps,
                           int first_slot, int n_slots, unsigned long long seq0, int order, int lead, int lag, hipStream_t st) {
  AuvDev d = d0;
  d.act_f64 = dtype == AUV_F64;
  d.ring_slots = 1;
  const uint32_t lds = (uint32_t)k2_slice_bytes(d);
  const int nk = 8 * ((d.ne + 63) / 64), nb = 8 * ((d.ne + 7) / 8);
  const int C = nk / 8;
  if (order == 1 && C >= 3 && d.ne % 64 == 0) {
    if (lead < 1) lead = 1;
    if (lag < 1) lag = 1;
    while (lead + lag > C - 1) {                          // every producer ahead of its consumer, also across steps
      if (lag > lead && lag > 1) lag--;
      else if (lead > 1) lead--;
      else lag--;
    }
    const dim3 grid((unsigned)(n_steps * C + lead + lag) * 144u), block(AUV_WAVE);
    const unsigned magic = (unsigned)((0x100000000ull + (unsigned)C - 1) / (unsigned)C);
    hipLaunchKernelGGL(k_step_multi, grid, block, lds, st, d, actions, obs, reward, done, n_steps, first_slot, n_slots, seq0, lead, lag, magic);
    return;
  }
  const dim3 grid((unsigned)n_steps * (unsigned)(2 * nk + 2 * nb)), block(AUV_WAVE);
  hipLaunchKernelGGL(k_step_multi, grid, block, lds, st, d, actions, obs, reward, done, n_steps, first_slot, n_slots, seq0, -1, 0, 0u);
}

// The stage's capacity for a bank (S, k_max, m_max known): the largest that gives the one-launch step its best occupancy (see
// k2_lidar.hip) and still leaves room for the search role's chunk list; asked of the runtime, not computed (allocation granule).
int auv_pick_seg_cap(const AuvDev& d0) {
  AuvDev d = d0;
  int best = K2_SEG_CAP, best_occ = -1;
  for (int cap = K2_SEG_CAP; cap >= K2_SEG_CAP_MIN; cap -= 2) {
    d.seg_cap = cap;
    const size_t b = k2_slice_bytes(d);
    if (b < NAV_SCRATCH_BYTES(d.nch_max)) break;
    int occ = 0;
    if (b > 64 * 1024 || hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_step_roles, AUV_WAVE, b) != hipSuccess) occ = 0;
    if (occ > best_occ) best_occ = occ, best = cap;
  }
  return best;
}

uint32_t auv_step_lds_bytes(const AuvDev& d) { return (uint32_t)k2_slice_bytes(d); }

hipError_t auv_step_fused_prepare(const AuvDev& d) {
  const size_t b = k2_slice_bytes(d);
  if (b <= 64 * 1024) return hipSuccess;
  hipError_t e = hipFuncSetAttribute((const void*)k23_lidar_nav, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)k_step_multi, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)k_step_roles, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
}
