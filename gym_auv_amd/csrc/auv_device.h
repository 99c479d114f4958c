// auv_device.h — device-side data layout and small fp64 helpers shared by the three kernels.
// gfx950 (MI355X) only.  All arithmetic is IEEE fp64 with contraction off (Makefile:
// -ffp-contract=off) so that results agree with the CPU oracle / the fp64 NumPy reference to
// ~1e-12 and the integer flags (collision, reached_goal, done) are bit-exact in practice.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/auv_hip.h"

#define AUV_PI 3.141592653589793
#define AUV_WAVE 64
#define AUV_BLOCK 256          // K2/K3: one wave per environment, 4 environments per workgroup
#define AUV_ENVS_PER_BLOCK (AUV_BLOCK / AUV_WAVE)
#define AUV_MOVER_NSEG 5
#define AUV_CHUNK 64           // polyline segments per bounding-circle chunk (= one wave pass)
#define AUV_FRESH_GRID 256     // workgroups of the reset-pass kernels (they loop over the fresh list)

// Where the tables of an environment's current world start and how long they are: one 64-byte
// record per environment, rewritten whenever the environment is bound to a world, so the step
// kernels reach their tables with one load instead of world_idx -> per-world offsets -> table.
struct __attribute__((aligned(16))) EnvDesc {
  long long k0, m0, p0, c0, kn0;   // first obstacle / mover / polyline vertex / chunk / knot
  int K, M, P, nch, nk, w;         // counts, and the world index itself
};

// Everything the kernels read, by value in the kernel argument buffer (no constant memory,
// no host round trip; hipGraph-capturable).
struct AuvDev {
  auv_config_t cfg;
  int32_t n;        // environments
  int32_t e0, ne;   // the slice [e0, e0 + ne) of environments a step launch covers (the whole batch: 0, n) -- sub-batches
                    // of one handle stepped on different streams (auv_step_slice) overlap each other's head and tail
  int32_t act_f64;  // the action buffer of this launch holds doubles (else floats)
  int32_t n_worlds;
  int32_t k_max, m_max;
  // ---- world bank (HBM, read-only during step).  Every table is addressed by a start offset
  // and a count per world, so the same kernels serve a packed CSR upload (auv_load_worlds) and
  // the fixed-capacity slots that on-device generation fills (auv_generate_worlds) ----
  const int32_t* poly_cnt;     // [W] vertices of the dense polyline
  const int32_t* chunk_cnt;    // [W] bounding-circle chunks
  const int32_t* knot_cnt;     // [W] PCHIP knots
  const int32_t* obs_cnt;      // [W] obstacles
  const int32_t* mv_cnt;       // [W] movers
  const int32_t* mv_vtab_len;  // [sum M] velocity-table entries per mover
  const int64_t* poly_off;
  const double2* poly_xy;
  const double* poly_cum;
  const int64_t* chunk_off;    // [W+1] offsets into chunk_bound (derived at load time)
  const double4* chunk_bound;  // cx, cy, inflated radius, - : circle around AUV_CHUNK segments
  int32_t nch_max;             // max chunks of any world
  int32_t seg_cap;             // boundary segments the LiDAR wave stages per batch (<= AUV_SEG_CAP_MAX; picked per bank: k2_lidar.hip)
  const int64_t* knot_off;
  const double* knot_s;
  const double* knot_coef;     // [.][8]
  const double* world_scalar;  // [W][8]
  const int64_t* obs_off;
  const int4* obs_meta;        // kind, seg_off, nseg, mover idx
  const double* obs_cull;      // [.][3]
  const double4* seg;          // ax, ay, bx, by
  const int64_t* mv_off;
  const double4* mv_param;     // width, pos0x, pos0y, n_vel
  const double4* mv_init;      // px, py, heading, counter
  const int64_t* mv_vtab_off;
  const double2* mv_vtab;
  // ---- environment state (HBM, SoA where per-env scalars) ----
  double* state;       // [6][N]
  int32_t* world_idx;  // [N]
  EnvDesc* env_desc;   // [N]  table starts / counts of the bound world (kept in step with world_idx)
  int4* counters;      // [N] t_step, step_counter, episodes, fresh-flag
  double* lidar_d;     // [N][S]
  double* obs64;       // [N][6+S]
  double* reward64;    // [N]
  double* info64;      // [N][8]
  double* nav64;       // [N][8]
  double4* mover;      // [N][Mmax]
  uint8_t* nearby;     // [N][Kmax]
  double* episode;     // [N][4]
  int2* limits;        // [N][Kmax]
  uint8_t* collision;  // [N]
  double* ep_log;      // [ep_log_cap][8] finished episodes in completion order (reward phase; auv_episode_log)
  unsigned long long* ep_log_count;  // [1] episodes logged so far, 64-bit (the ring position is count & (cap - 1))
  int32_t ep_log_cap;  // a power of two
  double* step_info;   // [N][4] info of the last step (terminal values survive an auto-reset)
  unsigned long long* pair_word; // [N] one-launch step: what the LiDAR wave leaves for the finish wave (k_step_fused.hip)
  int32_t* pair_error; // [4] mapped HOST memory: [0] set when a wave gave up polling for a hand-over (1 sweep's word, 2 state packet, 3 packet /
                       //     search record; 4 / 5: a rendezvous kernel of auv_step_async / _wait), [1] / [2] e0 / ne of the launch that reported
  int32_t* abort_flag; // [1] device: raised with pair_error; from then on the dynamics role of every launch hands out ABORT packets, i.e. launches
                       //     queued behind a time-out do nothing (and leave every environment they cover consistent) until the host has recovered
  uint8_t* broken;     // [N] environments whose step a wave that gave up has left unfinished: what the recovery resets
  unsigned long long* k1_pkt;  // [N][8] one-launch step: the state the dynamics role hands to the other two (k_step_roles)
  unsigned long long* nav_hand; // [N][8] one-launch step: the nearest path segment the navigation role's search hands to the finish role
  unsigned long long* carry;   // [N][24] launches of several steps (k_step_multi): what a step's finish wave hands the next step's roles
  int32_t* k1_done;            // [1] one-launch step in a captured graph: dynamics waves that have read the ring position
  int32_t cut_lidar, cut_nav;  // diagnostic build only (-DAUV_CUTS, tools/valu_budget.py): phases from this number on are skipped
  int32_t pair_skew;   // one-launch step, test-hook build only: idle workgroups between the roles (an environment's waves on different XCDs)
  int32_t pair_fault;  // one-launch step, test-hook build only: the launch's first sweep never publishes its word (the poll must run out)
  const struct AuvDev* self;  // this struct in device memory (as of the last bank load): the one-launch step's restore path
                              // reads its ~25 table pointers through it at the point of use -- as kernel arguments
                              // they would all be fetched (and spilled) at the entry of every wave of both roles
  int32_t* fresh_count; // [1]  } work list of the load-time pass that computes the reset rows
  int32_t* fresh_list;  // [N]  }
  // ---- per-world reset rows (derived once at load time by running the reset observation of
  //      every world through K2/K3): what reset() / auto-reset copy instead of recomputing ----
  double* w_obs64;     // [W][6+S]
  double* w_lidar;     // [W][S]
  double* w_info;      // [W][8]
  double* w_nav;       // [W][8]
  uint8_t* w_nearby;   // [W][Kmax]
  int2* w_limits;      // [W][Kmax]
  uint8_t* w_collision;// [W]
  int32_t w_ready;     // 0 while the rows are being computed
  double* rew_path;     // [N] path-following term of the reward, left by the navigation phase
  double* rew_lidar;    // [N] LiDAR term of the Colav reward, left by K2
  const double2* beam_cs; // [S] cos, sin of the body-frame beam angles -pi + (i + 1) 2 pi / S (vessel.py:66-68)
  double* beam_w;         // [S] gamma_theta weight of each beam, 1 / (1 + |10 angle|) (rewarder.py:205-222)
  double* derived;        // [8] per-config constants formed once on the device: log(1 + R), R exp(-0.1 R),
                          //     sum of beam_w (in the wave-reduction order), -
  int32_t* ring_pos;    // [1]  current slot of the action ring (advanced once per step by K3)
  int32_t ring_slots;   // 1 = plain action buffer
  int32_t ring_slot_host; // -1: read ring_pos and advance it; -2: read it, another kernel of the step advances it; >= 0: the host names the slot
  unsigned long long* stamps;  // [N][16] per-env phase cycle counts (diagnostic builds, -DAUV_STAMPS)
  // ---- a fresh world on every reset (auv_fresh_worlds_create; SURVEY 8(f) F1: generation joined to auto-reset) ----
  // The bank is D slots per environment (slot e + j N belongs to environment e); an environment whose episode ends moves
  // to its next slot IF that slot holds a world nobody has seen (READY), marks the slot it leaves STALE and queues it; a
  // refill pass on a side stream (auv_capi.hip: fw_refill) pops the queue, rebuilds exactly those slots -- tables
  // (k5_generate) and reset rows (the step's own kernels on a few shadow environments) -- and a publish kernel ON THE
  // ENVIRONMENT'S OWN STREAM, enqueued after the host has seen the pass complete, flips them to READY: every step launch that
  // can bind a regenerated slot started after the slot's tables were complete (kernel-boundary visibility, no fences).
  int32_t* fw_state;     // [W] AUV_FW_READY / _IN_USE / _STALE, or nullptr: the bank cycles (w + N) % W as before
  int32_t* fw_serial;    // [W] which world of its environment the slot holds: the world of (seed, environment, serial)
  int32_t* fw_queue;     // [fw_cap] ring of stale slots (-1: empty entry); a slot is in it at most once
  unsigned int* fw_ctl;  // [8] [0] tail (producers: finish waves of any chain), [1] head (the refill pass), [2] episodes that had to
                         //     start in the world they had just finished because the next slot was not READY yet (stale re-use)
  int32_t fw_cap;
};

enum { AUV_FW_READY = 0, AUV_FW_IN_USE = 1, AUV_FW_STALE = 2 };

// The world an environment whose episode has just ended is bound to next (one lane calls this; environment.py:176-218 reset ->
// _generate: the reference builds a new scenario on every reset).  Bank cycling: (w + N) % W.  Fresh worlds: the same slot
// arithmetic, but only onto a READY slot, and the slot left behind is queued for regeneration; if the next slot is not READY
// (the refill pass has fallen behind an episode that lasted only a few steps) the environment starts over in the world it has
// just finished and the event is counted -- never silently.
__device__ __forceinline__ int auv_next_world(const AuvDev& d, const int w) {
  const int w2 = (int)(((long long)w + d.n) % d.n_worlds);
  if (!d.fw_state) return w2;
  if (w2 == w) return w;
  if (__hip_atomic_load(d.fw_state + w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != AUV_FW_READY) {
    __hip_atomic_fetch_add(d.fw_ctl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return w;
  }
  __hip_atomic_store(d.fw_state + w2, AUV_FW_IN_USE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(d.fw_state + w, AUV_FW_STALE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned int at = __hip_atomic_fetch_add(d.fw_ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(d.fw_queue + (at % (unsigned int)d.fw_cap), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return w2;
}
// the same for a whole wave with a wave-uniform `w`: lane 0 takes the decision (and the queue entry), every lane gets it
__device__ __forceinline__ int auv_next_world_wave(const AuvDev& d, const int w, const int lane) {
  if (!d.fw_state) return (int)(((long long)w + d.n) % d.n_worlds);
  int w2 = 0;
  if (lane == 0) w2 = auv_next_world(d, w);
  return __builtin_amdgcn_readfirstlane(w2);
}

// State handed from one phase to the next inside a launch (registers instead of
// a global-memory round trip): the advanced vessel state and the env's counters.
struct EnvPre {
  double s[6];     // x, y, psi, u, v, r after Vessel.step
  int4 cnt;        // t_step, vessel step counter (already incremented), episodes, -
  const EnvDesc* ed = nullptr;   // the environment's descriptor where the caller has fetched it already
};

// Cumulative phase cuts (diagnostic build only, -DAUV_CUTS): AUV_RUN_L(d, n) is false when the LiDAR role's phases
// from n on are switched off (cut_lidar = n), likewise AUV_RUN_N for the navigation role.  Differences of the SQ
// counters between consecutive cut levels are the phases' dynamic instruction budgets (tools/valu_budget.py).
// The product build compiles them to `true`.
#ifdef AUV_CUTS
#define AUV_RUN_L(d, n) ((d).cut_lidar == 0 || (d).cut_lidar > (n))
#define AUV_RUN_N(d, n) ((d).cut_nav == 0 || (d).cut_nav > (n))
#else
#define AUV_RUN_L(d, n) true
#define AUV_RUN_N(d, n) true
#endif

// words of `stamps` per environment (-DAUV_STAMPS_MULTI: a second half, where the steps of a multi-step launch that are not looked at put theirs)
#ifdef AUV_STAMPS_MULTI
#define AUV_STAMP_WORDS 32
#else
#define AUV_STAMP_WORDS 16
#endif
// In-kernel phase stamps (diagnostic build only: make STAMPS=1).  The stamp values leave the
// kernel through `stamps` alone; no output is computed from them.
#ifdef AUV_STAMPS
#define AUV_STAMP_DECL unsigned long long _st[9]; int _si = 0; _st[_si++] = clock64();
#define AUV_STAMP() _st[_si++] = clock64();
#define AUV_STAMP_FLUSH(e, base)                                                        \
  if (lane == 0) {                                                                      \
    for (int _k = 1; _k < _si; _k++) d.stamps[(size_t)(e) * 16 + (base) + _k - 1] = _st[_k] - _st[_k - 1]; \
  }
#else
#define AUV_STAMP_DECL
#define AUV_STAMP()
#define AUV_STAMP_FLUSH(e, base)
#endif

// One wave works on one environment: its index is the same in every lane.  Saying so lets the
// compiler keep everything derived from it (table offsets, counts, the pose) in scalar registers
// and fetch it through the scalar cache.
__device__ __forceinline__ int auv_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// COH: every load agent-scope (sc1) -- the tables of a slot that a refill pass on another stream may have rebuilt while this
// launch was already running (fresh-world mode; see restore_env)
template <bool COH = false> __device__ __forceinline__ EnvDesc auv_make_desc(const AuvDev& d, int w) {
  EnvDesc ed;
  if constexpr (COH) {
    ed.k0 = d.obs_off[w], ed.m0 = d.mv_off[w], ed.p0 = d.poly_off[w], ed.c0 = d.chunk_off[w], ed.kn0 = d.knot_off[w];   // (slot offsets never change)
    ed.K = __hip_atomic_load(d.obs_cnt + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), ed.M = __hip_atomic_load(d.mv_cnt + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ed.P = __hip_atomic_load(d.poly_cnt + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), ed.nch = __hip_atomic_load(d.chunk_cnt + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ed.nk = __hip_atomic_load(d.knot_cnt + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    ed.k0 = d.obs_off[w], ed.m0 = d.mv_off[w], ed.p0 = d.poly_off[w], ed.c0 = d.chunk_off[w], ed.kn0 = d.knot_off[w];
    ed.K = d.obs_cnt[w], ed.M = d.mv_cnt[w], ed.P = d.poly_cnt[w], ed.nch = d.chunk_cnt[w], ed.nk = d.knot_cnt[w];
  }
  ed.w = w;
  return ed;
}

__device__ __forceinline__ double auv_princip(double a) {
  // ((a + pi) % (2 pi)) - pi with Python's sign convention (utils/geomutils.py:4-5).
  // x = a + pi almost always lies within one period of [0, 2 pi); there fmod is the identity
  // or one exact subtraction (Sterbenz), so the short cuts below are bit-identical to
  // fmod(x, 2 pi) followed by the sign fix-up, without the slow fp64 fmod.
  const double TWO_PI = 2.0 * AUV_PI;
  const double x = a + AUV_PI;
  double m;
  if (x >= 0.0 && x < TWO_PI) m = x;
  else if (x >= TWO_PI && x < 2.0 * TWO_PI) m = x - TWO_PI;
  else if (x < 0.0 && x > -TWO_PI) m = x + TWO_PI;
  else {
    m = fmod(x, TWO_PI);
    if (m < 0.0) m += TWO_PI;
  }
  return m - AUV_PI;
}

__device__ __forceinline__ double auv_clip(double x, double lo, double hi) {
  return x < lo ? lo : (x > hi ? hi : x);
}

// JTS/GEOS Distance::pointToSegment (what Point.distance / LineString.project evaluate):
//     r = (p - a).(b - a) / |b - a|^2;  r <= 0: |p - a|;  r >= 1: |p - b|;  else |cross(a - p, b - a) / |b - a|^2| * |b - a|
// with the same operations on the same operands, hence the same bits, in the order a wave executes cheaply: the two
// comparisons of r are taken on its numerator (r <= 0 <=> dot <= 0 and r >= 1 <=> dot >= len2 hold exactly for a
// correctly rounded quotient with len2 > 0: the quotient of two doubles rounds to 1 or above only if it is at least 1),
// which removes a division, and the three cases share ONE square root (of |p - a|^2, |p - b|^2 or |b - a|^2) instead of each
// lane group running its own when the lanes of a wave disagree -- they always do along a chunk of the path.  ~70
// instead of ~140 fp64 instructions per call.
__device__ __forceinline__ double auv_pt_seg_dist(double px, double py, double ax, double ay, double bx,
                                                  double by) {
  const double dxa = px - ax, dya = py - ay;
  const double ex = bx - ax, ey = by - ay;
  const double len2 = ex * ex + ey * ey;
  const double dot = dxa * ex + dya * ey;
  const bool at_a = (ax == bx && ay == by) || dot <= 0.0;
  const bool at_b = !at_a && dot >= len2;
  const bool interior = !(at_a || at_b);
  const double dxb = px - bx, dyb = py - by;
  const double da2 = dxa * dxa + dya * dya, db2 = dxb * dxb + dyb * dyb;
  const double root = sqrt(interior ? len2 : (at_a ? da2 : db2));
  double s = 1.0;
  if (interior) s = fabs(((ay - py) * ex - (ax - px) * ey) / len2);
  return interior ? s * root : root;
}

// Python floor-mod for ints (sensor.py:93: idx_max_ray % n_rays)
__device__ __forceinline__ int auv_pymod(int a, int s) {
  int r = a % s;
  return r < 0 ? r + s : r;
}

#define AUV_PAIR_EMPTY 0x7ff8dead00000001ull       // one-launch step: "no word yet" (a NaN payload no arithmetic produces)
#define AUV_PAIR_COLLISION 0x7ff8dead00000002ull   //              "the sweep found a collision"

// Stores and loads that are coherent over the whole device one by one (relaxed agent-scope atomics: the
// `sc1` forms, written through / read past the XCD's L2, which is not coherent with the other seven).  WT = false:
// plain accesses.  Used by the one-launch step, where two waves on possibly different XCDs hand rows to each other
// inside one launch (k_step_fused.hip: pair_finish).
template <bool WT> __device__ __forceinline__ void auv_st(double* p, const double v) {
  if constexpr (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
template <bool WT> __device__ __forceinline__ void auv_st(float* p, const float v) {
  if constexpr (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
template <bool WT> __device__ __forceinline__ void auv_st(uint8_t* p, const uint8_t v) {
  if constexpr (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
template <bool WT> __device__ __forceinline__ void auv_st(int2* p, const int2 v) {
  if constexpr (WT) {
    __hip_atomic_store((unsigned long long*)p, ((unsigned long long)(unsigned)v.y << 32) | (unsigned)v.x, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  } else {
    *p = v;
  }
}
template <bool WT> __device__ __forceinline__ void auv_st(double2* p, const double2 v) {
  if constexpr (WT) {
    __hip_atomic_store(&p->x, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&p->y, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    *p = v;
  }
}
template <bool WT> __device__ __forceinline__ void auv_st(double4* p, const double4 v) {
  if constexpr (WT) {
    __hip_atomic_store(&p->x, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&p->y, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&p->z, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&p->w, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    *p = v;
  }
}
template <bool WT, typename T> __device__ __forceinline__ T auv_ld(const T* p) {
  if constexpr (WT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else return *p;
}
template <bool WT> __device__ __forceinline__ double4 auv_ld4(const double4* p) {
  if constexpr (WT)
    return make_double4(__hip_atomic_load(&p->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(&p->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                        __hip_atomic_load(&p->z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(&p->w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  else return *p;
}
template <bool WT> __device__ __forceinline__ int2 auv_ld2i(const int2* p) {
  if constexpr (WT) {
    const unsigned long long v = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_int2((int)(unsigned)v, (int)(unsigned)(v >> 32));
  } else return *p;
}
// every global store this wave has issued so far is complete (for the sc1 forms: visible to the whole device)
__device__ __forceinline__ void auv_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// LDS hand-off between lanes of the SAME wave: a wave's LDS operations execute in order, so
// only outstanding operations must be waited for and the compiler kept from reordering.
__device__ __forceinline__ void auv_wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// wave64 reductions.  Inside a row of 16 lanes the partner comes by DPP (a VALU move, no trip through the LDS
// crossbar: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror -- each pairs lanes that hold different
// partial results, and a + b == b + a bit for bit, so all 16 lanes end with the same value); the four row results are
// read into scalar registers and combined in one fixed order, so every lane returns the same bits.
template <int CTRL> __device__ __forceinline__ double auv_dpp_f64(const double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double auv_readlane_f64(const double v, const int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
// inclusive prefix sum over the wave's 64 lanes, all by DPP: four shifted adds inside each row of 16, then the last lane
// of row 0 / 2 onto rows 1 / 3 (row_bcast:15) and lane 31 onto rows 2 and 3 (row_bcast:31).  Integer adds: any order.
__device__ __forceinline__ int auv_wave_scan_incl(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);    // row_shr:1 (lanes shifted in from outside the row: 0)
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);    // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);    // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);    // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ int auv_wave_last(const int v) { return __builtin_amdgcn_readlane(v, AUV_WAVE - 1); }
// the same for doubles (k5_generate's block-wide prefix sums): lanes shifted in from outside a row contribute +0.0
template <int CTRL, int ROWS, bool BOUND> __device__ __forceinline__ double auv_dpp_f64_rows(const double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xF, BOUND);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWS, 0xF, BOUND);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double auv_wave_scan_incl_f64(double v) {
  v += auv_dpp_f64_rows<0x111, 0xF, true>(v);    // row_shr:1
  v += auv_dpp_f64_rows<0x112, 0xF, true>(v);    // row_shr:2
  v += auv_dpp_f64_rows<0x114, 0xF, true>(v);    // row_shr:4
  v += auv_dpp_f64_rows<0x118, 0xF, true>(v);    // row_shr:8
  v += auv_dpp_f64_rows<0x142, 0xA, false>(v);   // row_bcast:15 -> rows 1, 3
  v += auv_dpp_f64_rows<0x143, 0xC, false>(v);   // row_bcast:31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ double auv_wave_sum(double v) {
  v += auv_dpp_f64<0xB1>(v);
  v += auv_dpp_f64<0x4E>(v);
  v += auv_dpp_f64<0x141>(v);
  v += auv_dpp_f64<0x140>(v);
  return (auv_readlane_f64(v, 0) + auv_readlane_f64(v, 16)) + (auv_readlane_f64(v, 32) + auv_readlane_f64(v, 48));
}
__device__ __forceinline__ double auv_wave_min(double v) {
  v = fmin(v, auv_dpp_f64<0xB1>(v));
  v = fmin(v, auv_dpp_f64<0x4E>(v));
  v = fmin(v, auv_dpp_f64<0x141>(v));
  v = fmin(v, auv_dpp_f64<0x140>(v));
  return fmin(fmin(auv_readlane_f64(v, 0), auv_readlane_f64(v, 16)), fmin(auv_readlane_f64(v, 32), auv_readlane_f64(v, 48)));
}
