// K2 — moving-obstacle update + LiDAR sweep: ONE WAVE (64 lanes) PER ENVIRONMENT, no workgroup
// barriers (everything is wave-synchronous).  The step path runs it as the first role of the
// one-wave workgroups of k23_lidar_nav (k_step_fused.hip); k2_lidar below is the stand-alone
// kernel of the per-kernel API (four environments per 256-thread workgroup).
//
// Reference: BaseEnvironment._update              gym_auv/environment.py:386-392
//            VesselObstacle._update / boundary     objects/obstacles.py:195-233
//            Vessel.perceive                       objects/vessel/vessel.py:249-368
//            find_rays_to_simulate_for_obstacles   objects/vessel/sensor.py:74-97
//            _find_limit_angle_rays                sensor.py:41-71
//            simulate_sensor                       sensor.py:140-159
//            ColavRewarder (LiDAR term)            objects/rewarder.py:205-222
//
// Work decomposition inside the wave:
//   phase A  lanes <-> movers: advance kinematics, rebuild the 5 pentagon segments + cull
//            circle in the wave's LDS slice.
//   phase C  lanes <-> rays (S/64 passes): ray vectors into LDS -- one sincos(psi) per environment,
//            the beam angles' cos/sin come from a per-config table (addition theorem).
//   phase B0 (every 25th vessel step) lanes <-> obstacles: refresh of the cached nearby mask.
//   phase B  lanes <-> obstacles: cull window [i_min-1, i_max % S) with the reference's
//            Python-range / negative-index semantics; obstacles with a non-empty window are
//            compacted (ballot + popcount).
//   phase S  the boundary segments of the surviving obstacles are looked at in flattened,
//            coalesced passes; the FRONT-FACING ones (a ray from outside a simple closed boundary
//            first meets an edge that faces p0) are compacted into LDS (vessel-relative: a - p0,
//            b - a), together with the point-in-polygon predicates of filled obstacles (LDS
//            xor/or per obstacle, all edges) and the conservative range of ray indices each staged
//            segment can possibly be hit by (its angular span seen from p0, fp32 atan2, widened
//            by two rays on both sides).
//   phase D  work items = (staged segment, run of <= K2_ITEM_RAYS (6) rays of its span), lanes <-> items: the rays
//            that also lie in the obstacle's window (the reference's culling decides visibility;
//            the span only skips pairs that cannot intersect) get the exact fp64 ray/segment
//            test, a hit does an LDS atomic-min on the ray's t (non-negative fp64 as uint64).
//   phase E  lanes <-> rays: distance from the min t, closeness (fp64 + the float32 observation
//            columns), ballot -> collision, and the LiDAR term of the Colav reward.
// Two rules the phases keep (round 4): (1) a wave's global loads and stores count down one counter in issue order, so a
// phase requests everything it reads before its first store (k2_back stores phase B's limit rows for that reason); (2) lane
// exchanges that are not data-dependent go by DPP (auv_wave_scan_incl, auv_wave_sum), not through the LDS crossbar.
// Work per environment is ~(rays subtended by the nearby front-facing boundaries), ~100 pair
// tests instead of S x G = 99 k, and does not depend on how wide the reference's windows are.
// Roofline: HBM.  Algorithmic bytes per env-step (fp64 layout): 32*G (segments, G per env)
// + 24*K (cull circles) + 16*K (meta) + 24 (pose) + 16*S (d + closeness out) + K (nearby) + 4*S
// (float32 closeness).
#include "auv_device.h"

namespace {

#define K2_KIND_MASK 0xff
#define K2_PIP_FLAG 0x100   // p0 lies within the enclosing circle: the point-in-polygon predicates are needed
struct ObsLds {        // per-obstacle scratch in LDS (24 B)
  int kind;            // AUV_OBS_* | K2_PIP_FLAG
  int seg_off;         // absolute index into seg[] (static) or mover slot*5 (mover)
  int nseg;
  int start;           // first ray index of the window (may be negative, > -2S)
  int count;           // number of rays in the window (0 = culled / not nearby)
  int wind;            // +1 / -1: simple ring, counter-clockwise / clockwise (back faces skipped); 0: keep all edges
};

struct EnvHdr {        // head of each wave's LDS slice: what the pair sweep needs to know
  double px, py;
  double cpsi, spsi;   // cos / sin of the heading, parked here between their (early) evaluation and phase C
  int n_act;           // obstacles with a non-empty ray window
  int pad[3];
};

__device__ __forceinline__ unsigned long long d2u(double x) { return (unsigned long long)__double_as_longlong(x); }
__device__ __forceinline__ double u2d(unsigned long long x) { return __longlong_as_double((long long)x); }

__device__ __forceinline__ int wrap_ray(int i, int S) {   // Python list index for i in (-2S, 2S)
  if (i < 0) i += S;
  if (i < 0) i += S;
  if (i >= S) i -= S;
  return i;
}

// The five boundary segments of a mover (obstacles.py:217-233) formed on demand from its pose, with
// the arithmetic phase A used to form them: LDS keeps 40 bytes per mover instead of 160.
struct MoverSegs {
  double c, s, x, y, wd;    // snapped cos / sin of the heading, position, width
  __device__ __forceinline__ void vertex(int k, double& vx, double& vy) const {
    const double bx = (k <= 1) ? -wd / 2 : (k == 3 ? 3.0 / 2 * wd : wd / 2);
    const double by = (k == 0 || k == 4) ? -wd / 2 : (k == 3 ? 0.0 : wd / 2);
    const double x0 = 5.0 * wd / 18.0;
    const double xo = x0 - x0 * c, yo = 0.0 - x0 * s;
    vx = (c * bx + -s * by + xo) + x;
    vy = (s * bx + c * by + yo) + y;
  }
  __device__ __forceinline__ double4 operator[](int i) const {
    double ax, ay, bx, by;
    vertex(i, ax, ay);
    vertex(i == 4 ? 0 : i + 1, bx, by);
    return make_double4(ax, ay, bx, by);
  }
};

#ifndef K2_SEG_CAP
#define K2_SEG_CAP 96    // the LARGEST stage: segments per wave and batch (180 beams, 50 obstacles: slice <= 10 KiB -> 16 waves per CU)
#endif
#define K2_SEG_CAP_MIN 32
#ifndef K2_HIT_S
#define K2_HIT_S 256     // k2_back: most beams whose (index, weight) lists are kept in LDS (Slice::hit_list; else every pass does everything)
#endif
// A handle's stage holds d.seg_cap segments, K2_SEG_CAP_MIN <= seg_cap <= K2_SEG_CAP, picked per bank (auv_pick_seg_cap): the
// largest that leaves the one-launch step its best occupancy.  EVERY role of that launch is charged the sweep's slice, so at 256
// beams + 47 obstacles + 17 movers a 96-segment stage (12.7 KB) means 12 waves per CU for all of them; a 34-segment one 16:
// 140.5 -> 162-165 M env-steps/s at 8192 x 256 (profiles/r05/ab_seg_cap_8192x256.jsonl: compile-time 96 / 64 / 32).  Results do not depend on it (a crowded
// environment's sweep takes more batches; every beam keeps the minimum over all of them).
#ifndef K2_RAW_CAP
#define K2_RAW_CAP 192   // boundary segments looked at per batch; only the front-facing ones are staged
#endif
#ifndef K2_ITEM_RAYS
#define K2_ITEM_RAYS 6    // rays per work item of the pair sweep (4, 5, 6, 8, 12 measured: profiles/r04/ab_item_rays.jsonl)
#endif

// per-wave LDS slice (decreasing alignment):
//   EnvHdr | [Mmax] double4 mover pose (cos, sin, x, y) | [CAP] double4 staged (wx,wy,sx,sy) |
//   [S] double2 ray vectors | [Mmax] double2 mover cull centre | [S] u64 min-t bits | [Mmax] double
//   mover width | [Kmax] ObsLds | [CAP] short2 ray span | [Kmax] int active list | [Kmax+1] int
//   segment prefix | [Kmax] int inside flags | [CAP] u16 owner | [CAP+1] u16 work-item prefix
__host__ __device__ __forceinline__ size_t k2_slice_bytes(int S, int k_max, int m_max, int cap) {
  size_t b = sizeof(EnvHdr) + (size_t)m_max * 32 + (size_t)cap * 32 + (size_t)S * 16 + (size_t)m_max * 16 +
             (size_t)S * 8 + (size_t)m_max * 8 + (size_t)k_max * sizeof(ObsLds) + (size_t)cap * 4 +
             (size_t)k_max * 4 + (size_t)(k_max + 1) * 4 + (size_t)k_max * 4 + (size_t)cap * 2 +
             (size_t)(cap + 2) * 2;
  return (b + 15) & ~(size_t)15;
}
__host__ __device__ __forceinline__ size_t k2_slice_bytes(const AuvDev& d) { return k2_slice_bytes(d.cfg.n_sensors, d.k_max, d.m_max, d.seg_cap); }

struct Slice {
  EnvHdr* hdr;
  double4* mvrot;
  double4* stage;
  double2* rayv;
  double2* mvcull;
  unsigned long long* dbits;
  double* mvw;
  ObsLds* obs;
  short2* span;
  int* act;
  int* sbase;
  int* par;
  unsigned short* owner;
  unsigned short* ioff;
  int cap;        // segments the stage holds (d.seg_cap)
  int* hits;      // k2_back: the returns' beam indices and weights, in regions that are idle by then (nullptr: they do not fit --
  double* hitw;   // every pass of k2_back then does everything)
};

__device__ __forceinline__ MoverSegs mover_segs(const double4 rot, const double wd) {
  MoverSegs ms;
  ms.c = rot.x, ms.s = rot.y, ms.x = rot.z, ms.y = rot.w, ms.wd = wd;
  return ms;
}

__device__ __forceinline__ Slice carve(unsigned char* p, int S, int k_max, int m_max, int cap) {
  Slice s;
  s.hdr = (EnvHdr*)p;
  s.mvrot = (double4*)(s.hdr + 1);
  s.stage = s.mvrot + m_max;
  s.rayv = (double2*)(s.stage + cap);
  s.mvcull = s.rayv + S;
  s.dbits = (unsigned long long*)(s.mvcull + m_max);
  s.mvw = (double*)(s.dbits + S);
  s.obs = (ObsLds*)(s.mvw + m_max);
  s.span = (short2*)(s.obs + k_max);
  s.act = (int*)(s.span + cap);
  s.sbase = s.act + k_max;
  s.par = s.sbase + k_max + 1;
  s.owner = (unsigned short*)(s.par + k_max);
  s.ioff = s.owner + cap;
  s.cap = cap;
  // k2_back's lists, 4 B of beam index + 8 B of weight per beam, in what is idle once the pair sweep is through: both in
  // R1 = [mvrot, rayv) = the movers' rotations + the stage if they fit there, else the indices in R1 and the weights in
  // R2 = [mvw, end of the slice) = everything behind the beams' distance words
  const size_t r1 = 32 * (size_t)(m_max + cap), r2 = (size_t)((unsigned char*)(s.ioff + cap + 2) - (unsigned char*)s.mvw);
  const size_t hb = ((size_t)4 * S + 7) & ~(size_t)7;
  s.hits = nullptr, s.hitw = nullptr;
  if (S <= K2_HIT_S && hb + (size_t)8 * S <= r1) s.hits = (int*)s.mvrot, s.hitw = (double*)((unsigned char*)s.mvrot + hb);
  else if (S <= K2_HIT_S && hb <= r1 && (size_t)8 * S <= r2) s.hits = (int*)s.mvrot, s.hitw = s.mvw;
  return s;
}
__device__ __forceinline__ Slice carve(unsigned char* p, const AuvDev& d) { return carve(p, d.cfg.n_sensors, d.k_max, d.m_max, d.seg_cap); }

// exact test of one (ray, boundary segment) pair, sensor.py:140-159; w = (wx, wy, sx, sy) with
// w = a - p0, s = b - a.  A hit keeps min t on the ray: the reference's distance
// |p0 + t r - p0| is (weakly) monotone in t for a fixed ray, also in floating point, so the min
// over hits of the distance == the distance at the min t (formed once per ray in phase E).
__device__ __forceinline__ void test_pair(const double4 w, const double tn, const double2 r,
                                          unsigned long long* slot) {
  const double den = r.x * w.w - r.y * w.z;
  const double un = w.x * r.y - w.y * r.x;              // u = un/den along the boundary segment
  // 0 <= tn/den <= 1 and 0 <= un/den <= 1 decided without dividing (exactly equivalent for
  // correctly rounded IEEE division); den < 0 handled by flipping all three signs (exact)
  const bool neg = den < 0.0;
  const double dn = neg ? -den : den, t1 = neg ? -tn : tn, u1 = neg ? -un : un;
  const bool hit = (dn != 0.0) & (t1 >= 0.0) & (t1 <= dn) & (u1 >= 0.0) & (u1 <= dn);
  if (hit) atomicMin(slot, d2u(tn / den));
}

// The same for the rays of one work item, whose segment (hence tn) is fixed: a hit needs t = tn / den
// >= 0, i.e. den of the sign of tn, so the flip is decided once per item (sg = -1 for tn < 0: exact) and
// a ray whose den has the other sign fails `dn > 0`.  tn == 0 (p0 on the segment's line: t = 0 for any
// non-parallel ray) has no preferred sign and takes the general form above.
// `ws` = sg * w, formed once per item: sg (r.x w.w - r.y w.z) == r.x (sg w.w) - r.y (sg w.z) bit for bit (negation commutes
// with every rounding), so the two multiplications by sg per ray are gone.
__device__ __forceinline__ void test_pair_signed(const double4 ws, const double ta, const double2 r, unsigned long long* slot) {
  const double dn = r.x * ws.w - r.y * ws.z;
  const double u1 = ws.x * r.y - ws.y * r.x;
  const bool hit = (dn > 0.0) & (ta <= dn) & (u1 >= 0.0) & (u1 <= dn);
  if (hit) atomicMin(slot, d2u(ta / dn));                  // = tn / den: both signs flipped, exact
}

// (window x boundary) pairs of an obstacle that did not fit the LDS stage, straight from HBM
template <typename SegPtr>
__device__ __forceinline__ void sweep_unstaged(const ObsLds& o, int tid, int nthr, int S, double px, double py,
                                               const double2* rayv, unsigned long long* dbits, SegPtr g) {
  const int total = o.count * o.nseg;
  for (int q = tid; q < total; q += nthr) {
    const int ro = q / o.nseg, si = q - ro * o.nseg;
    const int i = wrap_ray(o.start + ro, S);
    const double4 s = g[si];
    const double4 w = make_double4(s.x - px, s.y - py, s.z - s.x, s.w - s.y);
    test_pair(w, w.x * w.w - w.y * w.z, rayv[i], &dbits[i]);
  }
}

// inside flag of one filled obstacle straight from its boundary in HBM/LDS, lanes <-> segments
// (ballot: any on-boundary, parity of crossings).  "On the boundary" (GEOS distance == 0) is decided with
// the same predicates as Distance::pointToSegment == 0, without its divisions / square roots
template <typename SegPtr>
__device__ __forceinline__ int inside_flag_wave(double px, double py, SegPtr g, int nseg, int lane) {
  int on_any = 0, cross = 0;
  for (int sb = 0; sb < nseg; sb += AUV_WAVE) {
    const int si = sb + lane;
    bool on = false, cr = false;
    if (si < nseg) {
      const double4 s = g[si];
      const double sx = s.z - s.x, sy = s.w - s.y, dxa = px - s.x, dya = py - s.y;
      const double len2 = sx * sx + sy * sy, dot = dxa * sx + dya * sy;
      if (len2 == 0.0 || dot <= 0.0) on = (dxa == 0.0 && dya == 0.0);
      else if (dot >= len2) on = (px == s.z && py == s.w);
      else on = ((s.y - py) * sx - (s.x - px) * sy) == 0.0;
      if ((s.y > py) != (s.w > py)) cr = px < s.x + (py - s.y) * (s.z - s.x) / (s.w - s.y);
    }
    on_any |= __any(on);
    cross += __popcll(__ballot(cr));
  }
  return (on_any ? 2 : 0) | (cross & 1);
}

// ---- phase C: ray vectors, vessel.py:66-68, :317 (only if some obstacle has rays to test) ----
// cos / sin of (beam angle + psi) by the addition theorem from the per-config table of beam
// angles (built at load time): one sincos per environment instead of one per ray
// `staged`: the table entries are in the ray slots already (k2_stage_beams, run ahead of time by the one-launch step)
__device__ __forceinline__ void k2_stage_beams(const AuvDev& d, const int lane, const Slice& L) {
  const int S = d.cfg.n_sensors;
  for (int i = lane; i < S; i += AUV_WAVE) L.rayv[i] = d.beam_cs[i];
}

__device__ __forceinline__ void k2_rays(const AuvDev& d, const int lane, const Slice& L, const bool staged = false) {
  const int S = d.cfg.n_sensors;
  const double R = d.cfg.sensor_range;
  const double px = L.hdr->px, py = L.hdr->py;
  double sin_psi, cos_psi;
  if (S <= 4 * AUV_WAVE) {
    // the usual shapes: all passes' table entries are requested before the sincos, so the passes
    // do not each wait for their own trip to memory
    double2 b[4];
    if (!staged) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int i = q * AUV_WAVE + lane;
        b[q] = d.beam_cs[i < S ? i : 0];                    // cos, sin of -pi + (i + 1) * dangle
      }
    }
    cos_psi = L.hdr->cpsi, sin_psi = L.hdr->spsi;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (q * AUV_WAVE >= S) break;                         // (uniform: 180 beams are three passes)
      const int i = q * AUV_WAVE + lane;
      if (i < S) {
        if (staged) b[q] = L.rayv[i];                       // (already in this very slot: no need to hold all four)
        const double c = cos_psi * b[q].x - sin_psi * b[q].y, s = sin_psi * b[q].x + cos_psi * b[q].y;
        // end point minus origin, formed exactly as the reference forms the end point
        double ex = px + c * R, ey = py + s * R;
        L.rayv[i] = make_double2(ex - px, ey - py);
        L.dbits[i] = d2u(2.0);   // no hit yet (hits have t in [0, 1])
      }
    }
  } else {
    cos_psi = L.hdr->cpsi, sin_psi = L.hdr->spsi;
    for (int i = lane; i < S; i += AUV_WAVE) {
      const double2 b = staged ? L.rayv[i] : d.beam_cs[i];
      const double c = cos_psi * b.x - sin_psi * b.y, s = sin_psi * b.x + cos_psi * b.y;
      double ex = px + c * R, ey = py + s * R;
      L.rayv[i] = make_double2(ex - px, ey - py);
      L.dbits[i] = d2u(2.0);
    }
  }
  auv_wave_lds_sync();
}

// phase A: movers (obstacles.py:195-233) -- kinematics, then pose and cull-circle centre into LDS.  Needs nothing
// of the vessel: the one-launch step runs it while the dynamics role is still integrating.
// LD: the mover rows are read with agent-scope loads -- inside a launch of several steps (k_step_multi) the wave that wrote them
// one step earlier ran on another CU, possibly another XCD, and no kernel boundary lies in between
template <bool WT = false, bool LD = false>
__device__ __forceinline__ void k2_movers(const AuvDev& d, const int e, const int lane, const Slice& L, const EnvDesc& ed,
                                          const int advance_movers) {
  const long long m0 = ed.m0;
  const int M = ed.M;
  for (int m = lane; m < M; m += AUV_WAVE) {
    double4 st = auv_ld4<LD>(&d.mover[(size_t)e * d.m_max + m]);
    const double4 par_m = d.mv_param[m0 + m];
    if (advance_movers) {
      const double dt = d.cfg.dt;
      const long long voff = d.mv_vtab_off[m0 + m];
      const int vlen = d.mv_vtab_len[m0 + m];
      st.w += dt;
      int idx = (int)floor(st.w);
      if (idx >= (int)par_m.w - 1) {
        st.w = 0.0;
        idx = 0;
        st.x = par_m.y;
        st.y = par_m.z;
      }
      if (idx > vlen - 1) idx = vlen - 1;
      double2 v = d.mv_vtab[voff + idx];
      double dx = dt * v.x, dy = dt * v.y;
      st.z = atan2(dy, dx);
      st.x = st.x + dx;
      st.y = st.y + dy;
      auv_st<WT>(&d.mover[(size_t)e * d.m_max + m], st);
    }
    const double wd = par_m.x;
    double s, c;
    sincos(st.z, &s, &c);
    // closed form of enclosing_circle for the pentagon (MRR = body box): tests/test_world.py
    const double x0 = 5.0 * wd / 18.0, dxc = wd / 2.0 - x0;
    L.mvcull[m] = make_double2(st.x + x0 + c * dxc, st.y + s * dxc);   // radius: wd * sqrt(5) / 2
    if (fabs(c) < 2.5e-16) c = 0.0;   // shapely.affinity.rotate snaps tiny cos/sin
    if (fabs(s) < 2.5e-16) s = 0.0;
    L.mvrot[m] = make_double4(c, s, st.x, st.y);   // the pentagon's segments are formed on demand (MoverSegs)
    L.mvw[m] = wd;
  }
}

// what phase B needs of this lane's obstacle (the first 64 of the world), fetched ahead of time
struct K2Pre {
  int4 meta;
  double cx, cy, rho;
  uint8_t near;
};
template <bool LD = false>
__device__ __forceinline__ K2Pre k2_prefetch(const AuvDev& d, const int e, const int lane, const EnvDesc& ed) {
  K2Pre p;
  p.meta = make_int4(0, 0, 0, 0), p.cx = p.cy = p.rho = 0.0, p.near = 0;
  if (lane < ed.K) {
    p.meta = d.obs_meta[ed.k0 + lane];
    p.cx = d.obs_cull[3 * (ed.k0 + lane)], p.cy = d.obs_cull[3 * (ed.k0 + lane) + 1], p.rho = d.obs_cull[3 * (ed.k0 + lane) + 2];
    p.near = auv_ld<LD>(&d.nearby[(size_t)e * d.k_max + lane]);
  }
  return p;
}

// phases A, C, B, S for one environment, by one wave
// movers_done / kp: phase A has been run / the first 64 obstacle records have been fetched by the caller already
template <bool WT = false>
__device__ __forceinline__ int k2_front(const AuvDev& d, const int e, const int lane, const Slice& L, const int advance_movers,
                        const EnvPre* pre = nullptr, const int movers_done = 0,
                        const K2Pre* kp = nullptr, const bool beams_staged = false, int2* lim_defer = nullptr) {
  const int S = d.cfg.n_sensors;
  const size_t n = (size_t)d.n;
  const int4 cnt = pre ? pre->cnt : d.counters[e];
  const double px = pre ? pre->s[0] : d.state[0 * n + e], py = pre ? pre->s[1] : d.state[1 * n + e],
               psi = pre ? pre->s[2] : d.state[2 * n + e];
  const EnvDesc ed = (pre && pre->ed) ? *pre->ed : d.env_desc[e];
  // cos / sin of the heading for the ray table (phase C): formed right here, as soon as psi is known, so that
  // the ~100 dependent fp64 instructions are off the chain that follows phase B (the obstacle records were also
  // requested ahead of it once: 11 more live registers, 10 spilled, and the gain was gone)
  {
    double sn, co;
    sincos(psi, &sn, &co);
    if (lane == 0) L.hdr->cpsi = co, L.hdr->spsi = sn;               // (read back after the LDS sync that ends phase B)
  }
  const long long k0 = ed.k0;
  const int K = ed.K;
  const double R = d.cfg.sensor_range, W = d.cfg.vessel_width;
  // 2 pi / S and S / (2 pi): per-config constants (k_derive forms them with these very divisions), two scalar loads
  // instead of two fp64 divisions (~32 instructions) per wave
  const double dangle = d.derived[4], inv_dangle = d.derived[5];
  if (lane == 0) {
    L.hdr->px = px, L.hdr->py = py;
    L.hdr->n_act = 0;
  }

  // ---- phase A: movers (unless the caller has run it already) ----
  if (!movers_done) k2_movers<WT>(d, e, lane, L, ed, advance_movers);
  if (!d.cfg.use_lidar) return 0;   // lidar_d stays at sensor_range from reset; collision stays 0

  // ---- phase B0 (every sensor_interval_load_obstacles-th vessel step, vessel.py:266-273): refresh
  //      the cached nearby mask.  Kept apart from the cull-window pass below so that the exact
  //      distance loops do not sit inside its register budget ----
  const bool refresh = (cnt.y % d.cfg.sensor_interval_load_obstacles) == 0;
  if (refresh) {
    for (int kb = 0; kb < K; kb += AUV_WAVE) {
      const int k = kb + lane;
      // Point.distance(boundary) - width < range (filled: 0 inside).  The boundary lies inside the
      // obstacle's enclosing circle (c, rho), so
      //   |p0 - c| - rho <= distance(p0, boundary) <= |p0 - c| + rho ;
      // only obstacles the two bounds cannot classify need the exact distance (same answer,
      // far fewer segment loops).  1e-9 m of slack keeps rounding on the exact side.
      int4 meta = make_int4(0, 0, 0, 0);
      double dc = 0.0, rho = 0.0;
      int state = 0;                                          // 0 far, 1 near, 2 undecided
      if (k < K) {
        meta = d.obs_meta[k0 + k];
        double cx, cy;
        if (meta.x == AUV_OBS_MOVER) {
          const double2 c2 = L.mvcull[meta.w];
          cx = c2.x, cy = c2.y, rho = L.mvw[meta.w] * sqrt(5.0) / 2.0;
        } else {
          cx = d.obs_cull[3 * (k0 + k)], cy = d.obs_cull[3 * (k0 + k) + 1], rho = d.obs_cull[3 * (k0 + k) + 2];
        }
        dc = sqrt((px - cx) * (px - cx) + (py - cy) * (py - cy));
        state = (dc - rho - W >= R + 1e-9) ? 0 : ((dc + rho - W < R - 1e-9) ? 1 : 2);
      }
      // The undecided few.  An environment that refreshes is otherwise the launch's straggler (the
      // slowest wave ends the kernel), so their boundaries are walked by the whole wave, flattened:
      // lanes <-> (obstacle, segment) pairs, every pass one trip to memory.  The decision
      // fl(min_t - W) < R equals OR over segments of fl(t - W) < R (rounding is monotone), so an LDS
      // flag per obstacle replaces the min.  act / sbase / par / obs are free scratch here (phase B
      // rewrites them).
      // (rare) a filled obstacle whose enclosing circle contains p0: exact inside test first
      unsigned long long pin = __ballot(state == 2 && meta.x != AUV_OBS_RING && dc <= rho * (1.0 + 1e-9) + 1e-9);
      while (pin) {
        const int src = __ffsll((long long)pin) - 1;
        pin &= pin - 1;
        const int kind = __shfl(meta.x, src, AUV_WAVE), soff = __shfl(meta.y, src, AUV_WAVE),
                  nseg = __shfl(meta.z, src, AUV_WAVE), mi = __shfl(meta.w, src, AUV_WAVE);
        const int in = (kind == AUV_OBS_MOVER) ? inside_flag_wave(px, py, mover_segs(L.mvrot[mi], L.mvw[mi]), nseg, lane)
                                               : inside_flag_wave(px, py, d.seg + soff, nseg, lane);
        if (in && lane == src) state = 1;                       // distance 0
      }
      const unsigned long long und = __ballot(state == 2);
      if (und) {
        const int nu = __popcll(und), pos = __popcll(und & ((1ull << lane) - 1ull));
        const int incl = auv_wave_scan_incl((state == 2) ? meta.z : 0);
        const int total = auv_wave_last(incl);
        if (state == 2) {
          ObsLds o;
          o.kind = meta.x, o.seg_off = (meta.x == AUV_OBS_MOVER) ? meta.w : meta.y, o.nseg = meta.z;
          o.start = 0, o.count = 0, o.wind = 0;
          L.obs[pos] = o;
          L.sbase[pos + 1] = incl;
          L.par[pos] = 0;
        }
        if (lane == 0) L.sbase[0] = 0;
        auv_wave_lds_sync();
        for (int tb = 0; tb < total; tb += AUV_WAVE) {
          const int t = tb + lane;
          if (t < total) {
            int lo = 0, hi = nu;                                 // largest a with sbase[a] <= t
            while (hi - lo > 1) {
              const int mid = (lo + hi) >> 1;
              if (L.sbase[mid] <= t) lo = mid; else hi = mid;
            }
            const ObsLds o = L.obs[lo];
            const int si = t - L.sbase[lo];
            const double4 sg = (o.kind == AUV_OBS_MOVER) ? mover_segs(L.mvrot[o.seg_off], L.mvw[o.seg_off])[si] : d.seg[o.seg_off + si];
            // fp32 screen of dist - W < R: the point-segment distance of vessel-relative coordinates
            // (< ~1 km) is good to well under a millimetre in fp32; only within 1 cm of the threshold
            // does the exact fp64 distance decide
            const float ax = (float)(sg.x - px), ay = (float)(sg.y - py), bx = (float)(sg.z - px), by = (float)(sg.w - py);
            const float ex = bx - ax, ey = by - ay, l2 = ex * ex + ey * ey;
            const float dt = -(ax * ex + ay * ey);
            float d2f;
            if (l2 == 0.0f || dt <= 0.0f) d2f = ax * ax + ay * ay;
            else if (dt >= l2) d2f = bx * bx + by * by;
            else {
              const float cr = ax * ey - ay * ex;
              d2f = cr * cr / l2;
            }
            const float thr = (float)(R + W);
            const float df = sqrtf(d2f);
            int verdict = df < thr - 1e-2f ? 1 : (df > thr + 1e-2f ? 0 : 2);
            if (__any(verdict == 2)) {
              if (verdict == 2) verdict = (auv_pt_seg_dist(px, py, sg.x, sg.y, sg.z, sg.w) - W < R) ? 1 : 0;
            }
            if (verdict == 1) atomicOr(&L.par[lo], 1);
          }
        }
        auv_wave_lds_sync();
        if (state == 2) state = L.par[pos] ? 1 : 0;
        auv_wave_lds_sync();                                     // (the scratch is reused by the next block of obstacles)
      }
      if (k < K) auv_st<WT>(&d.nearby[(size_t)e * d.k_max + k], (uint8_t)state);   // read back below by the same lane
    }
    if constexpr (WT) auv_stores_done();   // (the read below goes past the caches that a write-through store does not update)
  }

  // ---- phase B: cull windows of the nearby obstacles; compaction of obstacles with a window ----
  int n_act = 0;
  for (int kb = 0; kb < K; kb += AUV_WAVE) {
    const int k = kb + lane;
    bool active = false;
    if (k < K) {
      const bool have = kp && kb == 0;                     // (the first 64 obstacles were fetched ahead of time)
      const int4 meta = have ? kp->meta : d.obs_meta[k0 + k];
      const bool mover = meta.x == AUV_OBS_MOVER;
      // static cull circle (unused for movers)
      const double scx = have ? kp->cx : d.obs_cull[3 * (k0 + k)], scy = have ? kp->cy : d.obs_cull[3 * (k0 + k) + 1],
                   srho = have ? kp->rho : d.obs_cull[3 * (k0 + k) + 2];
      ObsLds o;
      o.kind = meta.x;
      o.seg_off = mover ? meta.w * AUV_MOVER_NSEG : meta.y;
      o.nseg = meta.z;
      o.start = 0;
      o.count = 0;
      o.wind = mover ? -1 : (meta.w == -2 ? 1 : (meta.w == -3 ? -1 : 0));   // mover pentagon: clockwise
      // a ring is hollow: from inside, every edge is seen from behind and all of them count.  Its
      // cull circle is its circumcircle, so strictly outside that circle is strictly outside the ring.
      {
        // p0 on or inside a filled obstacle needs p0 within its enclosing circle (centre of the minimum
        // rotated rectangle, half its diagonal): only then do the staging passes evaluate the
        // point-in-polygon predicates of its edges
        double ccx = scx, ccy = scy, crho = srho;
        if (mover) {
          const double2 c2 = L.mvcull[meta.w];
          ccx = c2.x, ccy = c2.y, crho = L.mvw[meta.w] * sqrt(5.0) / 2.0;
        }
        const double rx = ccx - px, ry = ccy - py;
        const bool outside = rx * rx + ry * ry > crho * crho * (1.0 + 1e-9) + 1e-9;
        if (meta.x == AUV_OBS_RING) {
          if (!outside) o.wind = 0;
        } else if (!outside) {
          o.kind |= K2_PIP_FLAG;
        }
      }
      // (after a write-through refresh the flag is read past the caches; otherwise it is an ordinary load)
      const uint8_t near = (WT && refresh) ? auv_ld<WT>(&d.nearby[(size_t)e * d.k_max + k])
                                           : ((have && !refresh) ? kp->near : d.nearby[(size_t)e * d.k_max + k]);
      int2 lim = make_int2(INT32_MIN, INT32_MIN);
      if (near) {
        int start, stop;
        if (d.cfg.cull_mode == AUV_CULL_EXACT) {
          start = 0;
          stop = S;
        } else {
          double cx, cy, rho;
          if (mover) {
            const double2 c2 = L.mvcull[meta.w];
            cx = c2.x, cy = c2.y, rho = L.mvw[meta.w] * sqrt(5.0) / 2.0;
          } else {
            cx = scx, cy = scy, rho = srho;
          }
          const double relx = cx - px, rely = cy - py;
          // The two limits are floor / ceil of (pi + bearing -+ f) / dangle with bearing = atan2(rel) - psi
          // (not wrapped, sensor.py:54) and f = asin(rho / dist) (pi inside the circle, sensor.py:34-36).
          // fp32 atan2f / asinf give both quotients to a few 1e-4 of a ray; unless one of them comes that
          // close to an integer, floor / ceil of the fp32-based value IS the fp64 result, and the fp64
          // atan2 / asin / sqrt / division chain (a quarter of this phase) is not needed.  Error budget in
          // radians, twice what OpenCL guarantees for atan2f (6 ulp of <= pi) and asinf (4 ulp of <= pi/2)
          // plus the input roundings, the latter amplified by 1 / sqrt(1 - q^2); doubled once more below.
          int imin = 0, imax = 0;
          bool exact = true;
          {
            const float rxf = (float)relx, ryf = (float)rely;
            // (hardware square root / reciprocal / reciprocal square root, ~1 ulp each, instead of the correctly rounded
            // library forms: ~50 instructions fewer; their error is inside the budget below)
            const float df = __builtin_amdgcn_sqrtf(rxf * rxf + ryf * ryf);
            const float qf = (float)rho * __builtin_amdgcn_rcpf(df);
            const bool outside = qf < 0.999f, inside = qf > 1.001f;       // (NaN / inf: neither)
            if ((outside || inside) && df > 1e-3f) {
              const float th = atan2f(ryf, rxf);
              double f64 = AUV_PI, err = 3.0e-6;
              if (outside) {
                const float amp = __builtin_amdgcn_rsqf(1.0f - qf * qf);
                f64 = (double)asinf(qf);
                err += 1.0e-6 + 1.2e-6 * (double)amp;
              }
              const double b64 = (double)th - psi;
              // (a product with 1 / dangle instead of two fp64 divisions: the screen only accepts quotients that stay
              // `tol` ~ 1e-4 rays away from an integer, the product differs from the quotient by ~1e-13 rays)
              const double xmin = (AUV_PI + (b64 - f64)) * inv_dangle, xmax = (AUV_PI + (b64 + f64)) * inv_dangle;
              const double fl = floor(xmin), ce = ceil(xmax);
              const double tol = 2.0 * err * inv_dangle + 1e-9;
              if (xmin - fl > tol && (fl + 1.0) - xmin > tol && ce - xmax > tol && xmax - (ce - 1.0) > tol) {
                imin = (int)fl, imax = (int)ce;
                exact = false;
              }
            }
          }
          if (__any(exact)) {
            if (exact) {
              double bearing = atan2(rely, relx) - psi;          // not wrapped (sensor.py:54)
              double dist = sqrt(relx * relx + rely * rely);
              double safe = dist > 1e-8 ? dist : 1e-8;
              double q = rho / safe;
              double f = (q > 1.0 || q < -1.0 || isnan(q)) ? AUV_PI : asin(q);   // NaN -> pi (sensor.py:34-36)
              // |bearing| < 2 pi and f <= pi, so both indices lie in (-S, 2S): int32 is ample
              imin = (int)floor((AUV_PI + (bearing - f)) / dangle);
              imax = (int)ceil((AUV_PI + (bearing + f)) / dangle);
            }
          }
          lim = make_int2(imin, imax);
          start = imin - 1;
          stop = auv_pymod(imax, S);                           // range(i_min - 1, i_max % S)
        }
        if (stop > start) {
          o.start = start;
          o.count = stop - start;
          active = true;
        }
      }
      // (lim_defer: the first 64 obstacles' rows are stored by k2_back -- a wait for a load, here the boundary segments of
      // the staging pass, also waits for every store issued before it, and a write-through store takes ~0.7 us)
      if (lim_defer && kb == 0) *lim_defer = lim;
      else auv_st<WT>(&d.limits[(size_t)e * d.k_max + k], lim);
      L.obs[k] = o;
    }
    const unsigned long long mask = __ballot(active);
    if (active) L.act[n_act + __popcll(mask & ((1ull << lane) - 1ull))] = k;
    n_act += __popcll(mask);
  }
  for (int k = ((lim_defer && K < AUV_WAVE) ? AUV_WAVE : K) + lane; k < d.k_max; k += AUV_WAVE)
    auv_st<WT>(&d.limits[(size_t)e * d.k_max + k], make_int2(INT32_MIN, INT32_MIN));
  if (lane == 0) L.hdr->n_act = n_act;
  auv_wave_lds_sync();
  // prefix of boundary-segment counts over the active list (wave scan, 64 entries per pass);
  // an obstacle is staged iff the prefix up to and including it fits the staging buffer
  {
    int carry = 0;
    for (int ab = 0; ab < n_act; ab += AUV_WAVE) {
      const int a = ab + lane;
      const int v = auv_wave_scan_incl((a < n_act) ? L.obs[L.act[a]].nseg : 0);
      if (a < n_act) {
        L.sbase[a + 1] = carry + v;
        L.par[a] = 0;
      }
      carry += auv_wave_last(v);
    }
    if (lane == 0) L.sbase[0] = 0;
  }
  if (n_act == 0) return 0;                                  // nothing in sight: no rays, no sweep (k2_back writes the free row)
  if (!AUV_RUN_L(d, 2)) return 0;
  k2_rays(d, lane, L, beams_staged);                         // phase C
  return n_act;
}

// phases S + D for one environment, by one wave, in batches of <= K2_SEG_CAP boundary segments
__device__ __forceinline__ void k2_stage_and_pairs(const AuvDev& d, const Slice& L, const int lane, const int n_act,
                                   const double psi, unsigned long long* sub = nullptr) {
  const int S = d.cfg.n_sensors;
  const double px = L.hdr->px, py = L.hdr->py;
#ifdef AUV_STAMPS
  unsigned long long c_stage = 0, c_prefix = 0, c_items = 0, n_it = 0, c0;
#define SUB_T0() c0 = clock64();
#define SUB_ADD(x) { unsigned long long c1 = clock64(); x += c1 - c0; c0 = c1; }
#else
#define SUB_T0()
#define SUB_ADD(x)
#endif
  for (int a0 = 0; a0 < n_act;) {
    const int base0 = L.sbase[a0];
    if (L.sbase[a0 + 1] - base0 > K2_RAW_CAP) {
      // a single boundary larger than a batch (never a circle or a mover): sweep it from HBM
      const ObsLds o = L.obs[L.act[a0]];
      const double4* g = d.seg + o.seg_off;
      if ((o.kind & K2_PIP_FLAG) && inside_flag_wave(px, py, g, o.nseg, lane) != 0) {
        for (int q = lane; q < o.count; q += AUV_WAVE) atomicMin(&L.dbits[wrap_ray(o.start + q, S)], 0ull);
      } else {
        sweep_unstaged(o, lane, AUV_WAVE, S, px, py, L.rayv, L.dbits, g);
      }
      a0 += 1;
      continue;
    }
    int a1 = a0 + 1;
    while (a1 < n_act && L.sbase[a1 + 1] - base0 <= K2_RAW_CAP) a1++;
    const int T_raw = L.sbase[a1] - base0;

    // ---- phase S: one flattened, coalesced pass over the batch's boundary segments; the ones that
    //      face p0 are compacted into the stage ----
    SUB_T0()
    int T = 0;                                             // staged segments
    // the segment of the NEXT pass is requested before this pass's segment is worked on, so a batch
    // of several passes (the heavy sweeps that end the launch) pays one trip to memory, not one per pass
    int a_nx = a0;
    ObsLds o_nx;
    double4 s_nx = make_double4(0.0, 0.0, 0.0, 0.0);
    auto fetch = [&](int tb_f) {
      const int t = tb_f + lane;
      if (t < T_raw) {                                     // (a lane's t only grows from pass to pass: the walk carries on)
        while (t + base0 >= L.sbase[a_nx + 1]) a_nx++;
        o_nx = L.obs[L.act[a_nx]];
        const int si = t + base0 - L.sbase[a_nx];
        // (a mover's seg_off is 5 x its slot)
        s_nx = ((o_nx.kind & K2_KIND_MASK) == AUV_OBS_MOVER) ? mover_segs(L.mvrot[o_nx.seg_off / AUV_MOVER_NSEG], L.mvw[o_nx.seg_off / AUV_MOVER_NSEG])[si]
                                            : d.seg[o_nx.seg_off + si];
      }
    };
    fetch(0);
    for (int tb = 0; tb < T_raw; tb += AUV_WAVE) {
      const int t = tb + lane;
      bool keep = false;
      const int a = a_nx;
      const ObsLds o = o_nx;
      const double4 s = s_nx;
      if (tb + AUV_WAVE < T_raw) fetch(tb + AUV_WAVE);
      double4 wv = make_double4(0.0, 0.0, 0.0, 0.0);
      short2 sp = make_short2(0, 0);
      if (t < T_raw) {
        const double wx = s.x - px, wy = s.y - py, sx = s.z - s.x, sy = s.w - s.y;
        wv = make_double4(wx, wy, sx, sy);
        // A ray from outside a simple closed boundary first meets it on an edge that faces p0, and
        // only the first meeting matters (min t; p0 inside is handled through the predicates below),
        // so an edge seen from behind is not staged.  tn = cross(a - p0, b - a) tells the side;
        // edges p0 (nearly) lies on the line of stay in.
        const double tn_side = wx * sy - wy * sx;
        keep = !((o.wind > 0 && tn_side > 1e-9) || (o.wind < 0 && tn_side < -1e-9));
        if (keep) {
          // conservative range of ray indices this segment can be hit by: the rays between the
          // bearings of its end points (shorter arc; a segment subtends < pi from any point off
          // its line), fp32 trigonometry, rays of slack on both sides (fp32 error ~1e-5 rays)
          const float ax = (float)wx, ay = (float)wy, bx = (float)(s.z - px), by = (float)(s.w - py);
          if ((fabsf(ax) + fabsf(ay) < 1e-6f) || (fabsf(bx) + fabsf(by) < 1e-6f)) {
            sp = make_short2(0, (short)S);                         // p0 (almost) on an end point: all rays
          } else {
            const float PI_F = 3.14159265358979f;
            const float ta = atan2f(ay, ax), tb2 = atan2f(by, bx);
            float dl = tb2 - ta;
            if (dl > PI_F) dl -= 2.0f * PI_F;
            if (dl <= -PI_F) dl += 2.0f * PI_F;
            const float ts = dl >= 0.0f ? ta : tb2;
            const float inv_da = (float)S / (2.0f * PI_F);
            const float f = (ts - (float)psi + PI_F) * inv_da - 1.0f;   // fractional ray index of the arc start
            const int n = (int)ceilf(fabsf(dl) * inv_da) + 4;
            const int klo = ((int)floorf(f) - 1) % S;               // any representative; wrapped per ray
            sp = make_short2((short)klo, (short)(n > S ? S : n));
          }
        }
        if (o.kind & K2_PIP_FLAG) {
          // point-in-polygon predicates of this boundary segment (same tests as inside_flag_wave)
          const double dxa = px - s.x, dya = py - s.y;
          const double len2 = sx * sx + sy * sy;
          const double dot = dxa * sx + dya * sy;
          bool on;
          if (len2 == 0.0 || dot <= 0.0) on = (dxa == 0.0 && dya == 0.0);
          else if (dot >= len2) on = (px == s.z && py == s.w);
          else on = ((s.y - py) * sx - (s.x - px) * sy) == 0.0;
          if (on) atomicOr(&L.par[a], 2);
          if ((s.y > py) != (s.w > py)) {
            const double xint = s.x + (py - s.y) * (s.z - s.x) / (s.w - s.y);
            if (px < xint) atomicXor(&L.par[a], 1);
          }
        }
      }
      const unsigned long long kmask = __ballot(keep);
      const int pos = T + __popcll(kmask & ((1ull << lane) - 1ull));
      if (keep) {
        if (pos < L.cap) {
          L.stage[pos] = wv;
          L.owner[pos] = (unsigned short)a;
          L.span[pos] = sp;
        } else {
          // (rare) more front-facing segments than the stage holds: this lane sweeps its own
          const double tn = wv.x * wv.w - wv.y * wv.z;
          const bool all = o.count >= S;
          const int start_w = wrap_ray(o.start, S);
          for (int j = 0; j < sp.y; j++) {
            const int r = wrap_ray(sp.x + j, S);
            int x = r - start_w;
            if (x < 0) x += S;
            if (all || x < o.count) test_pair(wv, tn, L.rayv[r], &L.dbits[r]);
          }
        }
      }
      T += __popcll(kmask);
    }
    if (T > L.cap) T = L.cap;
    auv_wave_lds_sync();
    SUB_ADD(c_stage)
    if (!AUV_RUN_L(d, 4)) {
      a0 = a1;
      continue;
    }

    // ---- phase D (i): obstacles containing p0 -> distance 0 on every ray of their window ----
    for (int ab = a0; ab < a1; ab += AUV_WAVE) {            // lanes <-> obstacles of the batch
      const int al = ab + lane;
      unsigned long long in_mask = __ballot(al < a1 && L.par[al] != 0 && (L.obs[L.act[al]].kind & K2_PIP_FLAG));
      while (in_mask) {                                      // (rare) whole wave per containing obstacle
        const int a = ab + __ffsll((long long)in_mask) - 1;
        in_mask &= in_mask - 1;
        const ObsLds o = L.obs[L.act[a]];
        for (int q = lane; q < o.count; q += AUV_WAVE) atomicMin(&L.dbits[wrap_ray(o.start + q, S)], 0ull);
      }
    }
    // ---- phase D (ii): work items = (staged segment, run of <= K2_ITEM_RAYS rays of its span),
    //      lanes <-> items, so a segment seen under a wide angle is shared by several lanes ----
    int n_items = 0;
    for (int tb = 0; tb < T; tb += AUV_WAVE) {           // wave prefix sum of items per segment
      const int t = tb + lane;
      int v = 0;
      if (t < T) {
        const int a = L.owner[t];
        const ObsLds o = L.obs[L.act[a]];
        if (!((o.kind & K2_PIP_FLAG) && L.par[a] != 0)) v = (L.span[t].y + K2_ITEM_RAYS - 1) / K2_ITEM_RAYS;
      }
      const int incl = auv_wave_scan_incl(v);
      if (t < T) L.ioff[t + 1] = (unsigned short)(n_items + incl);
      n_items += auv_wave_last(incl);
    }
    if (lane == 0) L.ioff[0] = 0;
    auv_wave_lds_sync();
    SUB_ADD(c_prefix)
#ifdef AUV_STAMPS
    n_it += n_items;
#endif
    if (!AUV_RUN_L(d, 5)) n_items = 0;
    for (int it = lane; it < n_items; it += AUV_WAVE) {
      // largest t < T with ioff[t] <= it (the prefix is non-decreasing), in three rounds of INDEPENDENT LDS reads --
      // strides 16, 4, 1 -- instead of the seven dependent ones of a bisection: three trips to LDS per pass, not seven
      int lo = 0;
      {
        int c = 0;
#pragma unroll
        for (int k = 16; k < K2_SEG_CAP; k += 16) c += (k < T && (int)L.ioff[k] <= it) ? 1 : 0;
        lo = 16 * c;
        c = 0;
#pragma unroll
        for (int k = 4; k < 16; k += 4) c += (lo + k < T && (int)L.ioff[lo + k] <= it) ? 1 : 0;
        lo += 4 * c;
        c = 0;
#pragma unroll
        for (int k = 1; k < 4; k++) c += (lo + k < T && (int)L.ioff[lo + k] <= it) ? 1 : 0;
        lo += c;
      }
      const int t = lo;
      const int a = L.owner[t];
      const ObsLds o = L.obs[L.act[a]];
      const double4 w = L.stage[t];
      const double tn = w.x * w.w - w.y * w.z;               // ray-independent numerator of t
      const short2 sp = L.span[t];
      const bool all = o.count >= S;
      const int start_w = wrap_ray(o.start, S);
      const int j0 = (it - (int)L.ioff[t]) * K2_ITEM_RAYS;
      // the item's rays are fetched together (independent LDS loads), then tested.  Ray index and position inside the
      // obstacle's window are wrapped ONCE per item; from ray to ray both go up by one and wrap at most once (the index
      // arithmetic of eight rays used to be 2.4x their fp64 arithmetic)
      double2 rv[K2_ITEM_RAYS];
      int rr[K2_ITEM_RAYS];
      const int r0 = wrap_ray(sp.x + j0, S);
      int x0 = r0 - start_w;                                  // position of the item's first ray inside the window
      if (x0 < 0) x0 += S;
      int n_left = (int)sp.y - j0;                            // rays of the span from this item's first on
      int r0m = r0 - S, x0m = x0 - S;                         // (index - S: "wrapped" is then the sign)
      // (opaque to the optimiser, which otherwise re-derives every ray's index from the five values above: six dependent
      // subtractions per ray instead of one add)
      asm volatile("" : "+v"(r0m), "+v"(x0m), "+v"(n_left));
#pragma unroll
      for (int jj = 0; jj < K2_ITEM_RAYS; jj++) {
        int r = r0m + jj, x = x0m + jj;                       // in [-S, 8): negative = not wrapped yet
        r = r < 0 ? r + S : r;
        x = x < 0 ? x + S : x;
        const bool ok = (jj < n_left) && (all || x < o.count);
        r = ok ? r : -1;
        rr[jj] = r;
        rv[jj] = L.rayv[ok ? r : 0];
      }
      if (tn != 0.0) {
        const double ta = fabs(tn);
        const double4 ws = tn < 0.0 ? make_double4(-w.x, -w.y, -w.z, -w.w) : w;
#pragma unroll
        for (int jj = 0; jj < K2_ITEM_RAYS; jj++)
          if (rr[jj] >= 0) test_pair_signed(ws, ta, rv[jj], &L.dbits[rr[jj]]);
      } else {
#pragma unroll
        for (int jj = 0; jj < K2_ITEM_RAYS; jj++)
          if (rr[jj] >= 0) test_pair(w, tn, rv[jj], &L.dbits[rr[jj]]);
      }
    }
    auv_wave_lds_sync();
    SUB_ADD(c_items)
    a0 = a1;
  }
#ifdef AUV_STAMPS
  if (sub) sub[0] = c_stage, sub[1] = c_prefix, sub[2] = c_items, sub[3] = n_it;
#endif
}

// phase E: outputs (vessel.py:88-95, :356-359) and the LiDAR term of the Colav reward
// (rewarder.py:205-222: sum of gamma_theta-weighted R exp(-0.1 d) over the beams; the velocity channel
// is identically zero, sensor.py:159) and the float32 closeness columns of the observation row.
// A beam without a return sits exactly at R: closeness 1 - x/x = 0 and its exp() is one per-config
// constant.  So the beams are split: one cheap pass stores the free beams' constants and compacts the
// indices of the beams WITH a return into LDS (ballot + popcount); the square root / log / exp are then
// evaluated over that dense list only (typically one pass of 64 instead of S / 64).
// n_act == 0 (no obstacle had a ray to test): the whole row is free, nothing was swept.
// Global loads and write-through stores of a wave count down ONE counter in issue order: a wait for a load also waits for
// every store issued before it (~0.7 us for a write-through one).  So everything this phase reads from memory -- the beam
// weights, the per-config constants -- is requested before its first store, the returns take their weights along through
// LDS, and the cull-limit rows of phase B (`lim0`: this lane's row of the first 64 obstacles) are stored here, not there.
template <bool WT = false>
__device__ __forceinline__ int k2_back(const AuvDev& d, const int e, const int lane, const Slice& L, const int n_act,
                       float* __restrict__ obs_out = nullptr, double* rew_lidar_out = nullptr, const int2* lim0 = nullptr) {
  const int S = d.cfg.n_sensors;
  const double R = d.cfg.sensor_range, W = d.cfg.vessel_width;
  const bool colav = d.cfg.rewarder == AUV_REWARD_COLAV;
  const int D = 6 + S * (d.cfg.obs_channels == 3 ? 3 : 1);   // row stride of obs_out (use_lidar is on here)
  double* dd = d.lidar_d + (size_t)e * S;
  double* ob = d.obs64 + (size_t)e * (6 + S) + 6;
  float* oo = obs_out ? obs_out + (size_t)e * D + 6 : nullptr;
  if (n_act == 0) {
    const double term = colav ? d.derived[3] : 0.0;
    if (lim0 && lane < d.k_max) auv_st<WT>(&d.limits[(size_t)e * d.k_max + lane], *lim0);
    for (int i = lane; i < S; i += AUV_WAVE) {
      auv_st<WT>(dd + i, R), auv_st<WT>(ob + i, 0.0);
      if (oo) auv_st<WT>(oo + i, 0.0f);
    }
    if (lane == 0) {
      auv_st<WT>(d.collision + e, (uint8_t)0);
      if (colav) auv_st<WT>(d.rew_lidar + e, term);
    }
    if (rew_lidar_out) *rew_lidar_out = term;
    return 0;
  }
  const double logR = d.derived[0];                        // log(1 + R)
  const double raw_free = d.derived[1];                    // R exp(-0.1 R)
  const double wsum = colav ? d.derived[2] : 1.0;          // sum of the beam weights
  const double px = L.hdr->px, py = L.hdr->py;
  int col = 0;
  double num = 0.0;
  if (L.hits) {
    int* hits = L.hits;
    double* hitw = L.hitw;
    constexpr int NQ = K2_HIT_S / AUV_WAVE;
    double bw[NQ];                                         // gamma_theta from the per-config table
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int i = q * AUV_WAVE + lane;
      bw[q] = (colav && i < S) ? d.beam_w[i] : 0.0;
    }
    // ---- which beams have a return; their indices compacted into LDS (no store yet) ----
    unsigned long long hm[NQ];
    int n_hit = 0;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      hm[q] = 0;
      if (q * AUV_WAVE >= S) continue;                     // (uniform)
      const int i = q * AUV_WAVE + lane;
      const bool hit = (i < S) && (u2d(L.dbits[i]) <= 1.0);
      hm[q] = __ballot(hit);
      if (hit) hits[n_hit + __popcll(hm[q] & ((1ull << lane) - 1ull))] = i;
      n_hit += __popcll(hm[q]);
    }
    // the weights are here before the first store goes out: no later wait in this phase has a load behind stores
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0) -- as the builtin, so that the compiler knows too
    asm volatile("" ::: "memory");
    if (lim0 && lane < d.k_max) auv_st<WT>(&d.limits[(size_t)e * d.k_max + lane], *lim0);
    // ---- free beams: constants; the returns' weights go to LDS beside their indices ----
    {
      int base = 0;
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        if (q * AUV_WAVE >= S) continue;
        const int i = q * AUV_WAVE + lane;
        const bool hit = (hm[q] >> lane) & 1ull;
        if (i < S && !hit) {
          auv_st<WT>(dd + i, R), auv_st<WT>(ob + i, 0.0);  // sensor.py:156; closeness 1 - clip(x / x) = 0
          if (oo) auv_st<WT>(oo + i, 0.0f);
          if (colav) num += bw[q] * raw_free;
        }
        if (hit) hitw[base + __popcll(hm[q] & ((1ull << lane) - 1ull))] = bw[q];
        base += __popcll(hm[q]);
      }
    }
    auv_wave_lds_sync();
    if (!AUV_RUN_L(d, 6)) n_hit = 0;
    // ---- the returns ----
    for (int h0 = 0; h0 < n_hit; h0 += AUV_WAVE) {
      const int h = h0 + lane;
      if (h < n_hit) {
        const int i = hits[h];
        const double wgt = hitw[h];
        const double t = u2d(L.dbits[i]);
        // intersection point, then Point.distance: exactly the reference's arithmetic
        const double2 r = L.rayv[i];
        const double X = px + t * r.x, Y = py + t * r.y;
        const double dx = X - px, dy = Y - py;
        const double di = sqrt(dx * dx + dy * dy);
        auv_st<WT>(dd + i, di);
        double cl = d.cfg.sensor_log_transform ? 1 - auv_clip(log(1 + di) / logR, 0.0, 1.0) : 1 - auv_clip(di / R, 0.0, 1.0);
        cl = auv_clip(cl, -1.0, 1.0);
        auv_st<WT>(ob + i, cl);
        if (oo) auv_st<WT>(oo + i, (float)cl);
        if (colav) num += wgt * ((di != R) ? R * exp(-0.1 * di) : raw_free);   // gamma_x
        col |= (di < W);
      }
    }
  } else {
    // (more beams than the list holds: every pass does everything)
    if (lim0 && lane < d.k_max) auv_st<WT>(&d.limits[(size_t)e * d.k_max + lane], *lim0);
    for (int i = lane; i < S; i += AUV_WAVE) {
      const double t = u2d(L.dbits[i]);
      double di = R;
      if (t <= 1.0) {
        const double2 r = L.rayv[i];
        const double X = px + t * r.x, Y = py + t * r.y;
        const double dx = X - px, dy = Y - py;
        di = sqrt(dx * dx + dy * dy);
      }
      auv_st<WT>(dd + i, di);
      double cl = 0.0;
      if (t <= 1.0) cl = d.cfg.sensor_log_transform ? 1 - auv_clip(log(1 + di) / logR, 0.0, 1.0) : 1 - auv_clip(di / R, 0.0, 1.0);
      cl = auv_clip(cl, -1.0, 1.0);
      auv_st<WT>(ob + i, cl);
      if (oo) auv_st<WT>(oo + i, (float)cl);
      if (colav) num += d.beam_w[i] * ((di != R) ? R * exp(-0.1 * di) : raw_free);
      col |= (di < W);
    }
  }
  col = __any(col);
  if (colav) num = auv_wave_sum(num);
  const double term = (colav && S > 0) ? -num / wsum : 0.0;   // the same in every lane
  if (lane == 0) {
    auv_st<WT>(d.collision + e, (uint8_t)(col != 0));
    if (colav) auv_st<WT>(d.rew_lidar + e, term);
  }
  if (rew_lidar_out) *rew_lidar_out = term;
  return col != 0;
}

#ifndef AUV_DEVICE_FUNCS_ONLY
// all environments: wave g handles env g
__global__ void __launch_bounds__(AUV_BLOCK, 4) k2_lidar(AuvDev d, int advance_movers) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int wave = threadIdx.x / AUV_WAVE, lane = threadIdx.x % AUV_WAVE;
  const int e = auv_uniform(blockIdx.x * AUV_ENVS_PER_BLOCK + wave);
  if (e >= d.n) return;
  const Slice L = carve(smem + wave * k2_slice_bytes(d), d);
  AUV_STAMP_DECL
#ifdef AUV_STAMPS
  const unsigned long long t_real0 = wall_clock64();
#endif
  const int n_act = k2_front(d, e, lane, L, advance_movers);
  if (!d.cfg.use_lidar) return;
  AUV_STAMP()
  k2_stage_and_pairs(d, L, lane, n_act, d.state[2 * (size_t)d.n + e]);
  AUV_STAMP()
  k2_back(d, e, lane, L, n_act);
  AUV_STAMP()
  AUV_STAMP_FLUSH(e, 0)   // 0: front (A, C, B, S)  1: pairs (D)  2: back (E)
#ifdef AUV_STAMPS
  if (lane == 0) d.stamps[(size_t)e * 16 + 3] = t_real0, d.stamps[(size_t)e * 16 + 4] = wall_clock64();
  if (lane == 0) d.stamps[(size_t)e * 16 + 5] = (unsigned long long)L.sbase[n_act], d.stamps[(size_t)e * 16 + 6] = (unsigned long long)n_act;
#endif
}

// reset pass: only the environments on the fresh list (reset() / auto-reset), movers not advanced
__global__ void __launch_bounds__(AUV_BLOCK) k2_lidar_fresh(AuvDev d) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int wave = threadIdx.x / AUV_WAVE, lane = threadIdx.x % AUV_WAVE;
  const int nf = *d.fresh_count;
  const Slice L = carve(smem + wave * k2_slice_bytes(d), d);
  for (int i = blockIdx.x * AUV_ENVS_PER_BLOCK + wave; i < nf; i += gridDim.x * AUV_ENVS_PER_BLOCK) {
    const int e = auv_uniform(d.fresh_list[i]);
    const int n_act = k2_front(d, e, lane, L, 0);
    if (d.cfg.use_lidar) {
      k2_stage_and_pairs(d, L, lane, n_act, d.state[2 * (size_t)d.n + e]);
      k2_back(d, e, lane, L, n_act);
    }
    auv_wave_lds_sync();
  }
}

#endif

}  // namespace

#ifndef AUV_DEVICE_FUNCS_ONLY
size_t auv_k2_lds_bytes(const AuvDev& d) {
  return k2_slice_bytes(d) * AUV_ENVS_PER_BLOCK;
}

// gfx950 has 160 KiB of LDS per CU; footprints above the 64 KiB default need the opt-in.
hipError_t auv_k2_prepare(const AuvDev& d) {
  const size_t b = auv_k2_lds_bytes(d);
  if (b <= 64 * 1024) return hipSuccess;
  hipError_t e = hipFuncSetAttribute((const void*)k2_lidar, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)k2_lidar_fresh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
}

void auv_launch_k2(const AuvDev& d, int advance_movers, hipStream_t st) {
  hipLaunchKernelGGL(k2_lidar, dim3((d.n + AUV_ENVS_PER_BLOCK - 1) / AUV_ENVS_PER_BLOCK), dim3(AUV_BLOCK),
                     auv_k2_lds_bytes(d), st, d, advance_movers);
}

void auv_launch_k2_fresh(const AuvDev& d, hipStream_t st) {
  int grid = (d.n + AUV_ENVS_PER_BLOCK - 1) / AUV_ENVS_PER_BLOCK;
  if (grid > AUV_FRESH_GRID) grid = AUV_FRESH_GRID;
  hipLaunchKernelGGL(k2_lidar_fresh, dim3(grid), dim3(AUV_BLOCK), auv_k2_lds_bytes(d), st, d);
}
#endif
