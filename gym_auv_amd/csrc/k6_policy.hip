// K6 -- the policy in the loop: actor and critic of the reference's PPO set-up evaluated for a sub-batch of
// environments by ONE launch on the sub-batch's stream, straight from the observation rows the step kernel has just
// written, with sampling, log-probability, action clipping and the rollout stores in the epilogue.
//
// Reference: scripts/run.py:332-357 -- PPO2(MlpPolicy, net_arch [256, 128, 64] for policy and value function, tanh),
// a diagonal Gaussian over the two actions; what stable-baselines evaluates per environment step through
// SubprocVecEnv (run.py:293-296).  The torch modules of examples/ppo.py (ActorCritic) are the numerical reference
// (tests/test_gpu_policy.py: <= 1e-5).
//
// Shape of the work: [rows, obs_dim] x [obs_dim -> 256 -> 128 -> 64 -> out] twice, rows ~ 1024 per call: small GEMMs
// whose weights (0.7 MB for both nets) every workgroup streams from L2 while 1.45 GFLOP per 4096 rows go to the
// matrix pipe -- which the environment's own kernels never touch (they are VALU / latency bound), so policy and
// sweeps of different chains overlap on a CU.  One workgroup = 16 rows (one MFMA M-tile) of ONE net, eight waves (the launch
// is bound by the latencies of its weight loads and MFMA chains, not by a pipe: four waves per workgroup left the
// matrix pipe 35 % busy -- profiles/r04/pmc_policy.txt -- so the SIMDs get more waves to switch between):
//   * the observation tile goes to LDS once (and, from there, into the rollout's O[t]);
//   * a layer's 16-column n-tiles are dealt round-robin to the waves; v_mfma_f32_16x16x4_f32 (exact f32, a k-ordered fmaf
//     chain per output) with A from LDS and B = the weight rows straight from global memory as float4 (lane (n, g) reads
//     W[n][16 J + 4 g .. + 3]: A's lane (m, g) reads the same four k, so a super-step of 16 k is one 16-byte load per
//     operand and four MFMAs per n-tile; the k order inside a super-step is permuted identically on both sides);
//   * bias + tanh in the epilogue of each layer, activations stay in LDS;
//   * last layer: one n-tile (outputs padded to 16); its epilogue samples a = mu + sigma eps (counter-based generator,
//     Box-Muller), forms log pi(a), maps a into the action space into the env's action buffer, and stores the transition.
// Optional (flag): nothing else -- a bf16 variant would run the matrix pipe 16x faster but is not the reference's arithmetic.
#include "auv_device.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef POL_MT
#define POL_MT 1                   // MFMA M-tiles (of 16 rows) per workgroup.  (2 -- every weight fragment a wave loads then serves two
#endif                             // tiles, half the L2 traffic -- was measured: 51 against 61 M env-steps/s in PPO rollouts: fewer, longer
                                   // workgroups; the launch is not bound by what it pulls out of L2)
#define POL_ROWS (16 * POL_MT)
#ifndef POL_WAVES
#define POL_WAVES 8                 // waves per workgroup: a layer's n-tiles are dealt round-robin to them
#endif
#define POL_THREADS (64 * POL_WAVES)
#define POL_H1 256
#define POL_H2 128
#define POL_H3 64
#define POL_OUT 16                 // the last layer's outputs (2 actions / 1 value) padded to one n-tile
#define POL_LOG_SQRT_2PI 0.9189385332046727f

__host__ __device__ inline int pol_pad16(int x) { return (x + 31) & ~31; }   // (the k-step of pol_layer: 32)
// floats of ONE packed net: W1 [H1][K0p] b1 [H1] W2 [H2][H1] b2 [H2] W3 [H3][H2] b3 [H3] W4 [16][H3] b4 [16]
__host__ __device__ inline size_t pol_net_floats(int obs_dim) {
  const size_t k0 = (size_t)pol_pad16(obs_dim);
  return POL_H1 * k0 + POL_H1 + (size_t)POL_H2 * POL_H1 + POL_H2 + (size_t)POL_H3 * POL_H2 + POL_H3 + (size_t)POL_OUT * POL_H3 + POL_OUT;
}
// LDS: X [ROWS][K0p + 8] | Y1 [ROWS][H1 + 8]; Y2 [ROWS][H2 + 8] re-uses X's place (dead after layer 1), Y3 [ROWS][H3 + 8]
// Y1's (dead after layer 2).  Row strides = 8 mod 64 floats: the 16-byte reads of an A operand -- 16 rows x 4 k-groups --
// then touch every bank once.
__host__ __device__ inline size_t pol_lds_x_floats(int obs_dim) {
  const size_t x = (size_t)pol_pad16(obs_dim) + 8, y2 = POL_H2 + 8;
  return POL_ROWS * (x > y2 ? x : y2);
}
__host__ __device__ inline size_t pol_lds_bytes(int obs_dim) {
  return sizeof(float) * (pol_lds_x_floats(obs_dim) + (size_t)POL_ROWS * (POL_H1 + 8));
}

struct PolicyArgs {
  auv_policy_io_t io;
  int32_t e0, ne;
  long long t_host, gstep_host;      // >= 0: the host names the rollout position and the generator's step (auv_policy_rollout: a plain
                                     // loop of launches); -1: both live on the device (io.ctr) and the launch moves them on itself
};

// splitmix64 finaliser: a counter-based generator -- (seed, step, row, component) -> 64 well-mixed bits
__device__ __forceinline__ unsigned long long pol_mix(unsigned long long z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ float pol_gauss(unsigned long long seed, unsigned long long step, unsigned int row, unsigned int comp) {
  const unsigned long long h = pol_mix(pol_mix(seed ^ (step * 0xd1342543de82ef95ull)) ^ (((unsigned long long)row << 1) | comp));
  const float u1 = ((float)(unsigned int)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);      // (0, 1): 24 bits
  const float u2 = ((float)(unsigned int)((h >> 8) & 0xffffffu)) * (1.0f / 16777216.0f);   // [0, 1)
  return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

// tanh: 1 - 2 / (exp(2 x) + 1) with the hardware's exp2 and reciprocal (absolute error ~2e-7, far inside the 1e-5 of the
// parity test; saturates correctly: exp -> inf gives 1, exp -> 0 gives -1) instead of the library's ~40-instruction tanhf
__device__ __forceinline__ float pol_tanh(const float x) {
#ifdef POL_LIBM_TANH
  return tanhf(x);
#else
  const float t = __expf(2.0f * x);
  return 1.0f - 2.0f * __frcp_rn(t + 1.0f);
#endif
}

// One layer for this wave's n-tiles: Y = tanh(X W^T + b) (or the raw pre-activation rows of the last layer).
//   X  LDS [16][ldx], columns [0, Kp) valid (zero padded); W global, the [N][Kp] matrix of torch's Linear.weight (K padded
//   to a multiple of 32) re-ordered into MFMA fragment order; wave w takes the n-tiles w, w + 4, ... (NT of them).
//   A k-step covers 32 k: lane (row or column l & 15, group g = l >> 4) holds k = 32 J + 8 g + j, j = 0..7, of its row of
//   X (LDS) and of its row of W; MFMA j of the step multiplies element j of every lane, i.e. the k set
//   {32 J + 8 g + j : g} -- A and B permuted alike.  Read from the row-major matrix, neighbouring lanes would sit in
//   different rows (768 bytes apart): 64 separate requests per load instruction, and the kernel is bound by that (36 us for
//   4096 rows, profiles/r04/policy_bench_rowmajor.log); in fragment order an instruction reads one contiguous kilobyte.
//   The weights of the next DEPTH k-steps are in flight while a step's MFMAs issue: a workgroup streams its net's 0.37 MB
//   from L2 with only a wave or two per SIMD to hide the latency (one 16-k step ahead: 60 us for 4096 rows; profiles/r04).
//   KS = 2: the k range alternates between two accumulators per n-tile (a single v_mfma_f32_16x16x4_f32 chain is
//   latency-bound: 40 cycles dependent against 32 of issue).
#ifndef POL_DEPTH
#define POL_DEPTH 2                // (4 was measured: 141 VGPRs, or 24 spilled under the 128 that two workgroups per CU allow: 27 against 23 us)
#endif
// the weight fragments of a wave's n-tiles for the next DEPTH k-steps, in flight or landed.  BF: the weights are stored as
// bf16 (auv_policy_io::params_bf16, optional): a k-step's fragment is then 16 bytes per lane and ONE
// v_mfma_f32_16x16x32_bf16 instead of eight f32 MFMAs (activations are rounded to bf16 on the way into the MFMA,
// accumulation, bias and tanh stay f32) -- NOT the reference's arithmetic: ~1e-2 on the means, behind a flag.
template <int NTILES, bool BF>
struct PolW {
  static constexpr int NT = (NTILES + POL_WAVES - 1) / POL_WAVES;   // n-tiles of this wave: wave, wave + WAVES, ...
  static constexpr int BLK = BF ? 256 : 512;                         // floats (4-byte units) per (n-tile, k-step) block
  float4 q[POL_DEPTH][NT][BF ? 1 : 2];
  const float* row[NT];
  float bias[NT];                    // the epilogue's bias of this lane's column of each n-tile, requested with the weights
};

// Request the first DEPTH k-steps of a layer's weights.  They depend on nothing the kernel computes, so the request for
// layer l + 1 goes out BEFORE layer l's epilogue and the barrier behind it (and layer 1's before the observation tile is
// fetched): a layer then starts on fragments that have landed instead of on a cold trip to L2.
template <int NTILES, bool BF>
__device__ __forceinline__ void pol_prefetch(PolW<NTILES, BF>& w, const float* __restrict__ W, const float* __restrict__ b,
                                             const int Kp, const int wave, const int lane) {
  if (wave >= NTILES) return;
  // (the biases too: asked for in a layer's epilogue they cost it a trip to L2 with nothing else to do; asked for here,
  // ahead of the barrier in front of the layer, they cannot be moved back down to their use)
#pragma unroll
  for (int t = 0; t < PolW<NTILES, BF>::NT; t++) w.bias[t] = b[((wave + POL_WAVES * t) % NTILES) * 16 + (lane & 15)];
  const int nJ = Kp / 32;
  constexpr int BLK = PolW<NTILES, BF>::BLK;
#pragma unroll
  for (int t = 0; t < PolW<NTILES, BF>::NT; t++) w.row[t] = W + (size_t)((wave + POL_WAVES * t) % NTILES) * nJ * BLK + 4 * lane;   // (% : a tile index
                                                                         // past the end re-reads a valid tile, its result is dropped)
#pragma unroll
  for (int d = 0; d < POL_DEPTH; d++) {
    const int Jd = d < nJ ? d : nJ - 1;
#pragma unroll
    for (int t = 0; t < PolW<NTILES, BF>::NT; t++) {
      w.q[d][t][0] = *(const float4*)(w.row[t] + BLK * Jd);
      if (!BF) w.q[d][t][BF ? 0 : 1] = *(const float4*)(w.row[t] + BLK * Jd + 256);
    }
  }
}

__device__ __forceinline__ bf16x8 pol_to_bf16(const float4 lo, const float4 hi) {
  bf16x8 v;
  v[0] = (__bf16)lo.x, v[1] = (__bf16)lo.y, v[2] = (__bf16)lo.z, v[3] = (__bf16)lo.w;
  v[4] = (__bf16)hi.x, v[5] = (__bf16)hi.y, v[6] = (__bf16)hi.z, v[7] = (__bf16)hi.w;
  return v;
}

template <int NTILES, int KS, bool LAST, bool BF>
__device__ __forceinline__ void pol_layer(const float* __restrict__ X, const int ldx, PolW<NTILES, BF>& w,
                                          const float* __restrict__ b, const int Kp, const int wave, const int lane,
                                          float* __restrict__ Y, const int ldy, f32x4 (*out)[POL_MT]) {
  constexpr int NT = PolW<NTILES, BF>::NT;
  constexpr int BLK = PolW<NTILES, BF>::BLK;
  if (wave >= NTILES) return;                                // (more waves than tiles in the narrow layers: nothing to do)
  const int m = lane & 15, g = lane >> 4;
  f32x4 acc[POL_MT][NT][KS];
#pragma unroll
  for (int u = 0; u < POL_MT; u++)
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
      for (int s = 0; s < KS; s++) acc[u][t][s] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  const float* xrow = X + m * ldx + 8 * g;                   // m-tile u: + 16 u rows
  const int nJ = Kp / 32;                                    // (a multiple of KS for every layer of this architecture)
  for (int J = 0; J < nJ; J += POL_DEPTH) {
#pragma unroll
    for (int d = 0; d < POL_DEPTH; d++) {
      const int Jc = J + d;
      if (Jc < nJ) {                                         // (uniform)
        float4 a[POL_MT][2];
#pragma unroll
        for (int u = 0; u < POL_MT; u++)
          a[u][0] = *(const float4*)(xrow + 16 * u * ldx + 32 * Jc), a[u][1] = *(const float4*)(xrow + 16 * u * ldx + 32 * Jc + 4);
        float4 cur[NT][2];
#pragma unroll
        for (int t = 0; t < NT; t++) cur[t][0] = w.q[d][t][0], cur[t][1] = w.q[d][t][BF ? 0 : 1];
        if (Jc + POL_DEPTH < nJ) {                           // refill this slot for step Jc + DEPTH
#pragma unroll
          for (int t = 0; t < NT; t++) {
            w.q[d][t][0] = *(const float4*)(w.row[t] + BLK * (Jc + POL_DEPTH));
            if (!BF) w.q[d][t][BF ? 0 : 1] = *(const float4*)(w.row[t] + BLK * (Jc + POL_DEPTH) + 256);
          }
        }
        const int s = (KS == 2) ? (d & 1) : 0;               // (DEPTH is even: step parity == slot parity)
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
          for (int u = 0; u < POL_MT; u++) {
            f32x4 c = acc[u][t][s];
            if (BF) {
              const bf16x8 av = pol_to_bf16(a[u][0], a[u][1]);
              const bf16x8 bv = *(const bf16x8*)&cur[t][0];
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c, 0, 0, 0);
            } else {
              c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][0].x, cur[t][0].x, c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][0].y, cur[t][0].y, c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][0].z, cur[t][0].z, c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][0].w, cur[t][0].w, c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][1].x, cur[t][1].x, c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][1].y, cur[t][1].y, c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][1].z, cur[t][1].z, c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][1].w, cur[t][1].w, c, 0, 0, 0);
            }
            acc[u][t][s] = c;
          }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; t++) {
    if (wave + POL_WAVES * t >= NTILES) break;
    const int n = (wave + POL_WAVES * t) * 16 + m;           // D: column on the lane (lane & 15), rows 4 g + i in the registers
    const float bias = w.bias[t];
#pragma unroll
    for (int u = 0; u < POL_MT; u++) {
      f32x4 c = acc[u][t][0];
      if (KS == 2) c += acc[u][t][KS - 1];
      c += bias;
      if (LAST) {
        out[t][u] = c;
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) Y[(16 * u + 4 * g + i) * ldy + n] = pol_tanh(c[i]);
      }
    }
  }
}

// grid (ceil(ne / 16), 2): blockIdx.y = 0 the policy net, 1 the value net.
template <bool BF>
__global__ void __launch_bounds__(POL_THREADS, (POL_WAVES >= 8 ? 4 : 2)) k6_policy_act(PolicyArgs pa) {   // (two workgroups per CU)
  extern __shared__ __align__(16) unsigned char smem[];
  const auv_policy_io_t& io = pa.io;
  const int tid = threadIdx.x, wave = tid / AUV_WAVE, lane = tid % AUV_WAVE;
  const int net = blockIdx.y;
  const int r0 = blockIdx.x * POL_ROWS;                          // first row of the tile inside the slice
  const int K0 = io.obs_dim, K0p = pol_pad16(K0);
  const int ldx = K0p + 8, ld1 = POL_H1 + 8, ld2 = POL_H2 + 8, ld3 = POL_H3 + 8;
  float* X = (float*)smem;
  float* Y1 = X + pol_lds_x_floats(K0);
  float* Y2 = X;                                                 // (X is dead once layer 1 is through: a barrier lies in between)
  float* Y3 = Y1;                                                // (Y1 is dead once layer 2 is through)
  // position in the rollout and the generator's step counter: read by every workgroup before anybody moves them on
  const bool host_counts = pa.t_host >= 0;
  const long long t = host_counts ? pa.t_host : io.ctr[0];
  const unsigned long long gstep = (unsigned long long)(host_counts ? pa.gstep_host : io.ctr[1]);
  const int T = io.T;
  const int cnt = pa.ne;
  const size_t ld = (size_t)io.ld;                               // environments per rollout row
  const int col0 = pa.e0 - io.env_base + r0;                     // the tile's first column in a rollout row
  // ---- the transition the environment's last step completed: reward and done of step t - 1 ----
  // (requested here, stored at the end: load - store - load - store at the head of the kernel were two trips to memory in
  // front of everything wave 0 does, and wave 0 is the one that samples)
  const bool trans = net == 0 && t >= 1 && t <= T && tid < POL_ROWS && r0 + tid < cnt;
  float trans_r = 0.0f;
  uint8_t trans_d = 0;
  if (trans) trans_r = io.reward_in[pa.e0 + r0 + tid], trans_d = io.done_in[pa.e0 + r0 + tid];
  const bool act = t < T;                                        // t == T: the flush call after a rollout's last step
  if (act) {
    const float* P = io.params + (size_t)net * pol_net_floats(K0);
    const float* W1 = P;
    const float* b1 = W1 + (size_t)POL_H1 * K0p;
    const float* W2 = b1 + POL_H1;
    const float* b2 = W2 + (size_t)POL_H2 * POL_H1;
    const float* W3 = b2 + POL_H2;
    const float* b3 = W3 + (size_t)POL_H3 * POL_H2;
    const float* W4 = b3 + POL_H3;
    const float* b4 = W4 + (size_t)POL_OUT * POL_H3;
    if (BF) {
      // bf16 weights: the four matrices of a net back to back, half the floats each (biases and log_std stay in `params`)
      const float* Q = (const float*)io.params_bf16 + (size_t)net * (pol_net_floats(K0) - POL_H1 - POL_H2 - POL_H3 - POL_OUT) / 2;
      W1 = Q;
      W2 = W1 + (size_t)POL_H1 * K0p / 2;
      W3 = W2 + (size_t)POL_H2 * POL_H1 / 2;
      W4 = W3 + (size_t)POL_H3 * POL_H2 / 2;
    }
    PolW<POL_H1 / 16, BF> w1;
    PolW<POL_H2 / 16, BF> w2;
    PolW<POL_H3 / 16, BF> w3;
    PolW<1, BF> w4;
    pol_prefetch(w1, W1, b1, K0p, wave, lane);                       // (in flight while the observation tile is fetched)
    // ---- observation tile -> LDS (zero padded), and into the rollout (policy workgroup) ----
    // the tile's rows are consecutive rows of the observation buffer: one contiguous range, read two floats per lane
    // (rows of an even number of columns start 8-byte aligned and no pair straddles two rows)
    const int rows = (cnt - r0 < POL_ROWS) ? cnt - r0 : POL_ROWS;
    for (int idx = tid; idx < POL_ROWS * (ldx / 2); idx += POL_THREADS) *(float2*)(X + 2 * idx) = make_float2(0.0f, 0.0f);
    __syncthreads();
    const float* src = io.obs + (size_t)(pa.e0 + r0) * K0;
    float* dstO = (net == 0 && io.O) ? io.O + ((size_t)t * ld + col0) * K0 : nullptr;
    if ((K0 & 1) == 0) {
      for (int idx = 2 * tid; idx < rows * K0; idx += 2 * POL_THREADS) {
        const float2 v = *(const float2*)(src + idx);
        const int row = idx / K0, c = idx - row * K0;
        *(float2*)(X + row * ldx + c) = v;
        if (dstO) *(float2*)(dstO + idx) = v;
      }
    } else {
      for (int idx = tid; idx < rows * K0; idx += POL_THREADS) {
        const float v = src[idx];
        const int row = idx / K0, c = idx - row * K0;
        X[row * ldx + c] = v;
        if (dstO) dstO[idx] = v;
      }
    }
    __syncthreads();
    pol_prefetch(w2, W2, b2, POL_H1, wave, lane);                    // (the next layer's weights: in flight during this layer)
    pol_layer<POL_H1 / 16, 1, false, BF>(X, ldx, w1, b1, K0p, wave, lane, Y1, ld1, nullptr);
    __syncthreads();
    pol_prefetch(w3, W3, b3, POL_H2, wave, lane);
    pol_layer<POL_H2 / 16, 1, false, BF>(Y1, ld1, w2, b2, POL_H1, wave, lane, Y2, ld2, nullptr);
    __syncthreads();
    pol_prefetch(w4, W4, b4, POL_H3, wave, lane);
    // (the sampling's log-std with them: ahead of the barrier, so that the request is not moved down to its use)
    const int nc = (lane & 15) < 2 ? (lane & 15) : 0;
    float ls = 0.0f;
    if (wave == 0 && net == 0) ls = (io.params + 2 * pol_net_floats(K0))[nc];
    pol_layer<POL_H3 / 16, 2, false, BF>(Y2, ld2, w3, b3, POL_H2, wave, lane, Y3, ld3, nullptr);
    __syncthreads();
    if (wave == 0) {
      f32x4 o[1][POL_MT];
      pol_layer<1, 2, true, BF>(Y3, ld3, w4, b4, POL_H3, 0, lane, nullptr, 0, o);
      const int n = lane & 15, g = lane >> 4;
      if (net == 0) {
        // ---- diagonal Gaussian: sample, log-probability, action ----
        const float sigma = expf(ls);
        // the action map of this lane's component: launch arguments picked by a select (indexed by the lane they are vector
        // loads from the argument segment, behind the stores of every row: four trips to memory in this epilogue)
        float cl0 = io.clip_lo[0], cl1 = io.clip_lo[1], ch0 = io.clip_hi[0], ch1 = io.clip_hi[1];
        float am0 = io.act_mid[0], am1 = io.act_mid[1], ah0 = io.act_half[0], ah1 = io.act_half[1];
        asm volatile("" : "+s"(cl0), "+s"(cl1), "+s"(ch0), "+s"(ch1), "+s"(am0), "+s"(am1), "+s"(ah0), "+s"(ah1));   // (or the selects become indexed loads again)
        const float clip_lo = nc ? cl1 : cl0, clip_hi = nc ? ch1 : ch0, act_mid = nc ? am1 : am0, act_half = nc ? ah1 : ah0;
#pragma unroll
        for (int u = 0; u < POL_MT; u++)
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const int rr = 16 * u + 4 * g + i;                   // row inside the tile
            const int row = r0 + rr;
            const int e = pa.e0 + row;
            const float mu = o[0][u][i];
            const float eps = pol_gauss(io.seed, gstep, (unsigned int)e, (unsigned int)nc);
            const float a = mu + sigma * eps;
            const float z = (a - mu) / sigma;
            float lp = -0.5f * z * z - ls - POL_LOG_SQRT_2PI;
            lp += __shfl_xor(lp, 1, AUV_WAVE);                   // the two components sit on neighbouring lanes
            if (n < 2 && row < cnt) {
              const size_t q = (size_t)t * ld + col0 + rr;
              io.A[2 * q + n] = a;
              if (n == 0) io.LP[q] = lp;
              const float ac = fminf(fmaxf(a, clip_lo), clip_hi);
              io.actions_out[2 * (size_t)e + n] = act_mid + act_half * ac;
              if (io.mu_out) io.mu_out[2 * (size_t)row + n] = mu;
              if (io.eps_out) io.eps_out[2 * (size_t)row + n] = eps;
            }
          }
      } else if (n == 0) {
#pragma unroll
        for (int u = 0; u < POL_MT; u++)
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const int rr = 16 * u + 4 * g + i;
            if (r0 + rr < cnt) io.V[(size_t)t * ld + col0 + rr] = o[0][u][i];
          }
      }
    }
  }
  if (trans) {
    float r = trans_r;
    if (io.reward_clip > 0.0f) r = fminf(fmaxf(r, -io.reward_clip), io.reward_clip);
    io.R[(size_t)(t - 1) * ld + col0 + tid] = r * io.reward_scale;
    io.Dn[(size_t)(t - 1) * ld + col0 + tid] = trans_d ? 1.0f : 0.0f;
  }
  // ---- the rollout position and the generator's counter move on ----
  if (host_counts) {
    // the host named them (nobody in this launch reads io.ctr): one workgroup keeps the device copies in step, for a later
    // auv_policy_act on the same buffers
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
      if (t <= T) io.ctr[0] = t + 1;
      io.ctr[1] = (long long)(gstep + 1);
    }
    return;
  }
  // on the device (a launch that may sit in a captured graph): every workgroup has read them by the time it counts itself off,
  // the last one moves them.  (512 returning atomics on one word: ~4 of the 8 us an EMPTY launch of this grid takes -- why
  // auv_policy_rollout, a plain host loop, names the counters itself.)
  __syncthreads();
  if (tid == 0) {
    const unsigned long long total = (unsigned long long)gridDim.x * gridDim.y;
    const unsigned long long old = atomicAdd((unsigned long long*)&io.ctr[2], 1ull);
    if (old == total - 1) {
      io.ctr[2] = 0;
      if (t <= T) io.ctr[0] = t + 1;                             // (a flush call leaves T + 1: further calls do nothing)
      io.ctr[1] = (long long)(gstep + 1);
    }
  }
}

// Generalised advantage estimation over a rollout (what stable-baselines' PPO2 runner does on the host after n_steps,
// scripts/run.py:341-346: gamma 0.999, lam 0.98): one lane per environment walks its T transitions backwards, the loads of a
// wave are consecutive environments of one rollout row.  adv[t] = delta_t + gamma lam (1 - done_t) adv[t + 1],
// delta_t = r_t + gamma v_{t+1} (1 - done_t) - v_t; ret = adv + v.
__global__ void __launch_bounds__(256) k6_gae(const float* __restrict__ R, const float* __restrict__ V, const float* __restrict__ Dn,
                                             const float* __restrict__ last_v, const float gamma, const float lam,
                                             float* __restrict__ adv, float* __restrict__ ret, const int T, const int N) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  float nv = last_v[e], gae = 0.0f;
  for (int t = T - 1; t >= 0; t--) {
    const size_t q = (size_t)t * N + e;
    const float v = V[q], nd = 1.0f - Dn[q];
    const float delta = R[q] + gamma * nv * nd - v;
    gae = delta + gamma * lam * nd * gae;
    adv[q] = gae;
    ret[q] = gae + v;
    nv = v;
  }
}

}  // namespace

void auv_launch_gae(const float* R, const float* V, const float* Dn, const float* last_v, float gamma, float lam, float* adv, float* ret,
                    int T, int N, hipStream_t st) {
  hipLaunchKernelGGL(k6_gae, dim3((N + 255) / 256), dim3(256), 0, st, R, V, Dn, last_v, gamma, lam, adv, ret, T, N);
}

size_t auv_policy_param_floats_impl(int obs_dim) { return 2 * pol_net_floats(obs_dim) + 4; }
size_t auv_policy_lds_bytes(int obs_dim) { return pol_lds_bytes(obs_dim); }

hipError_t auv_policy_prepare(int obs_dim) {
  const size_t b = pol_lds_bytes(obs_dim);
  if (b <= 64 * 1024) return hipSuccess;
  hipError_t e = hipFuncSetAttribute((const void*)k6_policy_act<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)k6_policy_act<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
}

void auv_launch_policy(const auv_policy_io_t& io, int e0, int ne, hipStream_t st, long long t_host, long long gstep_host) {
  PolicyArgs pa;
  pa.io = io, pa.e0 = e0, pa.ne = ne;
  pa.t_host = t_host, pa.gstep_host = gstep_host;
  const dim3 grid((ne + POL_ROWS - 1) / POL_ROWS, 2), block(POL_THREADS);
  if (io.params_bf16) hipLaunchKernelGGL(k6_policy_act<true>, grid, block, pol_lds_bytes(io.obs_dim), st, pa);
  else hipLaunchKernelGGL(k6_policy_act<false>, grid, block, pol_lds_bytes(io.obs_dim), st, pa);
}
