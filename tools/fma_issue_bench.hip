// fma_issue_bench.hip -- how many cycles does a SIMD of gfx950 spend per wave64 VALU instruction, by type and by the
// number of waves it has to choose from?  (DESIGN.md's "fp32 cannot pay on this part" rested on
// SQ_ACTIVE_INST_VALU, which counts quad-cycles and cannot tell a 2-cycle from a 4-cycle issue.)
//
// Each wave runs ITERS iterations of UNROLL dependent-free instructions of one kind (8 independent accumulator chains,
// so no instruction waits for the result of the one before it), bracketed by s_memtime; one workgroup of 256 w threads per
// CU (80 KiB of LDS requested: no second workgroup fits), i.e. w waves on every SIMD.  Output per (kind, w):
//   cyc/inst/wave  = cycles of one wave / instructions of one wave          (latency view of one wave)
//   cyc/inst/SIMD  = cycles / (w * instructions)                            (issue cost: what a SIMD pays per instruction)
// build:  hipcc -O2 --offload-arch=gfx950 tools/fma_issue_bench.hip -o tools/fma_issue_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITERS 256
#define CHAINS 8
#define REPS 8      // CHAINS * REPS instructions per loop iteration: the loop's own scalar instructions are < 3 % of the issue slots

enum Kind { FMA_F32 = 0, FMA_F64 = 1, PK_FMA_F32 = 2, MUL_F64 = 3, ADD_F64 = 4, FMA_F32_DEP = 5, FMA_F64_DEP = 6, N_KINDS = 7 };
static const char* kind_name[N_KINDS] = {"v_fma_f32", "v_fma_f64", "v_pk_fma_f32", "v_mul_f64", "v_add_f64", "v_fma_f32 (one dependent chain)",
                                         "v_fma_f64 (one dependent chain)"};

template <int KIND>
__global__ void __launch_bounds__(1024) k_issue(unsigned long long* __restrict__ out, float seed) {
  extern __shared__ unsigned char smem[];
  if (threadIdx.x == 0) smem[0] = 1;
  float a32[CHAINS];
  double a64[CHAINS];
  float2 ap[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; i++) a32[i] = seed + i, a64[i] = seed + i, ap[i] = make_float2(seed + i, seed - i);
  const float b32 = 1.0000001f, c32 = 1e-9f;
  const double b64 = 1.0000000001, c64 = 1e-12;
  const float2 bp = make_float2(b32, b32), cp = make_float2(c32, c32);
  __syncthreads();
  const unsigned long long t0 = clock64(), r0 = wall_clock64();
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int rep = 0; rep < REPS; rep++)
#pragma unroll
    for (int i = 0; i < CHAINS; i++) {
      if constexpr (KIND == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a32[i]) : "v"(b32), "v"(c32));
      if constexpr (KIND == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a64[i]) : "v"(b64), "v"(c64));
      if constexpr (KIND == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(ap[i]) : "v"(bp), "v"(cp));
      if constexpr (KIND == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a64[i]) : "v"(b64));
      if constexpr (KIND == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a64[i]) : "v"(c64));
      if constexpr (KIND == FMA_F32_DEP) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a32[0]) : "v"(b32), "v"(c32));
      if constexpr (KIND == FMA_F64_DEP) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a64[0]) : "v"(b64), "v"(c64));
    }
  }
  const unsigned long long t1 = clock64(), r1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CHAINS; i++) s += a32[i] + (float)a64[i] + ap[i].x + ap[i].y;
  if (s == 12345.678f) out[0] = 0;   // keep the chains alive
  if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 16 + (threadIdx.x >> 6)] = ((r1 - r0) << 32) | (t1 - t0);
}

template <int KIND>
static void run(int w, unsigned long long* d_out, int n_cu, double clock_ratio) {
  const int threads = 256 * w, waves = 4 * w;
  hipFuncSetAttribute((const void*)k_issue<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipMemset(d_out, 0, (1 + n_cu * 16) * sizeof(unsigned long long));
  hipLaunchKernelGGL(k_issue<KIND>, dim3(n_cu), dim3(threads), 80 * 1024, 0, d_out, 1.0f);   // warm-up (code object, clocks)
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_issue<KIND>, dim3(n_cu), dim3(threads), 80 * 1024, 0, d_out, 1.0f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(1 + n_cu * 16);
  hipMemcpy(h.data(), d_out, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double sum = 0, mx = 0, wall = 0;
  int cnt = 0;
  for (int b = 0; b < n_cu; b++)
    for (int k = 0; k < waves; k++) {
      const double c = (double)(h[1 + b * 16 + k] & 0xffffffffull);
      wall += (double)(h[1 + b * 16 + k] >> 32) * 10.0;     // wall_clock64: 100 MHz -> ns
      sum += c, cnt++;
      if (c > mx) mx = c;
    }
  const double inst = (double)ITERS * CHAINS * REPS;
  const double ticks_per_inst = sum / cnt / inst;   // clock64 ticks (s_memtime) per instruction of one wave
  printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ticks_per_inst_per_wave\": %.3f, \"ticks_per_inst_per_simd\": %.3f, "
         "\"max_over_mean\": %.3f, \"kernel_us\": %.1f, \"ns_per_inst_per_simd_by_kernel_time\": %.4f, \"ns_per_inst_per_simd_by_wave_wall_clock\": %.4f, "
         "\"clock64_ghz\": %.3f}\n",
         kind_name[KIND], w, ticks_per_inst, ticks_per_inst / w, mx / (sum / cnt), ms * 1e3, ms * 1e6 / (inst * w), wall / cnt / inst / w,
         sum / wall);
  (void)clock_ratio;
  hipEventDestroy(e0), hipEventDestroy(e1);
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  // argv[1]: workgroups (= busy CUs); default all of them.  With ONE busy CU the chip is far from its power limit and the
  // shader clock stays at its maximum: that run shows the architectural issue cycles, the all-CU run what a full chip sustains
  const int n_cu = argc > 1 ? atoi(argv[1]) : p.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"note\": \"clock64 = s_memtime; kernel_us / (ITERS * CHAINS * REPS * w) gives ns per instruction per SIMD "
         "independent of the counter's rate: at f GHz a 4-cycle issue is 4 / f ns\"}\n",
         p.gcnArchName, n_cu, p.clockRate);
  unsigned long long* d_out = nullptr;
  hipMalloc((void**)&d_out, (1 + n_cu * 16) * sizeof(unsigned long long));
  for (int w : {1, 2, 4}) {
    run<FMA_F32>(w, d_out, n_cu, 1.0);
    run<FMA_F64>(w, d_out, n_cu, 1.0);
    run<PK_FMA_F32>(w, d_out, n_cu, 1.0);
    run<MUL_F64>(w, d_out, n_cu, 1.0);
    run<ADD_F64>(w, d_out, n_cu, 1.0);
    run<FMA_F32_DEP>(w, d_out, n_cu, 1.0);
    run<FMA_F64_DEP>(w, d_out, n_cu, 1.0);
  }
  hipFree(d_out);
  return 0;
}
