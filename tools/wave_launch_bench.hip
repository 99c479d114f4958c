// wave_launch_bench.hip -- what does it cost to START a one-wave workgroup of the step kernel's footprint (64 threads, 128 VGPRs,
// ~10 KiB of LDS => 16 per CU, 4096 slots on the chip), and what does a persistent wave pay for taking its next work item from a
// ticket counter instead?  The multi-step launch runs 9216 such workgroups per step of 4096 environments, ~10 us each; if the slot
// stands empty for a microsecond between two of them, that is a tenth of the step.
//
// Every work item "works" for W us (a wall-clock wait, no memory traffic).  Three shapes process the same N items:
//   dispatch    one workgroup per item (what k_step_multi does): N workgroups through the hardware dispatcher
//   tickets     G persistent workgroups, each takes items from ONE agent-scope atomic counter until it runs out
//   tickets+1   the same, the next ticket requested BEFORE the current item is worked on (its latency hidden)
//   static      G persistent workgroups, workgroup g takes items g, g + G, g + 2G, ... (no atomics)
// Output per shape: total time, time per item-slot (total * slots / N) and the overhead over W.
// build:  hipcc -O2 --offload-arch=gfx950 tools/wave_launch_bench.hip -o tools/wave_launch_bench
// usage:  ./tools/wave_launch_bench [W us = 10] [items per slot = 144] [LDS bytes = 10016]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void touch_regs() {      // make the kernel ask for 128 VGPRs, as the step kernel does
  asm volatile("v_mov_b32 v127, 0" ::: "v127");
}

__device__ __forceinline__ void work(unsigned long long ticks, unsigned char* smem) {
  const unsigned long long t0 = wall_clock64();
  smem[threadIdx.x] = (unsigned char)t0;
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}

__global__ void __launch_bounds__(64) k_dispatch(unsigned long long ticks, unsigned* sink) {
  extern __shared__ unsigned char smem[];
  touch_regs();
  work(ticks, smem);
  if (smem[threadIdx.x] == 255 && ticks == 1) sink[0] = 1;
}

template <int MODE>      // 0 tickets, 1 tickets with the next one requested ahead, 2 static
__global__ void __launch_bounds__(64) k_persistent(unsigned long long ticks, unsigned* counter, unsigned n_items, unsigned* sink) {
  extern __shared__ unsigned char smem[];
  touch_regs();
  unsigned done = 0;
  if (MODE == 2) {
    for (unsigned p = blockIdx.x; p < n_items; p += gridDim.x) work(ticks, smem), done++;
  } else {
    unsigned p = 0;
    if (threadIdx.x == 0) p = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    p = __builtin_amdgcn_readfirstlane(p);
    while (p < n_items) {
      unsigned nxt = 0;
      if (MODE == 1) {
        if (threadIdx.x == 0) nxt = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        work(ticks, smem);
      } else {
        work(ticks, smem);
        if (threadIdx.x == 0) nxt = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      p = __builtin_amdgcn_readfirstlane(nxt);
      done++;
    }
  }
  if (smem[threadIdx.x] == 255 && ticks == 1) sink[0] = done;
}

int main(int argc, char** argv) {
  const double w_us = argc > 1 ? atof(argv[1]) : 10.0;
  const int per_slot = argc > 2 ? atoi(argv[2]) : 144;
  const int lds = argc > 3 ? atoi(argv[3]) : 10016;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  int per_cu = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_dispatch, 64, lds));
  const unsigned slots = (unsigned)(cus * per_cu), n_items = slots * (unsigned)per_slot;
  const unsigned long long ticks = (unsigned long long)(w_us * 100.0);      // wall_clock64: 100 MHz
  unsigned *counter, *sink;
  CHECK(hipMalloc(&counter, 64));
  CHECK(hipMalloc(&sink, 64));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  auto report = [&](const char* shape, float ms, unsigned grid) {
    const double per_item = ms * 1e3 * slots / n_items;
    printf("{\"shape\": \"%s\", \"work_us\": %.2f, \"lds_bytes\": %d, \"workgroups_per_cu\": %d, \"slots\": %u, \"items\": %u, \"grid\": %u, "
           "\"total_ms\": %.4f, \"us_per_item_slot\": %.3f, \"overhead_us_per_item\": %.3f}\n",
           shape, w_us, lds, per_cu, slots, n_items, grid, ms, per_item, per_item - w_us);
  };
  for (int rep = 0; rep < 2; rep++) {      // (the second round is the one to read)
    float ms;
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k_dispatch, dim3(n_items), dim3(64), lds, 0, ticks, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (rep) report("dispatch", ms, n_items);
    for (int mode = 0; mode < 3; mode++) {
      CHECK(hipMemsetAsync(counter, 0, 4));
      CHECK(hipEventRecord(a));
      if (mode == 0) hipLaunchKernelGGL(k_persistent<0>, dim3(slots), dim3(64), lds, 0, ticks, counter, n_items, sink);
      if (mode == 1) hipLaunchKernelGGL(k_persistent<1>, dim3(slots), dim3(64), lds, 0, ticks, counter, n_items, sink);
      if (mode == 2) hipLaunchKernelGGL(k_persistent<2>, dim3(slots), dim3(64), lds, 0, ticks, counter, n_items, sink);
      CHECK(hipEventRecord(b));
      CHECK(hipEventSynchronize(b));
      CHECK(hipEventElapsedTime(&ms, a, b));
      if (rep) report(mode == 0 ? "tickets" : mode == 1 ? "tickets, next one requested ahead" : "static", ms, slots);
    }
  }
  CHECK(hipGetLastError());
  return 0;
}
