// How many bytes per clock can ONE CU pull through its vector-memory path when the data is cache-resident?
// 256 workgroups (one per CU, 512 threads), each reading its own 512 KB window (L2-resident after the first sweep) `reps` times:
//   A  32-bit loads, a wave instruction = 2 x 128-byte rows (the access shape of pw_gemm_res_kernel's activation loads)
//   B  128-bit loads, a wave instruction = 1 KB contiguous
//   C  as A, but every workgroup reads the SAME window (L2 hit for all but the first)
// Prints GB/s per CU and bytes per clock at the measured kernel time (shader clock from wall_clock / s_memtime is not used: the
// rate is given against 2.4 GHz nominal).
// build: hipcc --offload-arch=gfx950 -O3 experiments/cu_load_rate.hip -o experiments/build/cu_load_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 512;
__global__ __launch_bounds__(kThreads) void load32(const float* __restrict__ src, float* __restrict__ sink, size_t window_floats, int reps, int shared_window) {
  const float* p = src + (shared_window ? 0 : (size_t)blockIdx.x * window_floats);
  float acc = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // lane (r = lane & 31, h = lane >> 5): row h of a pair of 32-float rows 8 rows apart, like the MFMA A-fragment loads
  for (int rep = 0; rep < reps; ++rep)
    for (size_t base = (size_t)wave * 512; base + 512 <= window_floats; base += 8 * 512) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = p[base + (size_t)j * 64 + (lane >> 5) * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += v[j];
    }
  if (acc == 1.2345e38f) sink[blockIdx.x] = acc;
}
__global__ __launch_bounds__(kThreads) void load128(const f4* __restrict__ src, float* __restrict__ sink, size_t window_f4, int reps, int shared_window) {
  const f4* p = src + (shared_window ? 0 : (size_t)blockIdx.x * window_f4);
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int rep = 0; rep < reps; ++rep)
    for (size_t i = threadIdx.x; i + 3 * kThreads < window_f4; i += 4 * kThreads) {
      const f4 a = p[i], b = p[i + kThreads], c = p[i + 2 * kThreads], d = p[i + 3 * kThreads];
      acc += (a + b) + (c + d);
    }
  if (acc.x + acc.y == 1.2345e38f) sink[blockIdx.x] = acc.x;
}
int main() {
  const int cus = 256;
  const size_t window = 512 * 1024;                    // bytes per workgroup
  const int reps = 64;
  float *buf, *sink;
  hipMalloc(&buf, cus * window); hipMalloc(&sink, 4096);
  hipMemset(buf, 0, cus * window);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes_per_cu = (double)window * reps;
    printf("%-44s %8.3f ms  %7.1f GB/s per CU  %6.1f B/clk per CU (at 2.4 GHz)  %6.2f TB/s aggregate\n", name, ms, bytes_per_cu / ms / 1e6,
           bytes_per_cu / (ms * 1e-3) / 2.4e9, bytes_per_cu * cus / ms / 1e9);
  };
  run("A 32-bit loads, own 512 KB window", [&] { hipLaunchKernelGGL(load32, dim3(cus), dim3(kThreads), 0, 0, buf, sink, window / 4, reps, 0); });
  run("B 128-bit loads, own 512 KB window", [&] { hipLaunchKernelGGL(load128, dim3(cus), dim3(kThreads), 0, 0, (const f4*)buf, sink, window / 16, reps, 0); });
  run("C 32-bit loads, one window for all CUs", [&] { hipLaunchKernelGGL(load32, dim3(cus), dim3(kThreads), 0, 0, buf, sink, window / 4, reps, 1); });
  run("D 128-bit loads, one window for all CUs", [&] { hipLaunchKernelGGL(load128, dim3(cus), dim3(kThreads), 0, 0, (const f4*)buf, sink, window / 16, reps, 1); });
  return 0;
}
