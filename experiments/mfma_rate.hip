// Rate of v_mfma_f32_32x32x16_bf16 on one SIMD as issued by the cu GEMM: W waves per SIMD, each a chain of dependent MFMAs on
// NACC accumulators; random or zero operand data (power / clock), shader clocks by s_memtime next to wall time by HIP events.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/mfma_rate.hip -o experiments/build/mfma_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include <cstring>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int WAVES, int NACC, int NREG>
__global__ __launch_bounds__(WAVES * 64, 1) void mfma_rate(const u4* __restrict__ data, float* __restrict__ out, long long* __restrict__ clocks, int iters) {
  u4 a[NREG], b[2];
  for (int i = 0; i < NREG; ++i) a[i] = data[(threadIdx.x + 64 * i) % 4096];
  b[0] = data[(threadIdx.x * 7 + 1) % 4096]; b[1] = data[(threadIdx.x * 3 + 5) % 4096];
  f16v acc[NACC];
  for (int k = 0; k < NACC; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NREG; ++i)
      acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a[i]), __builtin_bit_cast(bf8, b[i & 1]), acc[i % NACC], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int k = 0; k < NACC; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
  if (s == 1.2345e38f) out[threadIdx.x] = s;
  if (threadIdx.x == 0) clocks[blockIdx.x] = t1 - t0;
}

int main() {
  std::vector<unsigned> h(4096 * 4);
  u4* d; float* out; long long* clk;
  CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&out, 4096)); CK(hipMalloc(&clk, 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rnd = 0; rnd < 2; ++rnd) {
    srand(1);
    for (auto& v : h) { // bf16 pairs of moderate magnitude, or zeros
      if (!rnd) { v = 0; continue; }
      auto one = [] { float f = (float)rand() / RAND_MAX * 2.f - 1.f; union { float f_; unsigned u_; } cv; cv.f_ = f; return cv.u_ >> 16; };
      v = one() | (one() << 16);
    }
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
#define RUN(WAVES, NACC, NREG)                                                                                                     \
    do {                                                                                                                           \
      const int iters = 4000 / NREG * 8;                                                                                           \
      hipLaunchKernelGGL((mfma_rate<WAVES, NACC, NREG>), dim3(256), dim3(WAVES * 64), 0, 0, d, out, clk, iters);                   \
      CK(hipDeviceSynchronize());                                                                                                  \
      CK(hipEventRecord(e0));                                                                                                      \
      hipLaunchKernelGGL((mfma_rate<WAVES, NACC, NREG>), dim3(256), dim3(WAVES * 64), 0, 0, d, out, clk, iters);                   \
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                                                         \
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));                                                                              \
      long long c[256]; CK(hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost));                                                   \
      const double n_per_simd = (double)iters * NREG * WAVES / 4;                                                                  \
      printf("%-6s waves/CU %2d  accumulators %d  distinct A regs %2d: %7.1f us  %6.0f TFLOP/s  %5.1f shader clocks per MFMA per SIMD (counter: %.0f MHz)\n", \
             rnd ? "random" : "zeros", WAVES, NACC, NREG, ms * 1e3, 256.0 * 4 * n_per_simd * 32768 / (ms * 1e-3) / 1e12,          \
             (double)c[7] / n_per_simd, (double)c[7] / (ms * 1e3));                                                                \
    } while (0)
    RUN(4, 1, 32); RUN(8, 1, 32); RUN(8, 2, 32); RUN(8, 4, 32); RUN(16, 1, 32); RUN(16, 4, 8); RUN(4, 4, 32);
  }
  return 0;
}
