// HBM read rate of the cu GEMM's activation access pattern as a function of the bytes a CU keeps in flight.
// One persistent workgroup per CU walks 32-pixel tiles of a (B, 2C, HW) float tensor: a tile is 2C rows x 128 bytes at a row
// stride of HW*4 bytes; wave w loads rows [w*KPW, (w+1)*KPW) of both inputs with row-wise 16-byte loads (8 lanes = one line),
// R tiles ahead (R * 32 VGPRs per lane at 8 waves), and only sums what it loaded.  Prints us and TB/s for
// R = 1..3, 8 and 4 waves per workgroup, interleaved and contiguous tile order, default and nt cache policy.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/tile_read_rate.hip -o experiments/build/tile_read_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int WAVES, int R, int AUX, int INPUTS, int WRITE = 0>
__global__ __launch_bounds__(WAVES * 64, 1) void tile_read(const float* __restrict__ x, float* __restrict__ sink, int c, int hw, int nb, int contig, float* __restrict__ y = nullptr) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 3, q = lane & 7;
  const int kpw = c / WAVES;                       // rows per wave and input
  constexpr int MAXG = 2;                          // 32-row groups per wave (c = 256: 1 at 8 waves, 2 at 4)
  const int groups = kpw / 32;
  const int nwt = hw / 32, total = nb * nwt;
  const int per = (total + gridDim.x - 1) / gridDim.x;
  const int t0 = contig ? blockIdx.x * per : blockIdx.x, t_end = contig ? min(total, t0 + per) : total, step = contig ? 1 : gridDim.x;
  const int voff = ((wv * kpw + 4 * g) * hw + 4 * q) * 4, row_bytes = hw * 4;
  f4 r[R][INPUTS][MAXG][4];
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  auto issue = [&](auto slot, int t) {
    constexpr int S = decltype(slot)::value;
    const int b = t / nwt, p0 = (t - b * nwt) * 32;
#pragma unroll
    for (int i = 0; i < INPUTS; ++i) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + ((size_t)b * 2 + i) * c * hw), 0, (unsigned)((size_t)c * hw * 4), 0x00020000);
#pragma unroll
      for (int gi = 0; gi < MAXG; ++gi)
        if (gi < groups)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            r[S][i][gi][j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (32 * gi + j) * row_bytes + p0 * 4, AUX));
    }
  };
  auto consume = [&](auto slot, int t = 0) {
    constexpr int S = decltype(slot)::value;
    if (WRITE) {   // the epilogue's store pattern: the wave's 32 * groups output rows of the tile, 16 bytes per lane, 8 lanes per line
      const int b = t / nwt, p0 = (t - b * nwt) * 32;
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y + (size_t)b * c * hw, 0, (unsigned)((size_t)c * hw * 4), 0x00020000);
      const int svoff = ((wv * kpw + g) * hw + 4 * q) * 4;
#pragma unroll
      for (int gi = 0; gi < MAXG; ++gi)
        if (gi < groups)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const f4 v = r[S][0][gi][k] + (INPUTS > 1 ? r[S][INPUTS - 1][gi][k] : acc);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), ry, svoff, (32 * gi + 8 * k) * row_bytes + p0 * 4, 0);
          }
      return;
    }
#pragma unroll
    for (int i = 0; i < INPUTS; ++i)
#pragma unroll
      for (int gi = 0; gi < MAXG; ++gi)
        if (gi < groups)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc += r[S][i][gi][j];
  };
  int t = t0;
  // R tiles in flight; the loop body handles R tiles so that slots are compile-time
  auto ic = [](auto v) { return v; };
  if (R >= 1 && t + 0 * step < t_end) issue(std::integral_constant<int, 0>{}, t + 0 * step);
  if (R >= 2 && t + 1 * step < t_end) issue(std::integral_constant<int, 1 % R>{}, t + 1 * step);
  if (R >= 3 && t + 2 * step < t_end) issue(std::integral_constant<int, 2 % R>{}, t + 2 * step);
  for (; t < t_end; t += R * step) {
    if (t < t_end) { consume(std::integral_constant<int, 0>{}, t); if (t + R * step < t_end) issue(std::integral_constant<int, 0>{}, t + R * step); }
    if (R >= 2 && t + step < t_end) { consume(std::integral_constant<int, 1 % R>{}, t + step); if (t + (R + 1) * step < t_end) issue(std::integral_constant<int, 1 % R>{}, t + (R + 1) * step); }
    if (R >= 3 && t + 2 * step < t_end) { consume(std::integral_constant<int, 2 % R>{}, t + 2 * step); if (t + (R + 2) * step < t_end) issue(std::integral_constant<int, 2 % R>{}, t + (R + 2) * step); }
  }
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e38f) sink[threadIdx.x] = acc.x;
}

__global__ void linear_read(const f4* __restrict__ x, float* __restrict__ sink, size_t n4) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += __builtin_nontemporal_load(x + i);
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e38f) sink[threadIdx.x] = acc.x;
}

template <class K>
static double time_kernel(K launch, int reps = 11) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); launch();
  CK(hipDeviceSynchronize());
  std::vector<float> ms(reps);
  for (int r = 0; r < reps; ++r) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[r], e0, e1)); }
  std::sort(ms.begin(), ms.end());
  return ms[reps / 2] * 1e3;
}

int main() {
  const int B = 4, C = 256, HW = 40000;
  const size_t n = (size_t)B * 2 * C * HW;
  float *x, *sink;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&sink, 4096 * 4));
  CK(hipMemset(x, 0, n * 4));
  int cus = 256;
#define RUN(WAVES, R, AUX, INPUTS, contig)                                                                                          \
  do {                                                                                                                              \
    const double us = time_kernel([&] { hipLaunchKernelGGL((tile_read<WAVES, R, AUX, INPUTS>), dim3(cus), dim3(WAVES * 64), 0, 0, x, sink, C, HW, B, contig); }); \
    printf("waves %d  tiles in flight %d (%3d KB per CU)  aux %d  inputs %d  %-12s %7.1f us  %5.2f TB/s\n", WAVES, R, R * 32 * INPUTS, AUX, INPUTS,      \
           contig ? "contiguous" : "interleaved", us, (double)B * INPUTS * C * HW * 4 / us / 1e6);                                  \
  } while (0)
  {
    const double us = time_kernel([&] { hipLaunchKernelGGL(linear_read, dim3(cus * 8), dim3(512), 0, 0, (const f4*)x, sink, n / 4); });
    printf("linear nt read of the whole tensor (656 MB)  %7.1f us  %5.2f TB/s\n", us, (double)n * 4 / us / 1e6);
  }
  float* y;
  CK(hipMalloc(&y, (size_t)B * C * HW * 4));
#define RUNW(WAVES, R, AUX, INPUTS, contig)                                                                                         \
  do {                                                                                                                              \
    const double us = time_kernel([&] { hipLaunchKernelGGL((tile_read<WAVES, R, AUX, INPUTS, 1>), dim3(cus), dim3(WAVES * 64), 0, 0, x, sink, C, HW, B, contig, y); }); \
    printf("READ+WRITE waves %d  tiles in flight %d  aux %d  inputs %d  %-12s %7.1f us  %5.2f TB/s\n", WAVES, R, AUX, INPUTS,      \
           contig ? "contiguous" : "interleaved", us, (double)B * (INPUTS + 1) * C * HW * 4 / us / 1e6);                            \
  } while (0)
  RUNW(8, 1, 0, 2, 0); RUNW(8, 2, 0, 2, 0); RUNW(8, 3, 0, 2, 0);
  RUNW(8, 1, 0, 2, 1); RUNW(8, 2, 0, 2, 1); RUNW(8, 3, 0, 2, 1);
  RUNW(8, 2, 2, 2, 1); RUNW(4, 2, 0, 2, 1); RUNW(8, 2, 0, 1, 1); RUNW(8, 1, 0, 1, 1);
  RUN(8, 1, 0, 2, 0); RUN(8, 2, 0, 2, 0); RUN(8, 3, 0, 2, 0);
  RUN(8, 1, 0, 2, 1); RUN(8, 2, 0, 2, 1); RUN(8, 3, 0, 2, 1);
  RUN(8, 2, 2, 2, 0); RUN(8, 3, 2, 2, 0);
  RUN(4, 1, 0, 2, 0); RUN(4, 2, 0, 2, 0); RUN(4, 3, 0, 2, 0);
  RUN(8, 1, 0, 1, 0); RUN(8, 2, 0, 1, 0); RUN(8, 3, 0, 1, 0);
  return 0;
}
