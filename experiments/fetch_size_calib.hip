// FETCH_SIZE calibration for the access patterns of the MGHS writer.
// MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports HALF the bytes of a wide coalesced streaming read (16 B per lane) and is
// "uncalibrated" for other access widths.  profiles/collect.sh doubles FETCH_SIZE for every kernel; the writer's reads, however,
// are not 16-byte streams: every (segment, channel part) workgroup reads 64-byte pieces (16 channels x 4 B, one 4-byte load per
// lane) of 256-byte vsum rows.  This program reads a KNOWN number of bytes in three patterns, each as its own kernel so that
// `rocprofv3 --pmc FETCH_SIZE` reports them separately:
//   read_wide16     16 B per lane, fully coalesced                                   (the guide's pattern: expect bytes / 2)
//   read_piece64    4 parts; part q reads bytes [64 q, 64 q + 64) of every 256-B row, 4 B per lane (the writer's vsum reads)
//   read_dword      4 B per lane, fully coalesced (256 B per wave-instruction)
// and, for WRITE_SIZE, write_wide16_nt (the writer's 16-byte non-temporal stores) over the same byte count.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 experiments/fetch_size_calib.hip -o experiments/build/fetch_size_calib
// run  : rocprofv3 --kernel-trace --pmc FETCH_SIZE -- experiments/build/fetch_size_calib   (then WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void read_wide16(const f4* __restrict__ x, size_t n4, float* sink) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += x[i];
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e38f) sink[0] = acc.x;
}

// rows of 64 floats; block (row group, part): lanes 0-15 of a wave read the 16 floats of one row's part, 4 rows per wave-instruction
__global__ __launch_bounds__(256) void read_piece64(const float* __restrict__ x, size_t rows, float* sink) {
  const int part = blockIdx.x & 3;
  const size_t blk = blockIdx.x >> 2, nblk = gridDim.x >> 2;
  float acc = 0.f;
  const int lane = threadIdx.x & 15, sub = threadIdx.x >> 4;       // 16 rows per block-iteration
  for (size_t r = blk * 16 + sub; r < rows; r += nblk * 16) acc += x[r * 64 + part * 16 + lane];
  if (acc == 1.2345e38f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void read_dword(const float* __restrict__ x, size_t n, float* sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += x[i];
  if (acc == 1.2345e38f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void write_wide16_nt(f4* __restrict__ x, size_t n4) {
  const f4 z = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(z, x + i);
}

int main() {
  const size_t bytes = 768ull << 20;      // 768 MiB: three times the Infinity Cache
  float *x, *sink;
  CK(hipMalloc(&x, bytes));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(x, 0, bytes));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(read_wide16, dim3(8192), dim3(256), 0, 0, (const f4*)x, bytes / 16, sink);
    hipLaunchKernelGGL(read_piece64, dim3(8192), dim3(256), 0, 0, x, bytes / 256, sink);
    hipLaunchKernelGGL(read_dword, dim3(8192), dim3(256), 0, 0, x, bytes / 4, sink);
    hipLaunchKernelGGL(write_wide16_nt, dim3(8192), dim3(256), 0, 0, (f4*)x, bytes / 16);
  }
  CK(hipDeviceSynchronize());
  printf("each kernel moved %zu bytes (%.1f MiB = %.0f KB in the counters' unit)\n", bytes, bytes / 1048576.0, bytes / 1024.0);
  return 0;
}
