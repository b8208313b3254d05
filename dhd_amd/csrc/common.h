// Shared helpers for the gfx950 kernels of libdhd_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dhd_amd.h"

#define DHD_WAVE 64

// Launch-and-report: every entry point returns the first hipError_t it sees (as a positive int).
#define DHD_LAUNCH_CHECK()                      \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

#define DHD_HIP(call)                           \
  do {                                          \
    hipError_t e__ = (call);                    \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

static inline hipStream_t dhd_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int dhd_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Sum over the `width` (power of two, <= 64) lanes that share lane_id / width.
__device__ __forceinline__ float group_sum(float v, int width) {
  for (int m = width >> 1; m > 0; m >>= 1) v += __shfl_xor(v, m, DHD_WAVE);
  return v;
}

__device__ __forceinline__ int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, observed, speed only).
// Remap so that each XCD walks one contiguous range of logical tiles: neighbouring output rows
// then share an L2 and their partial cache lines meet there before write-back.
__device__ __forceinline__ int xcd_contiguous_tile(int block, int n_tiles) {
  const int nx = 8;
  int per = (n_tiles + nx - 1) / nx;
  int t = (block % nx) * per + block / nx;
  return t;  // may be >= n_tiles for the ragged tail: caller checks
}
