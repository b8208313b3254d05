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

// Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8: observed behaviour, used
// for speed only).  Tiles are handed out in groups of `group` consecutive tiles per XCD, groups
// round-robin over the XCDs: a group of 4 rows of a 200-float-wide tensor is exactly 25 cache
// lines, so every partially written line is completed inside one XCD's L2, while heavy and light
// rows still spread evenly over all 256 CUs.  Bijective on [0, 8*group*ceil(n/(8*group))).
__device__ __forceinline__ int xcd_grouped_tile(int block, int group) {
  const int xcd = block & 7, i = block >> 3;
  return ((i / group) * 8 + xcd) * group + (i % group);
}
static inline int xcd_grouped_blocks(int n_tiles, int group) {
  const int q = 8 * group;
  return (n_tiles + q - 1) / q * q;
}

// Wave64 sum with DPP cross-lane adds (no LDS traffic); the total lands in lane 63 and is
// broadcast through an SGPR.  quad_perm x2, row_shr:4, row_shr:8, row_bcast:15, row_bcast:31.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
  return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum_bcast(float v) {
  v = dpp_add<0xB1, 0xf>(v);
  v = dpp_add<0x4E, 0xf>(v);
  v = dpp_add<0x114, 0xf>(v);
  v = dpp_add<0x118, 0xf>(v);
  v = dpp_add<0x142, 0xa>(v);
  v = dpp_add<0x143, 0xc>(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
