// Device code shared by the stand-alone lift kernels (lift.hip) and the fused prologue of dhd_mghs_lift
// (mghs_prepare.hip): height argmax -> band id, and 64 x 64 transposition tiles of the context features.
#pragma once
#include "common.h"

namespace dhd {

constexpr int kLiftBlock = 256;
constexpr int kMaxHeightBins = 128;

struct BandLut {
  uint8_t band[kMaxHeightBins];  // band id per height bin, precomputed on the host in float32
};

// create_mask_3 (lss_heightmap.py:561-563) on float32 heights: [h_min,thr1) [thr1,thr2) [thr2,h_max)
inline void make_band_lut(const float* height_range, int n_height, const float* mask_range, BandLut* lut) {
  const float h_min = mask_range[0], t1 = mask_range[1], t2 = mask_range[2], h_max = mask_range[3];
  for (int k = 0; k < n_height; ++k) {
    const float h = height_range[k];
    uint8_t b = 255;
    if (h >= h_min && h < t1) b = 0;
    if (h >= t1 && h < t2) b = 1;
    if (h >= t2 && h < h_max) b = 2;
    lut->band[k] = b;
  }
}

// kBandLanes lanes per pixel, each scanning every kBandLanes-th height bin, then a lane-group
// argmax that keeps torch.argmax's "first maximum wins".  Lane l of a group reads pixel p's bin
// k*kBandLanes + l: a wave touches kBandLanes bin planes x 8 consecutive pixels per step.
constexpr int kBandLanes = 8;

// `block` = index among the dhd_cdiv(n_pix_total * kBandLanes, kLiftBlock) blocks of this role
__device__ __forceinline__ void height_band_block(int block, const float* __restrict__ height, int n_pix_total, int n_height, int hw,
                                                  const BandLut& lut, uint8_t* __restrict__ band) {
  const int gid = block * kLiftBlock + threadIdx.x;
  const int p = gid / kBandLanes, sub = gid % kBandLanes;
  const bool ok = p < n_pix_total;
  const int pp = ok ? p : 0;
  const int bn = pp / hw, i = pp % hw;
  const float* src = height + (size_t)bn * n_height * hw + i;
  float best = -INFINITY;
  int arg = 0x7fffffff;
  for (int k = sub; k < n_height; k += kBandLanes) {
    float v = src[(size_t)k * hw];
    if (v > best || arg == 0x7fffffff) { best = v; arg = k; }
  }
#pragma unroll
  for (int m = 1; m < kBandLanes; m <<= 1) {
    float ob = __shfl_xor(best, m, DHD_WAVE);
    int oa = __shfl_xor(arg, m, DHD_WAVE);
    if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
  }
  if (ok && sub == 0) band[p] = lut.band[arg < n_height ? arg : 0];
}

typedef float lift_f4 __attribute__((ext_vector_type(4)));

// One 64 x 64 tile of (batch, rows, cols) -> (batch, cols, rows) through a padded LDS tile, scalar accesses.
__device__ __forceinline__ void transpose_tile(float (*tile)[65], const float* __restrict__ src, float* __restrict__ dst, int rows,
                                               int cols, int b, int r0, int c0) {
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* s = src + (size_t)b * rows * cols;
  float* d = dst + (size_t)b * rows * cols;
  for (int j = ty; j < 64; j += kLiftBlock / 64) {
    int r = r0 + j, c = c0 + tx;
    if (r < rows && c < cols) tile[j][tx] = s[(size_t)r * cols + c];
  }
  __syncthreads();
  for (int j = ty; j < 64; j += kLiftBlock / 64) {
    int c = c0 + j, r = r0 + tx;
    if (r < rows && c < cols) d[(size_t)c * rows + r] = tile[tx][j];
  }
}

// The same with 16-byte accesses on both sides (rows, cols multiples of 4; 16-byte aligned tensors): a thread moves four
// float4 in and four out, i.e. 64 bytes in flight per thread on either side of the barrier -- the tensors are a few MB and
// the kernel is a latency chain (load -> LDS -> barrier -> LDS -> store), so bytes per instruction are what counts (9 -> 5 us
// for the 4.3 MB context tensor of DHD-S at B = 4).
__device__ __forceinline__ void transpose4_tile(float (*tile)[65], const float* __restrict__ src, float* __restrict__ dst, int rows,
                                                int cols, int b, int r0, int c0) {
  const float* s = src + (size_t)b * rows * cols;
  float* d = dst + (size_t)b * rows * cols;
  lift_f4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {          // tile row j, columns 4 q .. 4 q + 3
    const int idx = threadIdx.x + k * kLiftBlock, j = idx >> 4, q = idx & 15;
    const int r = r0 + j, c = c0 + 4 * q;
    v[k] = (r < rows && c < cols) ? *reinterpret_cast<const lift_f4*>(s + (size_t)r * cols + c) : lift_f4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int idx = threadIdx.x + k * kLiftBlock, j = idx >> 4, q = idx & 15;
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[j][4 * q + e] = v[k][e];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {          // output row = source column j, output columns = source rows 4 q .. 4 q + 3
    const int idx = threadIdx.x + k * kLiftBlock, j = idx >> 4, q = idx & 15;
    const int c = c0 + j, r = r0 + 4 * q;
    if (c < cols && r < rows) {
      const lift_f4 w = {tile[4 * q][j], tile[4 * q + 1][j], tile[4 * q + 2][j], tile[4 * q + 3][j]};
      *reinterpret_cast<lift_f4*>(d + (size_t)c * rows + r) = w;
    }
  }
}

inline bool transpose_vectorisable(const void* src, const void* dst, int rows, int cols) {
  return (rows & 3) == 0 && (cols & 3) == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0;
}

}  // namespace dhd
