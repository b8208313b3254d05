// Small producers on the lift side of MGHS (gfx950): height argmax -> band id, and the
// NCHW <-> NHWC re-layout of the context features.  All HBM/L2-bound, a few MB per batch.
//
// Reference: models/necks/lss_heightmap.py:528-564 (height_feature_to_height_map +
// create_mask_3), :290 (feat.permute(0,1,3,4,2)) and :436-442 (the three masked copies of
// tran_feat, which the band id makes unnecessary).
#include "common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxHeightBins = 128;

struct BandLut {
  uint8_t band[kMaxHeightBins];  // band id per height bin, precomputed on the host in float32
};

// kBandLanes lanes per pixel, each scanning every kBandLanes-th height bin, then a lane-group
// argmax that keeps torch.argmax's "first maximum wins".  Lane l of a group reads pixel p's bin
// k*kBandLanes + l: a wave touches kBandLanes bin planes x 8 consecutive pixels per step.
constexpr int kBandLanes = 8;

__global__ __launch_bounds__(kBlock) void height_band_kernel(const float* __restrict__ height, int n_pix_total, int n_height,
                                                             int hw, BandLut lut, uint8_t* __restrict__ band) {
  const int gid = blockIdx.x * kBlock + threadIdx.x;
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

// (bn, C, hw) -> (bn, hw, C) through a padded 64x64 LDS tile; both sides coalesced.
__global__ __launch_bounds__(kBlock) void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows,
                                                           int cols) {
  // src is (batch, rows, cols) row-major, dst is (batch, cols, rows)
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* s = src + (size_t)b * rows * cols;
  float* d = dst + (size_t)b * rows * cols;
  for (int j = ty; j < 64; j += kBlock / 64) {
    int r = r0 + j, c = c0 + tx;
    if (r < rows && c < cols) tile[j][tx] = s[(size_t)r * cols + c];
  }
  __syncthreads();
  for (int j = ty; j < 64; j += kBlock / 64) {
    int c = c0 + j, r = r0 + tx;
    if (r < rows && c < cols) d[(size_t)c * rows + r] = tile[tx][j];
  }
}

// The same with 16-byte accesses on both sides (rows, cols multiples of 4; 16-byte aligned tensors): a thread moves four
// float4 in and four out, i.e. 64 bytes in flight per thread on either side of the barrier -- the tensors are a few MB and
// the kernel is a latency chain (load -> LDS -> barrier -> LDS -> store), so bytes per instruction are what counts (9 -> 5 us
// for the 4.3 MB context tensor of DHD-S at B = 4).
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(kBlock) void transpose4_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows,
                                                            int cols) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const float* s = src + (size_t)b * rows * cols;
  float* d = dst + (size_t)b * rows * cols;
  f4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {          // tile row j, columns 4 q .. 4 q + 3
    const int idx = threadIdx.x + k * kBlock, j = idx >> 4, q = idx & 15;
    const int r = r0 + j, c = c0 + 4 * q;
    v[k] = (r < rows && c < cols) ? *reinterpret_cast<const f4*>(s + (size_t)r * cols + c) : f4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int idx = threadIdx.x + k * kBlock, j = idx >> 4, q = idx & 15;
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[j][4 * q + e] = v[k][e];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {          // output row = source column j, output columns = source rows 4 q .. 4 q + 3
    const int idx = threadIdx.x + k * kBlock, j = idx >> 4, q = idx & 15;
    const int c = c0 + j, r = r0 + 4 * q;
    if (c < cols && r < rows) {
      const f4 w = {tile[4 * q][j], tile[4 * q + 1][j], tile[4 * q + 2][j], tile[4 * q + 3][j]};
      *reinterpret_cast<f4*>(d + (size_t)c * rows + r) = w;
    }
  }
}

int launch_transpose(const float* src, float* dst, int batch, int rows, int cols, void* stream) {
  if (!src || !dst || batch <= 0 || rows <= 0 || cols <= 0) return DHD_EINVAL;
  dim3 grid(dhd_cdiv(cols, 64), dhd_cdiv(rows, 64), batch);
  if ((rows & 3) == 0 && (cols & 3) == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0) {
    hipLaunchKernelGGL(transpose4_kernel, grid, dim3(kBlock), 0, dhd_stream(stream), src, dst, rows, cols);
    DHD_LAUNCH_CHECK();
    return DHD_OK;
  }
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(kBlock), 0, dhd_stream(stream), src, dst, rows, cols);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

}  // namespace

extern "C" {

int dhd_height_band(const float* height, int bn, int n_height, int fh, int fw, const float* height_range,
                    const float* mask_range, uint8_t* band, void* stream) {
  if (!height || !band || !height_range || !mask_range || bn <= 0 || fh <= 0 || fw <= 0) return DHD_EINVAL;
  if (n_height <= 0 || n_height > kMaxHeightBins) return DHD_EUNSUPPORTED;
  BandLut lut;
  // create_mask_3 (lss_heightmap.py:561-563) on float32 heights: [h_min,thr1) [thr1,thr2) [thr2,h_max)
  const float h_min = mask_range[0], t1 = mask_range[1], t2 = mask_range[2], h_max = mask_range[3];
  for (int k = 0; k < n_height; ++k) {
    const float h = height_range[k];
    uint8_t b = 255;
    if (h >= h_min && h < t1) b = 0;
    if (h >= t1 && h < t2) b = 1;
    if (h >= t2 && h < h_max) b = 2;
    lut.band[k] = b;
  }
  const int n = bn * fh * fw;
  hipLaunchKernelGGL(height_band_kernel, dim3(dhd_cdiv((long)n * kBandLanes, kBlock)), dim3(kBlock), 0, dhd_stream(stream), height, n,
                     n_height, fh * fw, lut, band);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

int dhd_feat_nchw_to_nhwc(const float* src, float* dst, int bn, int c, int hw, void* stream) {
  return launch_transpose(src, dst, bn, c, hw, stream);
}

int dhd_feat_nhwc_to_nchw(const float* src, float* dst, int bn, int c, int hw, void* stream) {
  return launch_transpose(src, dst, bn, hw, c, stream);
}

}  // extern "C"
