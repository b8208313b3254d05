// NCHW <-> channels_last conversion of activations at the boundaries between the custom operators (NCHW: MGHS, the SFA stage) and the
// dense stacks that run in channels_last: a batched matrix transpose [b][rows][cols] -> [b][cols][rows] of 2- or 4-byte elements
// through a 64 x 64 LDS tile, so that both the loads and the stores of a wave are one contiguous 128 / 256-byte span (torch's
// strided copy kernel moves these tensors at ~1 TB/s).
#include "common.h"

namespace {

constexpr int kTile = 64;
constexpr int kLayoutBlock = 256;

template <typename E>
__global__ __launch_bounds__(kLayoutBlock) void transpose_batched(const E* __restrict__ in, E* __restrict__ out, int rows, int cols,
                                                                  int tiles_c, int tiles_per_image) {
  __shared__ unsigned tile[kTile][kTile + 1];
  const long b = blockIdx.x / tiles_per_image;
  const int t = blockIdx.x % tiles_per_image;
  const int r0 = (t / tiles_c) * kTile, c0 = (t % tiles_c) * kTile;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const E* src = in + (size_t)b * rows * cols;
  E* dst = out + (size_t)b * rows * cols;
#pragma unroll
  for (int j = 0; j < kTile / 4; ++j) {
    const int r = r0 + ty + 4 * j, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + 4 * j][tx] = (unsigned)src[(size_t)r * cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kTile / 4; ++j) {
    const int c = c0 + ty + 4 * j, r = r0 + tx;
    if (r < rows && c < cols) dst[(size_t)c * rows + r] = (E)tile[tx][ty + 4 * j];
  }
}

// 2-byte elements, rows and cols even: the same tile with two elements per lane on both sides (4-byte loads along cols, 4-byte
// stores along rows) -- half the memory instructions of the scalar form (2.7 TB/s on the 164 MB half tensors of the SFA stage).
__global__ __launch_bounds__(kLayoutBlock) void transpose_batched_pairs(const unsigned short* __restrict__ in, unsigned short* __restrict__ out,
                                                                        int rows, int cols, int tiles_c, int tiles_per_image) {
  __shared__ __attribute__((aligned(8))) unsigned tile[kTile / 2][kTile + 2];   // [column pair][row]
  const long b = blockIdx.x / tiles_per_image;
  const int t = blockIdx.x % tiles_per_image;
  const int r0 = (t / tiles_c) * kTile, c0 = (t % tiles_c) * kTile;
  const int lo = threadIdx.x & 31, hi = threadIdx.x >> 5;   // 32 x 8
  const unsigned short* src = in + (size_t)b * rows * cols;
  unsigned short* dst = out + (size_t)b * rows * cols;
#pragma unroll
  for (int j = 0; j < kTile / 8; ++j) {
    const int r = r0 + hi + 8 * j, c = c0 + 2 * lo;
    if (r < rows && c < cols) tile[lo][hi + 8 * j] = *reinterpret_cast<const unsigned*>(src + (size_t)r * cols + c);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kTile / 8; ++j) {
    const int cl = hi + 8 * j, c = c0 + cl, r = r0 + 2 * lo;      // output row c, output columns r, r + 1
    if (r < rows && c < cols) {
      const uint2 two = *reinterpret_cast<const uint2*>(&tile[cl >> 1][2 * lo]);   // (r, c pair), (r + 1, c pair)
      const unsigned sh = (cl & 1) * 16;
      const unsigned v = ((two.x >> sh) & 0xffffu) | (((two.y >> sh) & 0xffffu) << 16);
      *reinterpret_cast<unsigned*>(dst + (size_t)c * rows + r) = v;
    }
  }
}

}  // namespace

extern "C" {

int dhd_transpose_batched(const void* in, void* out, int elem_bytes, long batch, int rows, int cols, void* stream) {
  if (!in || !out) return DHD_EINVAL;
  if ((elem_bytes != 2 && elem_bytes != 4) || batch <= 0 || rows <= 0 || cols <= 0) return DHD_EUNSUPPORTED;
  const int tiles_r = dhd_cdiv(rows, kTile), tiles_c = dhd_cdiv(cols, kTile);
  const long blocks = batch * tiles_r * tiles_c;
  if (blocks > 0x7fffffffL) return DHD_EUNSUPPORTED;
  hipStream_t st = dhd_stream(stream);
  if (elem_bytes == 2 && !(rows & 1) && !(cols & 1))
    hipLaunchKernelGGL(transpose_batched_pairs, dim3((unsigned)blocks), dim3(kLayoutBlock), 0, st, (const unsigned short*)in,
                       (unsigned short*)out, rows, cols, tiles_c, tiles_r * tiles_c);
  else if (elem_bytes == 2)
    hipLaunchKernelGGL(transpose_batched<unsigned short>, dim3((unsigned)blocks), dim3(kLayoutBlock), 0, st, (const unsigned short*)in,
                       (unsigned short*)out, rows, cols, tiles_c, tiles_r * tiles_c);
  else
    hipLaunchKernelGGL(transpose_batched<unsigned>, dim3((unsigned)blocks), dim3(kLayoutBlock), 0, st, (const unsigned*)in, (unsigned*)out, rows,
                       cols, tiles_c, tiles_r * tiles_c);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

}  // extern "C"
