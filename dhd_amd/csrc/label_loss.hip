// Height / depth supervision of the view transformer (row a16), for gfx950.
//
// Reference: models/necks/lss_heightmap.py:595-622 (MGHS.get_height_loss), :625-667
// (get_downsampled_gt_depth), :670-701 (get_downsampled_gt_height), and the depth+height variant
// :859-897 (MGHS_Depth).  Per feature pixel (16x16 image pixels): the smallest non-zero value of the
// sparse LiDAR map, `(v - offset) / step` in float32, truncation to a bin, "bin 0 = no label"; pixels
// with a depth label are foreground; the loss is the binary cross entropy of the softmax maps against
// the one-hot bins, summed over foreground pixels and channels, / max(1, n_fg).
//
// Here the one-hot matrices are never built: a label is a bin index (int16) per feature pixel.
//   sparse_bin_labels   one wave per feature pixel: window minimum by DPP, the two bin indices
//   bin_bce_partial     one thread per pixel over the C channel planes (coalesced across pixels)
//   bin_bce_finalize    loss = weight * sum / max(1, n_fg)   (double)
//   bin_bce_grad        torch's binary_cross_entropy backward, (p - y) / max((1 - p) p, 1e-12)
#include "common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxBlocks = 256;

__device__ __forceinline__ float wave_min(float v) {
  for (int m = 32; m > 0; m >>= 1) v = fminf(v, __shfl_xor(v, m, DHD_WAVE));
  return v;
}

// trunc((v - offset) / step) if 0 <= . < n_bins + 1 else 0, in the reference's float32 operation order
__device__ __forceinline__ int bin_of(float v, float offset, float step, int n_bins) {
  const float g = __fdiv_rn(__fsub_rn(v, offset), step);
  return (g < (float)(n_bins + 1) && g >= 0.0f) ? (int)g : 0;
}

// spacing-increasing depth bins (sid=True, lss_heightmap.py:655-660): g = (log(v) - log(d0)) * (D - 1) / log((d1 - 1) / d0) + 1,
// every operation rounded to float32 in the reference's order; log_d0 and den are the float32 values torch computed on the
// host, log(v) is the device's float32 logarithm (<= 1 ulp: a value within an ulp of a bin boundary may fall on the other side)
__device__ __forceinline__ int bin_of_sid(float v, float log_d0, float den, int n_bins) {
  float g = __fsub_rn(logf(v), log_d0);
  g = __fdiv_rn(__fmul_rn(g, (float)(n_bins - 1)), den);
  g = __fadd_rn(g, 1.0f);
  return (g < (float)(n_bins + 1) && g >= 0.0f) ? (int)g : 0;
}

template <bool SID>
__global__ __launch_bounds__(kBlock) void sparse_bin_labels(const float* __restrict__ gt_depth, const float* __restrict__ gt_height,
                                                            int n_pix, int fh, int fw, int ds, float d_off, float d_step, int d_bins,
                                                            float h_off, float h_step, int h_bins, int16_t* __restrict__ dbin,
                                                            int16_t* __restrict__ hbin) {
  const int lane = threadIdx.x & 63;
  const int pix = blockIdx.x * (kBlock / DHD_WAVE) + (threadIdx.x >> 6);  // (bn, y, x) feature pixel
  if (pix >= n_pix) return;
  const int x = pix % fw, y = (pix / fw) % fh, bn = pix / (fw * fh);
  const int wi = fw * ds;
  const size_t base = ((size_t)bn * fh * ds + (size_t)y * ds) * wi + (size_t)x * ds;
  float md = 1e5f, mh = 1e5f;
  for (int i = lane; i < ds * ds; i += DHD_WAVE) {
    const size_t o = base + (size_t)(i / ds) * wi + (i % ds);
    const float d = gt_depth[o], h = gt_height[o];
    md = fminf(md, d == 0.0f ? 1e5f : d);
    mh = fminf(mh, h == 0.0f ? 1e5f : h);
  }
  md = wave_min(md);
  mh = wave_min(mh);
  if (lane == 0) {
    dbin[pix] = (int16_t)(SID ? bin_of_sid(md, d_off, d_step, d_bins) : bin_of(md, d_off, d_step, d_bins));
    hbin[pix] = (int16_t)bin_of(mh, h_off, h_step, h_bins);
  }
}

// partial[block] = sum over the block's foreground pixels and all channels of the BCE terms; cnt[block] = #fg
__global__ __launch_bounds__(kBlock) void bin_bce_partial(const float* __restrict__ pred, const int16_t* __restrict__ bin,
                                                          const int16_t* __restrict__ fg_bin, int n_pix, int c, int hw,
                                                          float* __restrict__ partial, float* __restrict__ cnt) {
  __shared__ float sm[2][kBlock / DHD_WAVE];
  float s = 0.f, n = 0.f;
  for (int pix = blockIdx.x * kBlock + threadIdx.x; pix < n_pix; pix += gridDim.x * kBlock) {
    if (fg_bin[pix] <= 0) continue;
    n += 1.f;
    const int label = bin[pix] - 1;  // -1: no channel is hot
    const float* p = pred + (size_t)(pix / hw) * c * hw + (pix % hw);
    for (int k = 0; k < c; ++k) {
      const float v = p[(size_t)k * hw];
      // F.binary_cross_entropy clamps both logs at -100
      s -= k == label ? fmaxf(logf(v), -100.f) : fmaxf(log1pf(-v), -100.f);
    }
  }
  s = wave_sum_bcast(s);
  n = wave_sum_bcast(n);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { sm[0][wv] = s; sm[1][wv] = n; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < kBlock / DHD_WAVE; ++w) { a += sm[0][w]; b += sm[1][w]; }
    partial[blockIdx.x] = a;
    cnt[blockIdx.x] = b;
  }
}

__global__ void bin_bce_finalize(const float* __restrict__ partial, const float* __restrict__ cnt, int n_blocks, float weight,
                                 float* __restrict__ loss, float* __restrict__ n_fg) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0, n = 0.0;
  for (int b = 0; b < n_blocks; ++b) { s += (double)partial[b]; n += (double)cnt[b]; }
  *loss = (float)((double)weight * s / fmax(1.0, n));
  *n_fg = (float)n;
}

__global__ __launch_bounds__(kBlock) void bin_bce_grad(const float* __restrict__ pred, const int16_t* __restrict__ bin,
                                                       const int16_t* __restrict__ fg_bin, int n_pix, int c, int hw, float weight,
                                                       const float* __restrict__ n_fg, const float* __restrict__ gl,
                                                       float* __restrict__ grad) {
  const int pix = blockIdx.x * kBlock + threadIdx.x;
  if (pix >= n_pix) return;
  const size_t base = (size_t)(pix / hw) * c * hw + (pix % hw);
  const bool fg = fg_bin[pix] > 0;
  const int label = bin[pix] - 1;
  const float scale = gl[0] * weight / fmaxf(1.0f, n_fg[0]);
  for (int k = 0; k < c; ++k) {
    float g = 0.f;
    if (fg) {
      const float v = pred[base + (size_t)k * hw];
      g = scale * (v - (k == label ? 1.f : 0.f)) / fmaxf((1.0f - v) * v, 1e-12f);
    }
    grad[base + (size_t)k * hw] = g;
  }
}

inline int blocks_for(int n_pix) {
  const int b = (n_pix + kBlock - 1) / kBlock;
  return b < kMaxBlocks ? b : kMaxBlocks;
}

}  // namespace

extern "C" {

int dhd_sparse_bin_labels(const float* gt_depth, const float* gt_height, int bn, int fh, int fw, int downsample, float depth_offset,
                          float depth_step, int depth_bins, float height_offset, float height_step, int height_bins, int16_t* depth_bin,
                          int16_t* height_bin, void* stream) {
  if (!gt_depth || !gt_height || !depth_bin || !height_bin || bn <= 0 || fh <= 0 || fw <= 0 || downsample <= 0) return DHD_EINVAL;
  if (depth_bins <= 0 || height_bins <= 0 || depth_bins > 32000 || height_bins > 32000 || depth_step == 0.f || height_step == 0.f)
    return DHD_EINVAL;
  const long n_pix = (long)bn * fh * fw;
  if (n_pix > (1L << 30)) return DHD_EUNSUPPORTED;
  hipLaunchKernelGGL(sparse_bin_labels<false>, dim3(dhd_cdiv(n_pix, kBlock / DHD_WAVE)), dim3(kBlock), 0, dhd_stream(stream), gt_depth,
                     gt_height, (int)n_pix, fh, fw, downsample, depth_offset, depth_step, depth_bins, height_offset, height_step,
                     height_bins, depth_bin, height_bin);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

int dhd_sparse_bin_labels_sid(const float* gt_depth, const float* gt_height, int bn, int fh, int fw, int downsample, float log_d0,
                              float log_ratio, int depth_bins, float height_offset, float height_step, int height_bins,
                              int16_t* depth_bin, int16_t* height_bin, void* stream) {
  if (!gt_depth || !gt_height || !depth_bin || !height_bin || bn <= 0 || fh <= 0 || fw <= 0 || downsample <= 0) return DHD_EINVAL;
  if (depth_bins <= 1 || height_bins <= 0 || depth_bins > 32000 || height_bins > 32000 || log_ratio == 0.f || height_step == 0.f)
    return DHD_EINVAL;
  const long n_pix = (long)bn * fh * fw;
  if (n_pix > (1L << 30)) return DHD_EUNSUPPORTED;
  hipLaunchKernelGGL(sparse_bin_labels<true>, dim3(dhd_cdiv(n_pix, kBlock / DHD_WAVE)), dim3(kBlock), 0, dhd_stream(stream), gt_depth,
                     gt_height, (int)n_pix, fh, fw, downsample, log_d0, log_ratio, depth_bins, height_offset, height_step,
                     height_bins, depth_bin, height_bin);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

size_t dhd_bin_bce_workspace_bytes(void) { return (size_t)(2 * kMaxBlocks + 8) * sizeof(float); }

int dhd_bin_bce_forward(const float* pred, const int16_t* bin, const int16_t* fg_bin, int bn, int c, int hw, float weight, float* loss,
                        void* workspace, void* stream) {
  if (!pred || !bin || !fg_bin || !loss || !workspace || bn <= 0 || c <= 0 || hw <= 0) return DHD_EINVAL;
  const long n_pix = (long)bn * hw;
  if (n_pix > (1L << 30)) return DHD_EUNSUPPORTED;
  float* ws = static_cast<float*>(workspace);
  const int nb = blocks_for((int)n_pix);
  hipStream_t st = dhd_stream(stream);
  hipLaunchKernelGGL(bin_bce_partial, dim3(nb), dim3(kBlock), 0, st, pred, bin, fg_bin, (int)n_pix, c, hw, ws, ws + kMaxBlocks);
  hipLaunchKernelGGL(bin_bce_finalize, dim3(1), dim3(64), 0, st, ws, ws + kMaxBlocks, nb, weight, loss, ws + 2 * kMaxBlocks);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

int dhd_bin_bce_backward(const float* pred, const int16_t* bin, const int16_t* fg_bin, int bn, int c, int hw, float weight,
                         const float* grad_loss, const void* workspace, float* grad_pred, void* stream) {
  if (!pred || !bin || !fg_bin || !grad_loss || !workspace || !grad_pred || bn <= 0 || c <= 0 || hw <= 0) return DHD_EINVAL;
  const long n_pix = (long)bn * hw;
  if (n_pix > (1L << 30)) return DHD_EUNSUPPORTED;
  const float* ws = static_cast<const float*>(workspace);
  hipLaunchKernelGGL(bin_bce_grad, dim3(dhd_cdiv(n_pix, kBlock)), dim3(kBlock), 0, dhd_stream(stream), pred, bin, fg_bin, (int)n_pix, c,
                     hw, weight, ws + 2 * kMaxBlocks, grad_loss, grad_pred);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// LiDAR points -> sparse per-camera depth / height maps (the dataloader-side label step,
// datasets/pipelines/loading_new.py:35-99: PointToMultiViewDepthandHeight.points2depthmap /
// points2heightmap).  The reference sorts all points by the float32 key  pixel_rank + d/100  and keeps the
// first point of every pixel; here that is a z-buffer: one 64-bit atomicMin per point on
// (key bits << 32 | point index), then one thread per pixel writes the winner's depth and height.  Equal keys
// (depths closer than the float32 spacing at that rank) resolve to the lowest point index = a stable sort;
// the reference's unstable argsort may keep any of the tied points there.
// ------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(kBlock) void raster_min_kernel(const float* __restrict__ pts, int n_pts, int h, int w, float inv_ds_div,
                                                            float d_lo, float d_hi, unsigned long long* __restrict__ zbuf) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const int cam = blockIdx.y;
  if (i >= n_pts) return;
  const float* p = pts + ((size_t)cam * n_pts + i) * 4;
  const float cu = rintf(__fdiv_rn(p[0], inv_ds_div)), cv = rintf(__fdiv_rn(p[1], inv_ds_div));  // torch.round: half to even
  const float d = p[2];
  if (!(cu >= 0.f && cu < (float)w && cv >= 0.f && cv < (float)h && d < d_hi && d >= d_lo)) return;
  const float rank = __fadd_rn(cu, __fmul_rn(cv, (float)w));
  const float key = __fadd_rn(rank, __fdiv_rn(d, 100.0f));  // positive: the bit pattern orders like the value
  const unsigned long long packed = ((unsigned long long)__float_as_uint(key) << 32) | (unsigned)i;
  atomicMin(zbuf + (size_t)cam * h * w + (size_t)((int)cv * w + (int)cu), packed);
}

__global__ __launch_bounds__(kBlock) void raster_write_kernel(const float* __restrict__ pts, int n_pts, int hw,
                                                              const unsigned long long* __restrict__ zbuf,
                                                              float* __restrict__ depth_map, float* __restrict__ height_map) {
  const int px = blockIdx.x * kBlock + threadIdx.x;
  const int cam = blockIdx.y;
  if (px >= hw) return;
  const unsigned long long z = zbuf[(size_t)cam * hw + px];
  float d = 0.f, hv = 0.f;
  if (z != ~0ull) {
    const float* p = pts + ((size_t)cam * n_pts + (unsigned)(z & 0xffffffffull)) * 4;
    d = p[2];
    hv = p[3];
  }
  depth_map[(size_t)cam * hw + px] = d;
  height_map[(size_t)cam * hw + px] = hv;
}

}  // namespace

extern "C" int dhd_points_to_maps(const float* points, int n_cams, int n_points, int height, int width, int downsample, float depth_lo,
                                  float depth_hi, float* depth_map, float* height_map, void* zbuffer, void* stream) {
  if ((!points && n_points > 0) || !depth_map || !height_map || !zbuffer || n_cams <= 0 || n_points < 0 || height <= 0 || width <= 0 ||
      downsample <= 0)
    return DHD_EINVAL;
  const int h = height / downsample, w = width / downsample;
  if (h <= 0 || w <= 0 || (long)h * w >= (1L << 24)) return DHD_EUNSUPPORTED;  // pixel ranks must be exact in float32
  hipStream_t st = dhd_stream(stream);
  unsigned long long* zb = static_cast<unsigned long long*>(zbuffer);
  DHD_HIP(hipMemsetAsync(zb, 0xff, (size_t)n_cams * h * w * sizeof(unsigned long long), st));
  if (n_points > 0)
    hipLaunchKernelGGL(raster_min_kernel, dim3(dhd_cdiv(n_points, kBlock), n_cams), dim3(kBlock), 0, st, points, n_points, h, w,
                       (float)downsample, depth_lo, depth_hi, zb);
  hipLaunchKernelGGL(raster_write_kernel, dim3(dhd_cdiv((long)h * w, kBlock), n_cams), dim3(kBlock), 0, st, points, n_points, h * w, zb,
                     depth_map, height_map);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}
