// Deformable-convolution sampling (DCN v1, stride 1, one deformable group) for gfx950: the "exotic" op of
// HeightNet / DepthNet (models/necks/depthnet.py:225-236, :466-477 build mmcv's `DCN`; mmcv-full 1.5.3
// ops/deform_conv: deformable_im2col / col2im / col2im_coord).  The sampled columns feed an ordinary GEMM
// with the layer's weight, which stays on the library path.
//
//   col[b, c*K + t, p] = bilinear(x[b, c], y_p + ky*dil - pad + off[b, 2t, p], x_p + kx*dil - pad + off[b, 2t+1, p])
// with K = k*k taps t = ky*k + kx, zero outside the image (corners outside contribute 0).
//
//   deform_im2col      thread = (b, t, p) x a chunk of channels: corner indices / weights once, then a channel loop
//   deform_col2im      block = (b, chunk of channel planes): scatter into LDS planes (the feature maps here are
//                      16x44 .. 32x88), then plain stores -- no global float atomics
//   deform_col2offset  thread = (b, t, p): channel loop of the coordinate gradients
#include "common.h"

namespace {

constexpr int kBlock = 256;

struct Tap {
  int i00, i01, i10, i11;   // flat indices into the H*W plane (valid ones only are used)
  float w00, w01, w10, w11; // bilinear weights, 0 for corners outside the image
  float ly, lx;             // fractional parts
  bool v00, v01, v10, v11, inside;
};

__device__ __forceinline__ Tap make_tap(float py, float px, int h, int w) {
  Tap t;
  t.inside = py > -1.0f && px > -1.0f && py < (float)h && px < (float)w;
  const float fy = floorf(py), fx = floorf(px);
  const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
  t.ly = py - fy;
  t.lx = px - fx;
  const float hy = 1.0f - t.ly, hx = 1.0f - t.lx;
  t.v00 = t.inside && y0 >= 0 && x0 >= 0;
  t.v01 = t.inside && y0 >= 0 && x1 <= w - 1;
  t.v10 = t.inside && y1 <= h - 1 && x0 >= 0;
  t.v11 = t.inside && y1 <= h - 1 && x1 <= w - 1;
  t.i00 = y0 * w + x0; t.i01 = y0 * w + x1; t.i10 = y1 * w + x0; t.i11 = y1 * w + x1;
  t.w00 = t.v00 ? hy * hx : 0.f;
  t.w01 = t.v01 ? hy * t.lx : 0.f;
  t.w10 = t.v10 ? t.ly * hx : 0.f;
  t.w11 = t.v11 ? t.ly * t.lx : 0.f;
  return t;
}

__device__ __forceinline__ Tap tap_of(const float* __restrict__ off_b, int t, int p, int h, int w, int k, int pad, int dil) {
  const int hw = h * w;
  const int y = p / w, x = p % w, ky = t / k, kx = t % k;
  const float py = (float)(y + ky * dil - pad) + off_b[(size_t)(2 * t) * hw + p];
  const float px = (float)(x + kx * dil - pad) + off_b[(size_t)(2 * t + 1) * hw + p];
  return make_tap(py, px, h, w);
}

__global__ __launch_bounds__(kBlock) void deform_im2col(const float* __restrict__ x, const float* __restrict__ off,
                                                        float* __restrict__ col, int c, int h, int w, int k, int pad, int dil,
                                                        int c_chunk) {
  const int hw = h * w, kk = k * k;
  const int i = blockIdx.x * kBlock + threadIdx.x;  // (t, p)
  if (i >= kk * hw) return;
  const int b = blockIdx.z, t = i / hw, p = i % hw;
  const Tap tp = tap_of(off + (size_t)b * 2 * kk * hw, t, p, h, w, k, pad, dil);
  const int c0 = blockIdx.y * c_chunk, c1 = min(c, c0 + c_chunk);
  const float* xb = x + ((size_t)b * c + c0) * hw;
  float* cb = col + (((size_t)b * c + c0) * kk + t) * hw + p;
  for (int ch = c0; ch < c1; ++ch, xb += hw, cb += (size_t)kk * hw) {
    float v = 0.f;
    if (tp.v00) v = fmaf(tp.w00, xb[tp.i00], v);
    if (tp.v01) v = fmaf(tp.w01, xb[tp.i01], v);
    if (tp.v10) v = fmaf(tp.w10, xb[tp.i10], v);
    if (tp.v11) v = fmaf(tp.w11, xb[tp.i11], v);
    *cb = v;
  }
}

// dx[b, ch] for a chunk of channel planes held in LDS
__global__ __launch_bounds__(kBlock) void deform_col2im(const float* __restrict__ dcol, const float* __restrict__ off,
                                                        float* __restrict__ dx, int c, int h, int w, int k, int pad, int dil,
                                                        int c_chunk) {
  extern __shared__ float planes[];  // [c_chunk][hw]
  const int hw = h * w, kk = k * k;
  const int b = blockIdx.y, c0 = blockIdx.x * c_chunk, nc = min(c_chunk, c - c0);
  for (int i = threadIdx.x; i < nc * hw; i += kBlock) planes[i] = 0.f;
  __syncthreads();
  const float* off_b = off + (size_t)b * 2 * kk * hw;
  for (int i = threadIdx.x; i < kk * hw; i += kBlock) {
    const int t = i / hw, p = i % hw;
    const Tap tp = tap_of(off_b, t, p, h, w, k, pad, dil);
    if (!tp.inside) continue;
    const float* g = dcol + (((size_t)b * c + c0) * kk + t) * hw + p;
    for (int j = 0; j < nc; ++j, g += (size_t)kk * hw) {
      const float gv = *g;
      float* pl = planes + j * hw;
      if (tp.v00) atomicAdd(pl + tp.i00, tp.w00 * gv);
      if (tp.v01) atomicAdd(pl + tp.i01, tp.w01 * gv);
      if (tp.v10) atomicAdd(pl + tp.i10, tp.w10 * gv);
      if (tp.v11) atomicAdd(pl + tp.i11, tp.w11 * gv);
    }
  }
  __syncthreads();
  float* out = dx + ((size_t)b * c + c0) * hw;
  for (int i = threadIdx.x; i < nc * hw; i += kBlock) out[i] = planes[i];
}

// doff[b, 2t, p] = sum_c dcol * d val / d py, doff[b, 2t+1, p] = ... / d px  (mmcv deformable_col2im_coord)
__global__ __launch_bounds__(kBlock) void deform_col2offset(const float* __restrict__ dcol, const float* __restrict__ x,
                                                            const float* __restrict__ off, float* __restrict__ doff, int c, int h,
                                                            int w, int k, int pad, int dil) {
  const int hw = h * w, kk = k * k;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= kk * hw) return;
  const int b = blockIdx.y, t = i / hw, p = i % hw;
  const Tap tp = tap_of(off + (size_t)b * 2 * kk * hw, t, p, h, w, k, pad, dil);
  float gy = 0.f, gx = 0.f;
  if (tp.inside) {
    const float hy = 1.0f - tp.ly, hx = 1.0f - tp.lx;
    const float* xb = x + (size_t)b * c * hw;
    const float* g = dcol + ((size_t)b * c * kk + t) * hw + p;
    for (int ch = 0; ch < c; ++ch, xb += hw, g += (size_t)kk * hw) {
      const float v00 = tp.v00 ? xb[tp.i00] : 0.f, v01 = tp.v01 ? xb[tp.i01] : 0.f;
      const float v10 = tp.v10 ? xb[tp.i10] : 0.f, v11 = tp.v11 ? xb[tp.i11] : 0.f;
      const float gv = *g;
      gy = fmaf(gv, (v10 - v00) * hx + (v11 - v01) * tp.lx, gy);
      gx = fmaf(gv, (v01 - v00) * hy + (v11 - v10) * tp.ly, gx);
    }
  }
  float* d = doff + (size_t)b * 2 * kk * hw;
  d[(size_t)(2 * t) * hw + p] = gy;
  d[(size_t)(2 * t + 1) * hw + p] = gx;
}

inline bool bad_shape(int b, int c, int h, int w, int k, int dil) { return b <= 0 || c <= 0 || h <= 0 || w <= 0 || k <= 0 || dil <= 0; }

}  // namespace

extern "C" {

int dhd_deform_im2col(const float* x, const float* offset, float* col, int b, int c, int h, int w, int k, int pad, int dil,
                      void* stream) {
  if (!x || !offset || !col || bad_shape(b, c, h, w, k, dil)) return DHD_EINVAL;
  if ((long)b * c * k * k * h * w >= (1L << 40)) return DHD_EUNSUPPORTED;
  const int c_chunk = c >= 32 ? 32 : c;
  hipLaunchKernelGGL(deform_im2col, dim3(dhd_cdiv((long)k * k * h * w, kBlock), dhd_cdiv(c, c_chunk), b), dim3(kBlock), 0,
                     dhd_stream(stream), x, offset, col, c, h, w, k, pad, dil, c_chunk);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

int dhd_deform_col2im(const float* dcol, const float* x, const float* offset, float* dx, float* doffset, int b, int c, int h, int w,
                      int k, int pad, int dil, void* stream) {
  if (!dcol || !x || !offset || !dx || !doffset || bad_shape(b, c, h, w, k, dil)) return DHD_EINVAL;
  const size_t plane = (size_t)h * w * sizeof(float);
  if (plane > 48 * 1024) return DHD_EUNSUPPORTED;  // a feature plane must fit the LDS scatter buffer
  int c_chunk = (int)(48 * 1024 / plane);
  if (c_chunk > 8) c_chunk = 8;
  if (c_chunk > c) c_chunk = c;
  hipStream_t st = dhd_stream(stream);
  hipLaunchKernelGGL(deform_col2im, dim3(dhd_cdiv(c, c_chunk), b), dim3(kBlock), (size_t)c_chunk * plane, st, dcol, offset, dx, c, h, w,
                     k, pad, dil, c_chunk);
  hipLaunchKernelGGL(deform_col2offset, dim3(dhd_cdiv((long)k * k * h * w, kBlock), b), dim3(kBlock), 0, st, dcol, x, offset, doffset, c,
                     h, w, k, pad, dil);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Temporal-stereo cost volume of DepthNet (models/necks/depthnet.py:307-361 in the reference; the
// BEVStereo matching cost): for every stereo-resolution pixel and depth hypothesis, the adjacent
// frame's feature map is sampled bilinearly at the hypothesis' reprojection and compared with the
// current frame's feature:
//     cost[bn,d,y,x] = sum_c | curr[bn,c,y,x] - sample(prev[bn,c], grid[bn,d,y,x]) |   (+ bias where the sample of
//                      the reference's last channel group came back exactly 0, i.e. outside the image)
//     out = softmax_d(-cost)
// The reference runs C/4 grid_sample calls over (BN, 4, D*H, W) tensors plus sub/abs/sum passes (29 % of a
// DHD-M training step on this GPU).  Here: features in NHWC, one wave per pixel: the current feature lives in
// registers (lane = 4 channels), each hypothesis costs four 1-KB tap loads, a DPP wave reduction, and the
// softmax over the D hypotheses is done in the same wave.  No gradient (the reference wraps it in no_grad).
// ------------------------------------------------------------------------------------------------
namespace {

using f32x4_t = __attribute__((ext_vector_type(4))) float;
constexpr int kCvMaxD = 256;

__global__ __launch_bounds__(kBlock) void stereo_cost_volume_kernel(const float* __restrict__ prev, const float* __restrict__ curr,
                                                                    const float* __restrict__ grid, int c, int h, int w, int nd,
                                                                    float bias, int flag_channel, float* __restrict__ out, int n_pix) {
  const int lane = threadIdx.x & 63;
  const int pix = blockIdx.x * (kBlock / DHD_WAVE) + (threadIdx.x >> 6);  // (bn, y, x)
  if (pix >= n_pix) return;
  const int hw = h * w;
  const int bn = pix / hw, yx = pix % hw;
  const int c4 = c >> 2;                       // float4 groups per pixel
  const float* pb = prev + (size_t)bn * hw * c;
  // this lane's channels: groups lane, lane + 64, ...
  f32x4_t cur[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int g = lane + 64 * q;
    cur[q] = g < c4 ? reinterpret_cast<const f32x4_t*>(curr + (size_t)pix * c)[g] : f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  const int flag_group = flag_channel >> 2, flag_comp = flag_channel & 3;
  float mine[kCvMaxD / DHD_WAVE];  // cost of hypothesis d lives in lane d % 64, register d / 64
#pragma unroll
  for (int q = 0; q < kCvMaxD / DHD_WAVE; ++q) mine[q] = 3.0e38f;
  const float* gp = grid + ((size_t)bn * nd * hw + yx) * 2;
  for (int d0 = 0; d0 < nd; d0 += DHD_WAVE) {
    // lane l fetches the sampling position of hypothesis d0 + l
    float gx = 0.f, gy = 0.f;
    if (d0 + lane < nd) {
      const float2 g2 = *reinterpret_cast<const float2*>(gp + (size_t)(d0 + lane) * hw * 2);
      gx = g2.x; gy = g2.y;
    }
    const int nb = min(DHD_WAVE, nd - d0);
    for (int i = 0; i < nb; ++i) {
      const float px = (__shfl(gx, i, DHD_WAVE) + 1.0f) * 0.5f * (float)(w - 1);   // align_corners=True
      const float py = (__shfl(gy, i, DHD_WAVE) + 1.0f) * 0.5f * (float)(h - 1);
      const Tap tp = make_tap(py, px, h, w);
      float acc = 0.f, flag = 1.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int g = lane + 64 * q;
        if (g >= c4) break;
        f32x4_t s = {0.f, 0.f, 0.f, 0.f};
        if (tp.v00) s += tp.w00 * reinterpret_cast<const f32x4_t*>(pb + (size_t)tp.i00 * c)[g];
        if (tp.v01) s += tp.w01 * reinterpret_cast<const f32x4_t*>(pb + (size_t)tp.i01 * c)[g];
        if (tp.v10) s += tp.w10 * reinterpret_cast<const f32x4_t*>(pb + (size_t)tp.i10 * c)[g];
        if (tp.v11) s += tp.w11 * reinterpret_cast<const f32x4_t*>(pb + (size_t)tp.i11 * c)[g];
        const f32x4_t df = cur[q] - s;
        acc += (fabsf(df.x) + fabsf(df.y)) + (fabsf(df.z) + fabsf(df.w));
        if (g == flag_group) flag = s[flag_comp];
      }
      float tot = wave_sum_bcast(acc);
      if (bias != 0.f) {
        const float fv = __shfl(flag, flag_group & 63, DHD_WAVE);
        if (fv == 0.f) tot += bias;
      }
      if (lane == i) mine[d0 >> 6] = tot;
    }
  }
  // softmax over the hypotheses of -cost: lanes/registers without a hypothesis hold +3e38 -> exp(-inf) = 0
  float mn = mine[0];
#pragma unroll
  for (int q = 1; q < kCvMaxD / DHD_WAVE; ++q) mn = fminf(mn, mine[q]);
  for (int m = 32; m > 0; m >>= 1) mn = fminf(mn, __shfl_xor(mn, m, DHD_WAVE));
  float e[kCvMaxD / DHD_WAVE], sum = 0.f;
#pragma unroll
  for (int q = 0; q < kCvMaxD / DHD_WAVE; ++q) {
    e[q] = mine[q] > 1.0e38f ? 0.f : __expf(mn - mine[q]);
    sum += e[q];
  }
  sum = wave_sum_bcast(sum);
  const float inv = 1.0f / sum;
  float* ob = out + (size_t)bn * nd * hw + yx;
#pragma unroll
  for (int q = 0; q < kCvMaxD / DHD_WAVE; ++q) {
    const int d = lane + 64 * q;
    if (d < nd) ob[(size_t)d * hw] = e[q] * inv;
  }
}

}  // namespace

extern "C" int dhd_stereo_cost_volume(const float* prev_nhwc, const float* curr_nhwc, const float* grid, int bn, int c, int h, int w,
                                      int n_depth, float bias, int flag_channel, float* out, void* stream) {
  if (!prev_nhwc || !curr_nhwc || !grid || !out || bn <= 0 || c <= 0 || h <= 0 || w <= 0 || n_depth <= 0) return DHD_EINVAL;
  if ((c & 3) != 0 || c > 1024 || n_depth > kCvMaxD || flag_channel < 0 || flag_channel >= c) return DHD_EUNSUPPORTED;
  const long n_pix = (long)bn * h * w;
  if (n_pix >= (1L << 31) / n_depth) return DHD_EUNSUPPORTED;
  hipLaunchKernelGGL(stereo_cost_volume_kernel, dim3(dhd_cdiv(n_pix, kBlock / DHD_WAVE)), dim3(kBlock), 0, dhd_stream(stream), prev_nhwc,
                     curr_nhwc, grid, c, h, w, n_depth, bias, flag_channel, out, (int)n_pix);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}
