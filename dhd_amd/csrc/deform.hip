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
