// The element-wise head of MGHS.forward (models/necks/lss_heightmap.py:484-489) as ONE launch each way (gfx950):
//
//     x_d = depth_net(x)                        (BN, D + C [+ ...], fH, fW)   float32 / half, NCHW or channels_last
//     depth     = x_d[:, :D].softmax(dim=1)     (BN, D, fH, fW)  float32
//     tran_feat = x_d[:, D:D+C]                 (BN, C, fH, fW)  float32, dense NCHW (what dhd_mghs_lift takes)
//     height    = height_net(x)[:, :H].softmax(dim=1)            float32
//     band      = create_mask_3(height_range[argmax_H(height)])  uint8 (0 / 1 / 2, 255 = no band; :528-564)
//
// In torch this is two strided slices made contiguous, two casts under autocast, two softmax kernels and, on the way back,
// two softmax-backward launches (+ a multiply each), two zero-filled (BN, D+C, fH, fW) tensors with a slice copied into each
// and the add that joins them -- ~8 launches each way around 3.6-7.3 MB of data.
//
// Arithmetic = torch's own softmax for this shape (aten SoftMax.cu, cunn_SpatialSoftMaxForward with one thread per pixel:
// inner size fH*fW > 64): float32 max over the bins in order, float32 sum of expf(x - max) in order, expf(x - max) / sum with a
// correctly rounded division.  One thread per pixel walks the bins in the same order, so the probabilities come out as
// torch's do (asserted bit for bit against torch on the GPU, tests/test_gpu_parity.py) and the band id is the argmax of
// exactly the probabilities the caller gets back (first maximum wins, as torch.argmax on CPU / the reference).
//
// Backward: gx = (g - sum_k g_k y_k) * y per pixel (float32, in bin order), written together with the context gradient into
// ONE (BN, CT, fH, fW) tensor of x_d's dtype and layout; channels past D + C get zeros.
#include "lift_device.h"

namespace {

using namespace dhd;

constexpr int kBlock = 256;
typedef __bf16 bf16_t;

// element (image, channel, pixel) of a (BN, CT, hw) tensor that is dense NCHW (nhwc = 0) or channels_last (nhwc = 1)
struct View {
  int ct, hw, nhwc;
  __device__ __forceinline__ size_t at(int bn, int ch, int p) const {
    return nhwc ? ((size_t)bn * hw + p) * ct + ch : ((size_t)bn * ct + ch) * hw + p;
  }
};

template <typename T> __device__ __forceinline__ float ld(const T* p, size_t i) { return (float)p[i]; }
template <typename T> __device__ __forceinline__ void st(T* p, size_t i, float v) { p[i] = (T)v; }

// softmax over `n` bins of pixel (bn, p); probabilities to out (BN, n, hw) NCHW float32; returns the first argmax
template <typename T>
__device__ __forceinline__ int pixel_softmax(const T* __restrict__ x, const View v, int bn, int p, int n, float* __restrict__ out) {
  float mx = -3.402823466e+38f;     // numeric_limits<float>::lowest(), as aten
  for (int k = 0; k < n; ++k) mx = fmaxf(mx, ld(x, v.at(bn, k, p)));
  float sum = 0.f;
  for (int k = 0; k < n; ++k) sum += expf(ld(x, v.at(bn, k, p)) - mx);
  float best = -1.f;
  int arg = 0;
  float* o = out + (size_t)bn * n * v.hw + p;
  for (int k = 0; k < n; ++k) {
    const float y = expf(ld(x, v.at(bn, k, p)) - mx) / sum;
    o[(size_t)k * v.hw] = y;
    if (y > best) { best = y; arg = k; }
  }
  return arg;
}

// blocks [0, nb_pix): depth softmax; [nb_pix, 2 nb_pix): height softmax + band; the rest: context copy (float32 NCHW)
template <typename TX, typename TH>
__global__ __launch_bounds__(kBlock) void dh_softmax_fwd(const TX* __restrict__ xd, View vx, const TH* __restrict__ hl, View vh, int bn_total,
                                                         int d, int c, int hb, BandLut lut, float* __restrict__ depth,
                                                         float* __restrict__ feat, float* __restrict__ height,
                                                         uint8_t* __restrict__ band, int nb_pix) {
  const int hw = vx.hw, n_pix = bn_total * hw;
  int blk = blockIdx.x;
  if (blk < 2 * nb_pix) {
    const bool is_h = blk >= nb_pix;
    if (is_h) blk -= nb_pix;
    const int i = blk * kBlock + threadIdx.x;
    if (i >= n_pix) return;
    const int bn = i / hw, p = i % hw;
    if (!is_h) {
      pixel_softmax(xd, vx, bn, p, d, depth);
    } else if (hl) {
      const int arg = pixel_softmax(hl, vh, bn, p, hb, height);
      if (band) band[i] = lut.band[arg];
    }
    return;
  }
  blk -= 2 * nb_pix;
  // context: thread = (bn, channel, pixel) of the NCHW output
  const long j = (long)blk * kBlock + threadIdx.x;
  if (j >= (long)bn_total * c * hw) return;
  const int p = (int)(j % hw), ch = (int)((j / hw) % c), bn = (int)(j / ((long)hw * c));
  feat[j] = ld(xd, vx.at(bn, d + ch, p));
}

// backward: thread per pixel for the two softmax parts (bins in order), thread per element for the rest of x_d's gradient
template <typename TX, typename TH>
__global__ __launch_bounds__(kBlock) void dh_softmax_bwd(const float* __restrict__ g_depth, const float* __restrict__ g_feat,
                                                         const float* __restrict__ g_height, const float* __restrict__ depth,
                                                         const float* __restrict__ height, int bn_total, int d, int c, int hb,
                                                         TX* __restrict__ gxd, View vx, TH* __restrict__ ghl, View vh, int nb_pix) {
  const int hw = vx.hw, n_pix = bn_total * hw;
  int blk = blockIdx.x;
  if (blk < 2 * nb_pix) {
    const bool is_h = blk >= nb_pix;
    if (is_h) blk -= nb_pix;
    const int i = blk * kBlock + threadIdx.x;
    if (i >= n_pix) return;
    const int bn = i / hw, p = i % hw;
    if (!is_h) {
      if (!gxd) return;
      if (!g_depth) {
        for (int k = 0; k < d; ++k) st(gxd, vx.at(bn, k, p), 0.f);
        return;
      }
      const float* g = g_depth + (size_t)bn * d * hw + p;
      const float* y = depth + (size_t)bn * d * hw + p;
      float s = 0.f;
      for (int k = 0; k < d; ++k) s = fmaf(g[(size_t)k * hw], y[(size_t)k * hw], s);
      for (int k = 0; k < d; ++k) st(gxd, vx.at(bn, k, p), (g[(size_t)k * hw] - s) * y[(size_t)k * hw]);
    } else {
      if (!ghl) return;
      if (!g_height) {
        for (int k = 0; k < vh.ct; ++k) st(ghl, vh.at(bn, k, p), 0.f);
        return;
      }
      const float* g = g_height + (size_t)bn * hb * hw + p;
      const float* y = height + (size_t)bn * hb * hw + p;
      float s = 0.f;
      for (int k = 0; k < hb; ++k) s = fmaf(g[(size_t)k * hw], y[(size_t)k * hw], s);
      for (int k = 0; k < hb; ++k) st(ghl, vh.at(bn, k, p), (g[(size_t)k * hw] - s) * y[(size_t)k * hw]);
      for (int k = hb; k < vh.ct; ++k) st(ghl, vh.at(bn, k, p), 0.f);
    }
    return;
  }
  if (!gxd) return;
  blk -= 2 * nb_pix;
  const int rest = vx.ct - d;    // channels [d, ct): the context gradient, then zeros
  const long j = (long)blk * kBlock + threadIdx.x;
  if (j >= (long)bn_total * rest * hw) return;
  // thread order follows the DESTINATION's fastest axis so that the stores coalesce
  int p, ch, bn;
  if (vx.nhwc) { ch = (int)(j % rest); p = (int)((j / rest) % hw); bn = (int)(j / ((long)rest * hw)); }
  else { p = (int)(j % hw); ch = (int)((j / hw) % rest); bn = (int)(j / ((long)hw * rest)); }
  const float v = (g_feat && ch < c) ? g_feat[((size_t)bn * c + ch) * hw + p] : 0.f;
  st(gxd, vx.at(bn, d + ch, p), v);
}

template <typename TX>
int fwd_h(int hl_dtype, dim3 grid, hipStream_t stq, const TX* xd, View vx, const void* hl, View vh, int bn, int d, int c, int hb,
          const BandLut& lut, float* depth, float* feat, float* height, uint8_t* band, int nb_pix) {
#define DHD_FWD(TH) hipLaunchKernelGGL((dh_softmax_fwd<TX, TH>), grid, dim3(kBlock), 0, stq, xd, vx, (const TH*)hl, vh, bn, d, c, hb, lut, depth, feat, height, band, nb_pix)
  if (hl_dtype == DHD_F32) DHD_FWD(float);
  else if (hl_dtype == DHD_F16) DHD_FWD(_Float16);
  else if (hl_dtype == DHD_BF16) DHD_FWD(bf16_t);
  else return DHD_EINVAL;
#undef DHD_FWD
  return DHD_OK;
}

template <typename TX>
int bwd_h(int hl_dtype, dim3 grid, hipStream_t stq, const float* gd, const float* gf, const float* gh, const float* depth, const float* height,
          int bn, int d, int c, int hb, TX* gxd, View vx, void* ghl, View vh, int nb_pix) {
#define DHD_BWD(TH) hipLaunchKernelGGL((dh_softmax_bwd<TX, TH>), grid, dim3(kBlock), 0, stq, gd, gf, gh, depth, height, bn, d, c, hb, gxd, vx, (TH*)ghl, vh, nb_pix)
  if (hl_dtype == DHD_F32) DHD_BWD(float);
  else if (hl_dtype == DHD_F16) DHD_BWD(_Float16);
  else if (hl_dtype == DHD_BF16) DHD_BWD(bf16_t);
  else return DHD_EINVAL;
#undef DHD_BWD
  return DHD_OK;
}

}  // namespace

extern "C" {

int dhd_mghs_softmax_forward(const void* xd, int xd_dtype, int xd_nhwc, int ct, const void* hl, int hl_dtype, int hl_nhwc, int ht, int bn,
                             int hw, int d, int c, int h_bins, const float* height_range, const float* mask_range, float* depth,
                             float* feat, float* height, uint8_t* band, void* stream) {
  if (!xd || !depth || !feat || bn <= 0 || hw <= 0 || d <= 0 || c <= 0 || ct < d + c) return DHD_EINVAL;
  if (hl && (!height || h_bins <= 0 || ht < h_bins)) return DHD_EINVAL;
  if (hl && band && (!height_range || !mask_range)) return DHD_EINVAL;
  if (hl && h_bins > kMaxHeightBins) return DHD_EUNSUPPORTED;
  if ((long)bn * ct * hw >= (1L << 31)) return DHD_EUNSUPPORTED;
  BandLut lut = {};
  if (hl && band) make_band_lut(height_range, h_bins, mask_range, &lut);
  const int nb_pix = dhd_cdiv((long)bn * hw, kBlock);
  const dim3 grid(2 * nb_pix + dhd_cdiv((long)bn * c * hw, kBlock));
  const View vx{ct, hw, xd_nhwc ? 1 : 0}, vh{ht, hw, hl_nhwc ? 1 : 0};
  hipStream_t stq = dhd_stream(stream);
  int rc;
  if (xd_dtype == DHD_F32) rc = fwd_h<float>(hl_dtype, grid, stq, (const float*)xd, vx, hl, vh, bn, d, c, h_bins, lut, depth, feat, height, band, nb_pix);
  else if (xd_dtype == DHD_F16) rc = fwd_h<_Float16>(hl_dtype, grid, stq, (const _Float16*)xd, vx, hl, vh, bn, d, c, h_bins, lut, depth, feat, height, band, nb_pix);
  else if (xd_dtype == DHD_BF16) rc = fwd_h<bf16_t>(hl_dtype, grid, stq, (const bf16_t*)xd, vx, hl, vh, bn, d, c, h_bins, lut, depth, feat, height, band, nb_pix);
  else return DHD_EINVAL;
  if (rc != DHD_OK) return rc;
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

int dhd_mghs_softmax_backward(const float* g_depth, const float* g_feat, const float* g_height, const float* depth, const float* height,
                              int bn, int hw, int d, int c, int h_bins, void* g_xd, int xd_dtype, int xd_nhwc, int ct, void* g_hl,
                              int hl_dtype, int hl_nhwc, int ht, void* stream) {
  if (bn <= 0 || hw <= 0 || d <= 0 || c <= 0 || (!g_xd && !g_hl)) return DHD_EINVAL;
  if (g_xd && (ct < d + c || (g_depth && !depth))) return DHD_EINVAL;
  if (g_hl && (h_bins <= 0 || ht < h_bins || (g_height && !height))) return DHD_EINVAL;
  if ((long)bn * (ct > ht ? ct : ht) * hw >= (1L << 31)) return DHD_EUNSUPPORTED;
  const int nb_pix = dhd_cdiv((long)bn * hw, kBlock);
  const dim3 grid(2 * nb_pix + (g_xd ? dhd_cdiv((long)bn * (ct - d) * hw, kBlock) : 0));
  const View vx{ct, hw, xd_nhwc ? 1 : 0}, vh{ht, hw, hl_nhwc ? 1 : 0};
  hipStream_t stq = dhd_stream(stream);
  int rc;
  if (!g_xd || xd_dtype == DHD_F32) rc = bwd_h<float>(hl_dtype, grid, stq, g_depth, g_feat, g_height, depth, height, bn, d, c, h_bins, (float*)g_xd, vx, g_hl, vh, nb_pix);
  else if (xd_dtype == DHD_F16) rc = bwd_h<_Float16>(hl_dtype, grid, stq, g_depth, g_feat, g_height, depth, height, bn, d, c, h_bins, (_Float16*)g_xd, vx, g_hl, vh, nb_pix);
  else if (xd_dtype == DHD_BF16) rc = bwd_h<bf16_t>(hl_dtype, grid, stq, g_depth, g_feat, g_height, depth, height, bn, d, c, h_bins, (bf16_t*)g_xd, vx, g_hl, vh, nb_pix);
  else return DHD_EINVAL;
  if (rc != DHD_OK) return rc;
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

}  // extern "C"
