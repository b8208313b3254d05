// SFA channel/spatial attention stage as ONE operator (forward and backward), for gfx950.
//
// Reference: models/necks/mix.py:8-59 (channel_spatial_stage).  With x = cat[x_bev, x_voxel]:
//   s   = mean_hw(x)                         a = sigmoid(fc(s))                     (mix.py:41-44)
//   u   = a*x_bev + (1-a)*x_voxel                                                    (mix.py:46-50)
//   y1  = conv1(u)   z1 = relu(bn1(y1))   y2 = conv2(z1)   s2 = bn2(y2)              (mix.py:51)
//   out = sigmoid(s2)*(a*x_bev) + (1-sigmoid(s2))*((1-a)*x_voxel)                    (mix.py:52-58)
//
// The two 1x1 convolutions are (C x C) x (C x B*HW) GEMMs on NCHW data (six of them per forward + backward).  They run on the
// matrix cores with everything element-wise FUSED into the operand path, so no intermediate but y1 and y2 is ever stored.
// Precision per call (dhd_sfa_weights.gemm / storage_dtype, include/dhd_amd.h) and the kernel family that serves it:
//   bf16x3 (default) bf16 MFMA on a two-way split of every float32 operand, three products per a*b:
//                    C = 128 / 256 (DHD-S / DHD-L: SFA(512, 256)): pw_gemm_cu (sfa_gemm_cu.h: one CU per pixel tile, weights in
//                    registers); C = 512 (DHD-M): pw_gemm_res<2> (sfa_gemm_res.h); weight gradients: pw_wgrad3 (this file)
//   bf16x6           exact three-way split, six products (float32-level accuracy): pw_gemm_res<3> at C = 128 / 256, pw_gemm6 at
//                    C = 512 and beyond (sfa_gemm_streamed.h); pw_wgrad6
//   f32              v_mfma_f32_32x32x2_f32, the float32 reference point of the precision table: pw_gemm / pw_wgrad (sfa_gemm_streamed.h)
//   half storage     x / y1 / y2 / g2 / g1 / du in fp16 or bf16, single half products with float32 accumulation -- the form under
//                    autocast: pw_gemm_cuh / pw_wgrad_h (sfa_half.h), element-wise passes and host side in sfa_stage_half.h
// This file: the small dense pieces (channel mean -> fc -> a), BatchNorm statistics and coefficient tables, the fused blends,
// pw_wgrad3 + the deterministic partial reduction, and the host side of the float32-storage operator.
// Common structure of the GEMMs:
//   * forward / dgrad: the activation operand passes through a per-(sample,channel) affine prologue act(c0*in0 + c1*in1 + c2)
//     -- which is blend1 (in0,in1 = x_bev,x_voxel), BatchNorm+ReLU (in0 = y1) or BatchNorm-backward (in0,in1 = g,y); epilogue:
//     bias + BatchNorm batch statistics (forward), the ReLU mask from the pass bits the forward recorded (dgrad 2), whole-line
//     16-byte stores along the pixel axis (store_b128_guarded, sfa_mfma.h);
//   * weight gradient: pixels are the reduction dimension; per-worker partial matrices reduced by a second small kernel
//     (deterministic, no float atomics);
//   * BatchNorm gradient sums are per-plane streaming reductions with double-precision finalisation; the blends are fused with
//     the BatchNorm affine and the sigmoid.
// Forward reads x three times and y1/y2 twice; nothing is transposed, there is no NHWC detour.
#include <stdlib.h>

#include "sfa_gemm_cu.h"
#include "sfa_half.h"
#include "sfa_mfma.h"

using namespace dhd_sfa;

namespace {

constexpr int kEwBlock = 256;     // element-wise / reduction kernels
constexpr int kPlaneChunks = 4;   // blocks per (b, c) plane
constexpr int kPwBlock = 256;     // pw_gemm: 4 waves
#ifndef DHD_PW_COT_SEL
#define DHD_PW_COT_SEL 8
#endif
constexpr int kPwStep = 16;       // input channels per weight image / pipeline step
constexpr int kWgBlock = 512;     // pw_wgrad: 8 waves
constexpr int kWgStride = 33;     // LDS row stride of a 32-pixel operand row (conflict-free column reads)
constexpr int kWgWorkers = 256;   // total pw_wgrad blocks (one per CU)

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }

__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = group_sum(v, DHD_WAVE);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[wv] = v;
  __syncthreads();
  float t = 0.f;
  for (int k = 0; k < (int)(blockDim.x / DHD_WAVE); ++k) t += sm[k];
  return t;
}

// [lo, hi) in float4 units of this block's share of a plane of hw floats (hw % 4 == 0).
__device__ __forceinline__ void chunk_range4(int hw, int* lo, int* hi) {
  const int n4 = hw >> 2, per = (n4 + kPlaneChunks - 1) / kPlaneChunks;
  *lo = blockIdx.x * per;
  *hi = min(n4, *lo + per);
}

// ------------------------------------------------------------------------------------------------
// small dense pieces: channel mean -> fc -> a, and its backward
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ void plane_mean_block(const float* __restrict__ x, float* __restrict__ part, int hw, float* sm) {
  const size_t plane = blockIdx.y;
  const f32x4* p4 = reinterpret_cast<const f32x4*>(x + plane * hw);
  int lo, hi;
  chunk_range4(hw, &lo, &hi);
  float a0 = 0.f, a1 = 0.f;
  int i = lo + threadIdx.x;
  for (; i + kEwBlock < hi; i += 2 * kEwBlock) {
    f32x4 v = __builtin_nontemporal_load(p4 + i), w = __builtin_nontemporal_load(p4 + i + kEwBlock);
    a0 += (v.x + v.y) + (v.z + v.w);
    a1 += (w.x + w.y) + (w.z + w.w);
  }
  if (i < hi) {
    f32x4 v = __builtin_nontemporal_load(p4 + i);
    a0 += (v.x + v.y) + (v.z + v.w);
  }
  float tot = block_sum(a0 + a1, sm);
  if (threadIdx.x == 0) part[plane * kPlaneChunks + blockIdx.x] = tot;
}

__global__ __launch_bounds__(kEwBlock) void plane_mean_kernel(const float* __restrict__ x, float* __restrict__ part, int hw) {
  __shared__ float sm[kEwBlock / DHD_WAVE];
  plane_mean_block(x, part, hw, sm);
}

// one block per sample: s = mean, h = relu(fc1 s), a = sigmoid(fc2 h); blend table (a, 1-a, 0).  The kernel is a
// chain of dependent L2 round trips on the stage's critical path, so each phase puts all its loads in flight at
// once: 16 waves x 4 rows of fc1 per pass, 16-byte loads of the fc2 rows.  (Round 3, tried: both weight matrices requested
// into registers before the first barrier -- 96 more registers in a 1024-thread block: 15.9 -> 20.5 us, reverted.)
constexpr int kFcBlock = 1024;
__global__ __launch_bounds__(kFcBlock) void fc_forward_kernel(const float* __restrict__ part, const float* __restrict__ w1,
                                                              const float* __restrict__ b1, const float* __restrict__ w2,
                                                              const float* __restrict__ b2, float* __restrict__ s,
                                                              float* __restrict__ hbuf, float* __restrict__ a,
                                                              float* __restrict__ tab, int c, int r, int hw,
                                                              int* __restrict__ ticks, int n_ticks) {
  extern __shared__ float sh[];  // s (2c) | h (r)
  if (blockIdx.x == 0)           // the arrival counters of this call's later kernels (SavedLayout::tick)
    for (int i = threadIdx.x; i < n_ticks; i += kFcBlock) ticks[i] = 0;
  float* ss = sh;
  float* hh = sh + 2 * c;
  const int b = blockIdx.x, c2 = 2 * c;
  for (int i = threadIdx.x; i < c2; i += kFcBlock) {
    const float* q = part + ((size_t)b * c2 + i) * kPlaneChunks;
    float v = ((q[0] + q[1]) + (q[2] + q[3])) / (float)hw;
    ss[i] = v;
    s[(size_t)b * c2 + i] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // four rows per wave and pass so that their loads are in flight together
  for (int j0 = wv * 4; j0 < r; j0 += 4 * (kFcBlock / DHD_WAVE)) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = lane; i < c2; i += DHD_WAVE) {
      const float sv = ss[i];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (j0 + q < r) acc[q] = fmaf(w1[(size_t)(j0 + q) * c2 + i], sv, acc[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float t = group_sum(acc[q], DHD_WAVE);
      if (lane == 0 && j0 + q < r) {
        const float v = fmaxf(t + b1[j0 + q], 0.f);
        hh[j0 + q] = v;
        hbuf[(size_t)b * r + j0 + q] = v;
      }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < c; k += kFcBlock) {
    float acc = b2[k];
    if ((r & 3) == 0) {
      const f32x4* wrow = reinterpret_cast<const f32x4*>(w2 + (size_t)k * r);
#pragma unroll 8
      for (int j = 0; j < (r >> 2); ++j) {
        const f32x4 wv4 = wrow[j];
        acc = fmaf(wv4.x, hh[4 * j], acc);
        acc = fmaf(wv4.y, hh[4 * j + 1], acc);
        acc = fmaf(wv4.z, hh[4 * j + 2], acc);
        acc = fmaf(wv4.w, hh[4 * j + 3], acc);
      }
    } else {
      for (int j = 0; j < r; ++j) acc = fmaf(w2[(size_t)k * r + j], hh[j], acc);
    }
    float v = sigmoidf_(acc);
    a[(size_t)b * c + k] = v;
    float* t = tab + (size_t)b * 3 * c;
    t[k] = v;
    t[c + k] = 1.0f - v;
    t[2 * c + k] = 0.f;
  }
}

// one block per sample: da (from the two partial sets) -> dpre2, dh, ds
__global__ __launch_bounds__(kEwBlock) void fc_backward_kernel(const float* __restrict__ da_p1, const float* __restrict__ da_p2,
                                                               const float* __restrict__ a, const float* __restrict__ hbuf,
                                                               const float* __restrict__ w1, const float* __restrict__ w2,
                                                               float* __restrict__ dpre2, float* __restrict__ dh,
                                                               float* __restrict__ ds, int c, int r) {
  extern __shared__ float sh[];  // dpre2 (c) | dh (r) | partials (kEwBlock)
  float* sp = sh;
  float* sd = sh + c;
  const int b = blockIdx.x, c2 = 2 * c;
  for (int k = threadIdx.x; k < c; k += kEwBlock) {
    float g = 0.f;
    for (int q = 0; q < kPlaneChunks; ++q)
      g += da_p1[((size_t)b * kPlaneChunks + q) * c + k] + da_p2[((size_t)b * kPlaneChunks + q) * c + k];
    const float av = a[(size_t)b * c + k];
    const float v = g * av * (1.0f - av);
    sp[k] = v;
    dpre2[(size_t)b * c + k] = v;
  }
  __syncthreads();
  if (r <= kEwBlock && kEwBlock % r == 0) {
    // all threads: thread (group, j) sums k = group, group + groups, ...; partials through LDS
    float* pp = sh + c + r;  // groups * r partials
    const int groups = kEwBlock / r, j = threadIdx.x % r, grp = threadIdx.x / r;
    float acc = 0.f;
#pragma unroll 8
    for (int k = grp; k < c; k += groups) acc = fmaf(w2[(size_t)k * r + j], sp[k], acc);
    pp[grp * r + j] = acc;
    __syncthreads();
    if (threadIdx.x < r) {
      float t = 0.f;
      for (int q = 0; q < groups; ++q) t += pp[q * r + threadIdx.x];
      const float v = hbuf[(size_t)b * r + threadIdx.x] > 0.f ? t : 0.f;
      sd[threadIdx.x] = v;
      dh[(size_t)b * r + threadIdx.x] = v;
    }
  } else {
    for (int j = threadIdx.x; j < r; j += kEwBlock) {
      float acc = 0.f;
      for (int k = 0; k < c; ++k) acc = fmaf(w2[(size_t)k * r + j], sp[k], acc);
      const float v = hbuf[(size_t)b * r + j] > 0.f ? acc : 0.f;
      sd[j] = v;
      dh[(size_t)b * r + j] = v;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < c2; i += kEwBlock) {
    float acc = 0.f;
#pragma unroll 8
    for (int j = 0; j < r; ++j) acc = fmaf(w1[(size_t)j * c2 + i], sd[j], acc);
    ds[(size_t)b * c2 + i] = acc;
  }
}

// parameter gradients of the two Linear layers: one thread per output element.  Runs as extra block rows of
// stage_gx_kernel's launch (both only need fc_backward_kernel's results): one launch less on the stage's critical path.
struct FcGradJob {
  const float* dpre2;
  const float* dh;
  const float* hbuf;
  const float* s;
  float* gw1;
  float* gb1;
  float* gw2;
  float* gb2;
  int nb, r;
};
__device__ __forceinline__ void fc_param_grad_block(const FcGradJob& J, int block, int c) {
  const float* __restrict__ dpre2 = J.dpre2;
  const float* __restrict__ dh = J.dh;
  const float* __restrict__ hbuf = J.hbuf;
  const float* __restrict__ s = J.s;
  float* __restrict__ gw1 = J.gw1;
  float* __restrict__ gb1 = J.gb1;
  float* __restrict__ gw2 = J.gw2;
  float* __restrict__ gb2 = J.gb2;
  const int nb = J.nb, r = J.r;
  const int c2 = 2 * c;
  const int n_w1 = r * c2, n_w2 = c * r;
  int i = block * kEwBlock + threadIdx.x;
  if (i < n_w1) {
    const int j = i / c2, k = i % c2;
    float acc = 0.f;
    for (int b = 0; b < nb; ++b) acc = fmaf(dh[(size_t)b * r + j], s[(size_t)b * c2 + k], acc);
    gw1[i] = acc;
    return;
  }
  i -= n_w1;
  if (i < n_w2) {
    const int k = i / r, j = i % r;
    float acc = 0.f;
    for (int b = 0; b < nb; ++b) acc = fmaf(dpre2[(size_t)b * c + k], hbuf[(size_t)b * r + j], acc);
    gw2[i] = acc;
    return;
  }
  i -= n_w2;
  if (i < r) {
    float acc = 0.f;
    for (int b = 0; b < nb; ++b) acc += dh[(size_t)b * r + i];
    gb1[i] = acc;
    return;
  }
  i -= r;
  if (i < c) {
    float acc = 0.f;
    for (int b = 0; b < nb; ++b) acc += dpre2[(size_t)b * c + i];
    gb2[i] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm statistics
// ------------------------------------------------------------------------------------------------

// per plane chunk: sum (y - K), sum (y - K)^2 with K = y[0, ch, 0] (a sample of the channel, so the
// shifted sums do not cancel).  part: [(b*chunks + chunk)][2][c]
__global__ __launch_bounds__(kEwBlock) void moments_kernel(const float* __restrict__ y, float* __restrict__ part, int c, int hw) {
  __shared__ float sm[kEwBlock / DHD_WAVE];
  const int plane = blockIdx.y, b = plane / c, ch = plane % c;
  const float k = y[(size_t)ch * hw];
  const f32x4* p4 = reinterpret_cast<const f32x4*>(y + (size_t)plane * hw);
  int lo, hi;
  chunk_range4(hw, &lo, &hi);
  float s1 = 0.f, s2 = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += kEwBlock) {
    f32x4 v = p4[i];
    const float d0 = v.x - k, d1 = v.y - k, d2 = v.z - k, d3 = v.w - k;
    s1 += (d0 + d1) + (d2 + d3);
    s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  s1 = block_sum(s1, sm);
  s2 = block_sum(s2, sm);
  if (threadIdx.x == 0) {
    float* q = part + ((size_t)(b * kPlaneChunks + blockIdx.x) * 2) * c;
    q[ch] = s1;
    q[c + ch] = s2;
  }
}

// Batch statistics -> mean, rstd, (scale, shift), prologue table (scale, 0, shift) per sample;
// running statistics updated like torch.nn.BatchNorm2d (biased variance normalises, unbiased feeds
// the running estimate).
__global__ __launch_bounds__(kEwBlock) void bn_train_finalize_kernel(const float* __restrict__ part, int n_part,
                                                                     const float* __restrict__ shift, int shift_stride,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     float* __restrict__ run_mean, float* __restrict__ run_var,
                                                                     float momentum, float eps, float* __restrict__ mean,
                                                                     float* __restrict__ rstd, float* __restrict__ scsh,
                                                                     float* __restrict__ tab, int nb, int c, int hw,
                                                                     long long* __restrict__ batches_tracked) {
  const int ch = blockIdx.x * kEwBlock + threadIdx.x;
  if (ch >= c) return;
  if (ch == 0 && batches_tracked) *batches_tracked += 1;   // nn.BatchNorm2d.num_batches_tracked
  double s1 = 0.0, s2 = 0.0;
#pragma unroll 8
  for (int q = 0; q < n_part; ++q) {
    s1 += (double)part[((size_t)q * 2) * c + ch];
    s2 += (double)part[((size_t)q * 2 + 1) * c + ch];
  }
  const double n = (double)nb * (double)hw;
  const double md = s1 / n;
  double var = s2 / n - md * md;
  if (var < 0.0) var = 0.0;
  const double mu = (double)shift[(size_t)ch * shift_stride] + md;
  const float rs = (float)(1.0 / sqrt(var + (double)eps));
  mean[ch] = (float)mu;
  rstd[ch] = rs;
  const float sc = gamma[ch] * rs, shf = beta[ch] - (float)mu * sc;
  scsh[ch] = sc;
  scsh[c + ch] = shf;
  for (int b = 0; b < nb; ++b) {
    float* t = tab + (size_t)b * 3 * c;
    t[ch] = sc;
    t[c + ch] = 0.f;
    t[2 * c + ch] = shf;
  }
  if (run_mean) {
    const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
    run_mean[ch] = (float)((1.0 - (double)momentum) * (double)run_mean[ch] + (double)momentum * mu);
    run_var[ch] = (float)((1.0 - (double)momentum) * (double)run_var[ch] + (double)momentum * unb);
  }
}

__global__ __launch_bounds__(kEwBlock) void bn_eval_coef_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ run_mean, const float* __restrict__ run_var,
                                                                float eps, float* __restrict__ mean, float* __restrict__ rstd,
                                                                float* __restrict__ scsh, float* __restrict__ tab, int nb, int c) {
  const int ch = blockIdx.x * kEwBlock + threadIdx.x;
  if (ch >= c) return;
  const float rs = 1.0f / sqrtf(run_var[ch] + eps);
  mean[ch] = run_mean[ch];
  rstd[ch] = rs;
  const float sc = gamma[ch] * rs, shf = beta[ch] - run_mean[ch] * sc;
  scsh[ch] = sc;
  scsh[c + ch] = shf;
  for (int b = 0; b < nb; ++b) {
    float* t = tab + (size_t)b * 3 * c;
    t[ch] = sc;
    t[c + ch] = 0.f;
    t[2 * c + ch] = shf;
  }
}

// BatchNorm backward coefficients: dy = c0*g + c1*y + c2 (per channel), dgamma, dbeta, and the
// gradient of the bias of the convolution feeding this BatchNorm (sum of dy).
//   training: dy = gamma*rstd*(g - S1/n - (y-mu)*rstd^2*S2/n);  eval: dy = gamma*rstd*g
// COHERENT: the partial sums were written by other workgroups of the SAME launch (see BnTail): read them past the
// non-coherent cache levels.
template <bool COHERENT>
__device__ __forceinline__ void bn_backward_coef_channel(const float* __restrict__ part, int ch, const float* __restrict__ gamma,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         int training, float* __restrict__ tab, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, float* __restrict__ dbias, int nb, int c, int hw,
                                                         double* __restrict__ sums_out, const double* __restrict__ sums_in,
                                                         const double* __restrict__ loc_fwd, const float* __restrict__ shift) {
  // Cross-rank statistics (nn.SyncBatchNorm): with `sums_out` only this rank's sums are written, [sum g][C] |
  // [sum g (y - mu)][C] | count, as doubles; with `sums_in` (their all-reduced values) the input-gradient coefficients use
  // the global sums and count, while dgamma / dbeta stay this rank's sums (the caller's DDP averages parameter gradients),
  // and the convolution-bias gradient is this rank's sum of dy, which no longer vanishes rank by rank:
  //   sum_local dy = c0 S1_local + c1 sum_local y + n_local c2,   sum_local y = loc_fwd[ch] + n_local shift[ch]  (the forward
  //   sums are those of y - shift).
  double s1 = 0.0, s2 = 0.0;
  if (COHERENT) {
    float v1[16], v2[16];   // in flight together; the sum keeps the order of the plain loop
    for (int q0 = 0; q0 < nb * kPlaneChunks; q0 += 16) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int q = min(q0 + u, nb * kPlaneChunks - 1);
        v1[u] = __hip_atomic_load(part + ((size_t)q * 2) * c + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v2[u] = __hip_atomic_load(part + ((size_t)q * 2 + 1) * c + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (q0 + u < nb * kPlaneChunks) { s1 += (double)v1[u]; s2 += (double)v2[u]; }
    }
  } else {
    for (int q = 0; q < nb * kPlaneChunks; ++q) {
      s1 += (double)part[((size_t)q * 2) * c + ch];
      s2 += (double)part[((size_t)q * 2 + 1) * c + ch];
    }
  }
  const double n_loc = (double)nb * (double)hw;
  if (sums_out) {
    sums_out[ch] = s1; sums_out[c + ch] = s2;
    if (ch == 0) sums_out[2 * c] = n_loc;
    return;
  }
  const double n = sums_in ? sums_in[2 * c] : n_loc;
  const double g1 = sums_in ? sums_in[ch] : s1, g2 = sums_in ? sums_in[c + ch] : s2;
  const double rs = (double)rstd[ch], mu = (double)mean[ch], ga = (double)gamma[ch];
  if (dgamma) dgamma[ch] = (float)(rs * s2);
  if (dbeta) dbeta[ch] = (float)s1;
  double c0 = ga * rs, c1 = 0.0, c2 = 0.0, db = c0 * s1;
  if (training) {
    c1 = -ga * rs * rs * rs * g2 / n;
    c2 = -ga * rs * g1 / n - c1 * mu;
    db = 0.0;  // sum of dy over the (whole) batch vanishes identically
    if (sums_in) db = c0 * s1 + c1 * (loc_fwd[ch] + n_loc * (double)shift[ch]) + n_loc * c2;
  }
  if (dbias) dbias[ch] = (float)db;
  for (int b = 0; b < nb; ++b) {
    float* t = tab + (size_t)b * 3 * c;
    t[ch] = (float)c0;
    t[c + ch] = (float)c1;
    t[2 * c + ch] = (float)c2;
  }
}

__global__ __launch_bounds__(kEwBlock) void bn_backward_coef_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                    int training, float* __restrict__ tab, float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta, float* __restrict__ dbias, int nb,
                                                                    int c, int hw, double* __restrict__ sums_out,
                                                                    const double* __restrict__ sums_in, const double* __restrict__ loc_fwd,
                                                                    const float* __restrict__ shift) {
  const int ch = blockIdx.x * kEwBlock + threadIdx.x;
  if (ch >= c) return;
  bn_backward_coef_channel<false>(part, ch, gamma, mean, rstd, training, tab, dgamma, dbeta, dbias, nb, c, hw, sums_out, sums_in,
                                  loc_fwd, shift);
}

// The coefficient kernel above as the TAIL of the pass that produces its sums (one launch and ~7 us of latency chain less
// per BatchNorm backward).  The nb * kPlaneChunks workgroups of a channel count themselves in ticket[ch]; the one that
// arrives last computes the channel's coefficients in the plain kernel's summation order (bit-identical results) and leaves
// the counter at zero for the next launch.  Visibility across the eight XCDs (their L2s are not coherent with each other
// inside a kernel): the partial sums are written with agent-scope atomic stores (write-through), their acknowledgement is
// awaited (s_waitcnt) before the agent-scope ticket, and the last workgroup reads them with agent-scope atomic loads.  NOT
// with __threadfence(): an agent-scope release writes back the XCD's whole L2, and in a pass that is streaming 160 MB of
// results through it that made the pass 2.8x (blend2_bn_bwd 134 -> 375 us) and 4.7x (pair_sums 56 -> 262 us) slower.
// The counters live in the forward's `saved` block, which the forward zeroes.  ticket == nullptr (cross-rank statistics:
// the sums go through an all-reduce first): plain stores, no tail.
struct BnTail {
  int* ticket;
  const float* gamma;
  const float* mean;
  const float* rstd;
  float* tab;
  float* dgamma;
  float* dbeta;
  float* dbias;
  int training, nb, hw;
};
// thread 0 of a workgroup: its partial sums (row q of `part`) and, if it was the channel's last, the coefficients
__device__ __forceinline__ void bn_backward_publish(const BnTail& t, float* __restrict__ part, size_t q, int ch, int c, float s1,
                                                    float s2) {
  float* p1 = part + (q * 2) * c + ch;
  float* p2 = part + (q * 2 + 1) * c + ch;
  if (t.ticket == nullptr) {
    *p1 = s1;
    *p2 = s2;
    return;
  }
  __hip_atomic_store(p1, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(p2, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (__hip_atomic_fetch_add(t.ticket + ch, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != t.nb * kPlaneChunks - 1) return;
  __hip_atomic_store(t.ticket + ch, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bn_backward_coef_channel<true>(part, ch, t.gamma, t.mean, t.rstd, t.training, t.tab, t.dgamma, t.dbeta, t.dbias, t.nb, c, t.hw,
                                 nullptr, nullptr, nullptr, nullptr);
}

// sums for BatchNorm backward: S1 = sum g, S2 = sum g*(y - mean).  part: [(b*chunks+chunk)][2][c]
__global__ __launch_bounds__(kEwBlock) void pair_sums_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                             const float* __restrict__ mean, float* __restrict__ part, int c, int hw,
                                                             BnTail tail) {
  __shared__ float sm[kEwBlock / DHD_WAVE];
  const int plane = blockIdx.y, b = plane / c, ch = plane % c;
  const float mu = mean[ch];
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g + (size_t)plane * hw);
  const f32x4* y4 = reinterpret_cast<const f32x4*>(y + (size_t)plane * hw);
  int lo, hi;
  chunk_range4(hw, &lo, &hi);
  float s1 = 0.f, s2 = 0.f, t1 = 0.f, t2 = 0.f;
  int i = lo + threadIdx.x;
  for (; i + kEwBlock < hi; i += 2 * kEwBlock) {  // two independent 16-byte streams per thread
    // streamed once here: non-temporal, like plane_mean (62 -> 52 us)
    f32x4 a = __builtin_nontemporal_load(g4 + i), v = __builtin_nontemporal_load(y4 + i), a2 = __builtin_nontemporal_load(g4 + i + kEwBlock),
          v2 = __builtin_nontemporal_load(y4 + i + kEwBlock);
    s1 += (a.x + a.y) + (a.z + a.w);
    s2 += (a.x * (v.x - mu) + a.y * (v.y - mu)) + (a.z * (v.z - mu) + a.w * (v.w - mu));
    t1 += (a2.x + a2.y) + (a2.z + a2.w);
    t2 += (a2.x * (v2.x - mu) + a2.y * (v2.y - mu)) + (a2.z * (v2.z - mu) + a2.w * (v2.w - mu));
  }
  if (i < hi) {
    f32x4 a = g4[i], v = y4[i];
    s1 += (a.x + a.y) + (a.z + a.w);
    s2 += (a.x * (v.x - mu) + a.y * (v.y - mu)) + (a.z * (v.z - mu) + a.w * (v.w - mu));
  }
  s1 += t1;
  s2 += t2;
  s1 = block_sum(s1, sm);
  s2 = block_sum(s2, sm);
  if (threadIdx.x == 0) {
    bn_backward_publish(tail, part, (size_t)(b * kPlaneChunks + blockIdx.x), ch, c, s1, s2);
  }
}

// ------------------------------------------------------------------------------------------------
// fused blends
// ------------------------------------------------------------------------------------------------

// Four consecutive elements of the stage's edge tensors (out, gout, gx: dhd_sfa_weights.io_dtype) as float32: 16 bytes of
// float32, 8 bytes of a half type (widened exactly / rounded to nearest even).
template <class T> __device__ __forceinline__ f32x4 ld4(const T* base, size_t i4);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* base, size_t i4) {
  return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base) + i4);
}
template <> __device__ __forceinline__ f32x4 ld4<_Float16>(const _Float16* base, size_t i4) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const h4 v = __builtin_nontemporal_load(reinterpret_cast<const h4*>(base) + i4);
  return f32x4{(float)v.x, (float)v.y, (float)v.z, (float)v.w};
}
template <> __device__ __forceinline__ f32x4 ld4<__bf16>(const __bf16* base, size_t i4) {
  const u32x2 w = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(base) + i4);
  return f32x4{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u)};
}
template <class T> __device__ __forceinline__ void st4(T* base, size_t i4, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* base, size_t i4, f32x4 v) {
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(base) + i4);
}
template <> __device__ __forceinline__ void st4<_Float16>(_Float16* base, size_t i4, f32x4 v) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const h4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
  __builtin_nontemporal_store(h, reinterpret_cast<h4*>(base) + i4);
}
template <> __device__ __forceinline__ void st4<__bf16>(__bf16* base, size_t i4, f32x4 v) {
  typedef __bf16 b4 __attribute__((ext_vector_type(4)));
  const b4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  __builtin_nontemporal_store(h, reinterpret_cast<b4*>(base) + i4);
}

// out = g*(a*xb) + (1-g)*((1-a)*xv),  g = sigmoid(sc*y2 + sh)
template <class TO>
__global__ __launch_bounds__(kEwBlock) void blend2_bn_kernel(const float* __restrict__ x, const float* __restrict__ a1,
                                                             const float* __restrict__ y2, const float* __restrict__ scsh,
                                                             TO* __restrict__ out, int c, int hw) {
  const int plane = blockIdx.y, b = plane / c, ch = plane % c;
  const float a = a1[plane], na = 1.0f - a, sc = scsh[ch], sh = scsh[c + ch];
  const f32x4* b4 = reinterpret_cast<const f32x4*>(x + ((size_t)b * 2 * c + ch) * hw);
  const f32x4* v4 = reinterpret_cast<const f32x4*>(x + ((size_t)b * 2 * c + c + ch) * hw);
  const f32x4* y4 = reinterpret_cast<const f32x4*>(y2 + (size_t)plane * hw);
  TO* o4 = out + (size_t)plane * hw;
  int lo, hi;
  chunk_range4(hw, &lo, &hi);
  for (int i = lo + threadIdx.x; i < hi; i += kEwBlock) {
    f32x4 p = __builtin_nontemporal_load(b4 + i), q = __builtin_nontemporal_load(v4 + i), s = y4[i], r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g = sigmoidf_(fmaf(sc, s[j], sh));
      r[j] = g * (a * p[j]) + (1.0f - g) * (na * q[j]);
    }
    st4<TO>(o4, i, r);
  }
}

// g2 = dL/d s2 = go*(a*xb - (1-a)*xv)*g*(1-g); sums for BatchNorm-2 backward; the go-part of dL/da:
// sum go*(g*xb - (1-g)*xv).   part: [(b*chunks+chunk)][2][c];  da_p1: [(b*chunks+chunk)][c]
template <class TO>
__global__ __launch_bounds__(kEwBlock) void blend2_bn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ a1,
                                                                 const float* __restrict__ y2, const float* __restrict__ scsh,
                                                                 const float* __restrict__ mean, const TO* __restrict__ go,
                                                                 float* __restrict__ g2, float* __restrict__ part,
                                                                 float* __restrict__ da_p1, int c, int hw, BnTail tail) {
  __shared__ float sm[kEwBlock / DHD_WAVE];
  const int plane = blockIdx.y, b = plane / c, ch = plane % c;
  const float a = a1[plane], na = 1.0f - a, sc = scsh[ch], sh = scsh[c + ch], mu = mean[ch];
  const f32x4* b4 = reinterpret_cast<const f32x4*>(x + ((size_t)b * 2 * c + ch) * hw);
  const f32x4* v4 = reinterpret_cast<const f32x4*>(x + ((size_t)b * 2 * c + c + ch) * hw);
  const f32x4* y4 = reinterpret_cast<const f32x4*>(y2 + (size_t)plane * hw);
  const TO* o4 = go + (size_t)plane * hw;
  f32x4* r4 = reinterpret_cast<f32x4*>(g2 + (size_t)plane * hw);
  int lo, hi;
  chunk_range4(hw, &lo, &hi);
  float s1 = 0.f, s2 = 0.f, sa = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += kEwBlock) {
    f32x4 p = __builtin_nontemporal_load(b4 + i), q = __builtin_nontemporal_load(v4 + i), s = y4[i], o = ld4<TO>(o4, i), r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g = sigmoidf_(fmaf(sc, s[j], sh));
      const float v = o[j] * (a * p[j] - na * q[j]) * g * (1.0f - g);
      r[j] = v;
      s1 += v;
      s2 = fmaf(v, s[j] - mu, s2);
      sa = fmaf(o[j], g * p[j] - (1.0f - g) * q[j], sa);
    }
    r4[i] = r;
  }
  s1 = block_sum(s1, sm);
  s2 = block_sum(s2, sm);
  sa = block_sum(sa, sm);
  if (threadIdx.x == 0) {
    const size_t q = (size_t)(b * kPlaneChunks + blockIdx.x);
    da_p1[q * c + ch] = sa;
    bn_backward_publish(tail, part, q, ch, c, s1, s2);
  }
}

// the du-part of dL/da: sum du*(xb - xv).   da_p2: [(b*chunks+chunk)][c]
__global__ __launch_bounds__(kEwBlock) void blend1_da_kernel(const float* __restrict__ x, const float* __restrict__ du,
                                                             float* __restrict__ da_p2, int c, int hw) {
  __shared__ float sm[kEwBlock / DHD_WAVE];
  const int plane = blockIdx.y, b = plane / c, ch = plane % c;
  const f32x4* b4 = reinterpret_cast<const f32x4*>(x + ((size_t)b * 2 * c + ch) * hw);
  const f32x4* v4 = reinterpret_cast<const f32x4*>(x + ((size_t)b * 2 * c + c + ch) * hw);
  const f32x4* d4 = reinterpret_cast<const f32x4*>(du + (size_t)plane * hw);
  int lo, hi;
  chunk_range4(hw, &lo, &hi);
  float acc = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += kEwBlock) {
    f32x4 p = __builtin_nontemporal_load(b4 + i), q = __builtin_nontemporal_load(v4 + i), d = d4[i];
    acc += (d.x * (p.x - q.x) + d.y * (p.y - q.y)) + (d.z * (p.z - q.z) + d.w * (p.w - q.w));
  }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) da_p2[(size_t)(b * kPlaneChunks + blockIdx.x) * c + ch] = acc;
}

// gx_bev = a*(go*g + du) + ds_bev/hw;  gx_vox = (1-a)*(go*(1-g) + du) + ds_vox/hw
template <class TO>
__global__ __launch_bounds__(kEwBlock) void stage_gx_kernel(const float* __restrict__ a1, const float* __restrict__ y2,
                                                            const float* __restrict__ scsh, const TO* __restrict__ go,
                                                            const float* __restrict__ du, const float* __restrict__ ds,
                                                            TO* __restrict__ gx, int c, int hw, int fc_rows, FcGradJob fc) {
  if ((int)blockIdx.y < fc_rows) {   // the first block rows: the Linear layers' parameter gradients (dispatched first, no tail)
    fc_param_grad_block(fc, (int)blockIdx.y * kPlaneChunks + (int)blockIdx.x, c);
    return;
  }
  const int plane = (int)blockIdx.y - fc_rows, b = plane / c, ch = plane % c;
  const float a = a1[plane], na = 1.0f - a, sc = scsh[ch], sh = scsh[c + ch];
  const float kb = ds[(size_t)b * 2 * c + ch] / (float)hw, kv = ds[(size_t)b * 2 * c + c + ch] / (float)hw;
  const f32x4* y4 = reinterpret_cast<const f32x4*>(y2 + (size_t)plane * hw);
  const TO* o4 = go + (size_t)plane * hw;
  const f32x4* d4 = reinterpret_cast<const f32x4*>(du + (size_t)plane * hw);
  TO* gb4 = gx + ((size_t)b * 2 * c + ch) * hw;
  TO* gv4 = gx + ((size_t)b * 2 * c + c + ch) * hw;
  int lo, hi;
  chunk_range4(hw, &lo, &hi);
  for (int i = lo + threadIdx.x; i < hi; i += kEwBlock) {
    f32x4 s = __builtin_nontemporal_load(y4 + i), o = ld4<TO>(o4, i), d = __builtin_nontemporal_load(d4 + i), rb, rv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g = sigmoidf_(fmaf(sc, s[j], sh));
      rb[j] = fmaf(a, fmaf(o[j], g, d[j]), kb);
      rv[j] = fmaf(na, fmaf(o[j], 1.0f - g, d[j]), kv);
    }
    st4<TO>(gb4, i, rb);
    st4<TO>(gv4, i, rv);
  }
}

#include "sfa_gemm_streamed.h"   // pw_gemm / pw_gemm6 / pw_wgrad / pw_wgrad6: the f32 reference point and bf16x6 at C = 512
#include "sfa_gemm_res.h"        // pw_gemm_res: bf16x3 at C = 512, bf16x6 at C = 128 / 256

// The forward's first launch: the channel means of x (blockIdx.y < n_planes) and, in rows of extra blocks, all FOUR weight
// images of the call -- conv1 / conv2 for the forward GEMMs and their transposes for the backward's data-gradient GEMMs, which
// round 2 packed in two separate launches (one per direction) on the critical path of either pass.
struct PackJob {
  const float* w[2];     // conv1, conv2
  u32x4* dst[4];         // conv1, conv2, conv1^T, conv2^T
  int c, cob, nt, blocks_each;
  int cu;                // images for pw_gemm_cu_kernel (sfa_gemm_cu.h: A fragments, one 32-channel tile after the other)
};

__global__ __launch_bounds__(kEwBlock) void plane_mean_pack_kernel(const float* __restrict__ x, float* __restrict__ part, int hw,
                                                                   int n_planes, PackJob job) {
  __shared__ float sm[kEwBlock / DHD_WAVE];
  if ((int)blockIdx.y < n_planes) { plane_mean_block(x, part, hw, sm); return; }
  const int pb = ((int)blockIdx.y - n_planes) * kPlaneChunks + (int)blockIdx.x;
  const int which = pb / job.blocks_each;
  if (which >= 4) return;
  if (job.cu) cu_pack_weight(job.w[which & 1], which >> 1, job.dst[which], job.c, (pb % job.blocks_each) * kEwBlock + (int)threadIdx.x);
  else pack_weight_res_block(job.w[which & 1], which >> 1, job.dst[which], job.c, job.cob, job.nt, pb % job.blocks_each);
}

// The GEMM epilogues' statistics rows [n][2][c] (n = samples x wave tiles: 5000 rows, 10 MB at B = 4) -> batch statistics and
// everything bn_train_finalize_kernel derives from them, in ONE launch (stat_reduce_kernel + bn_train_finalize_kernel took
// 9 + 7 us as two dependent launches).  A workgroup owns four channels: thread t sums rows t, t + 256, ... with two 16-byte
// loads per row (sum and sum of squares), the 256 partial sums meet in LDS as doubles, threads 0-3 finalize one channel each.
// Workgroup -> channel group is XCD-aware: the workgroups on one XCD (ids = x mod 8) take neighbouring channel groups, so that
// the 128-byte lines of a row are fetched into one L2 only.
__global__ __launch_bounds__(kEwBlock) void bn_stats_finalize_kernel(const float* __restrict__ part, int n,
                                                                     const float* __restrict__ shift, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, float* __restrict__ run_mean,
                                                                     float* __restrict__ run_var, float momentum, float eps,
                                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                                     float* __restrict__ scsh, float* __restrict__ tab, int nb, int c,
                                                                     int hw, double* __restrict__ sums_out, double* __restrict__ sums_out2,
                                                                     const double* __restrict__ sums_in, long long* __restrict__ batches_tracked) {
  // Cross-rank statistics (nn.SyncBatchNorm, dhd_sfa_stage_*_phase): with `sums_out` the kernel stops after the row
  // reduction and leaves this rank's shifted sums as doubles, [sum (y - shift)][C] | [sum (y - shift)^2][C] | count, in
  // sums_out and sums_out2; with `sums_in` it starts from such a vector (all-reduced by the caller) instead of the rows.
  __shared__ double sm[kEwBlock][8];
  const int nblk = gridDim.x, t = threadIdx.x;
  const int cg = nblk >= 8 ? (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3) : blockIdx.x;   // nblk is a multiple of 8 here
  const int ch0 = 4 * cg;
  const f32x4* p1 = reinterpret_cast<const f32x4*>(part + ch0);
  const f32x4* p2 = reinterpret_cast<const f32x4*>(part + c + ch0);
  const size_t row4 = (size_t)(2 * c) / 4;   // f32x4 units per row
  f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1, b1 = a1, b2 = a1;
  if (sums_in) n = 0;
  int i = t;
  for (; i + kEwBlock < n; i += 2 * kEwBlock) {
    a1 += p1[(size_t)i * row4];
    a2 += p2[(size_t)i * row4];
    b1 += p1[(size_t)(i + kEwBlock) * row4];
    b2 += p2[(size_t)(i + kEwBlock) * row4];
  }
  if (i < n) {
    a1 += p1[(size_t)i * row4];
    a2 += p2[(size_t)i * row4];
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sm[t][e] = (double)a1[e] + (double)b1[e];
    sm[t][4 + e] = (double)a2[e] + (double)b2[e];
  }
  __syncthreads();
  for (int s = kEwBlock / 2; s > 0; s >>= 1) {
    if (t < s) {
#pragma unroll
      for (int e = 0; e < 8; ++e) sm[t][e] += sm[t + s][e];
    }
    __syncthreads();
  }
  if (t >= 4) return;
  const int ch = ch0 + t;
  double s1 = sm[0][t], s2 = sm[0][4 + t];
  double cnt = (double)nb * (double)hw;
  if (sums_out) {
    sums_out[ch] = s1; sums_out[c + ch] = s2;
    sums_out2[ch] = s1; sums_out2[c + ch] = s2;
    if (ch == 0) { sums_out[2 * c] = cnt; sums_out2[2 * c] = cnt; }
    return;
  }
  if (sums_in) { s1 = sums_in[ch]; s2 = sums_in[c + ch]; cnt = sums_in[2 * c]; }
  if (ch == 0 && batches_tracked) *batches_tracked += 1;   // nn.BatchNorm2d.num_batches_tracked (one kernel launch less per BatchNorm)
  const double md = s1 / cnt;
  double var = s2 / cnt - md * md;
  if (var < 0.0) var = 0.0;
  const double mu = (double)shift[ch] + md;
  const float rs = (float)(1.0 / sqrt(var + (double)eps));
  mean[ch] = (float)mu;
  rstd[ch] = rs;
  const float sc = gamma[ch] * rs, shf = beta[ch] - (float)mu * sc;
  scsh[ch] = sc;
  scsh[c + ch] = shf;
  for (int b = 0; b < nb; ++b) {
    float* q = tab + (size_t)b * 3 * c;
    q[ch] = sc;
    q[c + ch] = 0.f;
    q[2 * c + ch] = shf;
  }
  if (run_mean) {
    const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
    run_mean[ch] = (float)((1.0 - (double)momentum) * (double)run_mean[ch] + (double)momentum * mu);
    run_var[ch] = (float)((1.0 - (double)momentum) * (double)run_var[ch] + (double)momentum * unb);
  }
}

// Weight gradient in the bf16x3 mode: G[co][ci] = sum_{b,p} A(co,p) * B(ci,p) with two bf16 parts per operand and the
// three products ah*bh + ah*bm + am*bh.  Same organisation as pw_wgrad6_kernel (items of 8 pixels are prologue'd and
// split ONCE by one thread and written to LDS in MFMA fragment order), but
//   * a step is 32 pixels = two MFMA k-steps: every channel row contributes one whole 128-byte line per step, so no line
//     is fetched twice -- the 16-pixel steps of the x6 kernel re-fetch the other half of each line one step later, after
//     the XCD's L2 has been turned over (PMC: 980 MB against 656 MB algorithmic) -- and there is one workgroup barrier
//     per 32 pixels instead of per 16.  Two parts instead of three make the 32-pixel double buffer fit in LDS (2 x 64 KB);
//   * the loads are line-coalesced: in one load instruction 8 adjacent lanes read the 8 four-pixel pieces of ONE row's
//     line (a wave instruction = 8 whole lines), instead of every lane reading 16 bytes of a different row (64 lines
//     touched per instruction, each line touched by four instructions).  An 8-pixel item is then assembled with one
//     lane-pair exchange (DPP quad_perm): of the rows loaded by instructions 2a and 2a+1, the even lane keeps its piece
//     of row 2a and takes its neighbour's, the odd lane does the same for row 2a+1.
template <int OT, bool A_TWO, bool B_TWO, bool B_RELU>
__global__ __launch_bounds__(kWgBlock, 1) void pw_wgrad3_kernel(const float* __restrict__ a0, const float* __restrict__ a1,
                                                                const float* __restrict__ acoef, size_t a_bstride,
                                                                const float* __restrict__ b0, const float* __restrict__ b1,
                                                                const float* __restrict__ bcoef, size_t b_bstride,
                                                                float* __restrict__ partial, int c, int hw, int nb, int n_workers) {
  constexpr int TA = OT / 64, TB = OT / 128;     // 32x32 tiles per wave
  constexpr int kTiles = OT / 32;                // 32-row tiles per operand
  constexpr int kOp = kTiles * 2 * 64;           // 16-byte units of one staged operand k-step: [tile][term][lane]
  constexpr int kBuf = 2 * 2 * kOp;              // [k-step 2][operand 2]
  constexpr int NJ = OT / 64;                    // load instructions per operand input and step (64 rows each)
  constexpr int NA = NJ / 2;                     // 8-pixel items per thread, operand and step
  extern __shared__ u32x4 ldsw[];                // [buf 2][k-step 2][operand 2][tile][term 2][lane]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nob = c / OT;
  const int ob_co = (blockIdx.y / nob) * OT, ob_ci = (blockIdx.y % nob) * OT;
  const int wco = (wv >> 2) * (OT / 2), wci = (wv & 3) * (OT / 4);
  const int sps = (hw + 31) >> 5;                // steps per sample
  const long n_steps = (long)nb * sps;
  // worker w takes the contiguous range [n_steps w / n_workers, n_steps (w + 1) / n_workers).  (Tried: steps w, w + n_workers,
  // ... so that at any moment the workers together read one contiguous span of every channel row: 141 / 130 vs 133 / 128 us.)
  const int w_id = blockIdx.x;
  const int first = (int)(n_steps * w_id / n_workers);
  const int count = (int)(n_steps * (w_id + 1) / n_workers) - first;
  auto step_of = [&](int k) { return first + min(k, count - 1); };   // past the end: the last step again

  // loads: instruction j reads rows 64 j + 8 wv + (lane >> 3), four pixels 4 (lane & 7) ..
  const int ld_row = 8 * wv + (lane >> 3), ld_px = 4 * (lane & 7);
  // items after the pair exchange: row 64 (2a + odd) + ld_row, pixels 8 chunk .. 8 chunk + 7, chunk = (lane & 7) >> 1
  const int odd = lane & 1, chunk = (lane & 7) >> 1;
  const int it_ks = chunk >> 1, it_h = chunk & 1;   // MFMA k-step and lane half of the item

  f32x16 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x4 raw[2][2][NJ];                           // [operand][input][load instruction]
  float cfa[NA][3], cfb[NA][3];                  // prologue coefficients of this thread's item rows
  int cur_b = -1;
  auto load_coefs = [&](int b) {
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const int row = 64 * (2 * a + odd) + ld_row;
      const float* ca = acoef + (size_t)b * 3 * c + ob_co + row;
      const float* cb = bcoef + (size_t)b * 3 * c + ob_ci + row;
#pragma unroll
      for (int q = 0; q < 3; ++q) { cfa[a][q] = ca[q * c]; cfb[a][q] = cb[q * c]; }
    }
    cur_b = b;
  };
  auto fetch = [&](int s) {
    const int b = s / sps, p = (s % sps) * 32 + ld_px;
    const size_t off = p < hw ? p : 0;
#pragma unroll
    for (int op = 0; op < 2; ++op) {
      const float* src0 = op ? b0 : a0;
      const float* src1 = op ? b1 : a1;
      const bool two = op ? B_TWO : A_TWO;
      const size_t base = (size_t)b * (op ? b_bstride : a_bstride) + (size_t)((op ? ob_ci : ob_co) + ld_row) * hw + off;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        raw[op][0][j] = *reinterpret_cast<const f32x4*>(src0 + base + (size_t)(64 * j) * hw);
        if (two) raw[op][1][j] = *reinterpret_cast<const f32x4*>(src1 + base + (size_t)(64 * j) * hw);
      }
    }
  };
  // neighbour lane's value (lanes 2k <-> 2k+1): DPP quad_perm [1,0,3,2]
  auto swap1 = [](float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true)); };
  auto stage = [&](int s, int buf) {
    const int b = s / sps, p = (s % sps) * 32 + 8 * chunk;
    if (b != cur_b) load_coefs(b);  // block-uniform, a few times per worker
    const bool in_lo = p < hw, in_hi = p + 4 < hw;
#pragma unroll
    for (int op = 0; op < 2; ++op) {
      const bool two = op ? B_TWO : A_TWO;
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const float k0 = op ? cfb[a][0] : cfa[a][0], k1 = op ? cfb[a][1] : cfa[a][1], k2 = op ? cfb[a][2] : cfa[a][2];
        const f32x2 k0v = {k0, k0}, k1v = {k1, k1}, k2v = {k2, k2};
        // assemble the item: [lo 4 pixels | hi 4 pixels] of this thread's row, per input
        float x[2][8];
#pragma unroll
        for (int in = 0; in < 2; ++in) {
          if (in == 1 && !two) continue;
          const f32x4 even_row = raw[op][in][2 * a], odd_row = raw[op][in][2 * a + 1];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float keep = odd ? odd_row[e] : even_row[e];
            const float recv = swap1(odd ? even_row[e] : odd_row[e]);
            x[in][e] = odd ? recv : keep;       // even lane holds the lower piece, odd lane the higher one
            x[in][4 + e] = odd ? keep : recv;
          }
        }
        u32x4 th, tm;
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
          const f32x2 x0 = {x[0][2 * jp], x[0][2 * jp + 1]};
          f32x2 t = __builtin_elementwise_fma(k0v, x0, k2v);
          if (two) {
            const f32x2 x1 = {x[1][2 * jp], x[1][2 * jp + 1]};
            t = __builtin_elementwise_fma(k1v, x1, t);
          }
          if (op == 1 && B_RELU) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); }
          if (!(jp < 2 ? in_lo : in_hi)) { t.x = 0.f; t.y = 0.f; }
          unsigned hh, mm;
          split2_hm(t.x, t.y, hh, mm);
          th[jp] = hh; tm[jp] = mm;
        }
        const int row = 64 * (2 * a + odd) + ld_row;
        u32x4* dst = ldsw + buf * kBuf + (it_ks * 2 + op) * kOp + (row >> 5) * 128 + (row & 31) + 32 * it_h;
        dst[0] = th;
        dst[64] = tm;
      }
    }
  };

  if (count > 0) {
    fetch(step_of(0));
    stage(step_of(0), 0);
    fetch(step_of(1));
  }
  __syncthreads();
  for (int k = 0; k < count; ++k) {
    const int buf = k & 1;
    // unconditional (indices clamped to the last step, whose re-staged copy nobody reads), see pw_wgrad6_kernel.
    // (tried: operand by operand -- stage A(s+1), request A(s+2), stage B(s+1), request B(s+2): no gain)
    stage(step_of(k + 1), buf ^ 1);
    fetch(step_of(k + 2));
    // keep the loads of step s + 2 ahead of this step's MFMAs (the scheduler sinks them to the end)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const u32x4* ta = ldsw + buf * kBuf + (ks * 2) * kOp + lane;
      const u32x4* tb = ta + kOp;
      u32x4 fb[TB][2];
#pragma unroll
      for (int j = 0; j < TB; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) fb[j][t] = tb[(((wci >> 5) + j) * 2 + t) * 64];
#pragma unroll
      for (int i = 0; i < TA; ++i) {
        u32x4 fa[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) fa[t] = ta[(((wco >> 5) + i) * 2 + t) * 64];
        // terms: 0 = high, 1 = mid; smallest products first
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = mfma_bf16(fa[1], fb[j][0], acc[i][j]);
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = mfma_bf16(fa[0], fb[j][1], acc[i][j]);
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = mfma_bf16(fa[0], fb[j][0], acc[i][j]);
      }
    }
    __syncthreads();
  }

  float* po = partial + (size_t)blockIdx.x * c * c;
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = ob_co + wco + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int ci = ob_ci + wci + 32 * j + r;
        po[(size_t)co * c + ci] = acc[i][j][e];
      }
}

// gw[i] = sum over workers of partial[w][i].  A workgroup covers 64 consecutive elements x 4 worker phases: wave
// p sums workers p, p+4, ... with 16 loads in flight per lane, the four phase sums meet in LDS (fixed order:
// deterministic).  One thread per element with four loads in flight left the 64 MB of partials at 3 TB/s.
__global__ __launch_bounds__(kEwBlock) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gw, int n,
                                                                int n_workers) {
  __shared__ float sm[kEwBlock];
  constexpr int kPhases = kEwBlock / DHD_WAVE;
  const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int i = blockIdx.x * DHD_WAVE + lane;
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  if (i < n) {
    int w = ph;
    for (; w + 15 * kPhases < n_workers; w += 16 * kPhases) {
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] += partial[(size_t)(w + k * kPhases) * n + i];
    }
    for (; w < n_workers; w += kPhases) acc[0] += partial[(size_t)w * n + i];
  }
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < 16; k += 4) t += (acc[k] + acc[k + 1]) + (acc[k + 2] + acc[k + 3]);
  sm[threadIdx.x] = t;
  __syncthreads();
  if (ph == 0 && i < n) gw[i] = (sm[lane] + sm[DHD_WAVE + lane]) + (sm[2 * DHD_WAVE + lane] + sm[3 * DHD_WAVE + lane]);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------

inline size_t align_up(size_t v) { return (v + 63) & ~(size_t)63; }  // in floats: 256-byte sections

struct SavedLayout {
  size_t s, h, a1, tab_a, mean1, rstd1, scsh1, tab1, mean2, rstd2, scsh2, loc1, loc2, tick, wp1t, wp2t, mask, y1, y2, total;
};
constexpr int kTickWords = 64;   // arrival counters besides the per-channel ones (see saved_layout)
SavedLayout saved_layout(int b, int c, int hw, int r) {
  SavedLayout L;
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o += align_up(n); return at; };
  L.s = take((size_t)b * 2 * c);
  L.h = take((size_t)b * r);
  L.a1 = take((size_t)b * c);
  L.tab_a = take((size_t)b * 3 * c);
  L.mean1 = take(c); L.rstd1 = take(c); L.scsh1 = take(2 * c); L.tab1 = take((size_t)b * 3 * c);
  L.mean2 = take(c); L.rstd2 = take(c); L.scsh2 = take(2 * c);
  L.loc1 = take(2 * (2 * (size_t)c + 1)); L.loc2 = take(2 * (2 * (size_t)c + 1));   // (2C + 1) doubles each: this rank's shifted sums + count (phased calls)
  // arrival counters of the kernels that finish their own reductions (BnTail): [c] BatchNorm-2 backward | [c] BatchNorm-1
  // backward | kTickWords others; zeroed by the forward (fc_forward_kernel), left at zero by every kernel that uses them
  L.tick = take(2 * (size_t)c + kTickWords);
  L.wp1t = take(2 * (size_t)c * c); L.wp2t = take(2 * (size_t)c * c);   // transposed weight images, packed by the forward for the backward
  // ReLU pass bits, one per activation: a word per (32 pixels, channel), [sample][wave tile][channel] (resident / streamed
  // kernels), or 16-bit words in the staging lanes' order (cu kernels, sfa_gemm_cu.h: cu_mask_words)
  L.mask = take((size_t)b * c * ((hw + 31) / 32));
  L.y1 = take((size_t)b * c * hw);
  L.y2 = take((size_t)b * c * hw);
  L.total = o;
  return L;
}

struct ScratchLayout {
  size_t wp1, wp2, part, da1, da2, tab_g2, tab_g1, dpre2, dh, ds, mean_part, stat_part, g2, g1, du, wpart, total;
};
ScratchLayout scratch_layout(int b, int c, int hw, int r) {
  ScratchLayout L;
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o += align_up(n); return at; };
  const size_t cc = (size_t)c * c, plane = (size_t)b * c * hw;
  L.wp1 = take(2 * cc); L.wp2 = take(2 * cc);  // f32 images: cc, bf16x6 images: 1.5 cc (the transposed ones: SavedLayout)
  L.part = take((size_t)b * kPlaneChunks * 2 * c);
  L.stat_part = take((size_t)b * ((hw + 31) / 32 + 64) * 2 * c);   // a row per (sample, wave tile), or per wave of every launch (<= tiles + 63 each)
  L.da1 = take((size_t)b * kPlaneChunks * c);
  L.da2 = take((size_t)b * kPlaneChunks * c);
  L.tab_g2 = take((size_t)b * 3 * c);
  L.tab_g1 = take((size_t)b * 3 * c);
  L.dpre2 = take((size_t)b * c);
  L.dh = take((size_t)b * r);
  L.ds = take((size_t)b * 2 * c);
  L.mean_part = take((size_t)b * 2 * c * kPlaneChunks);
  L.g2 = take(plane); L.g1 = take(plane); L.du = take(plane);
  L.wpart = take((size_t)kWgWorkers * cc);
  L.total = o;
  return L;
}

// output channels per block = 32 * cot
inline int pw_cot(int c) { return (DHD_PW_COT_SEL == 8 && c % 256 == 0) ? 8 : 4; }

// (one sample of the input, 2 C hw floats, must stay below 4 GiB: the GEMMs address a sample through 32-bit buffer offsets)
inline bool stage_supported(int c, int hw) {
  return (c == 128 || (c > 0 && c % 256 == 0)) && hw > 0 && (hw & 3) == 0 && (size_t)2 * c * hw * sizeof(float) <= 0xFFFFFFFFull;
}

// hipFuncSetAttribute is not a stream operation: doing it on every launch breaks stream capture (HIP graphs),
// so each kernel instantiation raises its dynamic-LDS limit once per device, on first use.
inline int device_index() {
  int d = 0;
  (void)hipGetDevice(&d);
  return d < 0 || d >= 64 ? 0 : d;
}
#define DHD_LDS_ATTR_ONCE(kern, bytes)                                                                              \
  do {                                                                                                              \
    static bool done__[64] = {};                                                                                    \
    const int dev__ = device_index();                                                                               \
    if (!done__[dev__]) {                                                                                           \
      DHD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                  (int)(bytes)));                                                                   \
      done__[dev__] = true;                                                                                         \
    }                                                                                                               \
  } while (0)

// GEMM precision of the current call (dhd_sfa_weights.gemm), carried in a thread-local for the duration of the entry
// point -- per call, nothing process-wide.  Internal numbering:
//   0  f32 MFMA (pw_gemm / pw_wgrad)                                                    DHD_SFA_GEMM_F32
//   1  bf16x6: weights resident in LDS (pw_gemm_res<3>) where that form covers the channel count, else streamed per
//      pixel tile (pw_gemm6, with the 128-channel tail launch); bit-identical results      DHD_SFA_GEMM_BF16X6
//   3  bf16x3, weights resident in LDS (pw_gemm_res<2>): three products per a*b, relative error <= 3 * 2^-18 per
//      product.  DEFAULT: measured against float64 at (2,512,200,200) the stage output is off by 2.2e-5 (bf16x6:
//      1.2e-6, plain PyTorch fp32: 1.35e-6), well inside the 1e-3 bar of the path.         DHD_SFA_GEMM_BF16X3
thread_local int g_gemm_mode = 3;
inline int set_call_mode(int gemm) {
  switch (gemm) {
    case DHD_SFA_GEMM_DEFAULT: case DHD_SFA_GEMM_BF16X3: g_gemm_mode = 3; return DHD_OK;
    case DHD_SFA_GEMM_BF16X6: g_gemm_mode = 1; return DHD_OK;
    case DHD_SFA_GEMM_F32: g_gemm_mode = 0; return DHD_OK;
    default: return DHD_EINVAL;
  }
}
inline bool mode_resident() { return g_gemm_mode == 1 || g_gemm_mode == 3; }
inline int mode_terms() { return g_gemm_mode == 3 ? 2 : 3; }

constexpr int kResWaves = 8;                       // waves per resident workgroup
constexpr size_t kResTrBytes = (size_t)kResWaves * 16 * kResTrPitch * sizeof(float);  // store patches of the waves
constexpr size_t kLdsBytes = 160 * 1024;           // per-CU LDS of gfx950
constexpr size_t kResWeightMax = 128 * 1024;       // budget for the weight fragments
// 32-channel output tiles per resident workgroup: the largest of 4 / 2 / 1 whose fragments fit; 0 = does not fit
inline int res_cob(int c, int nt) {
  for (int cob = 4; cob >= 1; cob >>= 1)
    if (32 * cob <= c && (size_t)(c / 16) * cob * nt * 1024 <= kResWeightMax) return cob;
  return 0;
}
inline bool res_supported(int c) {
  if (!mode_resident() || (c != 128 && c != 256 && c != 512)) return false;
  const int nt = mode_terms(), cob = res_cob(c, nt);
  if (c == 128) return cob == 4;
  if (c == 256) return nt == 2 ? (cob == 4 || cob == 2) : cob == 2;
  // C = 512: bf16x3 with teams of 8 still beats the streamed form (2.85 vs 3.43 ms per stage at B = 3); bf16x6 would need
  // teams of 16 (4.23 ms) and stays on the streamed kernels (bit-identical results)
  return nt == 2 && cob == 2;
}

int cu_count() {
  static int n[64] = {};
  const int dev = device_index();
  if (n[dev] == 0 && hipDeviceGetAttribute(&n[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n[dev] = -1;
  return n[dev];
}

// One-CU-per-pixel-tile kernels (sfa_gemm_cu.h): the default precision (bf16x3) at the channel counts whose weight fragments
// fit the register file of a CU -- C = 256 (8 waves x 32 channels) and C = 128 (4 waves).  C = 512 (1 MB of fragments) and the
// bf16x6 precision (three parts) stay on the LDS-resident / streamed kernels.
inline bool cu_supported(int c) { return g_gemm_mode == 3 && (c == 128 || c == 256); }

int launch_pw_gemm_cu(const float* in0, const float* in1, size_t in_bstride, int in_channels, const float* coef, bool relu, const float* wp,
                      const float* bias, unsigned* relu_mask, float* stat_part, float* y, int epi, int b, int c, int hw, hipStream_t st,
                      int* stat_rows) {
  const int waves = c == 256 ? 8 : 4;
  const int max_b = cu_max_batch(c, waves);
  if (max_b < 1) return DHD_EUNSUPPORTED;
  const bool two = in1 != nullptr;
  const unsigned in_bytes = (unsigned)((size_t)in_channels * hw * sizeof(float));
  const int nwt = (hw + 31) / 32;
  int cus = cu_count();
  if (cus <= 0) cus = 256;
  int rows_done = 0;
  for (int b0 = 0; b0 < b; b0 += max_b) {
    const int nb = b - b0 < max_b ? b - b0 : max_b;
    const long total = (long)nb * nwt;
    const int grid = (int)(total < cus ? total : cus);
    const size_t shmem = cu_lds_bytes(c, waves, nb);
    const float* i0 = in0 + (size_t)b0 * in_bstride;
    const float* i1 = two ? in1 + (size_t)b0 * in_bstride : nullptr;
    const float* cf = coef + (size_t)b0 * 3 * c;
    unsigned* rm = relu_mask ? relu_mask + cu_mask_words(b0, c, hw) : nullptr;
    float* sp = stat_part ? stat_part + (size_t)rows_done * 2 * c : nullptr;
    rows_done += grid;                                   // one statistics row per workgroup
    float* yo = y + (size_t)b0 * c * hw;
    // <KCN, WAVES, TWO_IN, RELU, EPI, RECORD, AUX = nt loads, R = 1, NACC = 1, ABL = 0, SAUX = 0, PP = ping-pong, BPF = 0, EORD>,
    // contiguous tile ranges.  EORD: with two inputs the epilogue goes before the staging (which waits for twice the loads), with
    // one input after it -- measured either way (experiments/gemm_cu_bench.hip): 100 vs 107 us (conv1), 96 vs 97 (dgrad), 82 vs 76 (conv2)
#define DHD_CU(KCN, WAVES, TWO, RELU, EPI, REC)                                                                          \
  do {                                                                                                                \
    auto kern = pw_gemm_cu_kernel<KCN, WAVES, TWO, RELU, EPI, REC, 2, 1, 1, 0, 0, true, 0, (TWO) ? 1 : 0>;           \
    DHD_LDS_ATTR_ONCE(kern, kLdsBytes);                                                                               \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), shmem, st, i0, i1, in_bstride, in_bytes, cf,               \
                       reinterpret_cast<const u32x4*>(wp), bias, rm, sp, yo, hw, nb, 1);                              \
  } while (0)
#define DHD_CU_V(KCN, WAVES)                                                                       \
  do {                                                                                             \
    if (epi == 0 && two && !relu) DHD_CU(KCN, WAVES, true, false, 0, false);                       \
    else if (epi == 0 && !two && relu && rm) DHD_CU(KCN, WAVES, false, true, 0, true);             \
    else if (epi == 0 && !two && relu) DHD_CU(KCN, WAVES, false, true, 0, false);                  \
    else if (epi == 1 && two && !relu) DHD_CU(KCN, WAVES, true, false, 1, false);                  \
    else if (epi == 2 && two && !relu) DHD_CU(KCN, WAVES, true, false, 2, false);                  \
    else return DHD_EUNSUPPORTED;                                                                  \
  } while (0)
    if (c == 256) DHD_CU_V(16, 8);
    else DHD_CU_V(8, 4);
#undef DHD_CU_V
#undef DHD_CU
    DHD_LAUNCH_CHECK();
  }
  if (stat_rows) *stat_rows = rows_done;
  return DHD_OK;
}

int launch_pack(const float* w, int transpose, float* packed, int c, hipStream_t st, const float* w2 = nullptr,
                float* packed2 = nullptr) {
  const int cot = pw_cot(c);
  if (res_supported(c)) {
    const int nt = mode_terms();
    hipLaunchKernelGGL(pack_weight_res_kernel, dim3(dhd_cdiv((c / 32) * (c / 16) * 64, kEwBlock), w2 ? 2 : 1), dim3(kEwBlock), 0, st, w,
                       w2, transpose, reinterpret_cast<u32x4*>(packed), reinterpret_cast<u32x4*>(packed2), c, res_cob(c, nt), nt);
    DHD_LAUNCH_CHECK();
    return DHD_OK;
  }
  if (w2) {   // the streamed / f32 forms pack one weight per launch
    const int rc = launch_pack(w, transpose, packed, c, st);
    return rc != DHD_OK ? rc : launch_pack(w2, transpose, packed2, c, st);
  }
  if (g_gemm_mode >= 1)
    hipLaunchKernelGGL(pack_weight6_kernel, dim3(dhd_cdiv((c / 32) * (c / 16) * 64, kEwBlock)), dim3(kEwBlock), 0, st, w, transpose,
                       reinterpret_cast<u32x4*>(packed), c, cot);
  else
    hipLaunchKernelGGL(pack_weight_kernel, dim3(dhd_cdiv(c * c, kEwBlock)), dim3(kEwBlock), 0, st, w, transpose, packed, c, cot);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

// Resident-weights launcher: persistent workgroups, one per CU, teams of C / (32 COB) on one XCD.
int launch_pw_gemm_res(const float* in0, const float* in1, size_t in_bstride, int in_channels, const float* coef, bool relu, const float* wp,
                       const float* bias, unsigned* relu_mask, float* stat_part, float* y, int epi, int b, int c, int hw, hipStream_t st,
                       int* stat_rows) {
  int rows_done = 0;   // statistics rows written so far (bf16x3: one per wave of every launch; bf16x6: one per (sample, wave tile))
  const int nt = mode_terms(), cob = res_cob(c, nt);
  const int groups = c / (32 * cob);
  const int kcn = c / 16, nwt = (hw + 31) / 32;
  const size_t wbytes = (size_t)kcn * cob * nt * 1024;
  const int max_b = (int)((kLdsBytes - wbytes - kResTrBytes - 64) / ((size_t)3 * c * sizeof(float)));  // samples whose tables fit next to the weights
  if (max_b < 1) return DHD_EUNSUPPORTED;
  const bool two = in1 != nullptr;
  const unsigned in_bytes = (unsigned)((size_t)in_channels * hw * sizeof(float));
  int cus = cu_count();
  if (cus <= 0) cus = 256;
  for (int b0 = 0; b0 < b; b0 += max_b) {
    const int nb = b - b0 < max_b ? b - b0 : max_b;
    const long total = (long)nb * nwt;
    int nteams = (cus / (8 * groups)) * 8;       // whole teams per XCD
    if (nteams < 8) nteams = 8;
    const int need = dhd_cdiv(total, kResWaves);
    if (need < nteams) nteams = dhd_cdiv(need, 8) * 8;
    const dim3 grid(nteams * groups);
    const size_t shmem = wbytes + (((size_t)nb * 3 * c + 3) & ~(size_t)3) * sizeof(float) + kResTrBytes;
    const float* i0 = in0 + (size_t)b0 * in_bstride;
    const float* i1 = two ? in1 + (size_t)b0 * in_bstride : nullptr;
    const float* cf = coef + (size_t)b0 * 3 * c;
    unsigned* rm = relu_mask ? relu_mask + (size_t)b0 * nwt * c : nullptr;
    float* sp = stat_part ? stat_part + (size_t)rows_done * 2 * c : nullptr;
    rows_done += nt == 2 ? nteams : (int)total;
    float* yo = y + (size_t)b0 * c * hw;
#define DHD_RES(NT, COB, KCN, TWO, RELU, EPI)                                                                            \
  do {                                                                                                                \
    auto kern = pw_gemm_res_kernel<NT, COB, KCN, TWO, RELU, EPI, kResWaves, 0, 4>;                                    \
    DHD_LDS_ATTR_ONCE(kern, kLdsBytes);                                                                               \
    hipLaunchKernelGGL(kern, grid, dim3(kResWaves * 64), shmem, st, i0, i1, in_bstride, in_bytes, cf,                 \
                       reinterpret_cast<const u32x4*>(wp), bias, rm, sp, yo, c, hw, nb, groups, nteams);              \
  } while (0)
#define DHD_RES_V(NT, COB, KCN)                                                   \
  do {                                                                            \
    if (epi == 0 && two && !relu) DHD_RES(NT, COB, KCN, true, false, 0);          \
    else if (epi == 0 && !two && relu) DHD_RES(NT, COB, KCN, false, true, 0);     \
    else if (epi == 1 && two && !relu) DHD_RES(NT, COB, KCN, true, false, 1);     \
    else if (epi == 2 && two && !relu) DHD_RES(NT, COB, KCN, true, false, 2);     \
    else return DHD_EUNSUPPORTED;                                                 \
  } while (0)
    // (terms, tiles per workgroup) by channel count: C = 128: (.,4);  C = 256: x6 (3,2), x3 (2,4) or capped;  C = 512: x6 (3,1), x3 (2,2)
    if (kcn == 8 && cob == 4 && nt == 2) DHD_RES_V(2, 4, 8);
    else if (kcn == 8 && cob == 4 && nt == 3) DHD_RES_V(3, 4, 8);
    else if (kcn == 16 && cob == 4 && nt == 2) DHD_RES_V(2, 4, 16);
    else if (kcn == 16 && cob == 2 && nt == 2) DHD_RES_V(2, 2, 16);
    else if (kcn == 16 && cob == 2 && nt == 3) DHD_RES_V(3, 2, 16);
    else if (kcn == 32 && cob == 2 && nt == 2) DHD_RES_V(2, 2, 32);
    else if (kcn == 32 && cob == 1 && nt == 3) DHD_RES_V(3, 1, 32);
    else return DHD_EUNSUPPORTED;
#undef DHD_RES_V
#undef DHD_RES
    DHD_LAUNCH_CHECK();
  }
  if (stat_rows) *stat_rows = rows_done;
  return DHD_OK;
}

// in0/in1 prologue GEMM launcher.  epi: 0 forward (+bias), 1 dgrad with ReLU mask, 2 dgrad plain
int launch_pw_gemm(const float* in0, const float* in1, size_t in_bstride, int in_channels, const float* coef, bool relu, const float* wp,
                   const float* bias, const float* aux, const float* aux_scsh, unsigned* relu_mask, float* stat_part, float* y, int epi, int b,
                   int c, int hw, hipStream_t st, int* stat_rows = nullptr) {
  if (stat_rows) *stat_rows = b * ((hw + 31) / 32);   // the streamed kernels: a row per (sample, wave tile)
  if (cu_supported(c))
    return launch_pw_gemm_cu(in0, in1, in_bstride, in_channels, coef, relu, wp, bias, relu_mask, stat_part, y, epi, b, c, hw, st, stat_rows);
  if (res_supported(c))
    return launch_pw_gemm_res(in0, in1, in_bstride, in_channels, coef, relu, wp, bias, relu_mask, stat_part, y, epi, b, c, hw, st,
                              stat_rows);
  const int cot = pw_cot(c);
  const int tps = dhd_cdiv(hw, 32 * (kPwBlock / DHD_WAVE));  // 128-pixel tiles per sample
  const bool two = in1 != nullptr;
  const unsigned in_bytes = (unsigned)((size_t)in_channels * hw * sizeof(float));  // one sample of in0 (and of in1, which follows it for x)
  const size_t shmem = g_gemm_mode >= 1 ? (size_t)2 * cot * 3 * 64 * 16 + (size_t)3 * c * sizeof(float)
                                        : (size_t)(2 * cot * 512 + 3 * c) * sizeof(float);
  // Two 256-channel workgroups fit a CU (accumulators), so the tiles run in rounds of 2 x CUs.  With less
  // than two rounds of work (small batches) a partly filled round is a large share of the time: the tiles
  // beyond the full round go to a second launch of the 128-channel kernel instead (twice the workgroups,
  // three per CU, each about half as long), reading the same packed weights.  Measured -7 % (B = 1) and
  // -3 % (B = 2) on the stage; with more rounds the split gains nothing (B = 4: 1.715 vs 1.714 ms).
  int t_main = tps;
  if (g_gemm_mode != 0 && cot == 8) {
    const int cus = cu_count();
    const long nrb = c / 256, n = (long)b * tps * nrb, slots = 2L * cus;
    if (cus > 0 && n < 2 * slots && n % slots != 0) {
      const int cand = (int)((n / slots) * slots / (b * nrb)) & ~7;
      if ((long)b * (tps - cand) * nrb * 2 <= 3L * cus) t_main = cand;
    }
  }
  const int t_tail8 = dhd_cdiv(tps - t_main, 8) * 8;  // whole groups of 8; surplus tiles exit at once
  const dim3 grid((t_main == tps ? dhd_cdiv(tps, 8) * 8 : t_main) * (c / (32 * cot)), b);
  const dim3 grid_tail(t_tail8 * (c / 128), b);
  const size_t shmem_tail = (size_t)2 * 4 * 3 * 64 * 16 + (size_t)3 * c * sizeof(float);
#define DHD_PW(COT, TWO, RELU, EPI)                                                                                    \
  do {                                                                                                                 \
    if (g_gemm_mode >= 1) {                                                                                            \
      if (grid.x > 0) {                                                                                                \
        auto kern = pw_gemm6_kernel<COT, TWO, RELU, EPI>;                                                              \
        DHD_LDS_ATTR_ONCE(kern, shmem);                                                                                \
        hipLaunchKernelGGL(kern, grid, dim3(kPwBlock), shmem, st, in0, in1, in_bstride, in_bytes, coef,                \
                           reinterpret_cast<const u32x4*>(wp), bias, relu_mask, stat_part, y, c, hw, 0, t_main, 0);    \
      }                                                                                                                \
      if (t_main < tps) {                                                                                              \
        auto tail = pw_gemm6_kernel<4, TWO, RELU, EPI>;                                                                \
        DHD_LDS_ATTR_ONCE(tail, shmem_tail);                                                                           \
        hipLaunchKernelGGL(tail, grid_tail, dim3(kPwBlock), shmem_tail, st, in0, in1, in_bstride, in_bytes, coef,      \
                           reinterpret_cast<const u32x4*>(wp), bias, relu_mask, stat_part, y, c, hw, t_main, tps, 1);  \
      }                                                                                                                \
    } else {                                                                                                           \
      auto kern = pw_gemm_kernel<COT, TWO, RELU, EPI>;                                                                 \
      DHD_LDS_ATTR_ONCE(kern, shmem);                    \
      hipLaunchKernelGGL(kern, grid, dim3(kPwBlock), shmem, st, in0, in1, in_bstride, coef, wp, bias, aux, aux_scsh, y, \
                         c, hw);                                                                                       \
    }                                                                                                                  \
  } while (0)
#define DHD_PW_COT(TWO, RELU, EPI)           \
  do {                                       \
    if (cot == 4) DHD_PW(4, TWO, RELU, EPI); \
    else DHD_PW(8, TWO, RELU, EPI);          \
  } while (0)
  if (epi == 0 && two && !relu) DHD_PW_COT(true, false, 0);
  else if (epi == 0 && !two && relu) DHD_PW_COT(false, true, 0);
  else if (epi == 1 && two && !relu) DHD_PW_COT(true, false, 1);
  else if (epi == 2 && two && !relu) DHD_PW_COT(true, false, 2);
  else return DHD_EUNSUPPORTED;
#undef DHD_PW_COT
#undef DHD_PW
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

int launch_pw_wgrad(const float* a0, const float* a1, const float* acoef, size_t a_bs, const float* b0, const float* b1,
                    const float* bcoef, size_t b_bs, bool b_relu, float* partial, float* gw, int b, int c, int hw,
                    hipStream_t st) {
  // (tried at C = 256 in the bf16x3 mode: 128 x 128 tiles, two workgroups per CU, operands read twice through L2: 233 / 189 us
  // against 166 / 138 us for one 256 x 256 tile per CU)
  const int ot = c == 128 ? 128 : 256;
  const int nob = (c / ot) * (c / ot);
  const int workers = kWgWorkers / nob > 0 ? kWgWorkers / nob : 1;
  const dim3 grid(workers, nob);
  const size_t shmem = (size_t)4 * ot * kWgStride * sizeof(float);
#define DHD_WG(OT, ATWO, BTWO, BRELU)                                                                              \
  do {                                                                                                             \
    auto kern = pw_wgrad_kernel<OT, ATWO, BTWO, BRELU>;                                                            \
    DHD_LDS_ATTR_ONCE(kern, shmem);                    \
    hipLaunchKernelGGL(kern, grid, dim3(kWgBlock), shmem, st, a0, a1, acoef, a_bs, b0, b1, bcoef, b_bs, partial, c, \
                       hw, b, workers);                                                                            \
  } while (0)
  const bool btwo = b1 != nullptr;
  if (a1 == nullptr) return DHD_EUNSUPPORTED;
  if (g_gemm_mode == 3) {
    const size_t shmem3 = (size_t)2 * 2 * 2 * (ot / 32) * 2 * 64 * 16;
#define DHD_WG3(OT, ATWO, BTWO, BRELU)                                                                             \
  do {                                                                                                             \
    auto kern = pw_wgrad3_kernel<OT, ATWO, BTWO, BRELU>;                                                           \
    DHD_LDS_ATTR_ONCE(kern, shmem3);                                                                               \
    hipLaunchKernelGGL(kern, grid, dim3(kWgBlock), shmem3, st, a0, a1, acoef, a_bs, b0, b1, bcoef, b_bs, partial,  \
                       c, hw, b, workers);                                                                         \
  } while (0)
    if (ot == 128) {
      if (btwo) DHD_WG3(128, true, true, false);
      else if (b_relu) DHD_WG3(128, true, false, true);
      else return DHD_EUNSUPPORTED;
    } else {
      if (btwo) DHD_WG3(256, true, true, false);
      else if (b_relu) DHD_WG3(256, true, false, true);
      else return DHD_EUNSUPPORTED;
    }
#undef DHD_WG3
  } else if (g_gemm_mode >= 1) {
    const size_t shmem6 = (size_t)2 * 2 * (ot / 32) * 3 * 64 * 16;
#define DHD_WG6(OT, ATWO, BTWO, BRELU)                                                                             \
  do {                                                                                                             \
    auto kern = pw_wgrad6_kernel<OT, ATWO, BTWO, BRELU>;                                                           \
    DHD_LDS_ATTR_ONCE(kern, shmem6);                    \
    hipLaunchKernelGGL(kern, grid, dim3(kWgBlock), shmem6, st, a0, a1, acoef, a_bs, b0, b1, bcoef, b_bs, partial,  \
                       c, hw, b, workers);                                                                         \
  } while (0)
    if (ot == 128) {
      if (btwo) DHD_WG6(128, true, true, false);
      else if (b_relu) DHD_WG6(128, true, false, true);
      else return DHD_EUNSUPPORTED;
    } else {
      if (btwo) DHD_WG6(256, true, true, false);
      else if (b_relu) DHD_WG6(256, true, false, true);
      else return DHD_EUNSUPPORTED;
    }
#undef DHD_WG6
  } else if (ot == 128) {
    if (btwo) DHD_WG(128, true, true, false);
    else if (b_relu) DHD_WG(128, true, false, true);
    else return DHD_EUNSUPPORTED;
  } else {
    if (btwo) DHD_WG(256, true, true, false);
    else if (b_relu) DHD_WG(256, true, false, true);
    else return DHD_EUNSUPPORTED;
  }
#undef DHD_WG
  DHD_LAUNCH_CHECK();
  const int n = c * c;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(dhd_cdiv(n, DHD_WAVE)), dim3(kEwBlock), 0, st, partial, gw, n, workers);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

#include "sfa_stage_half.h"

// dhd_sfa_weights.storage_dtype of a call: DHD_F32, or the half type every tensor of the stage is stored in (== io_dtype)
inline int storage_of(const dhd_sfa_weights* w, int c, int hw, int* storage) {
  *storage = w->storage_dtype;
  if (*storage == DHD_F32) return DHD_OK;
  if (*storage != DHD_F16 && *storage != DHD_BF16) return DHD_EINVAL;
  if (w->io_dtype != *storage) return DHD_EINVAL;
  return half_storage_supported(c, hw) ? DHD_OK : DHD_EUNSUPPORTED;
}

}  // namespace

extern "C" {


int dhd_sfa_stage_supported(int c, int hw) { return stage_supported(c, hw) ? 1 : 0; }

size_t dhd_sfa_stage_saved_bytes(int b, int c, int hw, int hidden) {
  if (b <= 0 || hidden <= 0 || !stage_supported(c, hw)) return 0;
  return saved_layout(b, c, hw, hidden).total * sizeof(float);
}

size_t dhd_sfa_stage_scratch_bytes(int b, int c, int hw, int hidden) {
  if (b <= 0 || hidden <= 0 || !stage_supported(c, hw)) return 0;
  return scratch_layout(b, c, hw, hidden).total * sizeof(float);
}

int dhd_sfa_stage_half_storage_supported(int c, int hw) { return half_storage_supported(c, hw) ? 1 : 0; }

int dhd_sfa_stage_workspace_bytes(int b, int c, int hw, int hidden, int storage_dtype, size_t* saved_bytes, size_t* scratch_bytes) {
  if (b <= 0 || hidden <= 0 || !saved_bytes || !scratch_bytes) return DHD_EINVAL;
  if (storage_dtype == DHD_F32) {
    if (!stage_supported(c, hw)) return DHD_EUNSUPPORTED;
    *saved_bytes = saved_layout(b, c, hw, hidden).total * sizeof(float);
    *scratch_bytes = scratch_layout(b, c, hw, hidden).total * sizeof(float);
    return DHD_OK;
  }
  if (storage_dtype != DHD_F16 && storage_dtype != DHD_BF16) return DHD_EINVAL;
  if (!half_storage_supported(c, hw)) return DHD_EUNSUPPORTED;
  *saved_bytes = saved_layout_h(b, c, hw, hidden).total;
  *scratch_bytes = scratch_layout_h(b, c, hw, hidden).total;
  return DHD_OK;
}

// Forward in up to three phases, cut at the two BatchNorm statistics points.  sync == nullptr: all phases in one call with
// this call's own statistics.  sync != nullptr (nn.SyncBatchNorm): phases [lo, hi]; a phase that ends at a statistics point
// leaves this rank's sums in `sync` ((2C + 1) doubles: [sum (y - bias)][C] | [sum (y - bias)^2][C] | count), the next phase
// starts from the caller's all-reduced vector in the same place.
static int stage_forward_impl(const void* xv, const dhd_sfa_weights* w, void* out, void* saved, void* scratch, int b, int c, int hw,
                              int lo, int hi, double* sync, void* stream) {
  if (!xv || !w || !saved || !scratch || b <= 0 || (hi == 2 && !out)) return DHD_EINVAL;
  if (!stage_supported(c, hw) || w->hidden <= 0) return DHD_EUNSUPPORTED;
  if (!w->fc1_w || !w->fc1_b || !w->fc2_w || !w->fc2_b || !w->conv1_w || !w->conv1_b || !w->bn1_w || !w->bn1_b || !w->conv2_w ||
      !w->conv2_b || !w->bn2_w || !w->bn2_b)
    return DHD_EINVAL;
  if (!w->training && (!w->bn1_mean || !w->bn1_var || !w->bn2_mean || !w->bn2_var)) return DHD_EINVAL;
  if (set_call_mode(w->gemm) != DHD_OK) return DHD_EINVAL;
  if (w->io_dtype != DHD_F32 && w->io_dtype != DHD_F16 && w->io_dtype != DHD_BF16) return DHD_EINVAL;
  hipStream_t st = dhd_stream(stream);
  int storage;
  if (int rcs = storage_of(w, c, hw, &storage); rcs != DHD_OK) return rcs;
  if (storage == DHD_F16) return stage_forward_half<_Float16>(xv, w, out, saved, scratch, b, c, hw, lo, hi, sync, st);
  if (storage == DHD_BF16) return stage_forward_half<__bf16>(xv, w, out, saved, scratch, b, c, hw, lo, hi, sync, st);
  const float* x = static_cast<const float*>(xv);
  const int r = w->hidden;
  const SavedLayout S = saved_layout(b, c, hw, r);
  const ScratchLayout T = scratch_layout(b, c, hw, r);
  float* sv = static_cast<float*>(saved);
  float* sc = static_cast<float*>(scratch);
  const dim3 planes2(kPlaneChunks, b * 2 * c), planes(kPlaneChunks, b * c);
  const dim3 per_ch(dhd_cdiv(c, kEwBlock));
  const bool fused_stats = w->training && g_gemm_mode >= 1;  // BatchNorm sums come out of the GEMM epilogue
  if (sync && !fused_stats) return DHD_EUNSUPPORTED;         // cross-rank statistics: training mode, bf16 GEMM precisions
  int stat_rows = 0;
  int rc;

  if (lo <= 0) {
    if (res_supported(c)) {
      PackJob job;
      job.w[0] = w->conv1_w; job.w[1] = w->conv2_w;
      job.dst[0] = reinterpret_cast<u32x4*>(sc + T.wp1); job.dst[1] = reinterpret_cast<u32x4*>(sc + T.wp2);
      job.dst[2] = reinterpret_cast<u32x4*>(sv + S.wp1t); job.dst[3] = reinterpret_cast<u32x4*>(sv + S.wp2t);
      job.c = c; job.nt = mode_terms(); job.cob = res_cob(c, job.nt); job.cu = cu_supported(c) ? 1 : 0;
      job.blocks_each = dhd_cdiv((c / 32) * (c / 16) * 64, kEwBlock);
      const dim3 grid(kPlaneChunks, b * 2 * c + dhd_cdiv(4 * job.blocks_each, kPlaneChunks));
      hipLaunchKernelGGL(plane_mean_pack_kernel, grid, dim3(kEwBlock), 0, st, x, sc + T.mean_part, hw, b * 2 * c, job);
    } else {
      hipLaunchKernelGGL(plane_mean_kernel, planes2, dim3(kEwBlock), 0, st, x, sc + T.mean_part, hw);
      rc = launch_pack(w->conv1_w, 0, sc + T.wp1, c, st, w->conv2_w, sc + T.wp2);
      if (rc != DHD_OK) return rc;
      rc = launch_pack(w->conv1_w, 1, sv + S.wp1t, c, st, w->conv2_w, sv + S.wp2t);
      if (rc != DHD_OK) return rc;
    }
    hipLaunchKernelGGL(fc_forward_kernel, dim3(b), dim3(kFcBlock), (size_t)(2 * c + r) * sizeof(float), st, sc + T.mean_part,
                       w->fc1_w, w->fc1_b, w->fc2_w, w->fc2_b, sv + S.s, sv + S.h, sv + S.a1, sv + S.tab_a, c, r, hw,
                       reinterpret_cast<int*>(sv + S.tick), 2 * c + kTickWords);
    DHD_LAUNCH_CHECK();
    // y1 = conv1(blend1(x))
    rc = launch_pw_gemm(x, x + (size_t)c * hw, (size_t)2 * c * hw, c, sv + S.tab_a, false, sc + T.wp1, w->conv1_b, nullptr, nullptr,
                        nullptr, fused_stats ? sc + T.stat_part : nullptr, sv + S.y1, 0, b, c, hw, st, &stat_rows);
    if (rc != DHD_OK) return rc;
    if (sync) {   // this rank's sums only (also kept in `saved` for the backward's convolution-bias gradient)
      hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(c / 4), dim3(kEwBlock), 0, st, sc + T.stat_part, stat_rows, w->conv1_b, w->bn1_w,
                         w->bn1_b, w->bn1_mean, w->bn1_var, w->momentum1, w->eps1, sv + S.mean1, sv + S.rstd1, sv + S.scsh1,
                         sv + S.tab1, b, c, hw, sync, reinterpret_cast<double*>(sv + S.loc1), nullptr, nullptr);
      DHD_LAUNCH_CHECK();
    }
  }
  if (hi <= 0) return DHD_OK;
  if (lo <= 1) {
    if (w->training) {
      if (fused_stats) {  // the GEMM epilogue left per-(sample, wave tile) sums shifted by the bias
        hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(c / 4), dim3(kEwBlock), 0, st, sc + T.stat_part, stat_rows, w->conv1_b, w->bn1_w,
                           w->bn1_b, w->bn1_mean, w->bn1_var, w->momentum1, w->eps1, sv + S.mean1, sv + S.rstd1, sv + S.scsh1,
                           sv + S.tab1, b, c, hw, nullptr, nullptr, sync, reinterpret_cast<long long*>(w->bn1_batches));
      } else {
        hipLaunchKernelGGL(moments_kernel, planes, dim3(kEwBlock), 0, st, sv + S.y1, sc + T.part, c, hw);
        hipLaunchKernelGGL(bn_train_finalize_kernel, per_ch, dim3(kEwBlock), 0, st, sc + T.part, b * kPlaneChunks, sv + S.y1, hw, w->bn1_w,
                           w->bn1_b, w->bn1_mean, w->bn1_var, w->momentum1, w->eps1, sv + S.mean1, sv + S.rstd1, sv + S.scsh1,
                           sv + S.tab1, b, c, hw, reinterpret_cast<long long*>(w->bn1_batches));
      }
    } else {
      hipLaunchKernelGGL(bn_eval_coef_kernel, per_ch, dim3(kEwBlock), 0, st, w->bn1_w, w->bn1_b, w->bn1_mean, w->bn1_var, w->eps1,
                         sv + S.mean1, sv + S.rstd1, sv + S.scsh1, sv + S.tab1, b, c);
    }
    DHD_LAUNCH_CHECK();
    // y2 = conv2(relu(bn1(y1)))
    rc = launch_pw_gemm(sv + S.y1, nullptr, (size_t)c * hw, c, sv + S.tab1, true, sc + T.wp2, w->conv2_b, nullptr, nullptr,
                        reinterpret_cast<unsigned*>(sv + S.mask), fused_stats ? sc + T.stat_part : nullptr, sv + S.y2, 0,
                        b, c, hw, st, &stat_rows);
    if (rc != DHD_OK) return rc;
    if (sync) {
      hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(c / 4), dim3(kEwBlock), 0, st, sc + T.stat_part, stat_rows, w->conv2_b, w->bn2_w,
                         w->bn2_b, w->bn2_mean, w->bn2_var, w->momentum2, w->eps2, sv + S.mean2, sv + S.rstd2, sv + S.scsh2,
                         sc + T.tab_g2, b, c, hw, sync, reinterpret_cast<double*>(sv + S.loc2), nullptr, nullptr);
      DHD_LAUNCH_CHECK();
    }
  }
  if (hi <= 1) return DHD_OK;
  float* tab_unused = sc + T.tab_g2;  // bn2 has no consumer GEMM in forward; table slot reused as a sink
  if (w->training) {
    if (fused_stats) {  // the GEMM epilogue left per-(sample, wave tile) sums shifted by the bias
      hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(c / 4), dim3(kEwBlock), 0, st, sc + T.stat_part, stat_rows, w->conv2_b, w->bn2_w,
                         w->bn2_b, w->bn2_mean, w->bn2_var, w->momentum2, w->eps2, sv + S.mean2, sv + S.rstd2, sv + S.scsh2,
                         tab_unused, b, c, hw, nullptr, nullptr, sync, reinterpret_cast<long long*>(w->bn2_batches));
    } else {
      hipLaunchKernelGGL(moments_kernel, planes, dim3(kEwBlock), 0, st, sv + S.y2, sc + T.part, c, hw);
      hipLaunchKernelGGL(bn_train_finalize_kernel, per_ch, dim3(kEwBlock), 0, st, sc + T.part, b * kPlaneChunks, sv + S.y2, hw, w->bn2_w,
                         w->bn2_b, w->bn2_mean, w->bn2_var, w->momentum2, w->eps2, sv + S.mean2, sv + S.rstd2, sv + S.scsh2,
                         tab_unused, b, c, hw, reinterpret_cast<long long*>(w->bn2_batches));
    }
  } else {
    hipLaunchKernelGGL(bn_eval_coef_kernel, per_ch, dim3(kEwBlock), 0, st, w->bn2_w, w->bn2_b, w->bn2_mean, w->bn2_var, w->eps2,
                       sv + S.mean2, sv + S.rstd2, sv + S.scsh2, tab_unused, b, c);
  }
  if (w->io_dtype == DHD_F16)
    hipLaunchKernelGGL(blend2_bn_kernel<_Float16>, planes, dim3(kEwBlock), 0, st, x, sv + S.a1, sv + S.y2, sv + S.scsh2, static_cast<_Float16*>(out), c, hw);
  else if (w->io_dtype == DHD_BF16)
    hipLaunchKernelGGL(blend2_bn_kernel<__bf16>, planes, dim3(kEwBlock), 0, st, x, sv + S.a1, sv + S.y2, sv + S.scsh2, static_cast<__bf16*>(out), c, hw);
  else
    hipLaunchKernelGGL(blend2_bn_kernel<float>, planes, dim3(kEwBlock), 0, st, x, sv + S.a1, sv + S.y2, sv + S.scsh2, static_cast<float*>(out), c, hw);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

// Backward in up to three phases, cut where the two BatchNorm backward passes need their sums (sum g, sum g (y - mu)); `sync`
// as in stage_forward_impl ((2C + 1) doubles: [sum g][C] | [sum g (y - mu)][C] | count).
static int stage_backward_impl(const void* xv, const dhd_sfa_weights* w, const void* saved, const void* gout, void* gx,
                               const dhd_sfa_grads* grads, void* scratch, int b, int c, int hw, int lo, int hi, double* sync,
                               void* stream) {
  if (!xv || !w || !saved || !gout || !grads || !scratch || b <= 0 || (hi == 2 && !gx)) return DHD_EINVAL;
  if (!stage_supported(c, hw) || w->hidden <= 0) return DHD_EUNSUPPORTED;
  if (!grads->fc1_w || !grads->fc1_b || !grads->fc2_w || !grads->fc2_b || !grads->conv1_w || !grads->conv1_b || !grads->bn1_w ||
      !grads->bn1_b || !grads->conv2_w || !grads->conv2_b || !grads->bn2_w || !grads->bn2_b)
    return DHD_EINVAL;
  if (set_call_mode(w->gemm) != DHD_OK) return DHD_EINVAL;
  if (w->io_dtype != DHD_F32 && w->io_dtype != DHD_F16 && w->io_dtype != DHD_BF16) return DHD_EINVAL;
  hipStream_t st = dhd_stream(stream);
  int storage;
  if (int rcs = storage_of(w, c, hw, &storage); rcs != DHD_OK) return rcs;
  if (storage == DHD_F16) return stage_backward_half<_Float16>(xv, w, saved, gout, gx, grads, scratch, b, c, hw, lo, hi, sync, st);
  if (storage == DHD_BF16) return stage_backward_half<__bf16>(xv, w, saved, gout, gx, grads, scratch, b, c, hw, lo, hi, sync, st);
  if (sync && !(w->training && g_gemm_mode >= 1)) return DHD_EUNSUPPORTED;
  const float* x = static_cast<const float*>(xv);
  const int r = w->hidden;
  const SavedLayout S = saved_layout(b, c, hw, r);
  const ScratchLayout T = scratch_layout(b, c, hw, r);
  const float* sv = static_cast<const float*>(saved);
  float* sc = static_cast<float*>(scratch);
  const dim3 planes(kPlaneChunks, b * c);
  const dim3 per_ch(dhd_cdiv(c, kEwBlock));
  const size_t cs = (size_t)c * hw;
  int rc;

  if (lo <= 0) {
    // g2 = dL/ds2, BatchNorm-2 sums, go-part of dL/da
    // without cross-rank statistics the pass finishes its own reduction (BnTail): no bn_backward_coef launch
    int* tick = reinterpret_cast<int*>(const_cast<float*>(sv + S.tick));
    const BnTail tail2 = {sync ? nullptr : tick, w->bn2_w, sv + S.mean2, sv + S.rstd2, sc + T.tab_g2, grads->bn2_w, grads->bn2_b,
                          grads->conv2_b, w->training, b, hw};
#define DHD_B2BWD(TO)                                                                                                        \
  hipLaunchKernelGGL(blend2_bn_bwd_kernel<TO>, planes, dim3(kEwBlock), 0, st, x, sv + S.a1, sv + S.y2, sv + S.scsh2, sv + S.mean2, \
                     static_cast<const TO*>(gout), sc + T.g2, sc + T.part, sc + T.da1, c, hw, tail2)
    if (w->io_dtype == DHD_F16) DHD_B2BWD(_Float16);
    else if (w->io_dtype == DHD_BF16) DHD_B2BWD(__bf16);
    else DHD_B2BWD(float);
#undef DHD_B2BWD
    if (sync)
      hipLaunchKernelGGL(bn_backward_coef_kernel, per_ch, dim3(kEwBlock), 0, st, sc + T.part, w->bn2_w, sv + S.mean2, sv + S.rstd2,
                         w->training, sc + T.tab_g2, grads->bn2_w, grads->bn2_b, grads->conv2_b, b, c, hw, sync, nullptr, nullptr, nullptr);
    DHD_LAUNCH_CHECK();
  }
  if (hi <= 0) return DHD_OK;
  if (lo <= 1) {
    if (sync)
      hipLaunchKernelGGL(bn_backward_coef_kernel, per_ch, dim3(kEwBlock), 0, st, sc + T.part, w->bn2_w, sv + S.mean2, sv + S.rstd2,
                         w->training, sc + T.tab_g2, grads->bn2_w, grads->bn2_b, grads->conv2_b, b, c, hw, nullptr, sync,
                         reinterpret_cast<const double*>(sv + S.loc2), w->conv2_b);
    DHD_LAUNCH_CHECK();
    // dW2 = dy2 . z1^T
    rc = launch_pw_wgrad(sc + T.g2, sv + S.y2, sc + T.tab_g2, cs, sv + S.y1, nullptr, sv + S.tab1, cs, true, sc + T.wpart,
                         grads->conv2_w, b, c, hw, st);
    if (rc != DHD_OK) return rc;
    // g1 = (W2^T dy2) * [z1 > 0]
    rc = launch_pw_gemm(sc + T.g2, sv + S.y2, cs, c, sc + T.tab_g2, false, sv + S.wp2t, nullptr, sv + S.y1, sv + S.scsh1,
                        reinterpret_cast<unsigned*>(const_cast<float*>(sv + S.mask)), nullptr, sc + T.g1, 1, b,
                        c, hw, st);
    if (rc != DHD_OK) return rc;
    int* tick = reinterpret_cast<int*>(const_cast<float*>(sv + S.tick)) + c;
    const BnTail tail1 = {sync ? nullptr : tick, w->bn1_w, sv + S.mean1, sv + S.rstd1, sc + T.tab_g1, grads->bn1_w, grads->bn1_b,
                          grads->conv1_b, w->training, b, hw};
    hipLaunchKernelGGL(pair_sums_kernel, planes, dim3(kEwBlock), 0, st, sc + T.g1, sv + S.y1, sv + S.mean1, sc + T.part, c, hw, tail1);
    if (sync)
      hipLaunchKernelGGL(bn_backward_coef_kernel, per_ch, dim3(kEwBlock), 0, st, sc + T.part, w->bn1_w, sv + S.mean1, sv + S.rstd1,
                         w->training, sc + T.tab_g1, grads->bn1_w, grads->bn1_b, grads->conv1_b, b, c, hw, sync, nullptr, nullptr, nullptr);
    DHD_LAUNCH_CHECK();
  }
  if (hi <= 1) return DHD_OK;
  if (sync)
    hipLaunchKernelGGL(bn_backward_coef_kernel, per_ch, dim3(kEwBlock), 0, st, sc + T.part, w->bn1_w, sv + S.mean1, sv + S.rstd1,
                       w->training, sc + T.tab_g1, grads->bn1_w, grads->bn1_b, grads->conv1_b, b, c, hw, nullptr, sync,
                       reinterpret_cast<const double*>(sv + S.loc1), w->conv1_b);
  DHD_LAUNCH_CHECK();
  // dW1 = dy1 . u^T
  rc = launch_pw_wgrad(sc + T.g1, sv + S.y1, sc + T.tab_g1, cs, x, x + cs, sv + S.tab_a, 2 * cs, false, sc + T.wpart, grads->conv1_w,
                       b, c, hw, st);
  if (rc != DHD_OK) return rc;
  // du = W1^T dy1
  rc = launch_pw_gemm(sc + T.g1, sv + S.y1, cs, c, sc + T.tab_g1, false, sv + S.wp1t, nullptr, nullptr, nullptr, nullptr, nullptr, sc + T.du, 2, b, c,
                      hw, st);
  if (rc != DHD_OK) return rc;
  hipLaunchKernelGGL(blend1_da_kernel, planes, dim3(kEwBlock), 0, st, x, sc + T.du, sc + T.da2, c, hw);
  hipLaunchKernelGGL(fc_backward_kernel, dim3(b), dim3(kEwBlock), (size_t)(c + r + kEwBlock) * sizeof(float), st, sc + T.da1, sc + T.da2,
                     sv + S.a1, sv + S.h, w->fc1_w, w->fc2_w, sc + T.dpre2, sc + T.dh, sc + T.ds, c, r);
  const int n_fc = r * 2 * c + c * r + r + c;
  const FcGradJob fcj = {sc + T.dpre2, sc + T.dh, sv + S.h, sv + S.s, grads->fc1_w, grads->fc1_b, grads->fc2_w, grads->fc2_b, b, r};
  const int fc_rows = dhd_cdiv(dhd_cdiv(n_fc, kEwBlock), kPlaneChunks);
  const dim3 planes_fc(kPlaneChunks, b * c + fc_rows);
#define DHD_GX(TO)                                                                                                           \
  hipLaunchKernelGGL(stage_gx_kernel<TO>, planes_fc, dim3(kEwBlock), 0, st, sv + S.a1, sv + S.y2, sv + S.scsh2,                 \
                     static_cast<const TO*>(gout), sc + T.du, sc + T.ds, static_cast<TO*>(gx), c, hw, fc_rows, fcj)
  if (w->io_dtype == DHD_F16) DHD_GX(_Float16);
  else if (w->io_dtype == DHD_BF16) DHD_GX(__bf16);
  else DHD_GX(float);
#undef DHD_GX
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

int dhd_sfa_stage_forward(const void* x, const dhd_sfa_weights* w, void* out, void* saved, void* scratch, int b, int c, int hw,
                          void* stream) {
  return stage_forward_impl(x, w, out, saved, scratch, b, c, hw, 0, 2, nullptr, stream);
}

int dhd_sfa_stage_backward(const void* x, const dhd_sfa_weights* w, const void* saved, const void* gout, void* gx,
                           const dhd_sfa_grads* grads, void* scratch, int b, int c, int hw, void* stream) {
  return stage_backward_impl(x, w, saved, gout, gx, grads, scratch, b, c, hw, 0, 2, nullptr, stream);
}

int dhd_sfa_stage_forward_phase(const void* x, const dhd_sfa_weights* w, void* out, void* saved, void* scratch, int b, int c, int hw,
                                int phase, double* sync_sums, void* stream) {
  if (phase < 0 || phase > 2 || !sync_sums || (reinterpret_cast<uintptr_t>(sync_sums) & 7)) return DHD_EINVAL;
  return stage_forward_impl(x, w, out, saved, scratch, b, c, hw, phase, phase, sync_sums, stream);
}

int dhd_sfa_stage_backward_phase(const void* x, const dhd_sfa_weights* w, const void* saved, const void* gout, void* gx,
                                 const dhd_sfa_grads* grads, void* scratch, int b, int c, int hw, int phase, double* sync_sums,
                                 void* stream) {
  if (phase < 0 || phase > 2 || !sync_sums || (reinterpret_cast<uintptr_t>(sync_sums) & 7)) return DHD_EINVAL;
  return stage_backward_impl(x, w, saved, gout, gx, grads, scratch, b, c, hw, phase, phase, sync_sums, stream);
}

}  // extern "C"
