// Training-mode BatchNorm2d (NCHW; float32, float16 or bfloat16 activations, float32 parameters and statistics)
// for the dense callers of the hot path (ResNet-50, FPN, UNets, BEV encoder: 11 % of a DHD-S training step on
// MIOpen's kernels, which move these tensors at 0.7-3.6 TB/s).  Forward: plane sums -> per-channel finalize
// (double) -> y = x * scale + shift.  Backward: plane sums of g and g * (x - mean) -> coefficients ->
// gx = c0 * g + c1 * x + c2.  Every pass streams 16-byte vectors; eight passes over the tensor in total.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "common.h"

namespace {

constexpr int kBnBlock = 256;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 16 bytes of activations <-> float lanes
template <typename T>
struct Vec;
template <>
struct Vec<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    const f32x4 t = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
  }
};
template <>
struct Vec<__half> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const __half* p, float (&v)[8]) {
    const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned w = t[i];
      const __half2 h = *reinterpret_cast<const __half2*>(&w);
      v[2 * i] = __low2float(h);
      v[2 * i + 1] = __high2float(h);
    }
  }
  static __device__ __forceinline__ void store(__half* p, const float (&v)[8]) {
    u32x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const __half2 h = __floats2half2_rn(v[2 * i], v[2 * i + 1]);
      t[i] = *reinterpret_cast<const unsigned*>(&h);
    }
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(p));
  }
};
template <>
struct Vec<__hip_bfloat16> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const __hip_bfloat16* p, float (&v)[8]) {
    const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(t[i] << 16);
      v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store(__hip_bfloat16* p, const float (&v)[8]) {
    u32x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      auto rne = [](float f) {  // round to nearest even, as PyTorch's float -> bfloat16
        const unsigned u = __float_as_uint(f);
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
      };
      t[i] = rne(v[2 * i]) | (rne(v[2 * i + 1]) << 16);
    }
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(p));
  }
};

__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = group_sum(v, DHD_WAVE);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < kBnBlock / DHD_WAVE; ++i) t += sm[i];
  return t;
}

// grid (chunks, n*c).  BWD == false: s1 = sum (x - shift), s2 = sum (x - shift)^2 with shift = the channel's first
// value (keeps the float32 sums small);  BWD == true: s1 = sum g, s2 = sum g * (x - mean).
// part: [(n_idx * chunks + chunk)][2][c]
template <typename T, bool BWD>
__global__ __launch_bounds__(kBnBlock) void bn_plane_sums(const T* __restrict__ x, const T* __restrict__ g, const float* __restrict__ mean,
                                                          float* __restrict__ part, int c, int hw) {
  __shared__ float sm[kBnBlock / DHD_WAVE];
  constexpr int N = Vec<T>::N;
  const int plane = blockIdx.y, ni = plane / c, ch = plane % c;
  const float shift = BWD ? mean[ch] : (float)x[(size_t)ch * hw];
  const int nvec = hw / N, per = (nvec + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * per, hi = min(nvec, lo + per);
  const T* xp = x + (size_t)plane * hw;
  const T* gp = BWD ? g + (size_t)plane * hw : nullptr;
  float s1 = 0.f, s2 = 0.f, t1 = 0.f, t2 = 0.f;
  int i = lo + threadIdx.x;
  for (; i + kBnBlock < hi; i += 2 * kBnBlock) {  // two independent 16-byte streams per thread
    float a[N], b[N], ga[N], gb[N];
    Vec<T>::load(xp + (size_t)i * N, a);
    Vec<T>::load(xp + (size_t)(i + kBnBlock) * N, b);
    if (BWD) {
      Vec<T>::load(gp + (size_t)i * N, ga);
      Vec<T>::load(gp + (size_t)(i + kBnBlock) * N, gb);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float da = a[k] - shift, db = b[k] - shift;
      if (BWD) { s1 += ga[k]; s2 = fmaf(ga[k], da, s2); t1 += gb[k]; t2 = fmaf(gb[k], db, t2); }
      else { s1 += da; s2 = fmaf(da, da, s2); t1 += db; t2 = fmaf(db, db, t2); }
    }
  }
  if (i < hi) {
    float a[N], ga[N];
    Vec<T>::load(xp + (size_t)i * N, a);
    if (BWD) Vec<T>::load(gp + (size_t)i * N, ga);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float da = a[k] - shift;
      if (BWD) { s1 += ga[k]; s2 = fmaf(ga[k], da, s2); }
      else { s1 += da; s2 = fmaf(da, da, s2); }
    }
  }
  s1 = block_sum(s1 + t1, sm);
  s2 = block_sum(s2 + t2, sm);
  if (threadIdx.x == 0) {
    float* q = part + ((size_t)(ni * gridDim.x + blockIdx.x) * 2) * c;
    q[ch] = s1;
    q[c + ch] = s2;
  }
}

// per channel: batch mean / biased variance from the shifted sums (double), running statistics, y = x*scale + shift
template <typename T>
__global__ __launch_bounds__(kBnBlock) void bn_forward_finalize(const float* __restrict__ part, int n_part, const T* __restrict__ x, int hw,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                float factor, float eps, float* __restrict__ save_mean,
                                                                float* __restrict__ save_rstd, float* __restrict__ coef, int n, int c) {
  const int ch = blockIdx.x * kBnBlock + threadIdx.x;
  if (ch >= c) return;
  double s1 = 0.0, s2 = 0.0;
  for (int q = 0; q < n_part; ++q) {
    s1 += (double)part[((size_t)q * 2) * c + ch];
    s2 += (double)part[((size_t)q * 2 + 1) * c + ch];
  }
  const double cnt = (double)n * (double)hw, shift = (double)(float)x[(size_t)ch * hw];
  const double m = s1 / cnt, var = fmax(s2 / cnt - m * m, 0.0), mean = shift + m;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  save_mean[ch] = (float)mean;
  save_rstd[ch] = (float)rstd;
  if (running_mean) running_mean[ch] = (float)((1.0 - factor) * running_mean[ch] + factor * mean);
  if (running_var) running_var[ch] = (float)((1.0 - factor) * running_var[ch] + factor * var * (cnt > 1.0 ? cnt / (cnt - 1.0) : 1.0));
  const double ga = gamma ? (double)gamma[ch] : 1.0, be = beta ? (double)beta[ch] : 0.0;
  coef[ch] = (float)(ga * rstd);
  coef[c + ch] = (float)(be - mean * ga * rstd);
}

// dgamma, dbeta and the coefficients of gx = c0*g + c1*x + c2
__global__ __launch_bounds__(kBnBlock) void bn_backward_finalize(const float* __restrict__ part, int n_part, const float* __restrict__ gamma,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                 float* __restrict__ coef, int n, int c, int hw) {
  const int ch = blockIdx.x * kBnBlock + threadIdx.x;
  if (ch >= c) return;
  double s1 = 0.0, s2 = 0.0;
  for (int q = 0; q < n_part; ++q) {
    s1 += (double)part[((size_t)q * 2) * c + ch];
    s2 += (double)part[((size_t)q * 2 + 1) * c + ch];
  }
  const double cnt = (double)n * (double)hw, rs = (double)rstd[ch], mu = (double)mean[ch], ga = gamma ? (double)gamma[ch] : 1.0;
  if (dgamma) dgamma[ch] = (float)(rs * s2);
  if (dbeta) dbeta[ch] = (float)s1;
  const double c0 = ga * rs, c1 = -ga * rs * rs * rs * s2 / cnt, c2 = -ga * rs * s1 / cnt - c1 * mu;
  coef[ch] = (float)c0;
  coef[c + ch] = (float)c1;
  coef[2 * c + ch] = (float)c2;
}

// out = k0[c]*a + k1[c]*b + k2[c]   (forward: a = x, k1 = 0;  backward: a = g, b = x).  grid (chunks, n*c)
template <typename T, bool TWO>
__global__ __launch_bounds__(kBnBlock) void bn_affine(const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ coef,
                                                      T* __restrict__ out, int c, int hw) {
  constexpr int N = Vec<T>::N;
  const int plane = blockIdx.y, ch = plane % c;
  const float k0 = coef[ch], k1 = coef[c + ch], k2 = TWO ? coef[2 * c + ch] : 0.f;
  const int nvec = hw / N, per = (nvec + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * per, hi = min(nvec, lo + per);
  const T* ap = a + (size_t)plane * hw;
  const T* bp = TWO ? b + (size_t)plane * hw : nullptr;
  T* op = out + (size_t)plane * hw;
  for (int i = lo + threadIdx.x; i < hi; i += kBnBlock) {
    float va[N], vb[N], r[N];
    Vec<T>::load(ap + (size_t)i * N, va);
    if (TWO) Vec<T>::load(bp + (size_t)i * N, vb);
#pragma unroll
    for (int k = 0; k < N; ++k) r[k] = TWO ? fmaf(k0, va[k], fmaf(k1, vb[k], k2)) : fmaf(k0, va[k], k1);
    Vec<T>::store(op + (size_t)i * N, r);
  }
}

inline int bn_chunks(int n, int c, int hw, int vec) {
  // enough workgroups to fill the chip, at least ~2 vectors per thread in a chunk
  const long planes = (long)n * c, nvec = hw / vec;
  int chunks = 1;
  while (chunks < 16 && planes * chunks < 4096 && nvec / (chunks * 2) >= 2 * kBnBlock) chunks *= 2;
  return chunks;
}

template <typename T>
int bn_forward_t(const T* x, int n, int c, int hw, const float* gamma, const float* beta, float* rm, float* rv, float factor, float eps,
                 T* y, float* save_mean, float* save_rstd, float* ws, hipStream_t st) {
  const int chunks = bn_chunks(n, c, hw, Vec<T>::N);
  float* part = ws;
  float* coef = ws + (size_t)n * chunks * 2 * c;
  const dim3 grid(chunks, n * c);
  hipLaunchKernelGGL((bn_plane_sums<T, false>), grid, dim3(kBnBlock), 0, st, x, (const T*)nullptr, (const float*)nullptr, part, c, hw);
  hipLaunchKernelGGL((bn_forward_finalize<T>), dim3(dhd_cdiv(c, kBnBlock)), dim3(kBnBlock), 0, st, part, n * chunks, x, hw, gamma, beta, rm, rv,
                     factor, eps, save_mean, save_rstd, coef, n, c);
  hipLaunchKernelGGL((bn_affine<T, false>), grid, dim3(kBnBlock), 0, st, x, (const T*)nullptr, coef, y, c, hw);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

template <typename T>
int bn_backward_t(const T* x, const T* gy, int n, int c, int hw, const float* gamma, const float* mean, const float* rstd, T* gx,
                  float* dgamma, float* dbeta, float* ws, hipStream_t st) {
  const int chunks = bn_chunks(n, c, hw, Vec<T>::N);
  float* part = ws;
  float* coef = ws + (size_t)n * chunks * 2 * c;
  const dim3 grid(chunks, n * c);
  hipLaunchKernelGGL((bn_plane_sums<T, true>), grid, dim3(kBnBlock), 0, st, x, gy, mean, part, c, hw);
  hipLaunchKernelGGL(bn_backward_finalize, dim3(dhd_cdiv(c, kBnBlock)), dim3(kBnBlock), 0, st, part, n * chunks, gamma, mean, rstd, dgamma,
                     dbeta, coef, n, c, hw);
  hipLaunchKernelGGL((bn_affine<T, true>), grid, dim3(kBnBlock), 0, st, gy, x, coef, gx, c, hw);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

inline bool bn_shape_ok(int dtype, int n, int c, int hw) {
  if (n <= 0 || c <= 0 || hw <= 0 || dtype < 0 || dtype > 2) return false;
  if ((long)n * c > 65535L * 16) return false;
  return hw % (dtype == 0 ? 4 : 8) == 0;
}

}  // namespace

extern "C" {

int dhd_bn_supported(int dtype, int n, int c, int hw) { return bn_shape_ok(dtype, n, c, hw) && (long)n * c <= 65535 ? 1 : 0; }

size_t dhd_bn_workspace_bytes(int n, int c, int hw) {
  if (n <= 0 || c <= 0 || hw <= 0) return 0;
  return ((size_t)n * 16 * 2 * c + 3 * (size_t)c) * sizeof(float);
}

int dhd_bn_train_forward(const void* x, int dtype, int n, int c, int hw, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float factor, float eps, void* y, float* save_mean, float* save_rstd, void* workspace,
                         void* stream) {
  if (!x || !y || !save_mean || !save_rstd || !workspace) return DHD_EINVAL;
  if (!dhd_bn_supported(dtype, n, c, hw)) return DHD_EUNSUPPORTED;
  hipStream_t st = dhd_stream(stream);
  float* ws = static_cast<float*>(workspace);
  switch (dtype) {
    case 0: return bn_forward_t<float>((const float*)x, n, c, hw, gamma, beta, running_mean, running_var, factor, eps, (float*)y, save_mean, save_rstd, ws, st);
    case 1: return bn_forward_t<__half>((const __half*)x, n, c, hw, gamma, beta, running_mean, running_var, factor, eps, (__half*)y, save_mean, save_rstd, ws, st);
    default: return bn_forward_t<__hip_bfloat16>((const __hip_bfloat16*)x, n, c, hw, gamma, beta, running_mean, running_var, factor, eps, (__hip_bfloat16*)y, save_mean, save_rstd, ws, st);
  }
}

int dhd_bn_train_backward(const void* x, const void* grad_y, int dtype, int n, int c, int hw, const float* gamma, const float* save_mean,
                          const float* save_rstd, void* grad_x, float* dgamma, float* dbeta, void* workspace, void* stream) {
  if (!x || !grad_y || !grad_x || !save_mean || !save_rstd || !workspace) return DHD_EINVAL;
  if (!dhd_bn_supported(dtype, n, c, hw)) return DHD_EUNSUPPORTED;
  hipStream_t st = dhd_stream(stream);
  float* ws = static_cast<float*>(workspace);
  switch (dtype) {
    case 0: return bn_backward_t<float>((const float*)x, (const float*)grad_y, n, c, hw, gamma, save_mean, save_rstd, (float*)grad_x, dgamma, dbeta, ws, st);
    case 1: return bn_backward_t<__half>((const __half*)x, (const __half*)grad_y, n, c, hw, gamma, save_mean, save_rstd, (__half*)grad_x, dgamma, dbeta, ws, st);
    default: return bn_backward_t<__hip_bfloat16>((const __hip_bfloat16*)x, (const __hip_bfloat16*)grad_y, n, c, hw, gamma, save_mean, save_rstd, (__hip_bfloat16*)grad_x, dgamma, dbeta, ws, st);
  }
}

}  // extern "C"
