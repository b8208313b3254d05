// Training-mode BatchNorm2d (NCHW and, second half of the file, channels_last; float32, float16 or bfloat16 activations, float32
// parameters and statistics)
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
  static __device__ __forceinline__ float rnd(float f) { return f; }   // the value `store` leaves in memory
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
  static __device__ __forceinline__ float rnd(float f) { return __half2float(__float2half_rn(f)); }
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
  static __device__ __forceinline__ float rnd(float f) {
    const unsigned u = __float_as_uint(f);
    return __uint_as_float(((u + 0x7fffu + ((u >> 16) & 1u)) >> 16) << 16);
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
  const double m = s1 / cnt, var = fmax(s2 / cnt - m * m, 0.0), mean = shift + m;   // (inf - inf clamps to 0, as torch's kernels do: an Inf input turns the channel's finite entries into -inf)
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

// =======================================================================================
// channels_last ("NHWC") tensors: x is a row-major matrix [rows = n*hw][c].  A thread owns one 16-byte vector of channels and
// walks down the rows (the kBnBlock threads of a workgroup cover vpb <= 16 vectors of R = 256 / vpb consecutive rows per step:
// cl_grid); its N sums stay in registers and meet those of the other row lanes in LDS once per workgroup.
// Optional fused epilogues of the dense callers' BatchNorm -> ReLU and BatchNorm -> (+ residual) -> ReLU pairs:
//   forward   y = max(0, k0*x + k1 [+ res])
//   backward  g' = g where the forward output was positive, else 0; the mask is recomputed from x and the saved (k0, k1)
//             (MASK 1: same expression, same rounding to T as the forward's store) or read from the saved output (MASK 2: the
//             residual form, which also hands g' to the residual branch).
// part: [2][c][n_part]
// =======================================================================================
struct ClGrid {
  int V, vpb, R, gy, rpb, n_part;
};
inline ClGrid cl_grid(long rows, int c, int vec) {
  // A workgroup takes 16 vectors (256 bytes) of every row it visits and R = 16 rows per step (narrower tensors: all V vectors and
  // 256 / V rows); wider tensors are split into gy column groups.  Its 2 * 128 partial sums are then a small fraction of what it
  // read whatever c is, and one channel's partials are contiguous for the finalize kernel ([2][c][n_part]).  At least 8 steps
  // (32 KB) per workgroup, about 2048 workgroups at most.
  ClGrid g;
  g.V = c / vec;
  g.vpb = g.V < 16 ? g.V : 16;
  g.R = kBnBlock / g.vpb;
  g.gy = (g.V + g.vpb - 1) / g.vpb;
  long rpb = (rows * g.gy + 2047) / 2048;
  if (rpb < 8L * g.R) rpb = 8L * g.R;
  rpb = (rpb + g.R - 1) / g.R * g.R;
  g.rpb = (int)rpb;
  g.n_part = (int)((rows + rpb - 1) / rpb);
  return g;
}

template <typename T, bool BWD, int MASK>
__global__ __launch_bounds__(kBnBlock) void bn_cl_sums(const T* __restrict__ x, const T* __restrict__ g, const T* __restrict__ y,
                                                       const float* __restrict__ mean, const float* __restrict__ fco,
                                                       float* __restrict__ part, int c, long rows, int rpb, int n_part, int vpb) {
  constexpr int N = Vec<T>::N;
  __shared__ float sm[2 * N * kBnBlock];
  const int V = c / N, R = kBnBlock / vpb;
  const int vl = threadIdx.x % vpb, r = threadIdx.x / vpb;
  const int v = blockIdx.y * vpb + vl;
  const long lo = (long)blockIdx.x * rpb, hi = min(rows, lo + (long)rpb);
  float s1[N], s2[N];
#pragma unroll
  for (int k = 0; k < N; ++k) s1[k] = s2[k] = 0.f;
  if (r < R && v < V) {
    const int ch0 = v * N;
    float sh[N], f0[N], f1[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      sh[k] = BWD ? mean[ch0 + k] : (float)x[ch0 + k];   // forward: the channel's first value keeps the float32 sums small
      f0[k] = MASK == 1 ? fco[ch0 + k] : 0.f;
      f1[k] = MASK == 1 ? fco[c + ch0 + k] : 0.f;
    }
    auto acc = [&](const float (&a)[N], const float (&ga)[N], const float (&ya)[N]) {
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const float da = a[k] - sh[k];
        if (BWD) {
          float gg = ga[k];
          if (MASK == 1) gg = Vec<T>::rnd(fmaf(f0[k], a[k], f1[k])) > 0.f ? gg : 0.f;
          if (MASK == 2) gg = ya[k] > 0.f ? gg : 0.f;
          s1[k] += gg;
          s2[k] = fmaf(gg, da, s2[k]);
        } else {
          s1[k] += da;
          s2[k] = fmaf(da, da, s2[k]);
        }
      }
    };
    const T* xp = x + ch0;
    const T* gp = BWD ? g + ch0 : nullptr;
    const T* yp = MASK == 2 ? y + ch0 : nullptr;
    constexpr int U = BWD ? 2 : 4;   // independent 16-byte loads per tensor and thread
    long row = lo + r;
    for (; row + (long)(U - 1) * R < hi; row += (long)U * R) {
      float a[U][N], ga[U][N], ya[U][N];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t o = (size_t)(row + (long)u * R) * c;
        Vec<T>::load(xp + o, a[u]);
        if (BWD) Vec<T>::load(gp + o, ga[u]);
        if (MASK == 2) Vec<T>::load(yp + o, ya[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc(a[u], ga[u], ya[u]);
    }
    for (; row < hi; row += R) {
      float a[N], ga[N], ya[N];
      const size_t o = (size_t)row * c;
      Vec<T>::load(xp + o, a);
      if (BWD) Vec<T>::load(gp + o, ga);
      if (MASK == 2) Vec<T>::load(yp + o, ya);
      acc(a, ga, ya);
    }
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    sm[(0 * N + k) * kBnBlock + threadIdx.x] = s1[k];
    sm[(1 * N + k) * kBnBlock + threadIdx.x] = s2[k];
  }
  __syncthreads();
  // one thread per (sum, channel of this workgroup's vectors): adds the R row lanes in a fixed order
  for (int o = threadIdx.x; o < 2 * N * vpb; o += kBnBlock) {
    const int q = o / (N * vpb), k = (o / vpb) % N, vv = o % vpb;
    if (blockIdx.y * vpb + vv >= V) continue;
    float t = 0.f;
    for (int rr = 0; rr < R; ++rr) t += sm[(q * N + k) * kBnBlock + rr * vpb + vv];
    part[((size_t)q * c + (size_t)(blockIdx.y * vpb + vv) * N + k) * n_part + blockIdx.x] = t;
  }
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, DHD_WAVE);
  return v;
}

// the n_part partial sums of channel ch, added in a fixed order by one wave (float loads four deep, double accumulation)
__device__ __forceinline__ void fold_parts(const float* __restrict__ part, int n_part, int c, int ch, int lane, double* s1, double* s2) {
  const float* p1 = part + (size_t)ch * n_part;
  const float* p2 = part + ((size_t)c + ch) * n_part;
  double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
  int q = lane;
  for (; q + 3 * DHD_WAVE < n_part; q += 4 * DHD_WAVE) {
    float u[4], w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { u[j] = p1[q + j * DHD_WAVE]; w[j] = p2[q + j * DHD_WAVE]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] += (double)u[j]; b[j] += (double)w[j]; }
  }
  for (; q < n_part; q += DHD_WAVE) { a[0] += (double)p1[q]; b[0] += (double)p2[q]; }
  *s1 = wave_sum_f64((a[0] + a[1]) + (a[2] + a[3]));
  *s2 = wave_sum_f64((b[0] + b[1]) + (b[2] + b[3]));
}

// one wave per channel: folds the workgroups' partial sums (double), then the per-channel arithmetic of bn_forward_finalize
template <typename T>
__global__ __launch_bounds__(kBnBlock) void bn_cl_forward_finalize(const float* __restrict__ part, int n_part, const T* __restrict__ x,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                   float factor, float eps, float* __restrict__ save_mean,
                                                                   float* __restrict__ save_rstd, float* __restrict__ coef, double cnt, int c) {
  const int ch = blockIdx.x * (kBnBlock / DHD_WAVE) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ch >= c) return;
  double s1, s2;
  fold_parts(part, n_part, c, ch, lane, &s1, &s2);
  if (lane) return;
  const double shift = (double)(float)x[ch];
  const double m = s1 / cnt, var = fmax(s2 / cnt - m * m, 0.0), mean = shift + m;   // (inf - inf clamps to 0, as torch's kernels do: an Inf input turns the channel's finite entries into -inf)
  const double rstd = 1.0 / sqrt(var + (double)eps);
  save_mean[ch] = (float)mean;
  save_rstd[ch] = (float)rstd;
  if (running_mean) running_mean[ch] = (float)((1.0 - factor) * running_mean[ch] + factor * mean);
  if (running_var) running_var[ch] = (float)((1.0 - factor) * running_var[ch] + factor * var * (cnt > 1.0 ? cnt / (cnt - 1.0) : 1.0));
  const double ga = gamma ? (double)gamma[ch] : 1.0, be = beta ? (double)beta[ch] : 0.0;
  coef[ch] = (float)(ga * rstd);
  coef[c + ch] = (float)(be - mean * ga * rstd);
}

__global__ __launch_bounds__(kBnBlock) void bn_cl_backward_finalize(const float* __restrict__ part, int n_part, const float* __restrict__ gamma,
                                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                    float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                    float* __restrict__ coef, double cnt, int c) {
  const int ch = blockIdx.x * (kBnBlock / DHD_WAVE) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ch >= c) return;
  double s1, s2;
  fold_parts(part, n_part, c, ch, lane, &s1, &s2);
  if (lane) return;
  const double rs = (double)rstd[ch], mu = (double)mean[ch], ga = gamma ? (double)gamma[ch] : 1.0;
  if (dgamma) dgamma[ch] = (float)(rs * s2);
  if (dbeta) dbeta[ch] = (float)s1;
  const double c0 = ga * rs, c1 = -ga * rs * rs * rs * s2 / cnt, c2 = -ga * rs * s1 / cnt - c1 * mu;
  coef[ch] = (float)c0;
  coef[c + ch] = (float)c1;
  coef[2 * c + ch] = (float)c2;
}

// y = k0*x + k1, optionally + res, optionally max(0, .)
template <typename T, bool RELU, bool ADD>
__global__ __launch_bounds__(kBnBlock) void bn_cl_apply_fwd(const T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ coef,
                                                            T* __restrict__ y, int c, long rows, int rpb, int vpb) {
  constexpr int N = Vec<T>::N;
  const int V = c / N, R = kBnBlock / vpb;
  const int vl = threadIdx.x % vpb, r = threadIdx.x / vpb;
  const int v = blockIdx.y * vpb + vl;
  if (r >= R || v >= V) return;
  const long lo = (long)blockIdx.x * rpb, hi = min(rows, lo + (long)rpb);
  const int ch0 = v * N;
  float k0[N], k1[N];
#pragma unroll
  for (int k = 0; k < N; ++k) { k0[k] = coef[ch0 + k]; k1[k] = coef[c + ch0 + k]; }
  constexpr int U = ADD ? 2 : 4;
  long row = lo + r;
  auto one = [&](const float (&a)[N], const float (&b)[N], size_t o) {
    float out[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      float t = fmaf(k0[k], a[k], k1[k]);
      if (ADD) t += b[k];
      out[k] = RELU ? (t < 0.f ? 0.f : t) : t;   // not fmaxf: a NaN must come out as NaN, as torch.relu gives (fmaxf(NaN, 0) = 0)
    }
    Vec<T>::store(y + ch0 + o, out);
  };
  for (; row + (long)(U - 1) * R < hi; row += (long)U * R) {
    float a[U][N], b[U][N];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t o = (size_t)(row + (long)u * R) * c;
      Vec<T>::load(x + ch0 + o, a[u]);
      if (ADD) Vec<T>::load(res + ch0 + o, b[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) one(a[u], b[u], (size_t)(row + (long)u * R) * c);
  }
  for (; row < hi; row += R) {
    float a[N], b[N];
    const size_t o = (size_t)row * c;
    Vec<T>::load(x + ch0 + o, a);
    if (ADD) Vec<T>::load(res + ch0 + o, b);
    one(a, b, o);
  }
}

// gx = c0*g' + c1*x + c2 with g' the masked gradient (MASK as in bn_cl_sums); MASK 2 also stores g' for the residual branch
template <typename T, int MASK>
__global__ __launch_bounds__(kBnBlock) void bn_cl_apply_bwd(const T* __restrict__ x, const T* __restrict__ g, const T* __restrict__ y,
                                                            const float* __restrict__ coef, const float* __restrict__ fco,
                                                            T* __restrict__ gx, T* __restrict__ gres, int c, long rows, int rpb, int vpb) {
  constexpr int N = Vec<T>::N;
  const int V = c / N, R = kBnBlock / vpb;
  const int vl = threadIdx.x % vpb, r = threadIdx.x / vpb;
  const int v = blockIdx.y * vpb + vl;
  if (r >= R || v >= V) return;
  const long lo = (long)blockIdx.x * rpb, hi = min(rows, lo + (long)rpb);
  const int ch0 = v * N;
  float c0[N], c1[N], c2[N], f0[N], f1[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    c0[k] = coef[ch0 + k]; c1[k] = coef[c + ch0 + k]; c2[k] = coef[2 * c + ch0 + k];
    f0[k] = MASK == 1 ? fco[ch0 + k] : 0.f;
    f1[k] = MASK == 1 ? fco[c + ch0 + k] : 0.f;
  }
  constexpr int U = 2;
  long row = lo + r;
  auto one = [&](const float (&a)[N], const float (&ga)[N], const float (&ya)[N], size_t o) {
    float out[N], gm[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      float gg = ga[k];
      if (MASK == 1) gg = Vec<T>::rnd(fmaf(f0[k], a[k], f1[k])) > 0.f ? gg : 0.f;
      if (MASK == 2) gg = ya[k] > 0.f ? gg : 0.f;
      gm[k] = gg;
      out[k] = fmaf(c0[k], gg, fmaf(c1[k], a[k], c2[k]));
    }
    Vec<T>::store(gx + ch0 + o, out);
    if (MASK == 2 && gres) Vec<T>::store(gres + ch0 + o, gm);
  };
  for (; row + (long)(U - 1) * R < hi; row += (long)U * R) {
    float a[U][N], ga[U][N], ya[U][N];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t o = (size_t)(row + (long)u * R) * c;
      Vec<T>::load(x + ch0 + o, a[u]);
      Vec<T>::load(g + ch0 + o, ga[u]);
      if (MASK == 2) Vec<T>::load(y + ch0 + o, ya[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) one(a[u], ga[u], ya[u], (size_t)(row + (long)u * R) * c);
  }
  for (; row < hi; row += R) {
    float a[N], ga[N], ya[N];
    const size_t o = (size_t)row * c;
    Vec<T>::load(x + ch0 + o, a);
    Vec<T>::load(g + ch0 + o, ga);
    if (MASK == 2) Vec<T>::load(y + ch0 + o, ya);
    one(a, ga, ya, o);
  }
}

inline bool bn_cl_shape_ok(int dtype, long rows, int c) {
  if (rows <= 0 || c <= 0 || dtype < 0 || dtype > 2) return false;
  if (c % (dtype == 0 ? 4 : 8)) return false;
  return rows <= (1L << 40) && c <= (1 << 20);
}

template <typename T>
int bn_cl_forward_t(const T* x, const T* res, long rows, int c, int flags, const float* gamma, const float* beta, float* rm, float* rv,
                    float factor, float eps, T* y, float* save_mean, float* save_rstd, float* save_affine, float* ws, hipStream_t st) {
  const ClGrid g = cl_grid(rows, c, Vec<T>::N);
  const dim3 grid(g.n_part, g.gy), blk(kBnBlock);
  hipLaunchKernelGGL((bn_cl_sums<T, false, 0>), grid, blk, 0, st, x, (const T*)nullptr, (const T*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, ws, c, rows, g.rpb, g.n_part, g.vpb);
  hipLaunchKernelGGL((bn_cl_forward_finalize<T>), dim3(dhd_cdiv(c, kBnBlock / DHD_WAVE)), blk, 0, st, ws, g.n_part, x, gamma, beta, rm, rv,
                     factor, eps, save_mean, save_rstd, save_affine, (double)rows, c);
  const bool relu = flags & (DHD_BN_RELU | DHD_BN_ADD), add = flags & DHD_BN_ADD;
  if (add) hipLaunchKernelGGL((bn_cl_apply_fwd<T, true, true>), grid, blk, 0, st, x, res, save_affine, y, c, rows, g.rpb, g.vpb);
  else if (relu) hipLaunchKernelGGL((bn_cl_apply_fwd<T, true, false>), grid, blk, 0, st, x, res, save_affine, y, c, rows, g.rpb, g.vpb);
  else hipLaunchKernelGGL((bn_cl_apply_fwd<T, false, false>), grid, blk, 0, st, x, res, save_affine, y, c, rows, g.rpb, g.vpb);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

template <typename T>
int bn_cl_backward_t(const T* x, const T* y, const T* gy, long rows, int c, int flags, const float* gamma, const float* mean,
                     const float* rstd, const float* save_affine, T* gx, T* gres, float* dgamma, float* dbeta, float* ws, hipStream_t st) {
  const ClGrid g = cl_grid(rows, c, Vec<T>::N);
  const dim3 grid(g.n_part, g.gy), blk(kBnBlock);
  float* coef = ws + (size_t)2 * c * g.n_part;
  const int mask = (flags & DHD_BN_ADD) ? 2 : (flags & DHD_BN_RELU) ? 1 : 0;
  if (mask == 2) hipLaunchKernelGGL((bn_cl_sums<T, true, 2>), grid, blk, 0, st, x, gy, y, mean, save_affine, ws, c, rows, g.rpb, g.n_part, g.vpb);
  else if (mask == 1) hipLaunchKernelGGL((bn_cl_sums<T, true, 1>), grid, blk, 0, st, x, gy, y, mean, save_affine, ws, c, rows, g.rpb, g.n_part, g.vpb);
  else hipLaunchKernelGGL((bn_cl_sums<T, true, 0>), grid, blk, 0, st, x, gy, y, mean, save_affine, ws, c, rows, g.rpb, g.n_part, g.vpb);
  hipLaunchKernelGGL(bn_cl_backward_finalize, dim3(dhd_cdiv(c, kBnBlock / DHD_WAVE)), blk, 0, st, ws, g.n_part, gamma, mean, rstd, dgamma, dbeta,
                     coef, (double)rows, c);
  if (mask == 2) hipLaunchKernelGGL((bn_cl_apply_bwd<T, 2>), grid, blk, 0, st, x, gy, y, coef, save_affine, gx, gres, c, rows, g.rpb, g.vpb);
  else if (mask == 1) hipLaunchKernelGGL((bn_cl_apply_bwd<T, 1>), grid, blk, 0, st, x, gy, y, coef, save_affine, gx, gres, c, rows, g.rpb, g.vpb);
  else hipLaunchKernelGGL((bn_cl_apply_bwd<T, 0>), grid, blk, 0, st, x, gy, y, coef, save_affine, gx, gres, c, rows, g.rpb, g.vpb);
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

int dhd_bn_nhwc_supported(int dtype, long rows, int c) { return bn_cl_shape_ok(dtype, rows, c) ? 1 : 0; }

size_t dhd_bn_nhwc_workspace_bytes(long rows, int c) {
  if (rows <= 0 || c <= 0) return 0;
  return ((size_t)2 * c * 2049 + 3 * (size_t)c) * sizeof(float);
}

int dhd_bn_nhwc_train_forward(const void* x, const void* residual, int dtype, long rows, int c, int flags, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, float factor, float eps, void* y,
                              float* save_mean, float* save_rstd, float* save_affine, void* workspace, void* stream) {
  if (!x || !y || !save_mean || !save_rstd || !save_affine || !workspace) return DHD_EINVAL;
  if ((flags & DHD_BN_ADD) && !residual) return DHD_EINVAL;
  if (!bn_cl_shape_ok(dtype, rows, c)) return DHD_EUNSUPPORTED;
  hipStream_t st = dhd_stream(stream);
  float* ws = static_cast<float*>(workspace);
  switch (dtype) {
    case 0: return bn_cl_forward_t<float>((const float*)x, (const float*)residual, rows, c, flags, gamma, beta, running_mean, running_var, factor, eps, (float*)y, save_mean, save_rstd, save_affine, ws, st);
    case 1: return bn_cl_forward_t<__half>((const __half*)x, (const __half*)residual, rows, c, flags, gamma, beta, running_mean, running_var, factor, eps, (__half*)y, save_mean, save_rstd, save_affine, ws, st);
    default: return bn_cl_forward_t<__hip_bfloat16>((const __hip_bfloat16*)x, (const __hip_bfloat16*)residual, rows, c, flags, gamma, beta, running_mean, running_var, factor, eps, (__hip_bfloat16*)y, save_mean, save_rstd, save_affine, ws, st);
  }
}

int dhd_bn_nhwc_train_backward(const void* x, const void* y, const void* grad_y, int dtype, long rows, int c, int flags,
                               const float* gamma, const float* save_mean, const float* save_rstd, const float* save_affine,
                               void* grad_x, void* grad_residual, float* dgamma, float* dbeta, void* workspace, void* stream) {
  if (!x || !grad_y || !grad_x || !save_mean || !save_rstd || !workspace) return DHD_EINVAL;
  if ((flags & DHD_BN_ADD) && !y) return DHD_EINVAL;
  if ((flags & DHD_BN_RELU) && !(flags & DHD_BN_ADD) && !save_affine) return DHD_EINVAL;
  if (!bn_cl_shape_ok(dtype, rows, c)) return DHD_EUNSUPPORTED;
  hipStream_t st = dhd_stream(stream);
  float* ws = static_cast<float*>(workspace);
  switch (dtype) {
    case 0: return bn_cl_backward_t<float>((const float*)x, (const float*)y, (const float*)grad_y, rows, c, flags, gamma, save_mean, save_rstd, save_affine, (float*)grad_x, (float*)grad_residual, dgamma, dbeta, ws, st);
    case 1: return bn_cl_backward_t<__half>((const __half*)x, (const __half*)y, (const __half*)grad_y, rows, c, flags, gamma, save_mean, save_rstd, save_affine, (__half*)grad_x, (__half*)grad_residual, dgamma, dbeta, ws, st);
    default: return bn_cl_backward_t<__hip_bfloat16>((const __hip_bfloat16*)x, (const __hip_bfloat16*)y, (const __hip_bfloat16*)grad_y, rows, c, flags, gamma, save_mean, save_rstd, save_affine, (__hip_bfloat16*)grad_x, (__hip_bfloat16*)grad_residual, dgamma, dbeta, ws, st);
  }
}

}  // extern "C"
