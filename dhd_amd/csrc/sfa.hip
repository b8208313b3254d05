// SFA channel/spatial attention stage, memory-bound parts, for gfx950.
//
// Reference: models/necks/mix.py:37-59 (channel_spatial_stage.forward).  In eager PyTorch this
// stage is one full reduction plus ~6 element-wise passes over 41-82 MB tensors per sample; here
// it is one reduction kernel and two fused blend kernels (forward), and three streaming kernels
// (backward).  The two 1x1 convolutions + BatchNorm between the blends stay on the dense
// (MFMA-backed) library path.
//
// x is (B, 2C, H, W): channels [0,C) = x_bev, [C,2C) = x_voxel.  One workgroup streams a chunk of
// one (b, c) plane with 16-byte accesses; the per-plane scalars (a1) are wave-uniform.
#include "common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kChunksPerPlane = 4;

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }

__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = group_sum(v, DHD_WAVE);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) sm[wv] = v;
  __syncthreads();
  float t = 0.f;
  for (int k = 0; k < kBlock / DHD_WAVE; ++k) t += sm[k];
  __syncthreads();
  return t;
}

// Chunk [lo, hi) of a plane of hw floats, in units of float4 when VEC.
__device__ __forceinline__ void chunk_range(int hw_units, int* lo, int* hi) {
  int per = (hw_units + kChunksPerPlane - 1) / kChunksPerPlane;
  *lo = blockIdx.x * per;
  *hi = min(hw_units, *lo + per);
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void channel_mean_kernel(const float* __restrict__ x, float* __restrict__ s, int hw) {
  __shared__ float sm[kBlock / DHD_WAVE];
  const size_t plane = blockIdx.x;
  const float* p = x + plane * hw;
  float acc = 0.f;
  if (VEC) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
    for (int i = threadIdx.x; i < hw / 4; i += kBlock) {
      float4 v = p4[i];
      acc += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (int i = threadIdx.x; i < hw; i += kBlock) acc += p[i];
  }
  float tot = block_sum(acc, sm);
  if (threadIdx.x == 0) s[plane] = tot / (float)hw;
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void blend1_kernel(const float* __restrict__ x, const float* __restrict__ a1,
                                                        float* __restrict__ u, int c, int hw) {
  const int plane = blockIdx.y;  // b*C + ch
  const int b = plane / c, ch = plane % c;
  const float a = a1[plane], na = 1.0f - a;
  const float* xb = x + ((size_t)b * 2 * c + ch) * hw;
  const float* xv = xb + (size_t)c * hw;
  float* o = u + (size_t)plane * hw;
  int lo, hi;
  if (VEC) {
    chunk_range(hw / 4, &lo, &hi);
    const float4* b4 = reinterpret_cast<const float4*>(xb);
    const float4* v4 = reinterpret_cast<const float4*>(xv);
    float4* o4 = reinterpret_cast<float4*>(o);
    for (int i = lo + threadIdx.x; i < hi; i += kBlock) {
      float4 p = b4[i], q = v4[i], r;
      r.x = a * p.x + na * q.x; r.y = a * p.y + na * q.y; r.z = a * p.z + na * q.z; r.w = a * p.w + na * q.w;
      o4[i] = r;
    }
  } else {
    chunk_range(hw, &lo, &hi);
    for (int i = lo + threadIdx.x; i < hi; i += kBlock) o[i] = a * xb[i] + na * xv[i];
  }
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void blend2_kernel(const float* __restrict__ x, const float* __restrict__ a1,
                                                        const float* __restrict__ s2, float* __restrict__ out, int c, int hw) {
  const int plane = blockIdx.y;
  const int b = plane / c, ch = plane % c;
  const float a = a1[plane], na = 1.0f - a;
  const float* xb = x + ((size_t)b * 2 * c + ch) * hw;
  const float* xv = xb + (size_t)c * hw;
  const float* sp = s2 + (size_t)plane * hw;
  float* o = out + (size_t)plane * hw;
  int lo, hi;
  auto f = [&](float p, float q, float s) {
    float g = sigmoidf_(s);
    return g * (a * p) + (1.0f - g) * (na * q);
  };
  if (VEC) {
    chunk_range(hw / 4, &lo, &hi);
    const float4* b4 = reinterpret_cast<const float4*>(xb);
    const float4* v4 = reinterpret_cast<const float4*>(xv);
    const float4* s4 = reinterpret_cast<const float4*>(sp);
    float4* o4 = reinterpret_cast<float4*>(o);
    for (int i = lo + threadIdx.x; i < hi; i += kBlock) {
      float4 p = b4[i], q = v4[i], s = s4[i], r;
      r.x = f(p.x, q.x, s.x); r.y = f(p.y, q.y, s.y); r.z = f(p.z, q.z, s.z); r.w = f(p.w, q.w, s.w);
      o4[i] = r;
    }
  } else {
    chunk_range(hw, &lo, &hi);
    for (int i = lo + threadIdx.x; i < hi; i += kBlock) o[i] = f(xb[i], xv[i], sp[i]);
  }
}

// d/d(s2), and the blend2 part of d/dx and d/da1.
//   out = g*xb1 + (1-g)*xv1,  g = sigmoid(s2), xb1 = a*xb, xv1 = (1-a)*xv
template <bool VEC>
__global__ __launch_bounds__(kBlock) void blend2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ a1,
                                                            const float* __restrict__ s2, const float* __restrict__ go,
                                                            float* __restrict__ gx, float* __restrict__ gs2,
                                                            float* __restrict__ ga1, int c, int hw) {
  __shared__ float sm[kBlock / DHD_WAVE];
  const int plane = blockIdx.y;
  const int b = plane / c, ch = plane % c;
  const float a = a1[plane], na = 1.0f - a;
  const size_t ob = ((size_t)b * 2 * c + ch) * hw, ov = ob + (size_t)c * hw, op = (size_t)plane * hw;
  float acc = 0.f;
  auto f = [&](float p, float q, float s, float g_out, float* gp, float* gq, float* gs) {
    float g = sigmoidf_(s);
    *gs = g_out * (a * p - na * q) * g * (1.0f - g);
    *gp = g_out * g * a;
    *gq = g_out * (1.0f - g) * na;
    acc += g_out * (g * p - (1.0f - g) * q);
  };
  int lo, hi;
  if (VEC) {
    chunk_range(hw / 4, &lo, &hi);
    const float4* b4 = reinterpret_cast<const float4*>(x + ob);
    const float4* v4 = reinterpret_cast<const float4*>(x + ov);
    const float4* s4 = reinterpret_cast<const float4*>(s2 + op);
    const float4* g4 = reinterpret_cast<const float4*>(go + op);
    float4* gb4 = reinterpret_cast<float4*>(gx + ob);
    float4* gv4 = reinterpret_cast<float4*>(gx + ov);
    float4* gs4 = reinterpret_cast<float4*>(gs2 + op);
    for (int i = lo + threadIdx.x; i < hi; i += kBlock) {
      float4 p = b4[i], q = v4[i], s = s4[i], g = g4[i], rp, rq, rs;
      f(p.x, q.x, s.x, g.x, &rp.x, &rq.x, &rs.x);
      f(p.y, q.y, s.y, g.y, &rp.y, &rq.y, &rs.y);
      f(p.z, q.z, s.z, g.z, &rp.z, &rq.z, &rs.z);
      f(p.w, q.w, s.w, g.w, &rp.w, &rq.w, &rs.w);
      gb4[i] = rp; gv4[i] = rq; gs4[i] = rs;
    }
  } else {
    chunk_range(hw, &lo, &hi);
    for (int i = lo + threadIdx.x; i < hi; i += kBlock) {
      float rp, rq, rs;
      f(x[ob + i], x[ov + i], s2[op + i], go[op + i], &rp, &rq, &rs);
      gx[ob + i] = rp; gx[ov + i] = rq; gs2[op + i] = rs;
    }
  }
  float tot = block_sum(acc, sm);
  if (threadIdx.x == 0) unsafeAtomicAdd(ga1 + plane, tot);
}

// u = a*xb + (1-a)*xv:  gx_b += a*gu, gx_v += (1-a)*gu, ga1 += sum gu*(xb - xv)
template <bool VEC>
__global__ __launch_bounds__(kBlock) void blend1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ a1,
                                                            const float* __restrict__ gu, float* __restrict__ gx,
                                                            float* __restrict__ ga1, int c, int hw) {
  __shared__ float sm[kBlock / DHD_WAVE];
  const int plane = blockIdx.y;
  const int b = plane / c, ch = plane % c;
  const float a = a1[plane], na = 1.0f - a;
  const size_t ob = ((size_t)b * 2 * c + ch) * hw, ov = ob + (size_t)c * hw, op = (size_t)plane * hw;
  float acc = 0.f;
  int lo, hi;
  if (VEC) {
    chunk_range(hw / 4, &lo, &hi);
    const float4* b4 = reinterpret_cast<const float4*>(x + ob);
    const float4* v4 = reinterpret_cast<const float4*>(x + ov);
    const float4* g4 = reinterpret_cast<const float4*>(gu + op);
    float4* gb4 = reinterpret_cast<float4*>(gx + ob);
    float4* gv4 = reinterpret_cast<float4*>(gx + ov);
    for (int i = lo + threadIdx.x; i < hi; i += kBlock) {
      float4 p = b4[i], q = v4[i], g = g4[i], rb = gb4[i], rv = gv4[i];
      rb.x += a * g.x; rb.y += a * g.y; rb.z += a * g.z; rb.w += a * g.w;
      rv.x += na * g.x; rv.y += na * g.y; rv.z += na * g.z; rv.w += na * g.w;
      acc += g.x * (p.x - q.x) + g.y * (p.y - q.y) + g.z * (p.z - q.z) + g.w * (p.w - q.w);
      gb4[i] = rb; gv4[i] = rv;
    }
  } else {
    chunk_range(hw, &lo, &hi);
    for (int i = lo + threadIdx.x; i < hi; i += kBlock) {
      float g = gu[op + i];
      gx[ob + i] += a * g;
      gx[ov + i] += na * g;
      acc += g * (x[ob + i] - x[ov + i]);
    }
  }
  float tot = block_sum(acc, sm);
  if (threadIdx.x == 0) unsafeAtomicAdd(ga1 + plane, tot);
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void mean_bwd_kernel(const float* __restrict__ gs, float* __restrict__ gx, int hw) {
  const size_t plane = blockIdx.y;
  const float add = gs[plane] / (float)hw;
  float* p = gx + plane * hw;
  int lo, hi;
  if (VEC) {
    chunk_range(hw / 4, &lo, &hi);
    float4* p4 = reinterpret_cast<float4*>(p);
    for (int i = lo + threadIdx.x; i < hi; i += kBlock) {
      float4 v = p4[i];
      v.x += add; v.y += add; v.z += add; v.w += add;
      p4[i] = v;
    }
  } else {
    chunk_range(hw, &lo, &hi);
    for (int i = lo + threadIdx.x; i < hi; i += kBlock) p[i] += add;
  }
}

inline bool bad(int b, int c, int hw) { return b <= 0 || c <= 0 || hw <= 0; }

}  // namespace

#define DHD_SFA_LAUNCH(kernel, grid, stream, ...)                                                   \
  do {                                                                                              \
    if ((hw & 3) == 0)                                                                              \
      hipLaunchKernelGGL(kernel<true>, grid, dim3(kBlock), 0, dhd_stream(stream), __VA_ARGS__);     \
    else                                                                                            \
      hipLaunchKernelGGL(kernel<false>, grid, dim3(kBlock), 0, dhd_stream(stream), __VA_ARGS__);    \
    DHD_LAUNCH_CHECK();                                                                             \
  } while (0)

extern "C" {

int dhd_sfa_channel_mean(const float* x, float* s, int b, int c2, int hw, void* stream) {
  if (!x || !s || bad(b, c2, hw)) return DHD_EINVAL;
  DHD_SFA_LAUNCH(channel_mean_kernel, dim3(b * c2), stream, x, s, hw);
  return DHD_OK;
}

int dhd_sfa_blend1(const float* x, const float* a1, float* u, int b, int c, int hw, void* stream) {
  if (!x || !a1 || !u || bad(b, c, hw)) return DHD_EINVAL;
  DHD_SFA_LAUNCH(blend1_kernel, dim3(kChunksPerPlane, b * c), stream, x, a1, u, c, hw);
  return DHD_OK;
}

int dhd_sfa_blend2(const float* x, const float* a1, const float* s2, float* out, int b, int c, int hw, void* stream) {
  if (!x || !a1 || !s2 || !out || bad(b, c, hw)) return DHD_EINVAL;
  DHD_SFA_LAUNCH(blend2_kernel, dim3(kChunksPerPlane, b * c), stream, x, a1, s2, out, c, hw);
  return DHD_OK;
}

int dhd_sfa_blend2_backward(const float* x, const float* a1, const float* s2, const float* go, float* gx, float* gs2,
                            float* ga1, int b, int c, int hw, void* stream) {
  if (!x || !a1 || !s2 || !go || !gx || !gs2 || !ga1 || bad(b, c, hw)) return DHD_EINVAL;
  DHD_HIP(hipMemsetAsync(ga1, 0, (size_t)b * c * 4, dhd_stream(stream)));
  DHD_SFA_LAUNCH(blend2_bwd_kernel, dim3(kChunksPerPlane, b * c), stream, x, a1, s2, go, gx, gs2, ga1, c, hw);
  return DHD_OK;
}

int dhd_sfa_blend1_backward(const float* x, const float* a1, const float* gu, float* gx, float* ga1, int b, int c, int hw,
                            void* stream) {
  if (!x || !a1 || !gu || !gx || !ga1 || bad(b, c, hw)) return DHD_EINVAL;
  DHD_SFA_LAUNCH(blend1_bwd_kernel, dim3(kChunksPerPlane, b * c), stream, x, a1, gu, gx, ga1, c, hw);
  return DHD_OK;
}

int dhd_sfa_mean_backward(const float* gs, float* gx, int b, int c2, int hw, void* stream) {
  if (!gs || !gx || bad(b, c2, hw)) return DHD_EINVAL;
  DHD_SFA_LAUNCH(mean_bwd_kernel, dim3(kChunksPerPlane, b * c2), stream, gs, gx, hw);
  return DHD_OK;
}

}  // extern "C"
