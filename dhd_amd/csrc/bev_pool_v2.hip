// Operator-level drop-in for the reference's bev_pool_v2 extension on gfx950.
//
// Replaces ops/bev_pool_v2/src/bev_pool_cuda.cu (kernels :21-50 and :69-123) behind the same
// two entry points as ops/bev_pool_v2/src/bev_pool.cpp:30-39,74-85.  Semantics are identical
// (channel-last (B,Dz,Dy,Dx,C) output, caller pre-zeroes, intervals given by the caller); the
// parallelisation is re-done for 64-wide wavefronts:
//   forward : one wave per interval, lanes = channels (sub-slots of lanes share a wave when
//             C < 64); the three index words of a point are loaded once per wave, 64 points at a
//             time, and broadcast across lanes instead of once per channel-thread (:41-46);
//   backward: one wave per pixel interval instead of one THREAD per pixel (:87-122), so the
//             depth-gradient dot product is a wave reduction and the feature gradient a
//             per-lane accumulator; out_grad rows are read as coalesced 4*C-byte segments.
#include "common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / DHD_WAVE;

__global__ __launch_bounds__(kBlock) void bev_pool_v2_fwd_kernel(int c, int n_intervals, const float* __restrict__ depth,
                                                                  const float* __restrict__ feat,
                                                                  const int* __restrict__ ranks_depth,
                                                                  const int* __restrict__ ranks_feat,
                                                                  const int* __restrict__ ranks_bev,
                                                                  const int* __restrict__ interval_starts,
                                                                  const int* __restrict__ interval_lengths,
                                                                  float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int iv = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (iv >= n_intervals) return;
  const int start = interval_starts[iv];
  const int len = interval_lengths[iv];
  const int CL = c >= DHD_WAVE ? DHD_WAVE : next_pow2(c);
  const int nsub = DHD_WAVE / CL;
  const int cl = lane % CL, sub = lane / CL;
  const int vox = ranks_bev[start];
  for (int c0 = 0; c0 < c; c0 += DHD_WAVE) {
    const int ch = c0 + cl;
    const bool ok = ch < c;
    float acc = 0.f;
    for (int s0 = 0; s0 < len; s0 += DHD_WAVE) {
      const int nb = min(DHD_WAVE, len - s0);
      int rf = 0;
      float dv = 0.f;
      if (lane < nb) {
        rf = ranks_feat[start + s0 + lane];
        dv = depth[ranks_depth[start + s0 + lane]];
      }
      const int steps = (nb + nsub - 1) / nsub;
      // eight independent row gathers in flight per lane (a dependent one-at-a-time loop left the kernel at the
      // latency of ~8 serial L2 round trips per interval: 118 -> see DESIGN.md)
      for (int k0 = 0; k0 < steps; k0 += 8) {
        float f[8], d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = (k0 + u) * nsub + sub;
          const bool live = i < nb;
          const int q = __shfl(rf, live ? i : 0, DHD_WAVE);
          d[u] = __shfl(dv, live ? i : 0, DHD_WAVE);
          f[u] = (live && ok) ? feat[(size_t)q * c + ch] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(f[u], d[u], acc);
      }
    }
    for (int m = CL; m < DHD_WAVE; m <<= 1) acc += __shfl_xor(acc, m, DHD_WAVE);
    if (sub == 0 && ok) out[(size_t)vox * c + ch] = acc;
  }
}

__global__ __launch_bounds__(kBlock) void bev_pool_v2_bwd_kernel(int c, int n_intervals, const float* __restrict__ out_grad,
                                                                  const float* __restrict__ depth,
                                                                  const float* __restrict__ feat,
                                                                  const int* __restrict__ ranks_depth,
                                                                  const int* __restrict__ ranks_feat,
                                                                  const int* __restrict__ ranks_bev,
                                                                  const int* __restrict__ interval_starts,
                                                                  const int* __restrict__ interval_lengths,
                                                                  float* __restrict__ depth_grad,
                                                                  float* __restrict__ feat_grad) {
  const int lane = threadIdx.x & 63;
  const int iv = blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (iv >= n_intervals) return;
  const int start = interval_starts[iv];
  const int len = interval_lengths[iv];
  const int pix = ranks_feat[start];  // every point of the interval shares the pixel (bev_pool.py:47-57)
  const int n_cc = (c + DHD_WAVE - 1) / DHD_WAVE;
  for (int s0 = 0; s0 < len; s0 += DHD_WAVE) {
    const int nb = min(DHD_WAVE, len - s0);
    int rb = 0, rd = 0;
    if (lane < nb) {
      rb = ranks_bev[start + s0 + lane];
      rd = ranks_depth[start + s0 + lane];
    }
    float mine = 0.f;
    for (int i = 0; i < nb; ++i) {
      const int vox = __shfl(rb, i, DHD_WAVE);
      float part = 0.f;
      for (int cc = 0; cc < n_cc; ++cc) {
        const int ch = cc * DHD_WAVE + lane;
        if (ch < c) part = fmaf(out_grad[(size_t)vox * c + ch], feat[(size_t)pix * c + ch], part);
      }
      float tot = group_sum(part, DHD_WAVE);
      if (lane == i) mine = tot;
    }
    if (lane < nb) depth_grad[rd] = mine;  // plain store, one writer per point (:104-106)
  }
  for (int cc = 0; cc < n_cc; ++cc) {
    const int ch = cc * DHD_WAVE + lane;
    float acc = 0.f;
    for (int s0 = 0; s0 < len; s0 += DHD_WAVE) {
      const int nb = min(DHD_WAVE, len - s0);
      int rb = 0;
      float dv = 0.f;
      if (lane < nb) {
        rb = ranks_bev[start + s0 + lane];
        dv = depth[ranks_depth[start + s0 + lane]];
      }
      for (int i = 0; i < nb; ++i) {
        const int vox = __shfl(rb, i, DHD_WAVE);
        const float d = __shfl(dv, i, DHD_WAVE);
        if (ch < c) acc = fmaf(out_grad[(size_t)vox * c + ch], d, acc);
      }
    }
    if (ch < c) feat_grad[(size_t)pix * c + ch] = acc;  // one writer per pixel (:120-121)
  }
}

// Backward, single pass, for C <= 64 * NCC: every out_grad row of the pixel's points is gathered ONCE (eight rows in
// flight), used for both gradients; the per-point dot product is a DPP wave reduction (no LDS crossbar).
template <int NCC>
__global__ __launch_bounds__(kBlock) void bev_pool_v2_bwd_fast_kernel(int c, int n_intervals, const float* __restrict__ out_grad,
                                                                       const float* __restrict__ depth,
                                                                       const float* __restrict__ feat,
                                                                       const int* __restrict__ ranks_depth,
                                                                       const int* __restrict__ ranks_feat,
                                                                       const int* __restrict__ ranks_bev,
                                                                       const int* __restrict__ interval_starts,
                                                                       const int* __restrict__ interval_lengths,
                                                                       float* __restrict__ depth_grad,
                                                                       float* __restrict__ feat_grad) {
  const int lane = threadIdx.x & 63;
  const int iv = blockIdx.x * kWaves + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (iv >= n_intervals) return;
  const int start = __builtin_amdgcn_readfirstlane(interval_starts[iv]);
  const int len = __builtin_amdgcn_readfirstlane(interval_lengths[iv]);
  const int pix = __builtin_amdgcn_readfirstlane(ranks_feat[start]);
  float fv[NCC], facc[NCC];
#pragma unroll
  for (int cc = 0; cc < NCC; ++cc) {
    const int ch = cc * DHD_WAVE + lane;
    fv[cc] = ch < c ? feat[(size_t)pix * c + ch] : 0.f;
    facc[cc] = 0.f;
  }
  for (int s0 = 0; s0 < len; s0 += DHD_WAVE) {
    const int nb = min(DHD_WAVE, len - s0);
    int rb = 0, rd = 0;
    float dv = 0.f;
    if (lane < nb) {
      rb = ranks_bev[start + s0 + lane];
      rd = ranks_depth[start + s0 + lane];
      dv = depth[rd];
    }
    float mine = 0.f;
    for (int i0 = 0; i0 < nb; i0 += 8) {
      float g[8][NCC], d[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = min(i0 + u, nb - 1);     // uniform; past the end: a harmless re-read of the last point
        const int vox = __builtin_amdgcn_readlane(rb, i);
        d[u] = i0 + u < nb ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dv), i)) : 0.f;
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) {
          const int ch = cc * DHD_WAVE + lane;
          g[u][cc] = ch < c ? out_grad[(size_t)vox * c + ch] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float part = 0.f;
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) {
          part = fmaf(g[u][cc], fv[cc], part);
          facc[cc] = fmaf(g[u][cc], d[u], facc[cc]);
        }
        const float tot = wave_sum_bcast(part);
        if (lane == i0 + u) mine = tot;
      }
    }
    if (lane < nb) depth_grad[rd] = mine;
  }
#pragma unroll
  for (int cc = 0; cc < NCC; ++cc) {
    const int ch = cc * DHD_WAVE + lane;
    if (ch < c) feat_grad[(size_t)pix * c + ch] = facc[cc];
  }
}

}  // namespace

extern "C" {

int dhd_bev_pool_v2_forward(const float* depth, const float* feat, float* out, const int32_t* ranks_depth,
                            const int32_t* ranks_feat, const int32_t* ranks_bev, const int32_t* interval_lengths,
                            const int32_t* interval_starts, int c, int n_intervals, void* stream) {
  if (c <= 0 || n_intervals < 0) return DHD_EINVAL;
  if (n_intervals == 0) return DHD_OK;
  if (!depth || !feat || !out || !ranks_depth || !ranks_feat || !ranks_bev || !interval_lengths || !interval_starts)
    return DHD_EINVAL;
  hipLaunchKernelGGL(bev_pool_v2_fwd_kernel, dim3(dhd_cdiv(n_intervals, kWaves)), dim3(kBlock), 0, dhd_stream(stream), c,
                     n_intervals, depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths, out);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

int dhd_bev_pool_v2_backward(const float* out_grad, float* depth_grad, float* feat_grad, const float* depth,
                             const float* feat, const int32_t* ranks_depth, const int32_t* ranks_feat,
                             const int32_t* ranks_bev, const int32_t* interval_lengths_bp,
                             const int32_t* interval_starts_bp, int c, int n_intervals_bp, void* stream) {
  if (c <= 0 || n_intervals_bp < 0) return DHD_EINVAL;
  if (n_intervals_bp == 0) return DHD_OK;
  if (!out_grad || !depth_grad || !feat_grad || !depth || !feat || !ranks_depth || !ranks_feat || !ranks_bev ||
      !interval_lengths_bp || !interval_starts_bp)
    return DHD_EINVAL;
  const dim3 grid(dhd_cdiv(n_intervals_bp, kWaves));
#define DHD_BWD(KERN)                                                                                                    \
  hipLaunchKernelGGL(KERN, grid, dim3(kBlock), 0, dhd_stream(stream), c, n_intervals_bp, out_grad, depth, feat, ranks_depth, \
                     ranks_feat, ranks_bev, interval_starts_bp, interval_lengths_bp, depth_grad, feat_grad)
  if (c <= 64) DHD_BWD(bev_pool_v2_bwd_fast_kernel<1>);
  else if (c <= 128) DHD_BWD(bev_pool_v2_bwd_fast_kernel<2>);
  else DHD_BWD(bev_pool_v2_bwd_kernel);
#undef DHD_BWD
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

}  // extern "C"
