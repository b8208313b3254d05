// Exponential moving average of the model state (projects/mmdet3d_plugin/core/hook/ema.py:48-59):
// the reference walks the state dict with two eager ops per tensor (v *= d; v += (1-d)*m), i.e.
// ~1000 launches per training iteration for DHD-S.  Here: one launch over a chunk table that covers
// every float32 tensor of the state, each element read and written once.
#include "common.h"

namespace {

constexpr int kEmaBlock = 256;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The reference's arithmetic, rounding for rounding: fl(fl(v*d) + fl(omd*m)); the build has
// -ffp-contract=off, so neither product is fused into the add.
__device__ __forceinline__ float ema1(float v, float m, float d, float omd) { return v * d + omd * m; }

// decay_dev != nullptr: the two factors are read from device memory {decay, 1 - decay} (a launch captured into a HIP graph
// keeps its kernel arguments: the ramping decay of the reference must then come from memory the host refreshes per replay)
__global__ __launch_bounds__(kEmaBlock) void ema_update_kernel(const uint64_t* __restrict__ ema_addr,
                                                               const uint64_t* __restrict__ model_addr,
                                                               const int* __restrict__ len, float d, float omd,
                                                               const float* __restrict__ decay_dev) {
  if (decay_dev) { d = decay_dev[0]; omd = decay_dev[1]; }
  const int chunk = blockIdx.x;
  float* __restrict__ e = reinterpret_cast<float*>(ema_addr[chunk]);
  const float* __restrict__ m = reinterpret_cast<const float*>(model_addr[chunk]);
  const int n = len[chunk];
  // 16-byte path when both chunk starts are aligned (always, for whole tensors of the caching allocator)
  const bool vec = (((ema_addr[chunk] | model_addr[chunk]) & 15) == 0);
  const int n4 = vec ? n >> 2 : 0;
  f32x4* e4 = reinterpret_cast<f32x4*>(e);
  const f32x4* m4 = reinterpret_cast<const f32x4*>(m);
  int i = threadIdx.x;
  for (; i + 3 * kEmaBlock < n4; i += 4 * kEmaBlock) {
    f32x4 a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[q] = e4[i + q * kEmaBlock];
      b[q] = __builtin_nontemporal_load(m4 + i + q * kEmaBlock);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 r;
      r.x = ema1(a[q].x, b[q].x, d, omd);
      r.y = ema1(a[q].y, b[q].y, d, omd);
      r.z = ema1(a[q].z, b[q].z, d, omd);
      r.w = ema1(a[q].w, b[q].w, d, omd);
      e4[i + q * kEmaBlock] = r;
    }
  }
  for (; i < n4; i += kEmaBlock) {
    const f32x4 a = e4[i], b = __builtin_nontemporal_load(m4 + i);
    f32x4 r;
    r.x = ema1(a.x, b.x, d, omd);
    r.y = ema1(a.y, b.y, d, omd);
    r.z = ema1(a.z, b.z, d, omd);
    r.w = ema1(a.w, b.w, d, omd);
    e4[i] = r;
  }
  for (int j = 4 * n4 + threadIdx.x; j < n; j += kEmaBlock) e[j] = ema1(e[j], m[j], d, omd);
}

}  // namespace

extern "C" int dhd_ema_update(const uint64_t* ema_addr, const uint64_t* model_addr, const int* len, int n_chunks, float decay,
                              float one_minus_decay, void* stream) {
  if (n_chunks < 0) return DHD_EINVAL;
  if (n_chunks == 0) return DHD_OK;
  if (!ema_addr || !model_addr || !len) return DHD_EINVAL;
  hipLaunchKernelGGL(ema_update_kernel, dim3(n_chunks), dim3(kEmaBlock), 0, dhd_stream(stream), ema_addr, model_addr, len, decay,
                     one_minus_decay, static_cast<const float*>(nullptr));
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

extern "C" int dhd_ema_update_dev(const uint64_t* ema_addr, const uint64_t* model_addr, const int* len, int n_chunks,
                                  const float* decay_pair, void* stream) {
  if (n_chunks < 0) return DHD_EINVAL;
  if (n_chunks == 0) return DHD_OK;
  if (!ema_addr || !model_addr || !len || !decay_pair) return DHD_EINVAL;
  hipLaunchKernelGGL(ema_update_kernel, dim3(n_chunks), dim3(kEmaBlock), 0, dhd_stream(stream), ema_addr, model_addr, len, 0.f, 0.f,
                     decay_pair);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}
