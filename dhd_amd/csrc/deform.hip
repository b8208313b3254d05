// Deformable-convolution sampling (DCN v1, stride 1, one deformable group) for gfx950: the "exotic" op of
// HeightNet / DepthNet (models/necks/depthnet.py:225-236, :466-477 build mmcv's `DCN`; mmcv-full 1.5.3
// ops/deform_conv: deformable_im2col / col2im / col2im_coord).  The sampled columns feed an ordinary GEMM
// with the layer's weight, which stays on the library path.
//
//   col[b, c*K + t, p] = bilinear(x[b, c], y_p + ky*dil - pad + off[b, 2t, p], x_p + kx*dil - pad + off[b, 2t+1, p])
// with K = k*k taps t = ky*k + kx, zero outside the image (corners outside contribute 0).
//
//   deform_im2col      thread = (b, t, p) x a chunk of channels: corner indices / weights once, then a channel loop
//   deform_col2im      (ABI <= 4 entry point, no workspace) block = (b, chunk of channel planes): scatter into LDS planes
//                      with LDS float atomics, then plain stores.  713 us at the DHD-S HeightNet size (24 x 256 x 16x44): bound
//                      by the LDS atomic rate, 155 M read-modify-writes whose neighbouring lanes hit the same cells
//   deform_tap_sort +  (ABI 5, dhd_deform_col2im_t) the GATHER form: one block per image groups the <= 4 k k hw bilinear
//   deform_col2im_gather  corner entries (source (t, p), weight) by the cell they land in (LDS counting sort: the taps are
//                      shared by all channels -- one deformable group); then a block per (image, chunk of channels) stages its
//                      dcol rows in LDS with coalesced 16-byte loads and every cell sums its own entry list from LDS --
//                      plain LDS reads, one writer per cell, no atomics
//   deform_col2offset  thread = (b, t, p): channel loop of the coordinate gradients
// The column matrix may be float32, float16 or bfloat16 (`col_dtype`): under autocast the GEMM behind it runs in half, so
// im2col emits half and the two backward kernels read the half gradient (155 MB -> 78 MB per pass at the DHD-S size, and
// no cast kernels in between); x is read, and its gradient written, as float32 or in the column type; the offsets and their
// gradient are float32, arithmetic is float32.
#include "common.h"

namespace {

constexpr int kBlock = 256;

typedef __bf16 bf16_t;
// element (image, channel, cell) of x / dx (dense NCHW: lanes = consecutive cells, so the corner gathers of a wave stay within a
// few lines per channel plane; a channels_last x measured 5-7x slower in these kernels and is converted by the caller instead)
__device__ __forceinline__ size_t x_at(int b, int ch, int i, int c, int hw) { return ((size_t)b * c + ch) * hw + i; }
template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }   // round to nearest even

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

template <typename TC, typename TX>
__global__ __launch_bounds__(kBlock) void deform_im2col(const TX* __restrict__ x, const float* __restrict__ off,
                                                        TC* __restrict__ col, int c, int h, int w, int k, int pad, int dil,
                                                        int c_chunk) {
  const int hw = h * w, kk = k * k;
  const int i = blockIdx.x * kBlock + threadIdx.x;  // (t, p)
  if (i >= kk * hw) return;
  const int b = blockIdx.z, t = i / hw, p = i % hw;
  const Tap tp = tap_of(off + (size_t)b * 2 * kk * hw, t, p, h, w, k, pad, dil);
  const int c0 = blockIdx.y * c_chunk, c1 = min(c, c0 + c_chunk);
  TC* cb = col + (((size_t)b * c + c0) * kk + t) * hw + p;
  for (int ch = c0; ch < c1; ++ch, cb += (size_t)kk * hw) {
    float v = 0.f;
    if (tp.v00) v = fmaf(tp.w00, to_f32(x[x_at(b, ch, tp.i00, c, hw)]), v);
    if (tp.v01) v = fmaf(tp.w01, to_f32(x[x_at(b, ch, tp.i01, c, hw)]), v);
    if (tp.v10) v = fmaf(tp.w10, to_f32(x[x_at(b, ch, tp.i10, c, hw)]), v);
    if (tp.v11) v = fmaf(tp.w11, to_f32(x[x_at(b, ch, tp.i11, c, hw)]), v);
    *cb = from_f32<TC>(v);
  }
}

// ---- gather form of col2im ---------------------------------------------------------------------------------------
// Workspace per image: cell_start[hw + 1] (int32), then up to 4 k k hw entries {source = t hw + p, weight bits}.
constexpr int kSortBlock = 1024;

__global__ __launch_bounds__(kSortBlock) void deform_tap_sort(const float* __restrict__ off, int* __restrict__ cell_start,
                                                              uint2* __restrict__ entries, int h, int w, int k, int pad, int dil) {
  extern __shared__ int sm[];           // cnt[hw] | cur[hw] | part[kSortBlock]
  const int hw = h * w, kk = k * k, b = blockIdx.x, tid = threadIdx.x;
  int* cnt = sm;
  int* cur = sm + hw;
  int* part = sm + 2 * hw;
  for (int i = tid; i < hw; i += kSortBlock) cnt[i] = 0;
  __syncthreads();
  const float* off_b = off + (size_t)b * 2 * kk * hw;
  for (int i = tid; i < kk * hw; i += kSortBlock) {
    const Tap tp = tap_of(off_b, i / hw, i % hw, h, w, k, pad, dil);
    if (tp.v00) atomicAdd(cnt + tp.i00, 1);
    if (tp.v01) atomicAdd(cnt + tp.i01, 1);
    if (tp.v10) atomicAdd(cnt + tp.i10, 1);
    if (tp.v11) atomicAdd(cnt + tp.i11, 1);
  }
  __syncthreads();
  // exclusive scan of cnt: a contiguous chunk of cells per thread, then a scan of the chunk sums by one thread per wave + wave 0
  const int chunk = (hw + kSortBlock - 1) / kSortBlock, lo = min(hw, tid * chunk), hi = min(hw, lo + chunk);
  int sum = 0;
  for (int i = lo; i < hi; ++i) sum += cnt[i];
  part[tid] = sum;
  __syncthreads();
  if (tid < DHD_WAVE) {                 // 16 partials per lane, wave-inclusive scan, back to exclusive prefixes
    int local[kSortBlock / DHD_WAVE], run = 0;
#pragma unroll
    for (int q = 0; q < kSortBlock / DHD_WAVE; ++q) { local[q] = run; run += part[tid * (kSortBlock / DHD_WAVE) + q]; }
    int incl = run;
    for (int m = 1; m < DHD_WAVE; m <<= 1) {
      const int o = __shfl_up(incl, m, DHD_WAVE);
      if (tid >= m) incl += o;
    }
    const int base = incl - run;
#pragma unroll
    for (int q = 0; q < kSortBlock / DHD_WAVE; ++q) part[tid * (kSortBlock / DHD_WAVE) + q] = base + local[q];
  }
  __syncthreads();
  int* start_b = cell_start + (size_t)b * (hw + 1);
  int run = part[tid];
  for (int i = lo; i < hi; ++i) {
    cur[i] = run;
    start_b[i] = run;
    run += cnt[i];
  }
  if (hi == hw && lo < hw) start_b[hw] = run;
  if (hw == 0 && tid == 0) start_b[0] = 0;
  __syncthreads();
  uint2* ent_b = entries + (size_t)b * 4 * kk * hw;
  for (int i = tid; i < kk * hw; i += kSortBlock) {
    const Tap tp = tap_of(off_b, i / hw, i % hw, h, w, k, pad, dil);
    if (tp.v00) ent_b[atomicAdd(cur + tp.i00, 1)] = make_uint2((unsigned)i, __float_as_uint(tp.w00));
    if (tp.v01) ent_b[atomicAdd(cur + tp.i01, 1)] = make_uint2((unsigned)i, __float_as_uint(tp.w01));
    if (tp.v10) ent_b[atomicAdd(cur + tp.i10, 1)] = make_uint2((unsigned)i, __float_as_uint(tp.w10));
    if (tp.v11) ent_b[atomicAdd(cur + tp.i11, 1)] = make_uint2((unsigned)i, __float_as_uint(tp.w11));
  }
}

// dx[b, c0 .. c0+NC) for one image and NC channels whose dcol rows (NC x kk x hw, contiguous) sit in LDS
template <typename TC, int NC, typename TX>
__global__ __launch_bounds__(kBlock) void deform_col2im_gather(const TC* __restrict__ dcol, const int* __restrict__ cell_start,
                                                               const uint2* __restrict__ entries, TX* __restrict__ dx, int c,
                                                               int hw, int kk, int vec_ok) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
  TC* tile = reinterpret_cast<TC*>(tile_raw);            // [NC][kk * hw]
  const int b = blockIdx.y, c0 = blockIdx.x * NC, nc = min(NC, c - c0), khw = kk * hw;
  const TC* src = dcol + ((size_t)b * c + c0) * khw;
  const int n = nc * khw;
  if (vec_ok) {
    constexpr int PER = 16 / (int)sizeof(TC);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4* s4 = reinterpret_cast<const u32x4*>(src);
    u32x4* t4 = reinterpret_cast<u32x4*>(tile);
    for (int i = threadIdx.x; i < n / PER; i += kBlock) t4[i] = __builtin_nontemporal_load(s4 + i);   // read once
  } else {
    for (int i = threadIdx.x; i < n; i += kBlock) tile[i] = src[i];
  }
  __syncthreads();
  const int* start_b = cell_start + (size_t)b * (hw + 1);
  const uint2* ent_b = entries + (size_t)b * 4 * khw;
  for (int cell = threadIdx.x; cell < hw; cell += kBlock) {
    const int s = start_b[cell], e = start_b[cell + 1];
    float acc[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) acc[j] = 0.f;
    for (int i = s; i < e; ++i) {
      const uint2 en = ent_b[i];
      const float wgt = __uint_as_float(en.y);
#pragma unroll
      for (int j = 0; j < NC; ++j)
        if (j < nc) acc[j] = fmaf(wgt, to_f32(tile[j * khw + (int)en.x]), acc[j]);
    }
#pragma unroll
    for (int j = 0; j < NC; ++j)
      if (j < nc) dx[x_at(b, c0 + j, cell, c, hw)] = from_f32<TX>(acc[j]);
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
template <typename TC, typename TX = float>
__global__ __launch_bounds__(kBlock) void deform_col2offset(const TC* __restrict__ dcol, const TX* __restrict__ x,
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
    const TC* g = dcol + ((size_t)b * c * kk + t) * hw + p;
    for (int ch = 0; ch < c; ++ch, g += (size_t)kk * hw) {
      const float v00 = tp.v00 ? to_f32(x[x_at(b, ch, tp.i00, c, hw)]) : 0.f, v01 = tp.v01 ? to_f32(x[x_at(b, ch, tp.i01, c, hw)]) : 0.f;
      const float v10 = tp.v10 ? to_f32(x[x_at(b, ch, tp.i10, c, hw)]) : 0.f, v11 = tp.v11 ? to_f32(x[x_at(b, ch, tp.i11, c, hw)]) : 0.f;
      const float gv = to_f32(*g);
      gy = fmaf(gv, (v10 - v00) * hx + (v11 - v01) * tp.lx, gy);
      gx = fmaf(gv, (v01 - v00) * hy + (v11 - v10) * tp.ly, gx);
    }
  }
  float* d = doff + (size_t)b * 2 * kk * hw;
  d[(size_t)(2 * t) * hw + p] = gy;
  d[(size_t)(2 * t + 1) * hw + p] = gx;
}

// LDS the gather kernel may use per block: several blocks per CU so that one block's load phase overlaps another's gather phase
static constexpr size_t kGatherLds = 64 * 1024;
static constexpr size_t kSortLdsMax = 64 * 1024;

static size_t deform_ws_offsets(int b, int h, int w, int k, size_t* entries_at) {
  const size_t hw = (size_t)h * w, starts = ((size_t)b * (hw + 1) * sizeof(int) + 255) / 256 * 256;
  if (entries_at) *entries_at = starts;
  return starts + (size_t)b * 4 * k * k * hw * sizeof(uint2);
}

template <typename TC, typename TX>
static void launch_gather(const TC* dcol, const int* starts, const uint2* entries, TX* dx, int b, int c, int hw, int kk,
                          hipStream_t st) {
  const size_t row = (size_t)kk * hw * sizeof(TC);
  const int vec_ok = (row % 16 == 0) && ((uintptr_t)dcol % 16 == 0);
  // channels per block: as many as keep the tile within kGatherLds (>= 1: a single row may take up to 144 KiB)
  int nc = (int)(kGatherLds / row);
  nc = nc >= 8 ? 8 : nc >= 4 ? 4 : nc >= 2 ? 2 : 1;
  if (nc > c) nc = c >= 4 ? 4 : c >= 2 ? 2 : 1;
  const dim3 grid(dhd_cdiv(c, nc), b);
  const size_t lds = (size_t)nc * row;
#define DHD_GATHER(NC)                                                                                                        \
  do {                                                                                                                        \
    if (lds > 48 * 1024)                                                                                                      \
      (void)hipFuncSetAttribute((const void*)deform_col2im_gather<TC, NC, TX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((deform_col2im_gather<TC, NC, TX>), grid, dim3(kBlock), lds, st, dcol, starts, entries, dx, c, hw, kk, vec_ok); \
  } while (0)
  if (nc == 8) DHD_GATHER(8);
  else if (nc == 4) DHD_GATHER(4);
  else if (nc == 2) DHD_GATHER(2);
  else DHD_GATHER(1);
#undef DHD_GATHER
}

// x may be float32 (any col_dtype) or of the column type itself; dense NCHW
static bool x_combo_ok(int x_dtype, int col_dtype) { return x_dtype == DHD_F32 || x_dtype == col_dtype; }

template <typename TC, typename TX>
static void launch_im2col(const void* x, const float* offset, void* col, dim3 grid, hipStream_t st, int c, int h, int w, int k, int pad,
                          int dil, int c_chunk) {
  hipLaunchKernelGGL((deform_im2col<TC, TX>), grid, dim3(kBlock), 0, st, (const TX*)x, offset, (TC*)col, c, h, w, k, pad, dil, c_chunk);
}

template <typename TC, typename TX>
static void launch_col2im(const void* dcol, const void* x, const float* offset, void* dx, float* doffset, const int* starts,
                          const uint2* entries, int b, int c, int h, int w, int k, int pad, int dil, hipStream_t st) {
  const int hw = h * w, kk = k * k;
  launch_gather<TC, TX>((const TC*)dcol, starts, entries, (TX*)dx, b, c, hw, kk, st);
  hipLaunchKernelGGL((deform_col2offset<TC, TX>), dim3(dhd_cdiv((long)kk * hw, kBlock), b), dim3(kBlock), 0, st, (const TC*)dcol, (const TX*)x,
                     offset, doffset, c, h, w, k, pad, dil);
}

inline bool bad_shape(int b, int c, int h, int w, int k, int dil) { return b <= 0 || c <= 0 || h <= 0 || w <= 0 || k <= 0 || dil <= 0; }

}  // namespace

extern "C" {

int dhd_deform_im2col_t(const void* x, int x_dtype, const float* offset, void* col, int col_dtype, int b, int c, int h, int w,
                        int k, int pad, int dil, void* stream) {
  if (!x || !offset || !col || bad_shape(b, c, h, w, k, dil)) return DHD_EINVAL;
  if (col_dtype < DHD_F32 || col_dtype > DHD_BF16 || !x_combo_ok(x_dtype, col_dtype)) return DHD_EINVAL;
  if ((long)b * c * k * k * h * w >= (1L << 40)) return DHD_EUNSUPPORTED;
  const int c_chunk = c >= 32 ? 32 : c;
  const dim3 grid(dhd_cdiv((long)k * k * h * w, kBlock), dhd_cdiv(c, c_chunk), b);
  hipStream_t st = dhd_stream(stream);
  if (col_dtype == DHD_F32) launch_im2col<float, float>(x, offset, col, grid, st, c, h, w, k, pad, dil, c_chunk);
  else if (col_dtype == DHD_F16 && x_dtype == DHD_F32) launch_im2col<_Float16, float>(x, offset, col, grid, st, c, h, w, k, pad, dil, c_chunk);
  else if (col_dtype == DHD_F16) launch_im2col<_Float16, _Float16>(x, offset, col, grid, st, c, h, w, k, pad, dil, c_chunk);
  else if (x_dtype == DHD_F32) launch_im2col<bf16_t, float>(x, offset, col, grid, st, c, h, w, k, pad, dil, c_chunk);
  else launch_im2col<bf16_t, bf16_t>(x, offset, col, grid, st, c, h, w, k, pad, dil, c_chunk);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

int dhd_deform_im2col(const float* x, const float* offset, float* col, int b, int c, int h, int w, int k, int pad, int dil,
                      void* stream) {
  return dhd_deform_im2col_t(x, DHD_F32, offset, col, DHD_F32, b, c, h, w, k, pad, dil, stream);
}

size_t dhd_deform_col2im_workspace_bytes(int b, int h, int w, int k) {
  if (b <= 0 || h <= 0 || w <= 0 || k <= 0) return 0;
  return deform_ws_offsets(b, h, w, k, nullptr);
}

int dhd_deform_col2im_gather_supported(int col_dtype, int h, int w, int k) {
  if (h <= 0 || w <= 0 || k <= 0 || col_dtype < DHD_F32 || col_dtype > DHD_BF16) return 0;
  const size_t esz = col_dtype == DHD_F32 ? 4 : 2, khw = (size_t)k * k * h * w;
  return khw * esz <= 144 * 1024 && (2 * (size_t)h * w + kSortBlock) * sizeof(int) <= kSortLdsMax && khw < (1u << 30);
}

int dhd_deform_col2im_t(const void* dcol, int col_dtype, const void* x, int x_dtype, const float* offset, void* dx,
                        float* doffset, int b, int c, int h, int w, int k, int pad, int dil, void* workspace, size_t workspace_bytes,
                        void* stream) {
  if (!dcol || !x || !offset || !dx || !doffset || !workspace || bad_shape(b, c, h, w, k, dil)) return DHD_EINVAL;
  if (!dhd_deform_col2im_gather_supported(col_dtype, h, w, k)) return DHD_EUNSUPPORTED;
  if (!x_combo_ok(x_dtype, col_dtype)) return DHD_EINVAL;
  size_t entries_at = 0;
  if (workspace_bytes < deform_ws_offsets(b, h, w, k, &entries_at) || ((uintptr_t)workspace & 15)) return DHD_EINVAL;
  int* starts = (int*)workspace;
  uint2* entries = (uint2*)((char*)workspace + entries_at);
  hipStream_t st = dhd_stream(stream);
  hipLaunchKernelGGL(deform_tap_sort, dim3(b), dim3(kSortBlock), (2 * (size_t)h * w + kSortBlock) * sizeof(int), st, offset, starts, entries,
                     h, w, k, pad, dil);
  if (col_dtype == DHD_F32) launch_col2im<float, float>(dcol, x, offset, dx, doffset, starts, entries, b, c, h, w, k, pad, dil, st);
  else if (col_dtype == DHD_F16 && x_dtype == DHD_F32) launch_col2im<_Float16, float>(dcol, x, offset, dx, doffset, starts, entries, b, c, h, w, k, pad, dil, st);
  else if (col_dtype == DHD_F16) launch_col2im<_Float16, _Float16>(dcol, x, offset, dx, doffset, starts, entries, b, c, h, w, k, pad, dil, st);
  else if (x_dtype == DHD_F32) launch_col2im<bf16_t, float>(dcol, x, offset, dx, doffset, starts, entries, b, c, h, w, k, pad, dil, st);
  else launch_col2im<bf16_t, bf16_t>(dcol, x, offset, dx, doffset, starts, entries, b, c, h, w, k, pad, dil, st);
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
  hipLaunchKernelGGL(deform_col2offset<float>, dim3(dhd_cdiv((long)k * k * h * w, kBlock), b), dim3(kBlock), 0, st, dcol, x, offset, doffset, c,
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
// registers (lane = 4 channels), each hypothesis costs up to four 1-KB tap loads (a register cache keeps the taps of the
// previous hypothesis: see the kernel), a DPP wave reduction, and the softmax over the D hypotheses is done in the same wave.  No gradient (the reference wraps it in no_grad).
// ------------------------------------------------------------------------------------------------
namespace {

using f32x4_t = __attribute__((ext_vector_type(4))) float;
constexpr int kCvMaxD = 256;

// softmax over the hypotheses of -cost (hypothesis d in lane d % 64, register d / 64); registers without a hypothesis hold +3e38
__device__ __forceinline__ void cv_softmax_store(const float (&mine)[kCvMaxD / DHD_WAVE], float* __restrict__ out, int bn, int nd, int hw,
                                                 int yx, int lane) {
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


// Round 5.  The first version computed every hypothesis' sampling cell in all 64 lanes (a wave-uniform value) and reduced the
// cost over the wave once per hypothesis: ~100 VALU instructions per hypothesis, and with 128 channels half of the lanes idle --
// 7.5 ms at the DHD-L stereo size even when every tap hits the same position (experiments/cost_volume_bench.py), 9.6 ms in the
// model.  Now: (1) lane l computes the cell (four flat indices, -1 = outside, and four weights) of hypothesis d0 + l, 64
// hypotheses at once; the hypothesis loop only broadcasts them; (2) the taps of the previous hypothesis stay in registers keyed
// by their index (consecutive hypotheses walk along the epipolar line in sub-pixel steps for all but the nearest bins);
// (3) PAIR mode for c <= 128: the two halves of the wave take two hypotheses per step.  The arithmetic of a hypothesis and its
// order (tap order, the DPP reduction tree) are those of the first version.  DHD-L stereo size, c = 128: 7.5 -> 3.5 ms with every
// tap in one place, 9.1 -> 4.4 ms for a half-pixel walk; in the DHD-L step 9.6 -> 4.3 ms per call, DHD-M (c = 256) 3.25 -> 2.7.
struct CellRegs {
  int idx[4];
  float wt[4];
};

__device__ __forceinline__ CellRegs lane_cell(const float* __restrict__ gp, int d, int nd, int hw, int h, int w) {
  CellRegs r;
  float gx = 0.f, gy = 0.f;
  const bool have = d < nd;
  if (have) {
    const float2 g2 = *reinterpret_cast<const float2*>(gp + (size_t)d * hw * 2);
    gx = g2.x; gy = g2.y;
  }
  const float px = (gx + 1.0f) * 0.5f * (float)(w - 1);   // align_corners=True
  const float py = (gy + 1.0f) * 0.5f * (float)(h - 1);
  const Tap tp = make_tap(py, px, h, w);
  r.idx[0] = have && tp.v00 ? tp.i00 : -1;
  r.idx[1] = have && tp.v01 ? tp.i01 : -1;
  r.idx[2] = have && tp.v10 ? tp.i10 : -1;
  r.idx[3] = have && tp.v11 ? tp.i11 : -1;
  r.wt[0] = tp.w00; r.wt[1] = tp.w01; r.wt[2] = tp.w10; r.wt[3] = tp.w11;
  return r;
}

__device__ __forceinline__ float pick(const f32x4_t& v, int comp) {
  return comp == 0 ? v.x : comp == 1 ? v.y : comp == 2 ? v.z : v.w;
}

// one hypothesis per step, all lanes on its channels (c > 128): Q 16-byte channel groups per lane
template <int Q>
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
  f32x4_t cur[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int g = lane + 64 * q;
    cur[q] = g < c4 ? reinterpret_cast<const f32x4_t*>(curr + (size_t)pix * c)[g] : f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  const int flag_group = flag_channel >> 2, flag_comp = flag_channel & 3;
  float mine[kCvMaxD / DHD_WAVE];  // cost of hypothesis d lives in lane d % 64, register d / 64
#pragma unroll
  for (int q = 0; q < kCvMaxD / DHD_WAVE; ++q) mine[q] = 3.0e38f;
  const float* gp = grid + ((size_t)bn * nd * hw + yx) * 2;
  f32x4_t cache[4][Q];
  int cidx[4] = {-1, -1, -1, -1};
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int q = 0; q < Q; ++q) cache[k][q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int d0 = 0; d0 < nd; d0 += DHD_WAVE) {
    const CellRegs cell = lane_cell(gp, d0 + lane, nd, hw, h, w);
    const int nb = min(DHD_WAVE, nd - d0);
    for (int i = 0; i < nb; ++i) {
      int nidx[4];
      float wt[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        nidx[k] = __builtin_amdgcn_readlane(cell.idx[k], i);
        wt[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cell.wt[k]), i));
      }
      if (nidx[0] != cidx[0] || nidx[1] != cidx[1] || nidx[2] != cidx[2] || nidx[3] != cidx[3]) {
        f32x4_t nv[4][Q];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int want = nidx[k];
          if (want < 0) {
#pragma unroll
            for (int q = 0; q < Q; ++q) nv[k][q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          } else if (want == cidx[0]) {
#pragma unroll
            for (int q = 0; q < Q; ++q) nv[k][q] = cache[0][q];
          } else if (want == cidx[1]) {
#pragma unroll
            for (int q = 0; q < Q; ++q) nv[k][q] = cache[1][q];
          } else if (want == cidx[2]) {
#pragma unroll
            for (int q = 0; q < Q; ++q) nv[k][q] = cache[2][q];
          } else if (want == cidx[3]) {
#pragma unroll
            for (int q = 0; q < Q; ++q) nv[k][q] = cache[3][q];
          } else {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
              const int g = lane + 64 * q;
              nv[k][q] = g < c4 ? reinterpret_cast<const f32x4_t*>(pb + (size_t)want * c)[g] : f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          cidx[k] = nidx[k];
#pragma unroll
          for (int q = 0; q < Q; ++q) cache[k][q] = nv[k][q];
        }
      }
      float acc = 0.f, flag = 1.f;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int g = lane + 64 * q;
        if (g >= c4) break;
        f32x4_t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (cidx[k] >= 0) s += wt[k] * cache[k][q];
        const f32x4_t df = cur[q] - s;
        acc += (fabsf(df.x) + fabsf(df.y)) + (fabsf(df.z) + fabsf(df.w));
        if (g == flag_group) flag = pick(s, flag_comp);
      }
      float tot = wave_sum_bcast(acc);
      if (bias != 0.f) {
        const float fv = __shfl(flag, flag_group & 63, DHD_WAVE);
        if (fv == 0.f) tot += bias;
      }
      if (lane == i) mine[d0 >> 6] = tot;
    }
  }
  cv_softmax_store(mine, out, bn, nd, hw, yx, lane);
}

// PAIR mode (c <= 128: at most 32 channel groups): lanes 0-31 take hypothesis 2 j, lanes 32-63 hypothesis 2 j + 1 of a step
__global__ __launch_bounds__(kBlock) void stereo_cost_volume_pair_kernel(const float* __restrict__ prev, const float* __restrict__ curr,
                                                                         const float* __restrict__ grid, int c, int h, int w, int nd,
                                                                         float bias, int flag_channel, float* __restrict__ out,
                                                                         int n_pix) {
  const int lane = threadIdx.x & 63, half = lane >> 5, g = lane & 31;
  const int pix = blockIdx.x * (kBlock / DHD_WAVE) + (threadIdx.x >> 6);
  if (pix >= n_pix) return;
  const int hw = h * w;
  const int bn = pix / hw, yx = pix % hw;
  const int c4 = c >> 2;
  const bool live = g < c4;
  const float* pb = prev + (size_t)bn * hw * c + 4 * g;
  const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4_t cur = live ? reinterpret_cast<const f32x4_t*>(curr + (size_t)pix * c)[g] : zero;
  const int flag_group = flag_channel >> 2, flag_comp = flag_channel & 3;
  float mine[kCvMaxD / DHD_WAVE];
#pragma unroll
  for (int q = 0; q < kCvMaxD / DHD_WAVE; ++q) mine[q] = 3.0e38f;
  const float* gp = grid + ((size_t)bn * nd * hw + yx) * 2;
  f32x4_t cache[4] = {zero, zero, zero, zero};
  int cidx[4] = {-1, -1, -1, -1};
  for (int d0 = 0; d0 < nd; d0 += DHD_WAVE) {
    const CellRegs cell = lane_cell(gp, d0 + lane, nd, hw, h, w);
    const int nb = min(DHD_WAVE, nd - d0);
    for (int i = 0; i < nb; i += 2) {
      const int src = i + half;                 // the lane that holds this half's hypothesis (beyond nb: every index -1)
      f32x4_t s = zero;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int want = __shfl(cell.idx[k], src, DHD_WAVE);
        const float wk = __shfl(cell.wt[k], src, DHD_WAVE);
        if (want != cidx[k]) {                   // per lane: the halves walk their own lines
          cache[k] = (want >= 0 && live) ? *reinterpret_cast<const f32x4_t*>(pb + (size_t)want * c) : zero;
          cidx[k] = want;
        }
        if (want >= 0) s += wk * cache[k];
      }
      const f32x4_t df = cur - s;
      float acc = live ? (fabsf(df.x) + fabsf(df.y)) + (fabsf(df.z) + fabsf(df.w)) : 0.f;
      // the first five steps of wave_sum_bcast: lane 31 = sum of lanes 0-31, lane 63 = sum of lanes 32-63
      acc = dpp_add<0xB1, 0xf>(acc);
      acc = dpp_add<0x4E, 0xf>(acc);
      acc = dpp_add<0x114, 0xf>(acc);
      acc = dpp_add<0x118, 0xf>(acc);
      acc = dpp_add<0x142, 0xa>(acc);
      float ta = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 31));
      float tb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 63));
      if (bias != 0.f) {
        const float flag = pick(s, flag_comp);
        const float fa = __shfl(flag, flag_group & 31, DHD_WAVE);
        const float fb = __shfl(flag, 32 + (flag_group & 31), DHD_WAVE);
        if (fa == 0.f) ta += bias;
        if (fb == 0.f) tb += bias;
      }
      if (lane == i) mine[d0 >> 6] = ta;
      if (lane == i + 1 && i + 1 < nb) mine[d0 >> 6] = tb;
    }
  }
  cv_softmax_store(mine, out, bn, nd, hw, yx, lane);
}

}  // namespace

extern "C" int dhd_stereo_cost_volume(const float* prev_nhwc, const float* curr_nhwc, const float* grid, int bn, int c, int h, int w,
                                      int n_depth, float bias, int flag_channel, float* out, void* stream) {
  if (!prev_nhwc || !curr_nhwc || !grid || !out || bn <= 0 || c <= 0 || h <= 0 || w <= 0 || n_depth <= 0) return DHD_EINVAL;
  if ((c & 3) != 0 || c > 1024 || n_depth > kCvMaxD || flag_channel < 0 || flag_channel >= c) return DHD_EUNSUPPORTED;
  const long n_pix = (long)bn * h * w;
  if (n_pix >= (1L << 31) / n_depth) return DHD_EUNSUPPORTED;
  const dim3 blocks(dhd_cdiv(n_pix, kBlock / DHD_WAVE)), threads(kBlock);
  hipStream_t st = dhd_stream(stream);
  const int q = ((c >> 2) + DHD_WAVE - 1) / DHD_WAVE;   // 16-byte channel groups per lane
#define DHD_CV(K) hipLaunchKernelGGL(K, blocks, threads, 0, st, prev_nhwc, curr_nhwc, grid, c, h, w, n_depth, bias, flag_channel, out, (int)n_pix)
  if ((c >> 2) <= 32) DHD_CV(stereo_cost_volume_pair_kernel);
  else if (q <= 1) DHD_CV(stereo_cost_volume_kernel<1>);
  else if (q == 2) DHD_CV(stereo_cost_volume_kernel<2>);
  else if (q == 3) DHD_CV(stereo_cost_volume_kernel<3>);
  else DHD_CV(stereo_cost_volume_kernel<4>);
#undef DHD_CV
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}
