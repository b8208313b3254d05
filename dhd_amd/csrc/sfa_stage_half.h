// Half-storage form of the SFA stage operator (dhd_sfa_weights.storage_dtype): element-wise passes and host orchestration.
// Included by sfa_stage.hip inside its unnamed namespace, after the float32 stage's helpers (block_sum, BnTail, FcGradJob,
// the layouts' align_up, DHD_LDS_ATTR_ONCE ...).  GEMM kernels: sfa_half.h.  Reference: models/necks/mix.py:37-59 under
// autocast (DHD-S.py:281).

// [lo, hi) in 8-element units of this block's share of a plane of hw elements (hw % 8 == 0)
__device__ __forceinline__ void chunk_range8(int hw, int* lo, int* hi) {
  const int n8 = hw >> 3, per = (n8 + kPlaneChunks - 1) / kPlaneChunks;
  *lo = blockIdx.x * per;
  *hi = min(n8, *lo + per);
}

struct PackJobH {
  const float* w[2];     // conv1, conv2
  u32x4* dst[4];         // conv1, conv2, conv1^T, conv2^T  (cuh_pack_weight)
  int c, blocks_each;
};

// channel means of x (block rows < n_planes) and, in rows of extra blocks, the four weight images of the call
template <class TS>
__global__ __launch_bounds__(kEwBlock) void plane_mean_pack_h_kernel(const TS* __restrict__ x, float* __restrict__ part, int hw,
                                                                     int n_planes, PackJobH job) {
  __shared__ float sm[kEwBlock / DHD_WAVE];
  if ((int)blockIdx.y >= n_planes) {
    const int pb = ((int)blockIdx.y - n_planes) * kPlaneChunks + (int)blockIdx.x;
    const int which = pb / job.blocks_each;
    if (which < 4)
      cuh_pack_weight<TS>(job.w[which & 1], which >> 1, job.dst[which], job.c, (pb % job.blocks_each) * kEwBlock + (int)threadIdx.x);
    return;
  }
  const size_t plane = blockIdx.y;
  const TS* p = x + plane * hw;
  int lo, hi;
  chunk_range8(hw, &lo, &hi);
  float a0 = 0.f, a1 = 0.f;
  int i = lo + threadIdx.x;
  for (; i + kEwBlock < hi; i += 2 * kEwBlock) {
    float v[8], w[8];
    ld8<TS>(p, i, v);
    ld8<TS>(p, i + kEwBlock, w);
    a0 += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    a1 += ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
  }
  if (i < hi) {
    float v[8];
    ld8<TS>(p, i, v);
    a0 += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  const float tot = block_sum(a0 + a1, sm);
  if (threadIdx.x == 0) part[plane * kPlaneChunks + blockIdx.x] = tot;
}

// out = g*(a*xb) + (1-g)*((1-a)*xv),  g = sigmoid(sc*y2 + sh)
template <class TS>
__global__ __launch_bounds__(kEwBlock) void blend2_bn_h_kernel(const TS* __restrict__ x, const float* __restrict__ a1,
                                                               const TS* __restrict__ y2, const float* __restrict__ scsh,
                                                               TS* __restrict__ out, int c, int hw) {
  const int plane = blockIdx.y, b = plane / c, ch = plane % c;
  const float a = a1[plane], na = 1.0f - a, sc = scsh[ch], sh = scsh[c + ch];
  const TS* xb = x + ((size_t)b * 2 * c + ch) * hw;
  const TS* xv = x + ((size_t)b * 2 * c + c + ch) * hw;
  const TS* yp = y2 + (size_t)plane * hw;
  TS* op = out + (size_t)plane * hw;
  int lo, hi;
  chunk_range8(hw, &lo, &hi);
  for (int i = lo + threadIdx.x; i < hi; i += kEwBlock) {
    float p[8], q[8], s[8], r[8];
    ld8<TS>(xb, i, p);
    ld8<TS>(xv, i, q);
    ld8<TS>(yp, i, s);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float g = sigmoidf_(fmaf(sc, s[j], sh));
      r[j] = g * (a * p[j]) + (1.0f - g) * (na * q[j]);
    }
    st8<TS>(op, i, r);
  }
}

// g2 = dL/d s2 = go*(a*xb - (1-a)*xv)*g*(1-g), stored in TS; the BatchNorm-2 backward sums are those of the STORED g2;
// the go-part of dL/da: sum go*(g*xb - (1-g)*xv).   part: [(b*chunks+chunk)][2][c];  da_p1: [(b*chunks+chunk)][c]
template <class TS>
__global__ __launch_bounds__(kEwBlock) void blend2_bn_bwd_h_kernel(const TS* __restrict__ x, const float* __restrict__ a1,
                                                                   const TS* __restrict__ y2, const float* __restrict__ scsh,
                                                                   const float* __restrict__ mean, const TS* __restrict__ go,
                                                                   TS* __restrict__ g2, float* __restrict__ part,
                                                                   float* __restrict__ da_p1, int c, int hw, BnTail tail) {
  __shared__ float sm[kEwBlock / DHD_WAVE];
  const int plane = blockIdx.y, b = plane / c, ch = plane % c;
  const float a = a1[plane], na = 1.0f - a, sc = scsh[ch], sh = scsh[c + ch], mu = mean[ch];
  const TS* xb = x + ((size_t)b * 2 * c + ch) * hw;
  const TS* xv = x + ((size_t)b * 2 * c + c + ch) * hw;
  const TS* yp = y2 + (size_t)plane * hw;
  const TS* gp = go + (size_t)plane * hw;
  TS* rp = g2 + (size_t)plane * hw;
  int lo, hi;
  chunk_range8(hw, &lo, &hi);
  float s1 = 0.f, s2 = 0.f, sa = 0.f;
  auto body = [&](const u32x4 wp, const u32x4 wq, const u32x4 ws, const u32x4 wo, int i) {
    float p[8], q[8], s[8], o[8], r[8];
    widen8<TS>(wp, p);
    widen8<TS>(wq, q);
    widen8<TS>(ws, s);
    widen8<TS>(wo, o);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float g = sigmoidf_(fmaf(sc, s[j], sh));
      r[j] = o[j] * (a * p[j] - na * q[j]) * g * (1.0f - g);
      sa = fmaf(o[j], g * p[j] - (1.0f - g) * q[j], sa);
    }
    const u32x4 pk = narrow8<TS>(r);
    __builtin_nontemporal_store(pk, reinterpret_cast<u32x4*>(rp) + i);
    widen8<TS>(pk, r);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s1 += r[j];
      s2 = fmaf(r[j], s[j] - mu, s2);
    }
  };
  auto ldv = [](const TS* base, int i) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base) + i); };
  int i = lo + threadIdx.x;
  for (; i + kEwBlock < hi; i += 2 * kEwBlock) {   // two 16-byte vectors of each of the four streams in flight per thread
    const u32x4 p0 = ldv(xb, i), q0 = ldv(xv, i), y0 = ldv(yp, i), o0 = ldv(gp, i);
    const u32x4 p1 = ldv(xb, i + kEwBlock), q1 = ldv(xv, i + kEwBlock), y1v = ldv(yp, i + kEwBlock), o1 = ldv(gp, i + kEwBlock);
    body(p0, q0, y0, o0, i);
    body(p1, q1, y1v, o1, i + kEwBlock);
  }
  if (i < hi) body(ldv(xb, i), ldv(xv, i), ldv(yp, i), ldv(gp, i), i);
  s1 = block_sum(s1, sm);
  s2 = block_sum(s2, sm);
  sa = block_sum(sa, sm);
  if (threadIdx.x == 0) {
    const size_t qi = (size_t)(b * kPlaneChunks + blockIdx.x);
    da_p1[qi * c + ch] = sa;
    bn_backward_publish(tail, part, qi, ch, c, s1, s2);
  }
}

// sums for BatchNorm backward: S1 = sum g, S2 = sum g*(y - mean)
template <class TS>
__global__ __launch_bounds__(kEwBlock) void pair_sums_h_kernel(const TS* __restrict__ g, const TS* __restrict__ y,
                                                               const float* __restrict__ mean, float* __restrict__ part, int c, int hw,
                                                               BnTail tail) {
  __shared__ float sm[kEwBlock / DHD_WAVE];
  const int plane = blockIdx.y, b = plane / c, ch = plane % c;
  const float mu = mean[ch];
  const TS* gp = g + (size_t)plane * hw;
  const TS* yp = y + (size_t)plane * hw;
  int lo, hi;
  chunk_range8(hw, &lo, &hi);
  float s1 = 0.f, s2 = 0.f, t1 = 0.f, t2 = 0.f;
  int i = lo + threadIdx.x;
  for (; i + kEwBlock < hi; i += 2 * kEwBlock) {
    float a[8], v[8], a2[8], v2[8];
    ld8<TS>(gp, i, a);
    ld8<TS>(yp, i, v);
    ld8<TS>(gp, i + kEwBlock, a2);
    ld8<TS>(yp, i + kEwBlock, v2);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s1 += a[j];
      s2 = fmaf(a[j], v[j] - mu, s2);
      t1 += a2[j];
      t2 = fmaf(a2[j], v2[j] - mu, t2);
    }
  }
  if (i < hi) {
    float a[8], v[8];
    ld8<TS>(gp, i, a);
    ld8<TS>(yp, i, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s1 += a[j];
      s2 = fmaf(a[j], v[j] - mu, s2);
    }
  }
  s1 = block_sum(s1 + t1, sm);
  s2 = block_sum(s2 + t2, sm);
  if (threadIdx.x == 0) bn_backward_publish(tail, part, (size_t)(b * kPlaneChunks + blockIdx.x), ch, c, s1, s2);
}

// the du-part of dL/da: sum du*(xb - xv).   da_p2: [(b*chunks+chunk)][c]
template <class TS>
__global__ __launch_bounds__(kEwBlock) void blend1_da_h_kernel(const TS* __restrict__ x, const TS* __restrict__ du,
                                                               float* __restrict__ da_p2, int c, int hw) {
  __shared__ float sm[kEwBlock / DHD_WAVE];
  const int plane = blockIdx.y, b = plane / c, ch = plane % c;
  const TS* xb = x + ((size_t)b * 2 * c + ch) * hw;
  const TS* xv = x + ((size_t)b * 2 * c + c + ch) * hw;
  const TS* dp = du + (size_t)plane * hw;
  int lo, hi;
  chunk_range8(hw, &lo, &hi);
  float acc = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += kEwBlock) {
    float p[8], q[8], d[8];
    ld8<TS>(xb, i, p);
    ld8<TS>(xv, i, q);
    ld8<TS>(dp, i, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = fmaf(d[j], p[j] - q[j], acc);
  }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) da_p2[(size_t)(b * kPlaneChunks + blockIdx.x) * c + ch] = acc;
}

// gx_bev = a*(go*g + du) + ds_bev/hw;  gx_vox = (1-a)*(go*(1-g) + du) + ds_vox/hw
template <class TS>
__global__ __launch_bounds__(kEwBlock) void stage_gx_h_kernel(const float* __restrict__ a1, const TS* __restrict__ y2,
                                                              const float* __restrict__ scsh, const TS* __restrict__ go,
                                                              const TS* __restrict__ du, const float* __restrict__ ds,
                                                              TS* __restrict__ gx, int c, int hw, int fc_rows, FcGradJob fc) {
  if ((int)blockIdx.y < fc_rows) {   // the first block rows: the Linear layers' parameter gradients
    fc_param_grad_block(fc, (int)blockIdx.y * kPlaneChunks + (int)blockIdx.x, c);
    return;
  }
  const int plane = (int)blockIdx.y - fc_rows, b = plane / c, ch = plane % c;
  const float a = a1[plane], na = 1.0f - a, sc = scsh[ch], sh = scsh[c + ch];
  const float kb = ds[(size_t)b * 2 * c + ch] / (float)hw, kv = ds[(size_t)b * 2 * c + c + ch] / (float)hw;
  const TS* yp = y2 + (size_t)plane * hw;
  const TS* gp = go + (size_t)plane * hw;
  const TS* dp = du + (size_t)plane * hw;
  TS* gb = gx + ((size_t)b * 2 * c + ch) * hw;
  TS* gv = gx + ((size_t)b * 2 * c + c + ch) * hw;
  int lo, hi;
  chunk_range8(hw, &lo, &hi);
  auto body = [&](const u32x4 ws, const u32x4 wo, const u32x4 wd, int i) {
    float s[8], o[8], d[8], rb[8], rv[8];
    widen8<TS>(ws, s);
    widen8<TS>(wo, o);
    widen8<TS>(wd, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float g = sigmoidf_(fmaf(sc, s[j], sh));
      rb[j] = fmaf(a, fmaf(o[j], g, d[j]), kb);
      rv[j] = fmaf(na, fmaf(o[j], 1.0f - g, d[j]), kv);
    }
    st8<TS>(gb, i, rb);
    st8<TS>(gv, i, rv);
  };
  auto ldv = [](const TS* base, int i) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base) + i); };
  int i = lo + threadIdx.x;
  for (; i + kEwBlock < hi; i += 2 * kEwBlock) {   // two vectors of each stream in flight per thread
    const u32x4 s0 = ldv(yp, i), o0 = ldv(gp, i), d0 = ldv(dp, i);
    const u32x4 s1 = ldv(yp, i + kEwBlock), o1 = ldv(gp, i + kEwBlock), d1 = ldv(dp, i + kEwBlock);
    body(s0, o0, d0, i);
    body(s1, o1, d1, i + kEwBlock);
  }
  if (i < hi) body(ldv(yp, i), ldv(gp, i), ldv(dp, i), i);
}

// ------------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------------

// C = 128 / 256 (the weight fragments of 32 channels x all K fit a wave's registers, one wave per 32 channels) and whole
// 16-byte vectors of the half type per plane
inline bool half_storage_supported(int c, int hw) { return (c == 128 || c == 256) && hw > 0 && (hw & 7) == 0 && stage_supported(c, hw); }

// Layouts in BYTES.  Small float32 tables first (same content as the float32 stage's), then the tensors in TS.
struct SavedLayoutH {
  size_t s, h, a1, tab_a, mean1, rstd1, scsh1, tab1, mean2, rstd2, scsh2, loc1, loc2, tick, wp1t, wp2t, mask, y1, y2, total;
};
inline SavedLayoutH saved_layout_h(int b, int c, int hw, int r) {
  SavedLayoutH L;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
  auto f = [&](size_t n) { return take(n * sizeof(float)); };
  L.s = f((size_t)b * 2 * c); L.h = f((size_t)b * r); L.a1 = f((size_t)b * c); L.tab_a = f((size_t)b * 3 * c);
  L.mean1 = f(c); L.rstd1 = f(c); L.scsh1 = f(2 * c); L.tab1 = f((size_t)b * 3 * c);
  L.mean2 = f(c); L.rstd2 = f(c); L.scsh2 = f(2 * c);
  L.loc1 = take((2 * (size_t)c + 1) * sizeof(double)); L.loc2 = take((2 * (size_t)c + 1) * sizeof(double));
  L.tick = f(2 * (size_t)c + kTickWords);
  L.wp1t = take((size_t)c * c * 2); L.wp2t = take((size_t)c * c * 2);
  L.mask = take(cuh_mask_words(b, c, hw) * sizeof(unsigned));
  L.y1 = take((size_t)b * c * hw * 2); L.y2 = take((size_t)b * c * hw * 2);
  L.total = o;
  return L;
}
struct ScratchLayoutH {
  size_t wp1, wp2, part, stat_part, da1, da2, tab_g2, tab_g1, dpre2, dh, ds, mean_part, g2, g1, du, wpart, total;
};
inline ScratchLayoutH scratch_layout_h(int b, int c, int hw, int r) {
  ScratchLayoutH L;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
  auto f = [&](size_t n) { return take(n * sizeof(float)); };
  const size_t plane = (size_t)b * c * hw * 2;
  L.wp1 = take((size_t)c * c * 2); L.wp2 = take((size_t)c * c * 2);
  L.part = f((size_t)b * kPlaneChunks * 2 * c);
  L.stat_part = f((size_t)(b / 8 + 2) * 1024 * 2 * c);   // one row per workgroup of every GEMM launch (>= 8 samples per launch, <= 1024 CUs)
  L.da1 = f((size_t)b * kPlaneChunks * c); L.da2 = f((size_t)b * kPlaneChunks * c);
  L.tab_g2 = f((size_t)b * 3 * c); L.tab_g1 = f((size_t)b * 3 * c);
  L.dpre2 = f((size_t)b * c); L.dh = f((size_t)b * r); L.ds = f((size_t)b * 2 * c);
  L.mean_part = f((size_t)b * 2 * c * kPlaneChunks);
  L.g2 = take(plane); L.g1 = take(plane); L.du = take(plane);
  L.wpart = f((size_t)kWgWorkers * c * c);
  L.total = o;
  return L;
}

template <class TS>
int launch_pw_gemm_cuh(const TS* in0, const TS* in1, size_t in_bstride, int in_channels, const float* coef, bool relu, const void* wp,
                       const float* bias, unsigned* relu_mask, float* stat_part, TS* y, int epi, int b, int c, int hw, hipStream_t st,
                       int* stat_rows) {
  const int waves = c / 32;
  const int max_b = cuh_max_batch(c, waves);
  if (max_b < 1) return DHD_EUNSUPPORTED;
  const bool two = in1 != nullptr;
  const unsigned in_bytes = (unsigned)((size_t)in_channels * hw * sizeof(TS));
  const int nwt = (hw + kCuhTile - 1) / kCuhTile;
  int cus = cu_count();
  if (cus <= 0) cus = 256;
  int rows_done = 0;
  for (int b0 = 0; b0 < b; b0 += max_b) {
    const int nb = b - b0 < max_b ? b - b0 : max_b;
    const long total = (long)nb * nwt;
    const int grid = (int)(total < cus ? total : cus);
    const size_t shmem = cuh_lds_bytes(c, waves, nb);
    const TS* i0 = in0 + (size_t)b0 * in_bstride;
    const TS* i1 = two ? in1 + (size_t)b0 * in_bstride : nullptr;
    const float* cf = coef + (size_t)b0 * 3 * c;
    unsigned* rm = relu_mask ? relu_mask + cuh_mask_words(b0, c, hw) : nullptr;
    float* sp = stat_part ? stat_part + (size_t)rows_done * 2 * c : nullptr;
    rows_done += grid;
    TS* yo = y + (size_t)b0 * c * hw;
    // EORD as in the float32 cu kernels: with two inputs the epilogue goes before the staging (which waits for twice the loads)
#define DHD_CUH(KCN, WAVES, TWO, RELU, EPI, REC)                                                                      \
  do {                                                                                                                \
    auto kern = pw_gemm_cuh_kernel<TS, KCN, WAVES, TWO, RELU, EPI, REC, (TWO) ? 1 : 0>;                               \
    DHD_LDS_ATTR_ONCE(kern, kLdsBytes);                                                                               \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), shmem, st, i0, i1, in_bstride, in_bytes, cf,               \
                       reinterpret_cast<const u32x4*>(wp), bias, rm, sp, yo, hw, nb);                                 \
  } while (0)
#define DHD_CUH_V(KCN, WAVES)                                                                       \
  do {                                                                                              \
    if (epi == 0 && two && !relu) DHD_CUH(KCN, WAVES, true, false, 0, false);                       \
    else if (epi == 0 && !two && relu && rm) DHD_CUH(KCN, WAVES, false, true, 0, true);             \
    else if (epi == 0 && !two && relu) DHD_CUH(KCN, WAVES, false, true, 0, false);                  \
    else if (epi == 1 && two && !relu) DHD_CUH(KCN, WAVES, true, false, 1, false);                  \
    else if (epi == 2 && two && !relu) DHD_CUH(KCN, WAVES, true, false, 2, false);                  \
    else return DHD_EUNSUPPORTED;                                                                   \
  } while (0)
    if (c == 256) DHD_CUH_V(16, 8);
    else DHD_CUH_V(8, 4);
#undef DHD_CUH_V
#undef DHD_CUH
    DHD_LAUNCH_CHECK();
  }
  if (stat_rows) *stat_rows = rows_done;
  return DHD_OK;
}

template <class TS>
int launch_pw_wgrad_h(const TS* a0, const TS* a1, const float* acoef, size_t a_bs, const TS* b0, const TS* b1, const float* bcoef,
                      size_t b_bs, bool b_relu, float* partial, float* gw, int b, int c, int hw, hipStream_t st) {
  const int ot = c == 128 ? 128 : 256;
  const int workers = kWgWorkers;
  const dim3 grid(workers, 1);
  const size_t shmem = (size_t)2 * 4 * 2 * (ot / 32) * 64 * 16;
  const bool btwo = b1 != nullptr;
  if (a1 == nullptr || (btwo == b_relu)) return DHD_EUNSUPPORTED;
#define DHD_WGH(OT, BTWO, BRELU)                                                                                   \
  do {                                                                                                             \
    auto kern = pw_wgrad_h_kernel<TS, OT, BTWO, BRELU>;                                                            \
    DHD_LDS_ATTR_ONCE(kern, shmem);                                                                                \
    hipLaunchKernelGGL(kern, grid, dim3(512), shmem, st, a0, a1, acoef, a_bs, b0, b1, bcoef, b_bs, partial, c, hw, \
                       b, workers);                                                                                \
  } while (0)
  if (ot == 128) {
    if (btwo) DHD_WGH(128, true, false);
    else DHD_WGH(128, false, true);
  } else {
    if (btwo) DHD_WGH(256, true, false);
    else DHD_WGH(256, false, true);
  }
#undef DHD_WGH
  DHD_LAUNCH_CHECK();
  const int n = c * c;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(dhd_cdiv(n, DHD_WAVE)), dim3(kEwBlock), 0, st, partial, gw, n, workers);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

// Forward, phases [lo, hi] and `sync` as in stage_forward_impl.
template <class TS>
int stage_forward_half(const void* xv, const dhd_sfa_weights* w, void* outv, void* saved, void* scratch, int b, int c, int hw, int lo,
                       int hi, double* sync, hipStream_t st) {
  const TS* x = static_cast<const TS*>(xv);
  TS* out = static_cast<TS*>(outv);
  const int r = w->hidden;
  const SavedLayoutH S = saved_layout_h(b, c, hw, r);
  const ScratchLayoutH T = scratch_layout_h(b, c, hw, r);
  unsigned char* sv = static_cast<unsigned char*>(saved);
  unsigned char* sc = static_cast<unsigned char*>(scratch);
  auto SF = [&](size_t off) { return reinterpret_cast<float*>(sv + off); };
  auto TF = [&](size_t off) { return reinterpret_cast<float*>(sc + off); };
  TS* y1 = reinterpret_cast<TS*>(sv + S.y1);
  TS* y2 = reinterpret_cast<TS*>(sv + S.y2);
  const dim3 planes(kPlaneChunks, b * c), per_ch(dhd_cdiv(c, kEwBlock));
  const bool training = w->training != 0;
  if (sync && !training) return DHD_EUNSUPPORTED;
  const size_t cs = (size_t)c * hw;
  int stat_rows = 0, rc;

  if (lo <= 0) {
    PackJobH job;
    job.w[0] = w->conv1_w; job.w[1] = w->conv2_w;
    job.dst[0] = reinterpret_cast<u32x4*>(sc + T.wp1); job.dst[1] = reinterpret_cast<u32x4*>(sc + T.wp2);
    job.dst[2] = reinterpret_cast<u32x4*>(sv + S.wp1t); job.dst[3] = reinterpret_cast<u32x4*>(sv + S.wp2t);
    job.c = c;
    job.blocks_each = dhd_cdiv((c / 32) * (c / 16) * 64, kEwBlock);
    const dim3 grid(kPlaneChunks, b * 2 * c + dhd_cdiv(4 * job.blocks_each, kPlaneChunks));
    hipLaunchKernelGGL(plane_mean_pack_h_kernel<TS>, grid, dim3(kEwBlock), 0, st, x, TF(T.mean_part), hw, b * 2 * c, job);
    hipLaunchKernelGGL(fc_forward_kernel, dim3(b), dim3(kFcBlock), (size_t)(2 * c + r) * sizeof(float), st, TF(T.mean_part), w->fc1_w,
                       w->fc1_b, w->fc2_w, w->fc2_b, SF(S.s), SF(S.h), SF(S.a1), SF(S.tab_a), c, r, hw, reinterpret_cast<int*>(sv + S.tick),
                       2 * c + kTickWords);
    DHD_LAUNCH_CHECK();
    // y1 = conv1(blend1(x))
    rc = launch_pw_gemm_cuh<TS>(x, x + cs, 2 * cs, c, SF(S.tab_a), false, sc + T.wp1, w->conv1_b, nullptr,
                                training ? TF(T.stat_part) : nullptr, y1, 0, b, c, hw, st, &stat_rows);
    if (rc != DHD_OK) return rc;
    if (sync) {
      hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(c / 4), dim3(kEwBlock), 0, st, TF(T.stat_part), stat_rows, w->conv1_b, w->bn1_w,
                         w->bn1_b, w->bn1_mean, w->bn1_var, w->momentum1, w->eps1, SF(S.mean1), SF(S.rstd1), SF(S.scsh1), SF(S.tab1), b, c,
                         hw, sync, reinterpret_cast<double*>(sv + S.loc1), nullptr, nullptr);
      DHD_LAUNCH_CHECK();
    }
  }
  if (hi <= 0) return DHD_OK;
  if (lo <= 1) {
    if (training)
      hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(c / 4), dim3(kEwBlock), 0, st, TF(T.stat_part), stat_rows, w->conv1_b, w->bn1_w,
                         w->bn1_b, w->bn1_mean, w->bn1_var, w->momentum1, w->eps1, SF(S.mean1), SF(S.rstd1), SF(S.scsh1), SF(S.tab1), b, c,
                         hw, nullptr, nullptr, sync, reinterpret_cast<long long*>(w->bn1_batches));
    else
      hipLaunchKernelGGL(bn_eval_coef_kernel, per_ch, dim3(kEwBlock), 0, st, w->bn1_w, w->bn1_b, w->bn1_mean, w->bn1_var, w->eps1,
                         SF(S.mean1), SF(S.rstd1), SF(S.scsh1), SF(S.tab1), b, c);
    DHD_LAUNCH_CHECK();
    // y2 = conv2(relu(bn1(y1)))
    rc = launch_pw_gemm_cuh<TS>(y1, nullptr, cs, c, SF(S.tab1), true, sc + T.wp2, w->conv2_b, reinterpret_cast<unsigned*>(sv + S.mask),
                                training ? TF(T.stat_part) : nullptr, y2, 0, b, c, hw, st, &stat_rows);
    if (rc != DHD_OK) return rc;
    if (sync) {
      hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(c / 4), dim3(kEwBlock), 0, st, TF(T.stat_part), stat_rows, w->conv2_b, w->bn2_w,
                         w->bn2_b, w->bn2_mean, w->bn2_var, w->momentum2, w->eps2, SF(S.mean2), SF(S.rstd2), SF(S.scsh2), TF(T.tab_g2), b, c,
                         hw, sync, reinterpret_cast<double*>(sv + S.loc2), nullptr, nullptr);
      DHD_LAUNCH_CHECK();
    }
  }
  if (hi <= 1) return DHD_OK;
  float* tab_unused = TF(T.tab_g2);   // bn2 has no consumer GEMM in the forward
  if (training)
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(c / 4), dim3(kEwBlock), 0, st, TF(T.stat_part), stat_rows, w->conv2_b, w->bn2_w,
                       w->bn2_b, w->bn2_mean, w->bn2_var, w->momentum2, w->eps2, SF(S.mean2), SF(S.rstd2), SF(S.scsh2), tab_unused, b, c, hw,
                       nullptr, nullptr, sync, reinterpret_cast<long long*>(w->bn2_batches));
  else
    hipLaunchKernelGGL(bn_eval_coef_kernel, per_ch, dim3(kEwBlock), 0, st, w->bn2_w, w->bn2_b, w->bn2_mean, w->bn2_var, w->eps2, SF(S.mean2),
                       SF(S.rstd2), SF(S.scsh2), tab_unused, b, c);
  hipLaunchKernelGGL(blend2_bn_h_kernel<TS>, planes, dim3(kEwBlock), 0, st, x, SF(S.a1), y2, SF(S.scsh2), out, c, hw);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

template <class TS>
int stage_backward_half(const void* xv, const dhd_sfa_weights* w, const void* saved, const void* goutv, void* gxv,
                        const dhd_sfa_grads* grads, void* scratch, int b, int c, int hw, int lo, int hi, double* sync, hipStream_t st) {
  const TS* x = static_cast<const TS*>(xv);
  const TS* gout = static_cast<const TS*>(goutv);
  TS* gx = static_cast<TS*>(gxv);
  const int r = w->hidden;
  const SavedLayoutH S = saved_layout_h(b, c, hw, r);
  const ScratchLayoutH T = scratch_layout_h(b, c, hw, r);
  unsigned char* sv = const_cast<unsigned char*>(static_cast<const unsigned char*>(saved));
  unsigned char* sc = static_cast<unsigned char*>(scratch);
  auto SF = [&](size_t off) { return reinterpret_cast<float*>(sv + off); };
  auto TF = [&](size_t off) { return reinterpret_cast<float*>(sc + off); };
  const TS* y1 = reinterpret_cast<const TS*>(sv + S.y1);
  const TS* y2 = reinterpret_cast<const TS*>(sv + S.y2);
  TS* g2 = reinterpret_cast<TS*>(sc + T.g2);
  TS* g1 = reinterpret_cast<TS*>(sc + T.g1);
  TS* du = reinterpret_cast<TS*>(sc + T.du);
  const dim3 planes(kPlaneChunks, b * c), per_ch(dhd_cdiv(c, kEwBlock));
  const size_t cs = (size_t)c * hw;
  if (sync && !w->training) return DHD_EUNSUPPORTED;
  int rc;

  if (lo <= 0) {
    int* tick = reinterpret_cast<int*>(sv + S.tick);
    const BnTail tail2 = {sync ? nullptr : tick, w->bn2_w, SF(S.mean2), SF(S.rstd2), TF(T.tab_g2), grads->bn2_w, grads->bn2_b,
                          grads->conv2_b, w->training, b, hw};
    hipLaunchKernelGGL(blend2_bn_bwd_h_kernel<TS>, planes, dim3(kEwBlock), 0, st, x, SF(S.a1), y2, SF(S.scsh2), SF(S.mean2), gout, g2,
                       TF(T.part), TF(T.da1), c, hw, tail2);
    if (sync)
      hipLaunchKernelGGL(bn_backward_coef_kernel, per_ch, dim3(kEwBlock), 0, st, TF(T.part), w->bn2_w, SF(S.mean2), SF(S.rstd2), w->training,
                         TF(T.tab_g2), grads->bn2_w, grads->bn2_b, grads->conv2_b, b, c, hw, sync, nullptr, nullptr, nullptr);
    DHD_LAUNCH_CHECK();
  }
  if (hi <= 0) return DHD_OK;
  if (lo <= 1) {
    if (sync)
      hipLaunchKernelGGL(bn_backward_coef_kernel, per_ch, dim3(kEwBlock), 0, st, TF(T.part), w->bn2_w, SF(S.mean2), SF(S.rstd2), w->training,
                         TF(T.tab_g2), grads->bn2_w, grads->bn2_b, grads->conv2_b, b, c, hw, nullptr, sync,
                         reinterpret_cast<const double*>(sv + S.loc2), w->conv2_b);
    DHD_LAUNCH_CHECK();
    // dW2 = dy2 . z1^T
    rc = launch_pw_wgrad_h<TS>(g2, y2, TF(T.tab_g2), cs, y1, nullptr, SF(S.tab1), cs, true, TF(T.wpart), grads->conv2_w, b, c, hw, st);
    if (rc != DHD_OK) return rc;
    // g1 = (W2^T dy2) * [z1 > 0]
    rc = launch_pw_gemm_cuh<TS>(g2, y2, cs, c, TF(T.tab_g2), false, sv + S.wp2t, nullptr, reinterpret_cast<unsigned*>(sv + S.mask), nullptr,
                                g1, 1, b, c, hw, st, nullptr);
    if (rc != DHD_OK) return rc;
    int* tick = reinterpret_cast<int*>(sv + S.tick) + c;
    const BnTail tail1 = {sync ? nullptr : tick, w->bn1_w, SF(S.mean1), SF(S.rstd1), TF(T.tab_g1), grads->bn1_w, grads->bn1_b,
                          grads->conv1_b, w->training, b, hw};
    hipLaunchKernelGGL(pair_sums_h_kernel<TS>, planes, dim3(kEwBlock), 0, st, g1, y1, SF(S.mean1), TF(T.part), c, hw, tail1);
    if (sync)
      hipLaunchKernelGGL(bn_backward_coef_kernel, per_ch, dim3(kEwBlock), 0, st, TF(T.part), w->bn1_w, SF(S.mean1), SF(S.rstd1), w->training,
                         TF(T.tab_g1), grads->bn1_w, grads->bn1_b, grads->conv1_b, b, c, hw, sync, nullptr, nullptr, nullptr);
    DHD_LAUNCH_CHECK();
  }
  if (hi <= 1) return DHD_OK;
  if (sync)
    hipLaunchKernelGGL(bn_backward_coef_kernel, per_ch, dim3(kEwBlock), 0, st, TF(T.part), w->bn1_w, SF(S.mean1), SF(S.rstd1), w->training,
                       TF(T.tab_g1), grads->bn1_w, grads->bn1_b, grads->conv1_b, b, c, hw, nullptr, sync,
                       reinterpret_cast<const double*>(sv + S.loc1), w->conv1_b);
  DHD_LAUNCH_CHECK();
  // dW1 = dy1 . u^T
  rc = launch_pw_wgrad_h<TS>(g1, y1, TF(T.tab_g1), cs, x, x + cs, SF(S.tab_a), 2 * cs, false, TF(T.wpart), grads->conv1_w, b, c, hw, st);
  if (rc != DHD_OK) return rc;
  // du = W1^T dy1
  rc = launch_pw_gemm_cuh<TS>(g1, y1, cs, c, TF(T.tab_g1), false, sv + S.wp1t, nullptr, nullptr, nullptr, du, 2, b, c, hw, st, nullptr);
  if (rc != DHD_OK) return rc;
  hipLaunchKernelGGL(blend1_da_h_kernel<TS>, planes, dim3(kEwBlock), 0, st, x, du, TF(T.da2), c, hw);
  hipLaunchKernelGGL(fc_backward_kernel, dim3(b), dim3(kEwBlock), (size_t)(c + r + kEwBlock) * sizeof(float), st, TF(T.da1), TF(T.da2), SF(S.a1),
                     SF(S.h), w->fc1_w, w->fc2_w, TF(T.dpre2), TF(T.dh), TF(T.ds), c, r);
  const int n_fc = r * 2 * c + c * r + r + c;
  const FcGradJob fcj = {TF(T.dpre2), TF(T.dh), SF(S.h), SF(S.s), grads->fc1_w, grads->fc1_b, grads->fc2_w, grads->fc2_b, b, r};
  const int fc_rows = dhd_cdiv(dhd_cdiv(n_fc, kEwBlock), kPlaneChunks);
  const dim3 planes_fc(kPlaneChunks, b * c + fc_rows);
  hipLaunchKernelGGL(stage_gx_h_kernel<TS>, planes_fc, dim3(kEwBlock), 0, st, SF(S.a1), y2, SF(S.scsh2), gout, du, TF(T.ds), gx, c, hw, fc_rows,
                     fcj);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}
