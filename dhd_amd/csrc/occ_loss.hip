// Occupancy-head losses of DHD's `predictor` in two streaming passes, for gfx950.
//
// Reference: models/dense_heads/occ_head.py:102-139 (predictor.loss) = class-balanced, camera-masked
// softmax cross entropy (models/losses/cross_entropy_loss.py:12-63) + sem_scal_loss_with_mask
// (models/losses/semkitti_loss.py:171-226) + geo_scal_loss_with_mask (:136-169), over
// M = B*200*200*16 voxels x 18 classes.  In eager PyTorch that is a softmax, ~17 masked gathers with a
// device->host sync each (`if torch.sum(..) > 0`), and their autograd mirror images.
//
// Every term is a function of a few global sums over the valid voxels v (camera mask & label != 255):
//   T_i = #{v: t_v = i}     P_i = sum_v p_vi     N_i = sum_{v: t_v = i} p_vi     CE = sum_v w_t (-log p_vt)
// (e.g. the specificity numerator sum (1-p_i)(1-[t=i]) = n_valid - P_i - T_i + N_i).  So:
//   pass 1 (occ_loss_sums)  reads logits/labels/mask once, softmax in registers, 56 block-reduced sums;
//   occ_loss_finalize       double-precision totals -> the three losses;
//   pass 2 (occ_loss_grad)  re-reads the logits, recomputes the softmax and writes
//                           dL/dz_k = valid p_k (a_k + b_k [t=k] - sum_j p_j a_j - p_t b_t) + ce (p_k - [t=k]),
//                           a_i = dL/dP_i, b_i = dL/dN_i evaluated per block from the stored totals.
// Traffic: 2 x 72 B read + 72 B written per voxel (+2 B labels/mask each pass); no (M,18) temporaries.
// Logit tiles go through LDS so that global accesses are 16-byte coalesced although a voxel's 18 logits
// are 72 contiguous bytes: a thread reads its voxel as nine conflict-free ds_read_b64.
#include "common.h"

namespace {

using f32x4_t = __attribute__((ext_vector_type(4))) float;

constexpr int K = 18;              // classes (Occ3D-nuScenes: 17 semantic + free)
constexpr int kBlock = 256;        // one voxel per thread per tile
constexpr int kSums = 3 * K + 2;   // T, P, N per class; CE; n_valid
constexpr int kMaxBlocks = 1024;
constexpr int kFinBlock = 1024;    // finalize: 64 columns x 16 row phases

struct Voxel {
  float p[K];
  float lse;   // log sum exp (shifted back)
  float zt;    // logit of the label (if t < K)
};

__device__ __forceinline__ void softmax18(const float* row, int t, Voxel& v) {
  float z[K];
#pragma unroll
  for (int q = 0; q < K / 2; ++q) {
    const float2 two = reinterpret_cast<const float2*>(row)[q];
    z[2 * q] = two.x;
    z[2 * q + 1] = two.y;
  }
  float mx = z[0];
#pragma unroll
  for (int k = 1; k < K; ++k) mx = fmaxf(mx, z[k]);
  float s = 0.f;
  v.zt = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (k == t) v.zt = z[k];
    v.p[k] = __expf(z[k] - mx);
    s += v.p[k];
  }
  const float inv = 1.0f / s;
#pragma unroll
  for (int k = 0; k < K; ++k) v.p[k] *= inv;
  v.lse = mx + __logf(s);
}

// copy tile [first voxel v0, n_in voxels) of the (M,18) matrix into LDS with 16-byte accesses
__device__ __forceinline__ void load_tile(const float* __restrict__ logits, long v0, int n_in, float* tile) {
  const f32x4_t* src = reinterpret_cast<const f32x4_t*>(logits + v0 * K);  // v0 is a multiple of kBlock: 16-byte aligned
  const int n4 = (n_in * K) >> 2, rest = (n_in * K) & 3;
  for (int i = threadIdx.x; i < n4; i += kBlock) reinterpret_cast<f32x4_t*>(tile)[i] = src[i];
  if (threadIdx.x < rest) tile[4 * n4 + threadIdx.x] = logits[v0 * K + 4 * n4 + threadIdx.x];
}

__global__ __launch_bounds__(kBlock) void occ_loss_sums(const float* __restrict__ logits, const uint8_t* __restrict__ labels,
                                                        const uint8_t* __restrict__ mask, const float* __restrict__ cw, long m,
                                                        int ignore, float* __restrict__ partial) {
  __shared__ float tile[kBlock * K];
  __shared__ float red[kBlock / DHD_WAVE][kSums];
  float accP[K], accN[K], accT[K], ce = 0.f, nv = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) accP[k] = accN[k] = accT[k] = 0.f;
  const long n_tiles = (m + kBlock - 1) / kBlock;
  for (long tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
    const long v0 = tl * kBlock;
    const int n_in = (int)min((long)kBlock, m - v0);
    __syncthreads();
    load_tile(logits, v0, n_in, tile);
    __syncthreads();
    if ((int)threadIdx.x < n_in) {
      const long v = v0 + threadIdx.x;
      const int t = labels[v];
      const bool cam = mask[v] != 0;
      const bool valid = cam && t != ignore;
      if (valid) {
        Voxel x;
        softmax18(tile + threadIdx.x * K, t, x);
        nv += 1.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          accP[k] += x.p[k];
          if (k == t) {
            accN[k] += x.p[k];
            accT[k] += 1.f;
          }
        }
        if (t < K) ce += cw[t] * (x.lse - x.zt);
      }
    }
  }
  // block reduction of the 56 sums
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float a = wave_sum_bcast(accT[k]), b = wave_sum_bcast(accP[k]), c = wave_sum_bcast(accN[k]);
    if (lane == 0) {
      red[wv][k] = a;
      red[wv][K + k] = b;
      red[wv][2 * K + k] = c;
    }
  }
  {
    const float a = wave_sum_bcast(ce), b = wave_sum_bcast(nv);
    if (lane == 0) {
      red[wv][3 * K] = a;
      red[wv][3 * K + 1] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < kSums) {
    float s = 0.f;
    for (int w = 0; w < kBlock / DHD_WAVE; ++w) s += red[w][threadIdx.x];
    partial[(size_t)blockIdx.x * kSums + threadIdx.x] = s;
  }
}

__device__ __forceinline__ double nll_clamped(double x) {
  // binary_cross_entropy_with_logits(inverse_sigmoid(x), 1): inverse_sigmoid (semkitti_loss.py:8-16) takes the
  // float32 value into [1e-5, 1-1e-5) in steps of 1e-5 (one step for x in [0,1]); the result is -log(x')
  x = (double)(float)x;
  if (x >= 1.0 - 1e-5) x -= 1e-5;
  if (x < 1e-5) x += 1e-5;
  return -log(x);
}

// derivative of nll_clamped(num/den) with respect to num and den
__device__ __forceinline__ void d_nll(double num, double den, double* d_num, double* d_den) {
  double x = (double)(float)(num / den);
  if (x >= 1.0 - 1e-5) x -= 1e-5;
  if (x < 1e-5) x += 1e-5;
  *d_num = -1.0 / (x * den);
  *d_den = num / (x * den * den);
}

// totals[0..55] = T, P, N, CE, n_valid (double);  losses[0..2] = cross entropy, sem scal, geo scal
__global__ __launch_bounds__(kFinBlock) void occ_loss_finalize(const float* __restrict__ partial, int n_blocks, const float* __restrict__ cw,
                                                            int non_empty, double* __restrict__ totals, float* __restrict__ losses) {
  __shared__ double tot[kSums];
  constexpr int kPh = kFinBlock / 64;
  __shared__ double ph[kPh][64];
  {
    // thread = (column, row phase): 16 phases x four independent accumulators keep the loads in flight
    const int col = threadIdx.x & 63, phase = threadIdx.x >> 6;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (col < kSums) {
      int b = phase;
      for (; b + 3 * kPh < n_blocks; b += 4 * kPh) {
        s0 += (double)partial[(size_t)b * kSums + col];
        s1 += (double)partial[(size_t)(b + kPh) * kSums + col];
        s2 += (double)partial[(size_t)(b + 2 * kPh) * kSums + col];
        s3 += (double)partial[(size_t)(b + 3 * kPh) * kSums + col];
      }
      for (; b < n_blocks; b += kPh) s0 += (double)partial[(size_t)b * kSums + col];
    }
    ph[phase][col] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  if (threadIdx.x < kSums) {
    double s = 0.0;
    for (int q = 0; q < kPh; ++q) s += ph[q][threadIdx.x];
    tot[threadIdx.x] = s;
    totals[threadIdx.x] = s;
  }
  __syncthreads();
  // per-class terms in parallel (each is up to three double-precision logs), then one thread adds them up
  __shared__ double term[K + 3];
  const double* T = tot;
  const double* P = tot + K;
  const double* N = tot + 2 * K;
  const double nv = tot[3 * K + 1];
  const int e = non_empty;
  if (threadIdx.x < K - 1) {
    const int i = threadIdx.x;
    double sem = 0.0;
    if (T[i] > 0.0) {
      if (P[i] > 0.0) sem += nll_clamped(N[i] / (P[i] + 1e-5));
      sem += nll_clamped(N[i] / (T[i] + 1e-5));
      const double neg = nv - T[i];
      if (neg > 0.0) sem += nll_clamped((nv - P[i] - T[i] + N[i]) / (neg + 1e-5));
    }
    term[i] = sem;
  } else if (threadIdx.x >= 64 && threadIdx.x < 67) {
    const double inter = (nv - T[e]) - (P[e] - N[e]);
    const int q = threadIdx.x - 64;
    term[K + q] = q == 0 ? nll_clamped(inter / ((nv - P[e]) + 1e-5))
                : q == 1 ? nll_clamped(inter / ((nv - T[e]) + 1e-5)) : nll_clamped(N[e] / (T[e] + 1e-5));
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  double avg = 0.0, sem = 0.0, count = 0.0;
  for (int i = 0; i < K; ++i) avg += T[i] * (double)cw[i];
  for (int i = 0; i < K - 1; ++i) {
    sem += term[i];
    count += T[i] > 0.0 ? 1.0 : 0.0;
  }
  losses[0] = (float)(tot[3 * K] / avg);
  losses[1] = (float)(sem / count);
  losses[2] = (float)(term[K] + term[K + 1] + term[K + 2]);
}

__global__ __launch_bounds__(kBlock) void occ_loss_grad(const float* __restrict__ logits, const uint8_t* __restrict__ labels,
                                                        const uint8_t* __restrict__ mask, const float* __restrict__ cw, long m,
                                                        int ignore, int non_empty, const double* __restrict__ totals,
                                                        const float* __restrict__ gl, float* __restrict__ grad) {
  __shared__ float tile[kBlock * K];
  __shared__ float sa[K], sb[K], sce;
  // a_i = dL/dP_i, b_i = dL/dN_i from the totals and the upstream gradients gl[0..2]
  if (threadIdx.x < K) {
    const int i = threadIdx.x;
    const double* T = totals;
    const double* P = totals + K;
    const double* N = totals + 2 * K;
    const double nv = totals[3 * K + 1];
    double a = 0.0, b = 0.0;
    if (i < K - 1 && T[i] > 0.0) {
      double count = 0.0;
      for (int j = 0; j < K - 1; ++j) count += T[j] > 0.0 ? 1.0 : 0.0;
      const double g = (double)gl[1] / count;
      double dn, dd;
      if (P[i] > 0.0) {
        d_nll(N[i], P[i] + 1e-5, &dn, &dd);
        b += g * dn;
        a += g * dd;
      }
      d_nll(N[i], T[i] + 1e-5, &dn, &dd);
      b += g * dn;
      const double neg = nv - T[i];
      if (neg > 0.0) {
        d_nll(nv - P[i] - T[i] + N[i], neg + 1e-5, &dn, &dd);
        a -= g * dn;
        b += g * dn;
      }
    }
    if (i == non_empty) {
      const double g = (double)gl[2];
      const double inter = (nv - T[i]) - (P[i] - N[i]);
      double dn, dd;
      d_nll(inter, (nv - P[i]) + 1e-5, &dn, &dd);  // precision: inter and the denominator both fall with P
      a += g * (-dn - dd);
      b += g * dn;
      d_nll(inter, (nv - T[i]) + 1e-5, &dn, &dd);  // recall
      a -= g * dn;
      b += g * dn;
      d_nll(N[i], T[i] + 1e-5, &dn, &dd);          // specificity
      b += g * dn;
    }
    sa[i] = (float)a;
    sb[i] = (float)b;
  }
  if (threadIdx.x == 0) {
    double avg = 0.0;
    for (int i = 0; i < K; ++i) avg += totals[i] * (double)cw[i];
    sce = (float)((double)gl[0] / avg);
  }
  __syncthreads();
  float a[K], bq[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { a[k] = sa[k]; bq[k] = sb[k]; }
  const float ce_scale = sce;

  const long n_tiles = (m + kBlock - 1) / kBlock;
  for (long tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
    const long v0 = tl * kBlock;
    const int n_in = (int)min((long)kBlock, m - v0);
    __syncthreads();
    load_tile(logits, v0, n_in, tile);
    __syncthreads();
    if ((int)threadIdx.x < n_in) {
      const long v = v0 + threadIdx.x;
      const int t = labels[v];
      const bool valid = mask[v] != 0 && t != ignore;
      float* row = tile + threadIdx.x * K;
      float d[K];
      if (valid) {
        Voxel x;
        softmax18(row, t, x);
        float s = 0.f, bt = 0.f, wt = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          s = fmaf(x.p[k], a[k], s);
          if (k == t) { bt = bq[k]; wt = cw[k]; }
        }
        float pt = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) if (k == t) pt = x.p[k];
        s = fmaf(pt, bt, s);
        const float cs = ce_scale * wt;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const float hit = k == t ? 1.f : 0.f;
          d[k] = x.p[k] * (a[k] + hit * bt - s) + cs * (x.p[k] - hit);
        }
      } else {
#pragma unroll
        for (int k = 0; k < K; ++k) d[k] = 0.f;
      }
#pragma unroll
      for (int q = 0; q < K / 2; ++q) reinterpret_cast<float2*>(row)[q] = make_float2(d[2 * q], d[2 * q + 1]);
    }
    __syncthreads();
    {
      f32x4_t* dst = reinterpret_cast<f32x4_t*>(grad + v0 * K);
      const int n4 = (n_in * K) >> 2, rest = (n_in * K) & 3;
      for (int i = threadIdx.x; i < n4; i += kBlock) __builtin_nontemporal_store(reinterpret_cast<const f32x4_t*>(tile)[i], dst + i);
      if ((int)threadIdx.x < rest) grad[v0 * K + 4 * n4 + threadIdx.x] = tile[4 * n4 + threadIdx.x];
    }
  }
}

inline int n_blocks_for(long m) {
  const long tiles = (m + kBlock - 1) / kBlock;
  return (int)(tiles < kMaxBlocks ? tiles : kMaxBlocks);
}

}  // namespace

extern "C" {

size_t dhd_occ_loss_workspace_bytes(void) { return (size_t)kMaxBlocks * kSums * sizeof(float) + 64 * sizeof(double); }

int dhd_occ_loss_forward(const float* logits, const uint8_t* labels, const uint8_t* mask, const float* class_weight, int64_t n_voxels,
                         int n_classes, int ignore_index, int non_empty_idx, float* losses, void* workspace, void* stream) {
  if (!logits || !labels || !mask || !class_weight || !losses || !workspace || n_voxels <= 0) return DHD_EINVAL;
  if (n_classes != K || non_empty_idx < 0 || non_empty_idx >= K) return DHD_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(logits) & 15) != 0) return DHD_EINVAL;
  hipStream_t st = dhd_stream(stream);
  float* partial = static_cast<float*>(workspace);
  double* totals = reinterpret_cast<double*>(partial + (size_t)kMaxBlocks * kSums);
  const int nb = n_blocks_for(n_voxels);
  hipLaunchKernelGGL(occ_loss_sums, dim3(nb), dim3(kBlock), 0, st, logits, labels, mask, class_weight, (long)n_voxels, ignore_index,
                     partial);
  hipLaunchKernelGGL(occ_loss_finalize, dim3(1), dim3(kFinBlock), 0, st, partial, nb, class_weight, non_empty_idx, totals, losses);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

int dhd_occ_loss_backward(const float* logits, const uint8_t* labels, const uint8_t* mask, const float* class_weight, int64_t n_voxels,
                          int n_classes, int ignore_index, int non_empty_idx, const float* grad_losses, const void* workspace,
                          float* grad_logits, void* stream) {
  if (!logits || !labels || !mask || !class_weight || !grad_losses || !workspace || !grad_logits || n_voxels <= 0) return DHD_EINVAL;
  if (n_classes != K || non_empty_idx < 0 || non_empty_idx >= K) return DHD_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(logits) & 15) != 0 || (reinterpret_cast<uintptr_t>(grad_logits) & 15) != 0) return DHD_EINVAL;
  hipStream_t st = dhd_stream(stream);
  const float* partial = static_cast<const float*>(workspace);
  const double* totals = reinterpret_cast<const double*>(partial + (size_t)kMaxBlocks * kSums);
  hipLaunchKernelGGL(occ_loss_grad, dim3(n_blocks_for(n_voxels)), dim3(kBlock), 0, st, logits, labels, mask, class_weight,
                     (long)n_voxels, ignore_index, non_empty_idx, totals, grad_losses, grad_logits);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Evaluation side of the same logits: predictor.get_occ (occ_head.py:141-153: softmax -> argmax) and
// Metric_mIoU.hist_info (core/evaluation/occ_metrics.py:79-104: 18x18 confusion counts over the
// camera-visible voxels with a label in [0, 18)), in one pass over the (M,18) logits.
// argmax of the logits = argmax of their softmax (first maximum wins); the two can differ only where
// the two largest probabilities round to the same float.
// ------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(kBlock) void occ_argmax_hist(const float* __restrict__ logits, const uint8_t* __restrict__ labels,
                                                          const uint8_t* __restrict__ mask, long m, uint8_t* __restrict__ pred,
                                                          unsigned long long* __restrict__ hist) {
  __shared__ float tile[kBlock * K];
  __shared__ unsigned cnt[K * K];
  for (int i = threadIdx.x; i < K * K; i += kBlock) cnt[i] = 0;
  const long n_tiles = (m + kBlock - 1) / kBlock;
  for (long tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
    const long v0 = tl * kBlock;
    const int n_in = (int)min((long)kBlock, m - v0);
    __syncthreads();
    load_tile(logits, v0, n_in, tile);
    __syncthreads();
    if ((int)threadIdx.x < n_in) {
      const float* row = tile + threadIdx.x * K;
      float best = row[0];
      int arg = 0;
#pragma unroll
      for (int k = 1; k < K; ++k) {
        const float z = row[k];
        if (z > best) { best = z; arg = k; }
      }
      const long v = v0 + threadIdx.x;
      if (pred) pred[v] = (uint8_t)arg;
      if (hist) {
        const int t = labels[v];
        if (t < K && (mask == nullptr || mask[v] != 0)) atomicAdd(&cnt[t * K + arg], 1u);
      }
    }
  }
  __syncthreads();
  if (hist)
    for (int i = threadIdx.x; i < K * K; i += kBlock)
      if (cnt[i]) atomicAdd(&hist[i], (unsigned long long)cnt[i]);
}

}  // namespace

extern "C" int dhd_occ_argmax_hist(const float* logits, const uint8_t* labels, const uint8_t* mask, int64_t n_voxels, int n_classes,
                                   uint8_t* pred, int64_t* hist, void* stream) {
  if (!logits || n_voxels <= 0 || (!pred && !hist) || (hist && !labels)) return DHD_EINVAL;
  if (n_classes != K) return DHD_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(logits) & 15) != 0) return DHD_EINVAL;
  hipLaunchKernelGGL(occ_argmax_hist, dim3(n_blocks_for(n_voxels)), dim3(kBlock), 0, dhd_stream(stream), logits, labels, mask,
                     (long)n_voxels, pred, reinterpret_cast<unsigned long long*>(hist));
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}
