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
#include <type_traits>

#include "common.h"
#include "mghs_layout.h"

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
  if (len <= 0) return;               // an empty interval (dhd_bev_pool_v2_regroup lists every pixel): nothing to write
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
  if (len <= 0) return;               // empty interval: nothing to write
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

// ---------------------------------------------------------------------------------------------------------------
// Sub-wave forms for C = 4 L, L a power of two (C = 64: L = 16).  The operator is a gather over many short intervals
// (DHD-S, B = 4: 81 k voxels of 7.8 points on average, the longest 256; 17 k pixels of 37 points): a whole wave per
// interval leaves the kernel at the latency of its dependent chain index -> depth -> row -> store with ~10 rounds of
// resident waves.  Here L lanes serve one interval (a lane = four channels, 16-byte accesses, a 4 C-byte row = one contiguous
// piece per group) and a wave carries 64 / L intervals side by side: 4x fewer waves, 4x more gathers in flight per wave.
//
// Round 3: the kernel's time was the serial chain of its LONGEST interval (a batch of L points = three dependent round
// trips: index words -> depth value / row address -> row; a 256-point voxel = 16 batches one after the other = 40 us,
// whatever the other 81 k intervals did).  Now (a) the index words and depth values of kChunk batches are requested
// together before the first row gather (2 round trips per kChunk * L points instead of 2 per L), and (b) an interval
// longer than kChunk * L points is not walked by its own L lanes: after the short ones the whole wave takes it, every
// group a quarter of each 64-point batch, and the groups' partial sums are added in a fixed order (deterministic).
// ---------------------------------------------------------------------------------------------------------------
using pf4 = __attribute__((ext_vector_type(4))) float;

constexpr int kChunk = 4;   // batches of L points whose index words are in flight together

// Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8); the grid is padded to a multiple of 8 and
// every XCD takes a contiguous eighth of the wave positions, so that neighbouring intervals (voxels / pixels that
// gather the same rows) meet in one L2 instead of being fetched into all eight.
__device__ __forceinline__ int xcd_contiguous_block() {
  const int per = gridDim.x >> 3;
  return (blockIdx.x & 7) * per + (blockIdx.x >> 3);
}
static inline int xcd_padded_blocks(int n_blocks) { return (n_blocks + 7) / 8 * 8; }

// Sum over the L lanes of a group, result in every lane.  L = 16 is one DPP row: two rotations and two quad
// permutations on the VALU instead of four trips through the LDS crossbar.
template <int L>
__device__ __forceinline__ float lanes_sum(float v) {
  if constexpr (L == 16) {
    v = dpp_add<0x128, 0xf>(v);   // row_ror:8
    v = dpp_add<0x124, 0xf>(v);   // row_ror:4
    v = dpp_add<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    return v;
  } else {
#pragma unroll
    for (int m = 1; m < L; m <<= 1) v += __shfl_xor(v, m, DHD_WAVE);
    return v;
  }
}

// Value held by lane k of the caller's own group of L lanes.  L = 16 is one DPP row: row_share:k is a VALU move
// (the general form goes through the LDS crossbar, two of them per gathered row made the kernels issue-bound).
template <int L, int K>
__device__ __forceinline__ int group_lane_i(int v, int grp) {
  if constexpr (L == 16) return __builtin_amdgcn_update_dpp(0, v, 0x150 + K, 0xf, 0xf, false);
  else return __shfl(v, grp * L + K, DHD_WAVE);
}
template <int L, int K>
__device__ __forceinline__ float group_lane_f(float v, int grp) {
  return __int_as_float(group_lane_i<L, K>(__float_as_int(v), grp));
}

// compile-time loop: body(std::integral_constant<int, I>) for I in [0, N)
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

// acc += sum over the group's rows of the chunk: batch b, row k of this group sits in source lane grp * L + k of rf[b] /
// dv[b]; rows k >= cnt[b] do not exist (cnt is uniform within the group, dv is 0 there).  The rows are requested before
// the depth values are looked at: depth[] (asked for by the caller right before) and the rows travel together.
template <int L, int R>
__device__ __forceinline__ pf4 fwd_chunk(pf4 acc, const int (&rf)[kChunk], const float (&dv)[kChunk], const int (&cnt)[kChunk],
                                         const pf4* __restrict__ feat, int grp, int cl) {
  constexpr int U = L < R ? L : R;
#pragma unroll
  for (int b = 0; b < kChunk; ++b) {
    if (!__any(cnt[b] > 0)) break;       // wave-uniform: batches are filled in order
    bool done = false;
    static_for<L / U>([&](auto ks) {
      constexpr int k0 = decltype(ks)::value * U;
      if (done || (k0 > 0 && !__any(cnt[b] > k0))) { done = true; return; }
      pf4 f[U];
      float d[U];
      static_for<U>([&](auto us) {
        constexpr int u = decltype(us)::value;
        const int q = group_lane_i<L, k0 + u>(rf[b], grp);
        f[u] = k0 + u < cnt[b] ? feat[(size_t)q * L + cl] : pf4{0.f, 0.f, 0.f, 0.f};
      });
      __builtin_amdgcn_sched_barrier(0);
      static_for<U>([&](auto us) {
        constexpr int u = decltype(us)::value;
        d[u] = group_lane_f<L, k0 + u>(dv[b], grp);
      });
#pragma unroll
      for (int u = 0; u < U; ++u) acc += f[u] * d[u];
    });
  }
  return acc;
}

// MAPPED (the fused operator): the row of voxel v in `out` / `out_grad` is row_of[v] (the compact table vsum[slot][C] of the
// segment writer / reader, row_of = nzoff) instead of v itself.
template <int L, int R, int WPS, bool MAPPED = false>
__global__ __launch_bounds__(kBlock, WPS) void bev_pool_v2_fwd_vec_kernel(int n_intervals, const float* __restrict__ depth,
                                                                      const pf4* __restrict__ feat, const int* __restrict__ ranks_depth,
                                                                      const int* __restrict__ ranks_feat,
                                                                      const int* __restrict__ ranks_bev,
                                                                      const int* __restrict__ interval_starts,
                                                                      const int* __restrict__ interval_lengths, pf4* __restrict__ out,
                                                                      const int* __restrict__ row_of = nullptr, int n_rows = 0) {
  constexpr int G = DHD_WAVE / L;
  const int lane = threadIdx.x & 63, grp = lane / L, cl = lane % L;
  // Interval of this group.  The lists are sorted by voxel, so long intervals (voxels next to a camera: up to ~300
  // points against a mean of 8) come in clusters; G consecutive intervals per wave put four of them into one wave, whose
  // serial walk then was the kernel's time.  An XCD's contiguous range of wx * G intervals is cut into G sub-ranges,
  // group g walks sub-range g, rotated by g * (wx / 5) positions so that the same place of every sample (the rigs
  // are alike) does not meet in one wave either.
  const int wx = (gridDim.x >> 3) * kWaves;                        // waves per XCD
  const int wq = xcd_contiguous_block() * kWaves + (threadIdx.x >> 6);
  const int xcd = wq / wx, j = wq - xcd * wx;
  int jr = j + grp * (wx / 5 + 1);
  jr -= (jr / wx) * wx;
  const int iv = G > 1 ? (xcd * G + grp) * wx + jr : wq;
  const bool valid = iv < n_intervals;
  const int start = valid ? interval_starts[iv] : 0;
  const int len = valid ? interval_lengths[iv] : 0;
  int vox = (valid && len > 0) ? ranks_bev[start] : 0;           // requested with the first indices, not after the gathers
  bool keep = valid && len > 0;
  if constexpr (MAPPED) {
    keep = keep && (unsigned)vox < (unsigned)n_rows;               // voxels outside the grid were not counted: no row
    vox = keep ? row_of[vox] : 0;
  }
  const bool is_long = G > 1 && len > kChunk * L && keep;
  const int own = (is_long || !keep) ? 0 : len;    // points this group walks by itself
  pf4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; __any(s0 < own); s0 += kChunk * L) {
    int rf[kChunk], rd[kChunk], cnt[kChunk];
    float dv[kChunk];
#pragma unroll
    for (int b = 0; b < kChunk; ++b) {
      cnt[b] = min(L, own - s0 - b * L);           // group-uniform, <= 0 when the group has no such batch
      const bool live = cl < cnt[b];
      rf[b] = live ? ranks_feat[start + s0 + b * L + cl] : 0;
      rd[b] = live ? ranks_depth[start + s0 + b * L + cl] : -1;
    }
#pragma unroll
    for (int b = 0; b < kChunk; ++b) dv[b] = rd[b] >= 0 ? depth[rd[b]] : 0.f;
    acc = fwd_chunk<L, R>(acc, rf, dv, cnt, feat, grp, cl);
  }
  if (keep && !is_long) out[(size_t)vox * L + cl] = acc;
  if constexpr (G > 1) {
    // the long intervals of this wave, one after the other, all 64 lanes on each
    unsigned long long todo = __ballot(is_long && cl == 0);
    while (todo) {
      const int src = __builtin_ctzll(todo);       // lane 0 of the owning group
      todo &= todo - 1;
      const int s = __builtin_amdgcn_readlane(start, src), n = __builtin_amdgcn_readlane(len, src);
      const int v = __builtin_amdgcn_readlane(vox, src);
      pf4 part = {0.f, 0.f, 0.f, 0.f};
      for (int s0 = 0; s0 < n; s0 += kChunk * DHD_WAVE) {
        int rf[kChunk], rd[kChunk], cnt[kChunk];
        float dv[kChunk];
#pragma unroll
        for (int b = 0; b < kChunk; ++b) {
          const int left = n - s0 - b * DHD_WAVE;  // points of this 64-point batch and beyond
          cnt[b] = min(L, left - grp * L);
          const bool live = lane < left;
          rf[b] = live ? ranks_feat[s + s0 + b * DHD_WAVE + lane] : 0;
          rd[b] = live ? ranks_depth[s + s0 + b * DHD_WAVE + lane] : -1;
        }
#pragma unroll
        for (int b = 0; b < kChunk; ++b) dv[b] = rd[b] >= 0 ? depth[rd[b]] : 0.f;
        part = fwd_chunk<L, R>(part, rf, dv, cnt, feat, grp, cl);
      }
      // groups' partial sums, fixed order: ((g0 + g1) + (g2 + g3)) for G = 4
#pragma unroll
      for (int m = L; m < DHD_WAVE; m <<= 1) {
        part.x += __shfl_xor(part.x, m, DHD_WAVE);
        part.y += __shfl_xor(part.y, m, DHD_WAVE);
        part.z += __shfl_xor(part.z, m, DHD_WAVE);
        part.w += __shfl_xor(part.w, m, DHD_WAVE);
      }
      if (grp == 0) out[(size_t)v * L + cl] = part;
    }
  }
}

template <int L, int R, int WPS, bool MAPPED = false>
__global__ __launch_bounds__(kBlock, WPS) void bev_pool_v2_bwd_vec_kernel(int n_intervals, const pf4* __restrict__ out_grad,
                                                                      const float* __restrict__ depth, const pf4* __restrict__ feat,
                                                                      const int* __restrict__ ranks_depth,
                                                                      const int* __restrict__ ranks_feat,
                                                                      const int* __restrict__ ranks_bev,
                                                                      const int* __restrict__ interval_starts,
                                                                      const int* __restrict__ interval_lengths,
                                                                      float* __restrict__ depth_grad, pf4* __restrict__ feat_grad,
                                                                      const int* __restrict__ row_of = nullptr, int n_rows = 0) {
  constexpr int G = DHD_WAVE / L;
  const int lane = threadIdx.x & 63, grp = lane / L, cl = lane % L;
  // consecutive pixels per wave and per XCD: neighbouring pixels gather the same out_grad rows
  const int iv = (xcd_contiguous_block() * kWaves + (threadIdx.x >> 6)) * G + grp;
  const int len = iv < n_intervals ? interval_lengths[iv] : 0;
  const bool valid = len > 0;                      // empty intervals (and the padding of the last wave) write nothing
  const int start = valid ? interval_starts[iv] : 0;
  const int pix = valid ? ranks_feat[start] : 0;   // every point of the interval shares the pixel (bev_pool.py:47-57)
  const pf4 fv = valid ? feat[(size_t)pix * L + cl] : pf4{0.f, 0.f, 0.f, 0.f};
  pf4 facc = {0.f, 0.f, 0.f, 0.f};
  // a pixel's points (<= D per grid slice: 44 or 88) in chunks of kChunk batches whose index words and depth values are
  // requested together (round 2: per batch of L points, i.e. three dependent round trips per batch)
  for (int s0 = 0; __any(s0 < len); s0 += kChunk * L) {
    int rb[kChunk], rd[kChunk], cnt[kChunk];
    float dv[kChunk];
#pragma unroll
    for (int b = 0; b < kChunk; ++b) {
      cnt[b] = min(L, len - s0 - b * L);
      const bool live = cl < cnt[b];
      rb[b] = live ? ranks_bev[start + s0 + b * L + cl] : 0;
      rd[b] = live ? ranks_depth[start + s0 + b * L + cl] : -1;
    }
#pragma unroll
    for (int b = 0; b < kChunk; ++b) dv[b] = rd[b] >= 0 ? depth[rd[b]] : 0.f;
    if constexpr (MAPPED) {
      // a point whose voxel lies outside the grid has no row: its depth value becomes 0 (no feature gradient) and its row the
      // first one (finite values; the depth gradient it would receive is discarded below)
#pragma unroll
      for (int b = 0; b < kChunk; ++b) {
        const bool in = (unsigned)rb[b] < (unsigned)n_rows;
        if (!in) { dv[b] = 0.f; if (rd[b] >= 0) rd[b] = -2; }
        rb[b] = (in && rd[b] >= 0) ? row_of[rb[b]] : 0;
      }
    }
#pragma unroll
    for (int b = 0; b < kChunk; ++b) {
      if (!__any(cnt[b] > 0)) break;
      float mine = 0.f;
      constexpr int U = L < R ? L : R;
      bool done = false;
      static_for<L / U>([&](auto ks) {
        constexpr int k0 = decltype(ks)::value * U;
        if (done || (k0 > 0 && !__any(cnt[b] > k0))) { done = true; return; }
        pf4 g[U];
        float d[U];
        static_for<U>([&](auto us) {
          constexpr int u = decltype(us)::value;
          const int vox = group_lane_i<L, k0 + u>(rb[b], grp);
          g[u] = k0 + u < cnt[b] ? out_grad[(size_t)vox * L + cl] : pf4{0.f, 0.f, 0.f, 0.f};
        });
        __builtin_amdgcn_sched_barrier(0);
        static_for<U>([&](auto us) {
          constexpr int u = decltype(us)::value;
          d[u] = group_lane_f<L, k0 + u>(dv[b], grp);
        });
#pragma unroll
        for (int u = 0; u < U; ++u) {
          facc += g[u] * d[u];
          const float part = lanes_sum<L>((g[u].x * fv.x + g[u].y * fv.y) + (g[u].z * fv.z + g[u].w * fv.w));
          if (cl == k0 + u) mine = part;
        }
      });
      if (rd[b] >= 0) depth_grad[rd[b]] = mine;       // one writer per point (:104-106)
    }
  }
  if (valid) feat_grad[(size_t)pix * L + cl] = facc;  // one writer per pixel (:120-121)
}

// ------------------------------------------------------------------------------------------------
// dhd_bev_pool_v2_regroup: the re-grouping of the point lists by feature pixel that QuickCumsumCuda.backward performs with
// argsort + gathers + a run-length scan on every call (bev_pool.py:47-57), as a device counting sort: count per pixel ->
// exclusive scan -> scatter, then the points of every pixel ordered by ranks_depth so that the result (and the float sums of
// the backward kernel) do not depend on the arrival order of the counting atomics.  One interval per pixel, empty ones included
// (length 0): no count has to travel to the host.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void regroup_count_kernel(const int* __restrict__ ranks_feat, int n_points, int n_pixels,
                                                                int* __restrict__ count, int* __restrict__ rnk) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_points) return;
  const int p = ranks_feat[i];
  rnk[i] = (p >= 0 && p < n_pixels) ? atomicAdd(&count[p], 1) : -1;   // out-of-range pixels are dropped
}

constexpr int kScanBlock = 1024;
__global__ __launch_bounds__(kScanBlock) void regroup_scan_kernel(const int* __restrict__ count, int n_pixels, int* __restrict__ starts,
                                                                   int* __restrict__ lengths) {
  __shared__ int wsum[kScanBlock / DHD_WAVE];
  __shared__ int carry_s;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n_pixels; base += kScanBlock) {
    const int i = base + t;
    const int v = i < n_pixels ? count[i] : 0;
    int incl = v;
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, DHD_WAVE);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int off = carry_s;
    for (int k = 0; k < wv; ++k) off += wsum[k];
    if (i < n_pixels) { starts[i] = off + incl - v; lengths[i] = v; }
    __syncthreads();
    if (t == kScanBlock - 1) carry_s = off + incl;
    __syncthreads();
  }
}

__global__ __launch_bounds__(kBlock) void regroup_scatter_kernel(const int* __restrict__ rd, const int* __restrict__ rf,
                                                                  const int* __restrict__ rb, const int* __restrict__ rnk,
                                                                  const int* __restrict__ starts, int n_points,
                                                                  int* __restrict__ t_rd, int* __restrict__ t_rb) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_points) return;
  const int r = rnk[i];
  if (r < 0) return;
  const int pos = starts[rf[i]] + r;
  t_rd[pos] = rd[i];
  t_rb[pos] = rb[i];
}

// final order inside a pixel: ascending ranks_depth (unique per point of a grid); thread = one grouped entry
__global__ __launch_bounds__(kBlock) void regroup_order_kernel(const int* __restrict__ t_rd, const int* __restrict__ t_rb,
                                                                const int* __restrict__ starts, const int* __restrict__ lengths,
                                                                int n_pixels, int* __restrict__ o_rd, int* __restrict__ o_rf,
                                                                int* __restrict__ o_rb) {
  // one wave per pixel: its entries (a few dozen) are ranked against each other
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * (kBlock / DHD_WAVE) + (threadIdx.x >> 6);
  if (p >= n_pixels) return;
  const int s0 = starts[p], len = lengths[p];
  for (int a = lane; a < len; a += DHD_WAVE) {
    const int mine = t_rd[s0 + a], vox = t_rb[s0 + a];
    int r = 0;
    for (int k = 0; k < len; ++k) {
      const int other = t_rd[s0 + k];
      r += (other < mine) || (other == mine && k < a);
    }
    o_rd[s0 + r] = mine;
    o_rb[s0 + r] = vox;
    o_rf[s0 + r] = p;
  }
}

inline size_t regroup_align(size_t v) { return (v + 63) & ~(size_t)63; }   // ints: 256-byte sections

inline int vec_lanes(int c) {  // L = C / 4 when that is a power of two <= 64 and every row is 16-byte aligned, else 0
  if (c % 4 != 0) return 0;
  const int l = c / 4;
  return (l >= 1 && l <= 64 && (l & (l - 1)) == 0) ? l : 0;
}

// ------------------------------------------------------------------------------------------------
// Fused operator (dhd_bev_pool_v2_fused_*): bev_pool_v2 + `permute(0, 4, 1, 2, 3).contiguous()` (bev_pool.py:27,105) as
// interval sums into a compact table + the MGHS segment writer, which streams the final (B, C, Dz, Dy, Dx) tensor once, zeros
// included; the backward reads out_grad in that layout once (segment reader) and runs the pixel kernel on the table.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void fused_mark_kernel(int n_intervals, int n_voxels, const int* __restrict__ ranks_bev,
                                                             const int* __restrict__ interval_starts,
                                                             const int* __restrict__ interval_lengths, int* __restrict__ count) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_intervals) return;
  const int len = interval_lengths[i];
  if (len <= 0) return;
  const int v = ranks_bev[interval_starts[i]];
  if ((unsigned)v < (unsigned)n_voxels) count[v] = len;
}

// The single-grid layout of the writer / reader over a caller-provided state (nzoff, nzvox: kept for the backward) and
// scratch (count | scan_state, offset, vsum).
int fused_layout(int c, int batch, int dz, int dy, int dx, int n_intervals, void* state, size_t state_bytes, void* scratch,
                 size_t scratch_bytes, dhd::Layout* L, size_t* state_need, size_t* scratch_need) {
  using dhd::align_up;
  if (c != dhd::kTileC || batch <= 0 || dz <= 0 || dy <= 0 || dx <= 0 || n_intervals < 0) return DHD_EINVAL;
  if (dy % dhd::kSegRows || dx % 4 || dx * dhd::kSegRows > dhd::kSegMaxVox) return DHD_EUNSUPPORTED;
  const long v = (long)batch * dz * dy * dx;
  if (v > (1L << 30)) return DHD_EUNSUPPORTED;
  *L = dhd::Layout{};
  L->B = batch; L->C = c; L->G = 1; L->V = (int)v; L->R = (int)(v / dx);
  L->grid[0].n[0] = dx; L->grid[0].n[1] = dy; L->grid[0].n[2] = dz;
  for (int g = 1; g <= DHD_MAX_GRIDS; ++g) { L->vox_base[g] = L->V; L->row_base[g] = L->R; L->seg_base[g] = L->R / dhd::kSegRows; }
  L->compact = 1;
  L->n_segs = L->R / dhd::kSegRows;
  L->n_chunks = dhd_cdiv(v, dhd::kChunk);
  L->n_slots_max = n_intervals;
  size_t off = 0;
  char* base = static_cast<char*>(scratch);
  auto carve = [&](size_t n_words) { int* p = reinterpret_cast<int*>(base + off); off = align_up(off + n_words * 4, 256); return p; };
  L->count = carve((size_t)v);
  L->scan_state = reinterpret_cast<unsigned long long*>(carve(2 * (size_t)L->n_chunks));
  L->zero_bytes = off;
  L->offset = carve((size_t)v + 1);
  L->vsum = reinterpret_cast<float*>(carve(((size_t)n_intervals + 1) * dhd::kTileC));
  *scratch_need = off;
  off = 0;
  base = static_cast<char*>(state);
  L->nzoff = carve((size_t)v + 1);
  L->nzvox = carve((size_t)n_intervals + 1);
  *state_need = off;
  if (state || scratch) {
    if (!state || !scratch) return DHD_EINVAL;
    if ((reinterpret_cast<uintptr_t>(state) | reinterpret_cast<uintptr_t>(scratch)) & 255) return DHD_EINVAL;
    if (state_bytes < *state_need || scratch_bytes < *scratch_need) return DHD_ENOSPACE;
  }
  return DHD_OK;
}

// Channel parts per segment.  Measured on the full-height grid (200 x 200 x 1, B = 4: 200 segments, half of the voxels
// occupied): writer 17.0 / 21.7 / 32.7 us and reader 21.8 / 23.6 / 31.7 us with 4 / 8 / 16 parts (reader whole: 23.7) -- every
// part rebuilds the segment's slot map, which is what a small, dense grid pays for, not idle CUs.
constexpr int kFusedSplit = 4;

// (B, C, Dz, Dy, Dx): element (b, z, c, y, x) at b * C*Dz*plane + c * Dz*plane + z * plane + y * Dx + x
template <class Ptrs, class T>
void fused_view(const dhd::Layout& L, T* p, Ptrs* o) {
  *o = Ptrs{};
  const long plane = (long)L.grid[0].n[1] * L.grid[0].n[0], nz = L.grid[0].n[2];
  o->dtype = DHD_F32;
  o->p[0] = p; o->sb[0] = L.C * nz * plane; o->sc[0] = nz * plane; o->sz[0] = plane;
}

}  // namespace

extern "C" {

int dhd_bev_pool_v2_fused_workspace_bytes(int c, int batch, int dz, int dy, int dx, int n_intervals, size_t* state_bytes,
                                          size_t* scratch_bytes) {
  if (!state_bytes || !scratch_bytes) return DHD_EINVAL;
  dhd::Layout L;
  return fused_layout(c, batch, dz, dy, dx, n_intervals, nullptr, 0, nullptr, 0, &L, state_bytes, scratch_bytes);
}

int dhd_bev_pool_v2_fused_forward(const float* depth, const float* feat, float* out, const int32_t* ranks_depth,
                                  const int32_t* ranks_feat, const int32_t* ranks_bev, const int32_t* interval_lengths,
                                  const int32_t* interval_starts, int c, int n_intervals, int batch, int dz, int dy, int dx,
                                  void* state, size_t state_bytes, int state_valid, void* scratch, size_t scratch_bytes,
                                  void* stream) {
  dhd::Layout L;
  size_t sn, cn;
  if (!state || !scratch) return DHD_EINVAL;
  int rc = fused_layout(c, batch, dz, dy, dx, n_intervals, state, state_bytes, scratch, scratch_bytes, &L, &sn, &cn);
  if (rc) return rc;
  if (!out || (reinterpret_cast<uintptr_t>(out) & 15)) return DHD_EINVAL;
  if (n_intervals > 0 && (!depth || !feat || !ranks_depth || !ranks_feat || !ranks_bev || !interval_lengths || !interval_starts))
    return DHD_EINVAL;
  if (n_intervals > 0 && (reinterpret_cast<uintptr_t>(feat) & 15)) return DHD_EINVAL;
  hipStream_t st = dhd_stream(stream);
  if (!state_valid) {
    // voxel -> row map of these index lists (a static rig passes the same lists every call: the caller may keep `state`)
    DHD_HIP(hipMemsetAsync(L.count, 0, L.zero_bytes, st));
    if (n_intervals > 0) {
      hipLaunchKernelGGL(fused_mark_kernel, dim3(dhd_cdiv(n_intervals, kBlock)), dim3(kBlock), 0, st, n_intervals, L.V, ranks_bev,
                         interval_starts, interval_lengths, L.count);
      DHD_LAUNCH_CHECK();
    }
    if ((rc = dhd::launch_scan(L, st))) return rc;
  }
  if (n_intervals > 0) {
    constexpr int LL = dhd::kTileC / 4;
    hipLaunchKernelGGL((bev_pool_v2_fwd_vec_kernel<LL, 8, 6, true>), dim3(xcd_padded_blocks(dhd_cdiv(n_intervals, kWaves * (DHD_WAVE / LL)))),
                       dim3(kBlock), 0, st, n_intervals, depth, reinterpret_cast<const pf4*>(feat), ranks_depth, ranks_feat, ranks_bev,
                       interval_starts, interval_lengths, reinterpret_cast<pf4*>(L.vsum), L.nzoff, L.V);
    DHD_LAUNCH_CHECK();
  }
  dhd::OutPtrs o;
  fused_view<dhd::OutPtrs, float>(L, out, &o);
  return dhd::launch_stream_fwd(L, o, kFusedSplit, st, true);
}

int dhd_bev_pool_v2_fused_backward(const float* out_grad, float* depth_grad, float* feat_grad, const float* depth,
                                   const float* feat, const int32_t* ranks_depth, const int32_t* ranks_feat,
                                   const int32_t* ranks_bev, const int32_t* interval_lengths_bp,
                                   const int32_t* interval_starts_bp, int c, int n_intervals_bp, int n_intervals, int batch, int dz,
                                   int dy, int dx, void* state, size_t state_bytes, void* scratch, size_t scratch_bytes,
                                   void* stream) {
  dhd::Layout L;
  size_t sn, cn;
  if (!state || !scratch || n_intervals_bp < 0) return DHD_EINVAL;
  int rc = fused_layout(c, batch, dz, dy, dx, n_intervals, state, state_bytes, scratch, scratch_bytes, &L, &sn, &cn);
  if (rc) return rc;
  if (n_intervals_bp == 0 || n_intervals == 0) return DHD_OK;
  if (!out_grad || !depth_grad || !feat_grad || !depth || !feat || !ranks_depth || !ranks_feat || !ranks_bev ||
      !interval_lengths_bp || !interval_starts_bp)
    return DHD_EINVAL;
  if ((reinterpret_cast<uintptr_t>(out_grad) | reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(feat_grad)) & 15)
    return DHD_EINVAL;
  hipStream_t st = dhd_stream(stream);
  dhd::InPtrs in;
  fused_view<dhd::InPtrs, const float>(L, out_grad, &in);
  if ((rc = dhd::launch_stream_bwd(L, in, kFusedSplit, st, true))) return rc;
  constexpr int LL = dhd::kTileC / 4;
  hipLaunchKernelGGL((bev_pool_v2_bwd_vec_kernel<LL, 8, 5, true>), dim3(xcd_padded_blocks(dhd_cdiv(n_intervals_bp, kWaves * (DHD_WAVE / LL)))),
                     dim3(kBlock), 0, st, n_intervals_bp, reinterpret_cast<const pf4*>(L.vsum), depth, reinterpret_cast<const pf4*>(feat),
                     ranks_depth, ranks_feat, ranks_bev, interval_starts_bp, interval_lengths_bp, depth_grad,
                     reinterpret_cast<pf4*>(feat_grad), L.nzoff, L.V);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

int dhd_bev_pool_v2_forward(const float* depth, const float* feat, float* out, const int32_t* ranks_depth,
                            const int32_t* ranks_feat, const int32_t* ranks_bev, const int32_t* interval_lengths,
                            const int32_t* interval_starts, int c, int n_intervals, void* stream) {
  if (c <= 0 || n_intervals < 0) return DHD_EINVAL;
  if (n_intervals == 0) return DHD_OK;
  if (!depth || !feat || !out || !ranks_depth || !ranks_feat || !ranks_bev || !interval_lengths || !interval_starts)
    return DHD_EINVAL;
  const int lv = (((uintptr_t)feat | (uintptr_t)out) & 15) == 0 ? vec_lanes(c) : 0;
  // rows in flight per lane / waves per SIMD, measured at DHD-S B = 4: (16, 3) 28.8 us, (8, 6) 25.7, (4, 8) 26.2
#define DHD_FWD_VEC(LL) DHD_FWD_VEC_R(LL, 8, 6)
#define DHD_FWD_VEC_R(LL, RR, WW)                                                                                                       \
  hipLaunchKernelGGL((bev_pool_v2_fwd_vec_kernel<LL, RR, WW>), dim3(xcd_padded_blocks(dhd_cdiv(n_intervals, kWaves * (DHD_WAVE / LL)))), dim3(kBlock), 0, \
                     dhd_stream(stream), n_intervals, depth, reinterpret_cast<const pf4*>(feat), ranks_depth, ranks_feat,    \
                     ranks_bev, interval_starts, interval_lengths, reinterpret_cast<pf4*>(out))
  switch (lv) {
    case 1: DHD_FWD_VEC(1); break;
    case 2: DHD_FWD_VEC(2); break;
    case 4: DHD_FWD_VEC(4); break;
    case 8: DHD_FWD_VEC(8); break;
    case 16: DHD_FWD_VEC(16); break;
    case 32: DHD_FWD_VEC(32); break;
    case 64: DHD_FWD_VEC(64); break;
    default:
      hipLaunchKernelGGL(bev_pool_v2_fwd_kernel, dim3(dhd_cdiv(n_intervals, kWaves)), dim3(kBlock), 0, dhd_stream(stream), c,
                         n_intervals, depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts, interval_lengths, out);
  }
#undef DHD_FWD_VEC
#undef DHD_FWD_VEC_R
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

size_t dhd_bev_pool_v2_regroup_scratch_bytes(int n_points, int n_pixels) {
  if (n_points < 0 || n_pixels <= 0) return 0;
  return 4 * (regroup_align((size_t)n_pixels) + 3 * regroup_align((size_t)n_points));
}

int dhd_bev_pool_v2_regroup(const int32_t* ranks_depth, const int32_t* ranks_feat, const int32_t* ranks_bev, int n_points,
                            int n_pixels, int32_t* ranks_depth_bp, int32_t* ranks_feat_bp, int32_t* ranks_bev_bp,
                            int32_t* interval_starts_bp, int32_t* interval_lengths_bp, void* scratch, size_t scratch_bytes,
                            void* stream) {
  if (n_points < 0 || n_pixels <= 0) return DHD_EINVAL;
  if (!interval_starts_bp || !interval_lengths_bp || !scratch) return DHD_EINVAL;
  if (n_points > 0 && (!ranks_depth || !ranks_feat || !ranks_bev || !ranks_depth_bp || !ranks_feat_bp || !ranks_bev_bp))
    return DHD_EINVAL;
  if (scratch_bytes < dhd_bev_pool_v2_regroup_scratch_bytes(n_points, n_pixels)) return DHD_ENOSPACE;
  hipStream_t st = dhd_stream(stream);
  int* count = static_cast<int*>(scratch);
  int* rnk = count + regroup_align((size_t)n_pixels);
  int* t_rd = rnk + regroup_align((size_t)n_points);
  int* t_rb = t_rd + regroup_align((size_t)n_points);
  DHD_HIP(hipMemsetAsync(count, 0, (size_t)n_pixels * 4, st));
  if (n_points > 0) {
    hipLaunchKernelGGL(regroup_count_kernel, dim3(dhd_cdiv(n_points, kBlock)), dim3(kBlock), 0, st, ranks_feat, n_points, n_pixels, count,
                       rnk);
    DHD_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(regroup_scan_kernel, dim3(1), dim3(kScanBlock), 0, st, count, n_pixels, interval_starts_bp, interval_lengths_bp);
  DHD_LAUNCH_CHECK();
  if (n_points > 0) {
    hipLaunchKernelGGL(regroup_scatter_kernel, dim3(dhd_cdiv(n_points, kBlock)), dim3(kBlock), 0, st, ranks_depth, ranks_feat, ranks_bev,
                       rnk, interval_starts_bp, n_points, t_rd, t_rb);
    hipLaunchKernelGGL(regroup_order_kernel, dim3(dhd_cdiv(n_pixels, kBlock / DHD_WAVE)), dim3(kBlock), 0, st, t_rd, t_rb,
                       interval_starts_bp, interval_lengths_bp, n_pixels, ranks_depth_bp, ranks_feat_bp, ranks_bev_bp);
    DHD_LAUNCH_CHECK();
  }
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
  const int lv = (((uintptr_t)feat | (uintptr_t)out_grad | (uintptr_t)feat_grad) & 15) == 0 ? vec_lanes(c) : 0;
  // (16, 3) 17.9 us, (8, 5) 16.3, (4, 6) 16.6
#define DHD_BWD_VEC(LL) DHD_BWD_VEC_R(LL, 8, 5)
#define DHD_BWD_VEC_R(LL, RR, WW)                                                                                                       \
  hipLaunchKernelGGL((bev_pool_v2_bwd_vec_kernel<LL, RR, WW>), dim3(xcd_padded_blocks(dhd_cdiv(n_intervals_bp, kWaves * (DHD_WAVE / LL)))), dim3(kBlock), 0, \
                     dhd_stream(stream), n_intervals_bp, reinterpret_cast<const pf4*>(out_grad), depth,                          \
                     reinterpret_cast<const pf4*>(feat), ranks_depth, ranks_feat, ranks_bev, interval_starts_bp,                 \
                     interval_lengths_bp, depth_grad, reinterpret_cast<pf4*>(feat_grad))
  switch (lv) {
    case 1: DHD_BWD_VEC(1); break;
    case 2: DHD_BWD_VEC(2); break;
    case 4: DHD_BWD_VEC(4); break;
    case 8: DHD_BWD_VEC(8); break;
    case 16: DHD_BWD_VEC(16); break;
    case 32: DHD_BWD_VEC(32); break;
    case 64: DHD_BWD_VEC(64); break;
    default:
      if (c <= 64) DHD_BWD(bev_pool_v2_bwd_fast_kernel<1>);
      else if (c <= 128) DHD_BWD(bev_pool_v2_bwd_fast_kernel<2>);
      else DHD_BWD(bev_pool_v2_bwd_kernel);
  }
#undef DHD_BWD_VEC
#undef DHD_BWD_VEC_R
#undef DHD_BWD
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

}  // extern "C"
