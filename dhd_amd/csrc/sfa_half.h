// SFA attention stage with HALF STORAGE (dhd_sfa_weights.storage_dtype = DHD_F16 / DHD_BF16): the form a caller inside an
// autocast region gets (DHD-S.py:281 `fp16 = dict(loss_scale='dynamic')`; BASELINE configs[1] fp16, [3] / [4] bf16).
//
// Reference: models/necks/mix.py:37-59.  Under autocast the reference's own formulation keeps every (B,C,H,W) tensor of the
// stage in the half type -- x arrives in half from the encoders, both 1x1 convolutions (mix.py:51) run and store in half, and
// so does every tensor autograd saves or passes back.  Here: x, y1, y2, the ReLU pass bits, g2, g1, du, out, gout and gx
// are stored in the half type TS; everything BETWEEN two stores is float32 -- the affine prologues, the MFMA accumulation
// (v_mfma_f32_32x32x16_f16 / _bf16: ONE product per a*b, both operands rounded to TS once), the BatchNorm statistics (taken
// from the rounded values that are stored, so that forward, backward and the sums they share see the same tensor), the
// coefficient tables, the parameters and every parameter gradient.  That is never less accurate than autocast's op-by-op
// rounding, and it halves the bytes the float32 stage moves (11 + 28 tensor passes of B*C*HW elements per forward + backward).
//
// Kernels:
//   pw_gemm_cuh_kernel  the 1x1 convolution as a GEMM with one CU per 64-PIXEL tile (a pixel row of a tile = one 128-byte
//                       line of TS): the structure of pw_gemm_cu_kernel (sfa_gemm_cu.h: weights in registers, persistent
//                       workgroup, swizzled double-buffered LDS tile, ping-pong wave groups, branch-free tile loop), with
//                       a single operand part -- 64 weight registers per wave instead of 128, one LDS read per MFMA.
//   pw_wgrad_h_kernel   weight gradients, 64-pixel steps, one 256 x 256 (128 x 128) output tile per CU, per-worker partials.
//   *_h element-wise passes: 8 elements (16 bytes) per lane and load.
#pragma once
#include "sfa_gemm_cu.h"

namespace dhd_sfa {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;

template <class TS> struct HalfOps;
template <> struct HalfOps<_Float16> {
  static __device__ __forceinline__ f32x2 widen2(unsigned w) {
    const f16x2 h = __builtin_bit_cast(f16x2, w);
    return f32x2{(float)h.x, (float)h.y};
  }
  static __device__ __forceinline__ unsigned narrow2(f32x2 v) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
  }
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <> struct HalfOps<__bf16> {
  static __device__ __forceinline__ f32x2 widen2(unsigned w) { return unpack_bf16(w); }
  static __device__ __forceinline__ unsigned narrow2(f32x2 v) { return pack_bf16(v); }
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) { return mfma_bf16(a, b, c); }
};

// 8 consecutive elements of TS (16 bytes) <-> 8 floats (widened exactly / rounded to nearest even)
template <class TS> __device__ __forceinline__ void widen8(u32x4 w, float* v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 p = HalfOps<TS>::widen2(w[i]);
    v[2 * i] = p.x;
    v[2 * i + 1] = p.y;
  }
}
template <class TS> __device__ __forceinline__ u32x4 narrow8(const float* v) {
  u32x4 w;
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = HalfOps<TS>::narrow2(f32x2{v[2 * i], v[2 * i + 1]});
  return w;
}
template <class TS> __device__ __forceinline__ void ld8(const TS* base, size_t i8, float* v) {
  widen8<TS>(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base) + i8), v);
}
template <class TS> __device__ __forceinline__ void st8(TS* base, size_t i8, const float* v) {
  __builtin_nontemporal_store(narrow8<TS>(v), reinterpret_cast<u32x4*>(base) + i8);
}
// the value a float becomes when it is stored as TS and read back
template <class TS> __device__ __forceinline__ f32x2 round2(f32x2 v) { return HalfOps<TS>::widen2(HalfOps<TS>::narrow2(v)); }

// ------------------------------------------------------------------------------------------------------------------------
// pw_gemm_cuh_kernel
// ------------------------------------------------------------------------------------------------------------------------

constexpr int kCuhTile = 64;          // pixels per tile: a row of a tile is one 128-byte line of TS
constexpr int kCuhPatchPitch = 68;    // floats per channel row of a wave's 32 x 64 store patch

// LDS tile [pixel p][k] of TS, row pitch 2*C bytes, the 16-byte unit u of a row stored at u ^ cuh_swz(p).  Linear over bit
// vectors (cuh_swz(8q + e) = cuh_swz(8q) ^ cuh_swz(e)); with it both the 8-byte staging stores (lane = (row quad g, pixel
// oct q), one pixel 8q + e per instruction) and the 16-byte fragment reads (lane = (pixel n, k half h), rows n and n + 32) are
// bank-conflict free (tests/test_host_logic.py simulates both against MI355X_MICROARCH.md's rules).
constexpr int cuh_swz_c(int p) { return (((p >> 3) & 7) ^ ((p & 1) << 2)) | (((p >> 1) & 1) << 3); }
__host__ __device__ inline int cuh_swz(int p) { return (((p >> 3) & 7) ^ ((p & 1) << 2)) | (((p >> 1) & 1) << 3); }

inline size_t cuh_lds_bytes(int c, int waves, int nb) {
  return (size_t)2 * kCuhTile * 2 * c + (size_t)waves * 32 * kCuhPatchPitch * sizeof(float) + ((size_t)nb * 3 + 1) * c * sizeof(float);
}
inline int cuh_max_batch(int c, int waves) { return (int)((160 * 1024 - cuh_lds_bytes(c, waves, 0)) / ((size_t)3 * c * sizeof(float))); }
// ReLU pass bits: one 32-bit word per (tile, 32-row group, staging lane); bit 8 j + e = row 4 g + j of the group, pixel 8 q + e
inline size_t cuh_mask_words(int nb, int c, int hw) { return (size_t)nb * ((hw + kCuhTile - 1) / kCuhTile) * (c / 32) * 64; }

// Weight M (rows x k, or its transpose), float32 -> MFMA A fragments of TS, one 32-channel tile after the other:
//   wp[(ct * KCN + ks) * 64 + lane] = TS(M[32 ct + (lane & 31)][16 ks + 8 (lane >> 5) + j]), j = 0..7     (thread idx = (ct, ks, lane))
template <class TS>
__device__ __forceinline__ void cuh_pack_weight(const float* __restrict__ w, int transpose, u32x4* __restrict__ wp, int c, int idx) {
  const int kcn = c / 16;
  if (idx >= (c / 32) * kcn * 64) return;
  const int lane = idx & 63, ks = (idx >> 6) % kcn, ct = (idx >> 6) / kcn;
  const int row = 32 * ct + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = transpose ? w[(size_t)(k0 + j) * c + row] : w[(size_t)row * c + k0 + j];
  wp[(size_t)(ct * kcn + ks) * 64 + lane] = narrow8<TS>(v);
}

// y[b, co, p] = TS( sum_ci TS(W[co, ci]) * TS(act(c0[b,ci]*in0[b,ci,p] + c1[b,ci]*in1[b,ci,p] + c2[b,ci])) (+ epilogue) )
// EPI: 0 forward (+ bias; BatchNorm partial sums of the STORED values, shifted by the bias), 1 data gradient with the recorded
// ReLU pass bits, 2 plain.  RECORD (with RELU): the prologue leaves the pass bits of its ReLU.  EORD: epilogue before (1) or
// after (0) the staging of the next tile.
template <class TS, int KCN, int WAVES, bool TWO_IN, bool RELU, int EPI, bool RECORD, int EORD>
__global__ __launch_bounds__(WAVES * 64, 1) void pw_gemm_cuh_kernel(const TS* __restrict__ in0, const TS* __restrict__ in1, size_t in_bstride,
                                                                    unsigned in_bytes, const float* __restrict__ coef,
                                                                    const u32x4* __restrict__ wp, const float* __restrict__ bias,
                                                                    unsigned* __restrict__ relu_mask, float* __restrict__ stat_part,
                                                                    TS* __restrict__ y, int hw, int nb) {
  constexpr int C = 16 * KCN;
  static_assert(C == 32 * WAVES, "a wave owns 32 input rows / output channels");
  constexpr int ROWB = 2 * C, HALFB = 32 * ROWB, BUFB = kCuhTile * ROWB;
  constexpr int PP = kCuhPatchPitch;
  extern __shared__ __attribute__((aligned(16))) unsigned char cu_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 3, q = lane & 7;        // staging: rows 4g..4g+3 of the wave's 32, pixels 8q..8q+7; stores: channels g + 8k
  const int n = lane & 31, h = lane >> 5;       // MFMA: pixel n (+ 32 s), k half h (B operand) / channel half h (result)
  const int kbase = wv * 32;
  const int grp = __builtin_amdgcn_readfirstlane((int)(wv >= WAVES / 2));   // ping-pong: the late group runs one barrier behind
  float* patch = reinterpret_cast<float*>(cu_lds + 2 * BUFB) + wv * (32 * PP);

  u32x4 wf[KCN];                                 // weights -> registers (A fragments: lane = (channel n, k half h))
#pragma unroll
  for (int ks = 0; ks < KCN; ++ks) wf[ks] = wp[(size_t)(wv * KCN + ks) * 64 + lane];

  const int nwt = (hw + kCuhTile - 1) / kCuhTile, total = nb * nwt;
  const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t0 = (int)blockIdx.x * per;          // a contiguous range of tiles per workgroup
  const int t_end = min(total, t0 + per);
  const int n_mine = t0 < t_end ? t_end - t0 : 0;
  const int t_last = t0 + n_mine - 1;
  const int row_bytes = hw * 2;

  const int ld_voff = ((kbase + 4 * g) * hw + 8 * q) * 2;
  const int kq = (kbase >> 2) + g;                                   // 8-byte unit of a pixel row this lane writes
  const int wbase = 8 * q * ROWB + (((kq >> 1) ^ cuh_swz(8 * q)) << 4) + ((kq & 1) << 3);
  const int rbase = n * ROWB + ((h ^ cuh_swz(n)) << 4);              // fragment reads: unit (2 ks + h) ^ swz(n + 32 s)
  const int st_voff = ((kbase + g) * hw + 8 * q) * 2;

  float* cf_lds = reinterpret_cast<float*>(cu_lds + 2 * BUFB) + WAVES * (32 * PP);
  for (int i = tid; i < nb * 3 * C; i += WAVES * 64) cf_lds[i] = coef[i];
  float* bias_lds = cf_lds + nb * 3 * C;
  if (EPI == 0)
    for (int i = tid; i < C; i += WAVES * 64) bias_lds[i] = bias[i];

  // (the tile loop is branch-free and the pipeline fill mirrors a steady-state iteration: see pw_gemm_cu_kernel)
  u32x4 r0[4], r1[4];
  auto issue = [&](int t) {
    t = min(t, t_last);
    const int b = t / nwt, wt = t - b * nwt, p0 = wt * kCuhTile;
    const __amdgpu_buffer_rsrc_t s0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<TS*>(in0 + (size_t)b * in_bstride), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t s1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<TS*>((TWO_IN ? in1 : in0) + (size_t)b * in_bstride), 0, in_bytes, 0x00020000);
    // a pixel oct beyond the row's end (last tile, hw % 64 != 0) re-reads oct 0: its columns are never stored or summed
    const int voff = (p0 + 8 * q < hw) ? ld_voff : ld_voff - 16 * q;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int soff = j * row_bytes + p0 * 2;
      r0[j] = __builtin_amdgcn_raw_buffer_load_b128(s0, voff, soff, 2);
      if (TWO_IN) r1[j] = __builtin_amdgcn_raw_buffer_load_b128(s1, voff, soff, 2);
    }
  };

  auto stage = [&](int buf, int t) {
    t = min(t, t_last);
    unsigned char* dst = cu_lds + buf * BUFB;
    const float* cb = cf_lds + (t / nwt) * 3 * C + kbase + 4 * g;
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(cb);
    const f32x4 c1 = TWO_IN ? *reinterpret_cast<const f32x4*>(cb + C) : c0;
    const f32x4 c2 = *reinterpret_cast<const f32x4*>(cb + 2 * C);
    unsigned bits = 0;
    // two pixels (one 32-bit word of each of the four rows) at a time: 8 live values
#pragma unroll
    for (int ep = 0; ep < 4; ++ep) {
      f32x2 tv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x2 t2 = __builtin_elementwise_fma(f32x2{c0[j], c0[j]}, HalfOps<TS>::widen2(r0[j][ep]), f32x2{c2[j], c2[j]});
        if (TWO_IN) t2 = __builtin_elementwise_fma(f32x2{c1[j], c1[j]}, HalfOps<TS>::widen2(r1[j][ep]), t2);
        if (RELU) { t2.x = fmaxf(t2.x, 0.f); t2.y = fmaxf(t2.y, 0.f); }
        if (RECORD)   // v >= 0 here: v > 0 <=> its bits != 0
          bits |= (min(__float_as_uint(t2.x), 1u) << (8 * j + 2 * ep)) | (min(__float_as_uint(t2.y), 1u) << (8 * j + 2 * ep + 1));
        tv[j] = t2;
      }
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const int e = 2 * ep + o;
        const u32x2 pk = {HalfOps<TS>::narrow2(f32x2{tv[0][o], tv[1][o]}), HalfOps<TS>::narrow2(f32x2{tv[2][o], tv[3][o]})};
        int wa;
        asm("v_xor_b32 %0, %1, %2" : "=v"(wa) : "n"(cuh_swz_c(e) << 4), "v"(wbase));
        *reinterpret_cast<u32x2*>(dst + wa + e * ROWB) = pk;
      }
    }
    if (RECORD) relu_mask[((size_t)t * (C / 32) + wv) * 64 + lane] = bits;
  };

  float ws1[4], ws2[4];                          // BatchNorm partial sums (EPI 0): channel kbase + g + 8 k, this lane's pixel octs
#pragma unroll
  for (int k = 0; k < 4; ++k) ws1[k] = ws2[k] = 0.f;

  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): weights, tables
  cu_lds_barrier();

  auto tile = [&](auto parc, int t, bool live) {
    constexpr int PAR = decltype(parc)::value;
    t = min(t, t_last);
    const int b = t / nwt, wt = t - b * nwt, p0 = wt * kCuhTile;
    unsigned mask_r[4];
    if (EPI == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) mask_r[k] = relu_mask[((size_t)t * (C / 32) + wv) * 64 + 8 * ((g >> 2) + 2 * k) + q];
    }
    // ---- MFMA phase: D[channel][pixel] over all K, two 32-pixel halves -------------------------------------------------
    const unsigned char* src = cu_lds + PAR * BUFB;
    f32x16 acc[2];
    static_for<KCN>([&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      int a0, a1;
      asm("v_xor_b32 %0, %1, %2" : "=v"(a0) : "n"(((2 * ks) & 15) << 4), "v"(rbase));
      asm("v_xor_b32 %0, %1, %2" : "=v"(a1) : "n"((((2 * ks) & 15) ^ cuh_swz_c(32)) << 4), "v"(rbase));
      const u32x4 b0 = *reinterpret_cast<const u32x4*>(src + a0 + (ks >> 3) * 256);
      const u32x4 b1 = *reinterpret_cast<const u32x4*>(src + a1 + (ks >> 3) * 256 + HALFB);
      acc[0] = HalfOps<TS>::mfma(wf[ks], b0, ks == 0 ? zero : acc[0]);
      acc[1] = HalfOps<TS>::mfma(wf[ks], b1, ks == 0 ? zero : acc[1]);
      // The MFMA phase is a small part of a tile's time (32 MFMAs against ~8 k clocks of HBM time per tile and CU): fragments are
      // requested four K-steps at a time, so that the scheduler does not hoist all 2 KCN reads (128 registers) to the top
      if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    });
    cu_lds_barrier();
    auto next_tile = [&]() {
      // the late group is a whole tile ahead with its staging: tile t + 2 into the buffer it has just read
      stage((PAR ^ 1) ^ grp, t + 1 + grp);
      issue(t + 2 + grp);
    };
    auto epilogue = [&]() {
      const __amdgpu_buffer_rsrc_t ry =
          __builtin_amdgcn_make_buffer_rsrc(y + (size_t)b * C * hw, 0, (unsigned)((size_t)C * hw * sizeof(TS)), 0x00020000);
      const bool oct_ok = live && p0 + 8 * q < hw;                   // hw % 8 == 0: a pixel oct is inside or outside as a whole
      const int voff_st = oct_ok ? st_voff : 0x7ffffff0;             // beyond the buffer's range: the store is dropped
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int v = 0; v < 16; ++v)                                 // acc[s][v]: channel row 8 (v>>2) + 4 h + (v&3), pixel n + 32 s
          patch[(8 * (v >> 2) + 4 * h + (v & 3)) * PP + n + 32 * s] = acc[s][v];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(patch + (g + 8 * k) * PP + 8 * q);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(patch + (g + 8 * k) * PP + 8 * q + 4);
        float o[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (EPI == 1) {
          const int w8 = (int)(mask_r[k] >> (8 * (g & 3)));          // bit e = pixel 8q + e of channel g + 8k
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = __int_as_float(__float_as_int(o[e]) & __builtin_amdgcn_sbfe(w8, e, 1));
        }
        u32x4 pk;
        if (EPI == 0) {
          const float bs = bias_lds[kbase + g + 8 * k];
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            pk[i] = HalfOps<TS>::narrow2(f32x2{o[2 * i] + bs, o[2 * i + 1] + bs});
            const f32x2 d = HalfOps<TS>::widen2(pk[i]) - f32x2{bs, bs};   // what is stored, shifted by the bias
            s1 += d.x + d.y;
            s2 += d.x * d.x + d.y * d.y;
          }
          ws1[k] += oct_ok ? s1 : 0.f;
          ws2[k] += oct_ok ? s2 : 0.f;
        } else {
          pk = narrow8<TS>(o);
        }
        const int soff = 8 * k * row_bytes + p0 * 2;
        store_b128_guarded<0>(pk, ry, voff_st, soff);
      }
    };
    if (EORD == 0) { next_tile(); epilogue(); } else { epilogue(); next_tile(); }
    cu_lds_barrier();
  };

  if (n_mine > 0) {                                                  // workgroup-uniform
    auto fill_stores = [&]() {
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y, 0, 16, 0x00020000);
#pragma unroll
      for (int i = 0; i < 4; ++i) store_b128_guarded<0>(u32x4{0u, 0u, 0u, 0u}, ry, 0x7ffffff0, 0);
    };
    issue(t0);
    stage(0, t0);
    issue(t0 + 1);
    fill_stores();
    cu_lds_barrier();
    if (grp) {                                                       // the late group's first phase: its rows of tile 1
      stage(1, t0 + 1);
      issue(t0 + 2);
      fill_stores();
      cu_lds_barrier();
    }
    for (int i = 0, t = t0; i < n_mine; i += 2, t += 2) {
      tile(std::integral_constant<int, 0>{}, t, true);
      tile(std::integral_constant<int, 1>{}, t + 1, i + 1 < n_mine);
    }
    if (!grp) cu_lds_barrier();                                      // the early group meets the late group's last barrier
  }

  if (EPI == 0 && stat_part != nullptr) {                            // one row [2][C] per workgroup
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float s1 = ws1[k], s2 = ws2[k];
#pragma unroll
      for (int m = 1; m < 8; m <<= 1) {
        s1 += __shfl_xor(s1, m, DHD_WAVE);
        s2 += __shfl_xor(s2, m, DHD_WAVE);
      }
      if (q == 0) {
        float* row = stat_part + (size_t)blockIdx.x * 2 * C + kbase + g + 8 * k;
        row[0] = s1;
        row[C] = s2;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// weight gradients
// ------------------------------------------------------------------------------------------------------------------------

// G[co][ci] = sum_{b,p} TS(A(co,p)) * TS(B(ci,p)), A / B = act(c0*in0 + c1*in1 + c2) per (sample, row).  Pixels are the MFMA k
// dimension.  A step is 64 pixels = four MFMA k-steps: every channel row contributes one whole 128-byte line per step, loaded
// line-coalesced (8 adjacent lanes = one row's line); a lane's 16 bytes are 8 consecutive pixels of one row = exactly one
// lane's MFMA operand fragment, so the loading thread applies the prologue, rounds to TS once and writes the fragment to LDS in
// fragment order.  The eight lanes of a row write eight fragments whose LDS units are congruent modulo 8 (an 8-way conflict
// on ds_write_b128), so unit u of fragment (k-step ks, half hh) is stored at u ^ (2 ks + hh); the reads XOR the same constant.
// Block = 8 waves, output tile OT x OT (wave: OT/2 x OT/4), double-buffered LDS (2 x 64 KB at OT = 256), one barrier per step.
// Workers own contiguous step ranges; per-worker partial matrices are reduced by wgrad_reduce_kernel.
template <class TS, int OT, bool B_TWO, bool B_RELU>
__global__ __launch_bounds__(512, 1) void pw_wgrad_h_kernel(const TS* __restrict__ a0, const TS* __restrict__ a1,
                                                            const float* __restrict__ acoef, size_t a_bstride,
                                                            const TS* __restrict__ b0, const TS* __restrict__ b1,
                                                            const float* __restrict__ bcoef, size_t b_bstride,
                                                            float* __restrict__ partial, int c, int hw, int nb, int n_workers) {
  constexpr int TA = OT / 64, TB = OT / 128;     // 32x32 tiles per wave
  constexpr int kTiles = OT / 32;                // 32-row tiles per operand
  constexpr int kOp = kTiles * 64;               // 16-byte units of one staged operand k-step: [tile][lane]
  constexpr int kBuf = 4 * 2 * kOp;              // [k-step 4][operand 2]
  constexpr int NJ = OT / 64;                    // load instructions per operand input and step (64 rows each)
  extern __shared__ u32x4 ldsh[];                // [buf 2][k-step 4][operand 2][tile][lane]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nob = c / OT;
  const int ob_co = (blockIdx.y / nob) * OT, ob_ci = (blockIdx.y % nob) * OT;
  const int wco = (wv >> 2) * (OT / 2), wci = (wv & 3) * (OT / 4);
  const int sps = (hw + 63) >> 6;                // steps per sample
  const long n_steps = (long)nb * sps;
  const int w_id = blockIdx.x;
  const int first = (int)(n_steps * w_id / n_workers);
  const int count = (int)(n_steps * (w_id + 1) / n_workers) - first;
  auto step_of = [&](int k) { return first + min(k, count - 1); };   // past the end: the last step again

  // loads: instruction j reads rows 64 j + 8 wv + (lane >> 3), pixels 8 (lane & 7) .. + 7
  const int ld_row = 8 * wv + (lane >> 3), ch8 = lane & 7;
  const int it_ks = ch8 >> 1, it_h = ch8 & 1;    // MFMA k-step and lane half of the item

  f32x16 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  u32x4 raw[2][2][NJ];                           // [operand][input][load instruction]
  float cfa[NJ][3], cfb[NJ][3];                  // prologue coefficients of this thread's rows
  int cur_b = -1;
  auto load_coefs = [&](int b) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row = 64 * j + ld_row;
      const float* ca = acoef + (size_t)b * 3 * c + ob_co + row;
      const float* cb = bcoef + (size_t)b * 3 * c + ob_ci + row;
#pragma unroll
      for (int k = 0; k < 3; ++k) { cfa[j][k] = ca[k * c]; cfb[j][k] = cb[k * c]; }
    }
    cur_b = b;
  };
  auto fetch = [&](int s) {
    const int b = s / sps, p = (s % sps) * 64 + 8 * ch8;
    const size_t off = p < hw ? p : 0;
#pragma unroll
    for (int op = 0; op < 2; ++op) {
      const TS* src0 = op ? b0 : a0;
      const TS* src1 = op ? b1 : a1;
      const bool two = op ? B_TWO : true;
      const size_t base = (size_t)b * (op ? b_bstride : a_bstride) + (size_t)((op ? ob_ci : ob_co) + ld_row) * hw + off;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        raw[op][0][j] = *reinterpret_cast<const u32x4*>(src0 + base + (size_t)(64 * j) * hw);
        if (two) raw[op][1][j] = *reinterpret_cast<const u32x4*>(src1 + base + (size_t)(64 * j) * hw);
      }
    }
  };
  auto stage = [&](int s, int buf) {
    const int b = s / sps, p = (s % sps) * 64 + 8 * ch8;
    if (b != cur_b) load_coefs(b);               // block-uniform, a few times per worker
    const bool in = p < hw;                      // hw % 8 == 0: an item is inside or outside as a whole
#pragma unroll
    for (int op = 0; op < 2; ++op) {
      const bool two = op ? B_TWO : true;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float k0 = op ? cfb[j][0] : cfa[j][0], k1 = op ? cfb[j][1] : cfa[j][1], k2 = op ? cfb[j][2] : cfa[j][2];
        float x0[8], x1[8], v[8];
        widen8<TS>(raw[op][0][j], x0);
        if (two) widen8<TS>(raw[op][1][j], x1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = fmaf(k0, x0[e], k2);
          if (two) t = fmaf(k1, x1[e], t);
          if (op == 1 && B_RELU) t = fmaxf(t, 0.f);
          v[e] = in ? t : 0.f;
        }
        const int row = 64 * j + ld_row;
        const int unit = (row >> 5) * 64 + (((row & 31) + 32 * it_h) ^ ch8);
        ldsh[buf * kBuf + (it_ks * 2 + op) * kOp + unit] = narrow8<TS>(v);
      }
    }
  };

  if (count > 0) {
    fetch(step_of(0));
    stage(step_of(0), 0);
    fetch(step_of(1));
  }
  __syncthreads();
  const int lx = lane ^ h;                       // reads: lane's unit ^ (2 ks + h)
  for (int k = 0; k < count; ++k) {
    const int buf = k & 1;
    stage(step_of(k + 1), buf ^ 1);              // unconditional (indices clamped; the re-staged copy of the last step is never read)
    fetch(step_of(k + 2));
    __builtin_amdgcn_sched_barrier(0);           // keep the loads of step k + 2 ahead of this step's MFMAs
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const u32x4* ta = ldsh + buf * kBuf + (ks * 2) * kOp + (lx ^ (2 * ks));
      const u32x4* tb = ta + kOp;
      u32x4 fb[TB];
#pragma unroll
      for (int j = 0; j < TB; ++j) fb[j] = tb[((wci >> 5) + j) * 64];
#pragma unroll
      for (int i = 0; i < TA; ++i) {
        const u32x4 fa = ta[((wco >> 5) + i) * 64];
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = HalfOps<TS>::mfma(fa, fb[j], acc[i][j]);
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

}  // namespace dhd_sfa
