// 1x1 convolution of the SFA stage as a GEMM with ONE compute unit per pixel tile ("cu" kernels; bf16x3, C = 128 / 256).
//
// Reference: models/necks/mix.py:51 (spacial_leanring: conv1x1 - BN - ReLU - conv1x1 - BN) and its backward.
//
// pw_gemm_res_kernel (sfa_stage.hip) keeps the bf16 weight fragments of 128 output channels in LDS, so two workgroups
// ("team") on one XCD share every pixel tile: each activation byte is requested, prologue'd and split twice, and the SQ
// counters show the waves waiting for exactly those loads (profiles/r3/sfa_gemm_sq_counters.txt).  The limit there is LDS
// (128 KB of fragments per 128 channels); the register file of a CU is 512 KB.  Here
//   * the weights live in REGISTERS: a wave owns 32*MT output channels and holds their MFMA A fragments for all K
//     (16 K-steps x 2 bf16 parts x 4 VGPRs = 128 VGPRs per 32 channels), loaded once per workgroup lifetime;
//   * a workgroup (one per CU, persistent) owns whole 32-pixel tiles: its waves load the tile's C input rows ONCE
//     (row-wise 16-byte loads: 8 lanes = one 128-byte line), apply the affine prologue  act(c0*in0 + c1*in1 + c2),
//     cut the result into two bf16 parts and store it pixel-major into a double-buffered LDS tile (2 x 32 KB at C = 256);
//   * every wave then reads the tile's B fragments (one ds_read_b128 per K-step and part) against its own weights:
//     D[channel][pixel] += W_h X_h + W_h X_m + W_m X_h  (v_mfma_f32_32x32x16_bf16);
//   * the epilogue transposes a wave's 32 x 32 result through a wave-private LDS patch so that stores are whole lines
//     (16 bytes per lane, 8 lanes per line) and a lane sees 4 pixels of ONE channel: bias, BatchNorm partial sums
//     (kept per lane over all tiles of the workgroup, one row per workgroup at the end) and the ReLU pass bits are cheap there.
// One barrier per tile; R (1 or 2) register sets of raw rows: the loads of tile t + 1 + R are issued as soon as tile t + 1
// has been staged, so with R = 2 a CU always has at least one tile's loads in flight.
//
// LDS tile layout: [part][pixel p][k] bf16, row pitch 2*C bytes, the 16-byte unit u of a row stored at u ^ swz(p).  With
// swz(p) = ((p>>2) ^ (p&1)) & 7 | ((p>>1)&1) << 3 both the 8-byte stores of the staging lanes (lane = (row quad, pixel quad))
// and the 16-byte fragment reads (lane = (pixel, k half)) are bank-conflict free (tests/test_host_logic.py simulates both
// against the bank rules of MI355X_MICROARCH.md).
#pragma once
#include "sfa_mfma.h"

namespace dhd_sfa {

constexpr int kCuPatchPitch = 36;   // floats per channel row of a wave's 32 x 32 store patch

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence over ALL address spaces: hipcc
// puts `s_waitcnt vmcnt(0)` in front of its s_barrier, which would drain the activation loads in flight for the next tiles
// (and the result stores) once per tile -- measured: loads and compute strictly one after the other, 106 us instead of 7x us.
__device__ __forceinline__ void cu_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int cu_swz_c(int p) { return ((((p >> 2) & 7) ^ (p & 1)) | (((p >> 1) & 1) << 3)); }
__host__ __device__ inline int cu_swz(int p) { return ((((p >> 2) & 7) ^ (p & 1)) | (((p >> 1) & 1) << 3)); }

// dynamic LDS of a workgroup: the double-buffered activation tile + one store patch per wave + the prologue coefficient
// tables [nb][3][c] of the launch's samples + the bias [c]
inline size_t cu_lds_bytes(int c, int waves, int nb) {
  return (size_t)2 * 2 * 32 * 2 * c + (size_t)waves * 32 * kCuPatchPitch * sizeof(float) + ((size_t)nb * 3 + 1) * c * sizeof(float);
}
// samples per launch: as many as have their coefficient tables next to the tiles and patches in 160 KB of LDS
inline int cu_max_batch(int c, int waves) { return (int)((160 * 1024 - cu_lds_bytes(c, waves, 0)) / ((size_t)3 * c * sizeof(float))); }

// ReLU pass bits of the cu kernels: one 16-bit word per (tile, 32-row group, staging lane); bit 4*j + e = row 4*g + j of
// the group, pixel 4*q + e of the tile (lane = 8*g + q): one bit per activation, C 32-bit words per tile.
inline size_t cu_mask_words(int nb, int c, int hw) { return (size_t)nb * ((hw + 31) / 32) * c; }

// Weight M (rows x k, or its transpose) -> MFMA A fragments in the two-part split, one 32-channel tile after the other:
//   wp[((ct * KCN + ks) * 2 + part) * 64 + lane] = part(M[32 ct + (lane & 31)][16 ks + 8 (lane >> 5) + j]), j = 0..7
// (thread idx = (ct, ks, lane)).
__device__ __forceinline__ void cu_pack_weight(const float* __restrict__ w, int transpose, u32x4* __restrict__ wp, int c, int idx) {
  const int kcn = c / 16;
  if (idx >= (c / 32) * kcn * 64) return;
  const int lane = idx & 63, ks = (idx >> 6) % kcn, ct = (idx >> 6) / kcn;
  const int row = 32 * ct + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
  u32x4 hi, mid;
#pragma unroll
  for (int jp = 0; jp < 4; ++jp) {
    const int k = k0 + 2 * jp;
    const float a = transpose ? w[(size_t)k * c + row] : w[(size_t)row * c + k];
    const float b = transpose ? w[(size_t)(k + 1) * c + row] : w[(size_t)row * c + k + 1];
    unsigned hh, mm;
    split2_hm(a, b, hh, mm);
    hi[jp] = hh; mid[jp] = mm;
  }
  u32x4* dst = wp + ((size_t)(ct * kcn + ks) * 2) * 64 + lane;
  dst[0] = hi;
  dst[64] = mid;
}

// EPI: 0 forward (+ bias, BatchNorm partial sums of the un-biased result), 1 data gradient with the recorded ReLU pass bits,
// 2 plain.  RECORD (with RELU): the prologue leaves the pass bits of its ReLU for the backward's EPI 1.
// ABL (experiments/gemm_cu_bench.hip only; 0 in the product): 1 no MFMAs, 2 no result stores, 4 no epilogue at all,
// 8 no activation loads (stale registers are staged), 16 no staging, 32 no B-fragment LDS reads (MFMAs on registers),
// 64 shader clocks of the workgroup (s_memtime) into stat_part, 128 s_sleep in place of the MFMAs (with 1), 256 shader clocks per
// phase and wave into stat_part
template <int KCN, int WAVES, bool TWO_IN, bool RELU, int EPI, bool RECORD, int AUX, int R, int NACC, int ABL = 0, int SAUX = 0, bool PP = false,
          int BPF = 0, int EORD = 0>
__global__ __launch_bounds__(WAVES * 64, 1) void pw_gemm_cu_kernel(const float* __restrict__ in0, const float* __restrict__ in1,
                                                                   size_t in_bstride, unsigned in_bytes, const float* __restrict__ coef,
                                                                   const u32x4* __restrict__ wp, const float* __restrict__ bias,
                                                                   unsigned* __restrict__ relu_mask, float* __restrict__ stat_part,
                                                                   float* __restrict__ y, int hw, int nb, int contig) {
  constexpr int C = 16 * KCN;
  constexpr int MT = C / (32 * WAVES);          // 32-channel output tiles (= 32-row input groups) per wave
  static_assert(MT >= 1 && MT * 32 * WAVES == C, "C = 32 * MT * WAVES");
  constexpr int ROWB = 2 * C, PARTB = 32 * ROWB, BUFB = 2 * PARTB;
  extern __shared__ __attribute__((aligned(16))) unsigned char cu_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 3, q = lane & 7;        // staging role: rows 4g..4g+3 of a 32-row group, pixels 4q..4q+3; store role: channels g + 8k
  const int n = lane & 31, h = lane >> 5;       // MFMA role: pixel n, k half h (B operand) / channel half h (result)
  const int kbase = wv * 32 * MT;               // the input rows this wave stages == the output channels it computes
  // PP ("ping-pong"): the upper half of the waves (the second wave of every SIMD) runs the same loop ONE BARRIER LATER, so that
  // on each SIMD one wave is in its MFMA phase while the other stages / issues loads / stores results
  const int grp = PP ? __builtin_amdgcn_readfirstlane((int)(wv >= WAVES / 2)) : 0;
  static_assert(!PP || R == 1, "ping-pong: one register set");
  float* patch = reinterpret_cast<float*>(cu_lds + 2 * BUFB) + wv * (32 * kCuPatchPitch);

  // ---- weights -> registers (A fragments: lane = (channel n of the 32-tile, k half h)) -------------------------------
  u32x4 wh[KCN][MT], wm[KCN][MT];
#pragma unroll
  for (int ks = 0; ks < KCN; ++ks)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const u32x4* src = wp + ((size_t)((wv * MT + mt) * KCN + ks) * 2) * 64 + lane;
      wh[ks][mt] = src[0];
      wm[ks][mt] = src[64];
    }

  const long long abl_clk0 = (ABL & 64) ? (long long)__builtin_readcyclecounter() : 0;
  const int nwt = (hw + 31) >> 5, total = nb * nwt;
  const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t0 = contig ? (int)blockIdx.x * per : (int)blockIdx.x;
  const int t_end = contig ? min(total, t0 + per) : total;
  const int t_step = contig ? 1 : (int)gridDim.x;
  const int n_mine = t0 < t_end ? (t_end - t0 + t_step - 1) / t_step : 0;   // tiles of this workgroup
  const int t_last = t0 + (n_mine - 1) * t_step;
  const int row_bytes = hw * 4;

  // ---- per-lane constants -------------------------------------------------------------------------------------------
  const int ld_voff = ((kbase + 4 * g) * hw + 4 * q) * 4;            // staging loads: row kbase + 4g (+ j, + 32 rg as scalars)
  // staging stores: (row group rg, pixel 4q + e) -> byte in a part = p * ROWB + ((unit ^ swz(p)) << 4) + 8 * (kq & 1), with
  // swz(4q + e) = swz(4q) ^ swz(e) (cu_swz is linear over bit vectors): one base per row group, the pixel's part as an XOR
  // with a constant and an immediate offset
  int wbase[MT];
#pragma unroll
  for (int rg = 0; rg < MT; ++rg) {
    const int kq = (kbase >> 2) + 8 * rg + g;                        // 8-byte unit of the row
    wbase[rg] = 4 * q * ROWB + (((kq >> 1) ^ cu_swz(4 * q)) << 4) + ((kq & 1) << 3);
  }
  const int rbase = n * ROWB + ((h ^ cu_swz(n)) << 4);               // fragment reads: unit (2 ks + h) ^ swz(n)
  const int st_voff = ((kbase + g) * hw + 4 * q) * 4;                // result stores: channel kbase + g (+ 8k, + 32 mt), pixels 4q..

  // prologue coefficients of all samples -> LDS (read per tile: 12 registers less than keeping a sample's rows resident, and no
  // conditional reload inside the tile loop)
  float* cf_lds = reinterpret_cast<float*>(cu_lds + 2 * BUFB) + WAVES * (32 * kCuPatchPitch);
  for (int i = tid; i < nb * 3 * C; i += WAVES * 64) cf_lds[i] = coef[i];

  // THE TILE LOOP IS BRANCH-FREE ON PURPOSE.  hipcc's wait-count pass merges its per-register "pending load" state at every
  // control-flow join and gives up precision there: with `if (next tile exists) { stage; issue }` inside the loop it put
  // s_waitcnt vmcnt(7..0) in front of every staging -- a full drain of the queue, including the loads just issued for the tile
  // after next and the previous tile's stores, i.e. loads and compute strictly one after the other (measured with the MFMAs
  // replaced by s_sleep: same time).  So: every iteration stages and issues (past the workgroup's last tile: that tile again,
  // harmless), stores are never skipped (a lane outside the row, or a repeated tile, stores out of the buffer's range, which
  // the hardware drops), and the loop runs an even number of iterations.
  // raw rows in flight / being staged: R register sets, tile i of the workgroup uses set i % R
  f32x4 r0[R][MT][4], r1[R][MT][4];
  auto issue = [&](auto slotc, int t) {
    constexpr int S = decltype(slotc)::value;
    if (ABL & 8) return;
    t = min(t, t_last);
    const int b = t / nwt, wt = t - b * nwt, p0 = wt * 32;
    const __amdgpu_buffer_rsrc_t s0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in0 + (size_t)b * in_bstride), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t s1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((TWO_IN ? in1 : in0) + (size_t)b * in_bstride), 0, in_bytes, 0x00020000);
    // a pixel quad beyond the row's end (last tile, hw % 32 != 0) re-reads quad 0: its columns are never stored or summed
    const int voff = (p0 + 4 * q < hw) ? ld_voff : ld_voff - 16 * q;
#pragma unroll
    for (int rg = 0; rg < MT; ++rg)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int soff = (32 * rg + j) * row_bytes + p0 * 4;
        r0[S][rg][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s0, voff, soff, AUX));
        if (TWO_IN) r1[S][rg][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s1, voff, soff, AUX));
      }
  };

  // prologue + split + store of register set S into tile buffer `buf` (per lane: 4 rows x 4 pixels per row group; a store
  // takes the 4 rows of one pixel = 8 bytes per part); RECORD: the pass bits of tile t
  float abl_sink = 0.f;
  auto stage = [&](auto slotc, int buf, int t) {
    constexpr int S = decltype(slotc)::value;
    t = min(t, t_last);
    unsigned char* dst = cu_lds + buf * BUFB;
    const float* cb = cf_lds + (t / nwt) * 3 * C + kbase + 4 * g;
#pragma unroll
    for (int rg = 0; rg < MT; ++rg) {
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(cb + 32 * rg);
      const f32x4 c1 = TWO_IN ? *reinterpret_cast<const f32x4*>(cb + C + 32 * rg) : c0;
      const f32x4 c2 = *reinterpret_cast<const f32x4*>(cb + 2 * C + 32 * rg);
      unsigned bits = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (ABL & 16) {                                              // keep the loads alive
#pragma unroll
          for (int j = 0; j < 4; ++j) abl_sink += r0[S][rg][j][e] + (TWO_IN ? r1[S][rg][j][e] : 0.f);
          continue;
        }
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float tv = fmaf(c0[j], r0[S][rg][j][e], c2[j]);
          if (TWO_IN) tv = fmaf(c1[j], r1[S][rg][j][e], tv);
          v[j] = RELU ? fmaxf(tv, 0.f) : tv;
          if (RECORD) bits |= min(__float_as_uint(v[j]), 1u) << (4 * j + e);   // v >= 0 here: v > 0 <=> its bits != 0
        }
        unsigned h01, m01, h23, m23;
        split2_hm(v[0], v[1], h01, m01);
        split2_hm(v[2], v[3], h23, m23);
        const u32x2 hq = {h01, h23}, mq = {m01, m23};
        int wa;
        asm("v_xor_b32 %0, %1, %2" : "=v"(wa) : "n"(cu_swz_c(e) << 4), "v"(wbase[rg]));
        *reinterpret_cast<u32x2*>(dst + wa + e * ROWB) = hq;
        *reinterpret_cast<u32x2*>(dst + wa + e * ROWB + PARTB) = mq;
      }
      if (RECORD && !(ABL & 16))
        reinterpret_cast<unsigned short*>(relu_mask)[((size_t)t * (C / 32) + wv * MT + rg) * 64 + lane] = (unsigned short)bits;
    }
  };

  // BatchNorm partial sums (EPI 0): channels kbase + 32 mt + g + 8 k, this lane's pixel quads, all tiles of the workgroup
  float ws1[MT][4], ws2[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int k = 0; k < 4; ++k) ws1[mt][k] = ws2[mt][k] = 0.f;
  // the bias goes to LDS behind the coefficient tables (read per tile in the epilogue: 4 registers less)
  float* bias_lds = cf_lds + nb * 3 * C;
  if (EPI == 0)
    for (int i = tid; i < C; i += WAVES * 64) bias_lds[i] = bias[i];

  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // Everything requested so far (weights, bias, coefficient table) is complete before the tile loop, unconditionally: otherwise
  // the wait-count pass assumes the 32 weight loads may still be pending at the loop header and puts  s_waitcnt vmcnt(30) ...
  // vmcnt(0)  in front of the K-steps of every tile.  vmcnt(0), expcnt / lgkmcnt untouched (gfx9 encoding); the LDS barrier
  // publishes the coefficient table.
  __builtin_amdgcn_s_waitcnt(0x0F70);
  cu_lds_barrier();

  // One tile: MFMAs of tile t from buffer PAR, then the staging of tile t + 1 (register set (PAR + 1) % R) into the other
  // buffer and the loads of tile t + 1 + R into the set just freed, then the epilogue of tile t.  `live`: t is one of the
  // workgroup's tiles (the last iteration of an odd count repeats the last tile without storing or summing it).
  unsigned ph[6] = {0, 0, 0, 0, 0, 0};                              // ABL & 256: shader clocks per phase, summed over the tiles
  unsigned ph_t = 0;
  auto stamp = [&](int i) {
    if (!(ABL & 256)) return;
    const unsigned now = (unsigned)__builtin_readcyclecounter();
    ph[i] += now - ph_t;
    ph_t = now;
  };
  auto tile = [&](auto parc, int t, bool live) {
    constexpr int PAR = decltype(parc)::value;
    stamp(5);                                                        // (barrier wait of the previous tile)
    constexpr int SN = (PAR + 1) % R;                                // register set of the next tile
    t = min(t, t_last);
    const int b = t / nwt, wt = t - b * nwt, p0 = wt * 32;
    unsigned mask_r[MT][4];
    if (EPI == 1) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          mask_r[mt][k] = reinterpret_cast<const unsigned short*>(relu_mask)[((size_t)t * (C / 32) + wv * MT + mt) * 64 + 8 * ((g >> 2) + 2 * k) + q];
    }
    // ---- MFMA phase: D[channel][pixel] over all K ---------------------------------------------------------------------
    const unsigned char* src = cu_lds + PAR * BUFB;
    // NACC = 2: acc holds W_h X_h, acc2 the two small terms W_m X_h + W_h X_m (two independent MFMA chains, 16 more registers)
    f32x16 acc[MT], acc2[NACC == 2 ? MT : 1];
    // B fragments are requested BPF K-steps ahead of their MFMAs (BPF = 2: 8 more registers; in the ping-pong form a wave has no
    // partner's MFMAs to cover its LDS latency during the MFMA phase)
    u32x4 bfh[BPF + 1], bfm[BPF + 1];
    auto frag = [&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      if (BPF == 0 && ks > 0) return;
      // (the XOR as an opaque instruction: hipcc would otherwise keep all eight swizzled addresses in registers)
      int a;
      asm("v_xor_b32 %0, %1, %2" : "=v"(a) : "n"(((2 * ks) & 15) << 4), "v"(rbase));
      a += (ks >> 3) * 256;
      if (ABL & 32) {                                                // no fragment reads: the MFMAs run on a register
        bfh[ks % (BPF + 1)] = wh[(ks + 1) % KCN][0]; bfm[ks % (BPF + 1)] = wm[(ks + 1) % KCN][0];
      } else {
        bfh[ks % (BPF + 1)] = *reinterpret_cast<const u32x4*>(src + a);
        bfm[ks % (BPF + 1)] = *reinterpret_cast<const u32x4*>(src + PARTB + a);
      }
    };
    static_for<BPF>([&](auto ksc) { frag(ksc); });
    static_for<KCN>([&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      u32x4 bh, bm;
      if constexpr (BPF == 0) {                                      // requested where they are used: the compiler's own schedule
        int a;
        asm("v_xor_b32 %0, %1, %2" : "=v"(a) : "n"(((2 * ks) & 15) << 4), "v"(rbase));
        a += (ks >> 3) * 256;
        bh = *reinterpret_cast<const u32x4*>(src + a);
        bm = *reinterpret_cast<const u32x4*>(src + PARTB + a);
      } else {
        if constexpr (ks + BPF < KCN) frag(std::integral_constant<int, ks + BPF>{});
        bh = bfh[ks % (BPF + 1)]; bm = bfm[ks % (BPF + 1)];
      }
      if (ABL & 128) __builtin_amdgcn_s_sleep(4);                    // (with ABL & 1) the MFMA phase's duration without its MFMAs
      if (ABL & 1) {
        if (ks == 0) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) { acc[mt] = zero + bh[0]; if (NACC == 2) acc2[mt] = zero + bm[0]; }
        }
        return;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (NACC == 2) {
          acc2[mt] = mfma_bf16(wm[ks][mt], bh, ks == 0 ? zero : acc2[mt]);
          acc[mt] = mfma_bf16(wh[ks][mt], bh, ks == 0 ? zero : acc[mt]);
          acc2[mt] = mfma_bf16(wh[ks][mt], bm, acc2[mt]);
        } else {                                                     // smallest terms first
          acc[mt] = mfma_bf16(wm[ks][mt], bh, ks == 0 ? zero : acc[mt]);
          acc[mt] = mfma_bf16(wh[ks][mt], bm, acc[mt]);
          acc[mt] = mfma_bf16(wh[ks][mt], bh, acc[mt]);
        }
      }
    });
    stamp(0);                                                        // MFMA phase (issue; the last MFMAs may still run)
    if (PP) cu_lds_barrier();
    auto next_tile = [&]() {
    // ---- next tile: registers -> other buffer, then the set is free for the tile R further on ------------------------------
    // (ping-pong: the late group is a whole tile ahead with its staging -- tile t + 2 into the buffer it has just read)
    stage(std::integral_constant<int, SN>{}, (PAR ^ 1) ^ grp, t + (1 + grp) * t_step);
    stamp(1);                                                        // wait for the loads + staging
    issue(std::integral_constant<int, SN>{}, t + (1 + grp + R) * t_step);
    stamp(2);                                                        // load issue
    };
    auto epilogue = [&]() {
    // ---- epilogue: transposition through the wave's patch, row-wise 16-byte stores -------------------------------------
    const __amdgpu_buffer_rsrc_t ry =
        __builtin_amdgcn_make_buffer_rsrc(y + (size_t)b * C * hw, 0, (unsigned)((size_t)C * hw * sizeof(float)), 0x00020000);
    const bool quad_ok = live && p0 + 4 * q < hw;                    // hw % 4 == 0: a pixel quad is inside or outside as a whole
    const int voff_st = quad_ok ? st_voff : 0x7ffffff0;              // beyond the buffer's range: the store is dropped
    if (ABL & 4) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int v = 0; v < 16; ++v) abl_sink += acc[mt][v] + (NACC == 2 ? acc2[mt][v] : 0.f);
    }
#pragma unroll
    for (int mt = 0; mt < ((ABL & 4) ? 0 : MT); ++mt) {
#pragma unroll
      for (int v = 0; v < 16; ++v)                                   // acc[v]: channel row 8 (v>>2) + 4 h + (v&3), pixel n
        patch[(8 * (v >> 2) + 4 * h + (v & 3)) * kCuPatchPitch + n] = NACC == 2 ? acc[mt][v] + acc2[mt][v] : acc[mt][v];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f32x4 o = *reinterpret_cast<const f32x4*>(patch + (g + 8 * k) * kCuPatchPitch + 4 * q);
        if (EPI == 0) {
          const float s1 = (o.x + o.y) + (o.z + o.w), s2 = (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
          ws1[mt][k] += quad_ok ? s1 : 0.f;
          ws2[mt][k] += quad_ok ? s2 : 0.f;
          const float bs = bias_lds[kbase + 32 * mt + g + 8 * k];
          o.x += bs; o.y += bs; o.z += bs; o.w += bs;
        }
        if (EPI == 1) {
          const int w4 = (int)(mask_r[mt][k] >> (4 * (g & 3)));      // bit e = pixel 4q + e of channel g + 8k
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = __int_as_float(__float_as_int(o[e]) & __builtin_amdgcn_sbfe(w4, e, 1));
        }
        const int soff = (32 * mt + 8 * k) * row_bytes + p0 * 4;
        if (!(ABL & 2)) store_b128_guarded<SAUX>(__builtin_bit_cast(u32x4, o), ry, voff_st, soff);
      }
    }
    };
    // EORD = 1: the epilogue (which needs nothing from memory) before the staging (which waits for the next tile's loads)
    if (EORD == 0) { next_tile(); epilogue(); } else { epilogue(); next_tile(); }
    stamp(3);                                                        // epilogue incl. store issue
    cu_lds_barrier();
  };

  if (n_mine > 0) {                                                  // workgroup-uniform
    // The pipeline's fill issues the same sequence of vector-memory operations as a steady-state iteration (the loads of a
    // tile, then as many stores as an epilogue -- here out of range, i.e. dropped): the wait-count pass merges the state at the
    // loop header with the back edge's, and where the two disagree it assumes the FEWER operations behind a pending load, i.e.
    // it would make the first staging of every iteration wait for half of the following tile's loads as well.
    auto fill_stores = [&]() {
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y, 0, 16, 0x00020000);
#pragma unroll
      for (int i = 0; i < 4 * MT; ++i)
        if (!(ABL & 2)) store_b128_guarded<SAUX>(u32x4{0u, 0u, 0u, 0u}, ry, 0x7ffffff0, 0);
    };
    issue(std::integral_constant<int, 0>{}, t0);
    if (R == 2) {
      issue(std::integral_constant<int, 1 % R>{}, t0 + t_step);
      fill_stores();
    }
    stage(std::integral_constant<int, 0>{}, 0, t0);
    issue(std::integral_constant<int, 0>{}, t0 + R * t_step);
    fill_stores();
    cu_lds_barrier();
    if (PP && grp) {                                                 // the late group's first phase: its rows of tile 1
      stage(std::integral_constant<int, 0>{}, 1, t0 + t_step);
      issue(std::integral_constant<int, 0>{}, t0 + 2 * t_step);
      fill_stores();
      cu_lds_barrier();
    }
    for (int i = 0, t = t0; i < n_mine; i += 2, t += 2 * t_step) {
      tile(std::integral_constant<int, 0>{}, t, true);
      tile(std::integral_constant<int, 1>{}, t + t_step, i + 1 < n_mine);
    }
    if (PP && !grp) cu_lds_barrier();                                // the early group meets the late group's last barrier
  }

  if ((ABL & 256) && lane == 0) {
    unsigned* o = reinterpret_cast<unsigned*>(stat_part) + 1024 + ((size_t)blockIdx.x * WAVES + wv) * 8;
    for (int i = 0; i < 6; ++i) o[i] = ph[i];
    o[6] = (unsigned)n_mine;
  }
  if ((ABL & 64) && tid == 0) reinterpret_cast<long long*>(stat_part)[blockIdx.x] = (long long)__builtin_readcyclecounter() - abl_clk0;
  if (EPI == 0 && stat_part != nullptr) {                            // one row [2][C] per workgroup
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float s1 = ws1[mt][k], s2 = ws2[mt][k];
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {
          s1 += __shfl_xor(s1, m, DHD_WAVE);
          s2 += __shfl_xor(s2, m, DHD_WAVE);
        }
        if (q == 0) {
          float* row = stat_part + (size_t)blockIdx.x * 2 * C + kbase + 32 * mt + g + 8 * k;
          row[0] = s1;
          row[C] = s2;
        }
      }
  }
}

}  // namespace dhd_sfa
