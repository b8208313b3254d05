// SFA stage, 1x1 convolutions: the STREAMED GEMM families of rounds 1-2 (weights pass through LDS once per 128-pixel tile) and
// their weight-gradient kernels.  Included by sfa_stage.hip inside its unnamed namespace (after kPwBlock / kPwStep / kWgBlock /
// kWgStride / kEwBlock).  Reference: models/necks/mix.py:51 and its backward.
// Which calls still reach them (launch_pw_gemm / launch_pw_wgrad in sfa_stage.hip):
//   pw_gemm_kernel / pw_wgrad_kernel      gemm = DHD_SFA_GEMM_F32 at every supported C: the plain float32-MFMA reference point of
//                                         the precision table (tests/test_gpu_parity.py::test_sfa_stage_f32_mfma_mode_vs_torch)
//   pw_gemm6_kernel                       gemm = DHD_SFA_GEMM_BF16X6 at C = 512 and multiples of 256 beyond (three bf16 parts of 512
//                                         channels do not fit the LDS-resident form)
//   pw_wgrad6_kernel                      gemm = DHD_SFA_GEMM_BF16X6 at every supported C
// The default precision (bf16x3) never runs a kernel of this file; half storage (sfa_half.h) neither.
// ------------------------------------------------------------------------------------------------
// 1x1 convolution on the f32 MFMA
// ------------------------------------------------------------------------------------------------

// Weight (rows x k, row-major; or its transpose) -> LDS images for the MFMA A operand, one image
// per 16 input channels:
//   packed[((((rb*KC + kc)*COT + t)*2 + s4)*64 + lane)*4 + s] = M[rb*32*COT + 32 t + (lane&31)][16 kc + 8 s4 + 2 s + (lane>>5)]
// so that a wave reading (t, s4) with one ds_read_b128 per lane gets the A fragments of four
// consecutive 32x32x2 MFMAs.  M = W (forward, rows = output channels) or W^T (dgrad).
__global__ __launch_bounds__(kEwBlock) void pack_weight_kernel(const float* __restrict__ w, int transpose, float* __restrict__ packed,
                                                               int c, int cot) {
  const int idx = blockIdx.x * kEwBlock + threadIdx.x;
  if (idx >= c * c) return;
  const int kcn = c / kPwStep;
  int q = idx;
  const int s = q & 3; q >>= 2;
  const int lane = q & 63; q >>= 6;
  const int s4 = q & 1; q >>= 1;
  const int t = q % cot; q /= cot;
  const int kc = q % kcn;
  const int rb = q / kcn;
  const int row = rb * 32 * cot + 32 * t + (lane & 31);
  const int k = kPwStep * kc + 8 * s4 + 2 * s + (lane >> 5);
  packed[idx] = transpose ? w[(size_t)k * c + row] : w[(size_t)row * c + k];
}

// y[b, co, p] = sum_ci W[co, ci] * act(c0[b,ci]*in0[b,ci,p] + c1[b,ci]*in1[b,ci,p] + c2[b,ci])  (+ epilogue)
// EPI: 0 = + bias, 1 = ReLU mask from aux (keep where aux_sc*aux + aux_sh > 0), 2 = plain.
// A wave owns 32 pixels x 32*COT output channels; a step is 16 input channels = 8 k-pairs:
// 8 dword loads per input (issued one step ahead), 2*COT ds_read_b128, 8*COT MFMAs.  Two blocks per
// CU (<= 256 registers) so that one block's prologue / epilogue / barrier waits run under the other
// block's MFMAs.
template <int COT, bool TWO_IN, bool RELU, int EPI>
__global__ __launch_bounds__(kPwBlock, 2) void pw_gemm_kernel(const float* __restrict__ in0, const float* __restrict__ in1,
                                                              size_t in_bstride, const float* __restrict__ coef,
                                                              const float* __restrict__ wp, const float* __restrict__ bias,
                                                              const float* __restrict__ aux, const float* __restrict__ aux_scsh,
                                                              float* __restrict__ y, int c, int hw) {
  constexpr int kImg = COT * 2 * 64 * 4;  // floats per weight image (16 k x 32*COT rows)
  constexpr int kWst = COT / 2;           // float4 per thread per image
  constexpr int K2 = kPwStep / 2;         // k-pairs per step
  extern __shared__ float lds[];          // 2 images | coefficient table (3c)
  float* cf = lds + 2 * kImg;
  // blockIdx.x = (tile group of 8, row block, tile in group): the c/(32 COT) blocks that read the same
  // pixels are 8 apart in dispatch order, i.e. on the same XCD (shared L2) and close in time
  const int nrb = c / (32 * COT);
  const int b = blockIdx.y, rb = (blockIdx.x >> 3) % nrb;
  const int tile = (blockIdx.x / (8 * nrb)) * 8 + (blockIdx.x & 7);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  if (tile * (kPwBlock / DHD_WAVE) * 32 >= hw) return;  // padding tile of the last group (block-uniform)
  const int p = (tile * (kPwBlock / DHD_WAVE) + wv) * 32 + r;
  const bool live = p < hw;
  const int pc = live ? p : hw - 1;
  const int kcn = c / kPwStep;

  for (int i = tid; i < 3 * c; i += kPwBlock) cf[i] = coef[(size_t)b * 3 * c + i];

  const f32x4* wp4 = reinterpret_cast<const f32x4*>(wp) + (size_t)rb * kcn * (kImg / 4);
  f32x4 wst[kWst];
#pragma unroll
  for (int j = 0; j < kWst; ++j) wst[j] = wp4[j * kPwBlock + tid];
#pragma unroll
  for (int j = 0; j < kWst; ++j) reinterpret_cast<f32x4*>(lds)[j * kPwBlock + tid] = wst[j];

  const float* i0 = in0 + (size_t)b * in_bstride + pc + (size_t)h * hw;
  const float* i1 = TWO_IN ? in1 + (size_t)b * in_bstride + pc + (size_t)h * hw : nullptr;
  const size_t hw2 = (size_t)2 * hw;
  float raw0[K2], raw1[K2];
#pragma unroll
  for (int s = 0; s < K2; ++s) {
    raw0[s] = i0[s * hw2];
    if (TWO_IN) raw1[s] = i1[s * hw2];
  }
  __syncthreads();

  f32x16 acc[COT];
#pragma unroll
  for (int t = 0; t < COT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  for (int kc = 0; kc < kcn; ++kc) {
    float bv[K2];
#pragma unroll
    for (int s = 0; s < K2; ++s) {
      const int ci = kPwStep * kc + 2 * s + h;
      float v = fmaf(cf[ci], raw0[s], cf[2 * c + ci]);
      if (TWO_IN) v = fmaf(cf[c + ci], raw1[s], v);
      bv[s] = RELU ? fmaxf(v, 0.f) : v;
    }
    const bool more = kc + 1 < kcn;
    if (more) {
      i0 += (size_t)kPwStep * hw;
      if (TWO_IN) i1 += (size_t)kPwStep * hw;
#pragma unroll
      for (int s = 0; s < K2; ++s) {
        raw0[s] = i0[s * hw2];
        if (TWO_IN) raw1[s] = i1[s * hw2];
      }
#pragma unroll
      for (int j = 0; j < kWst; ++j) wst[j] = wp4[(size_t)(kc + 1) * (kImg / 4) + j * kPwBlock + tid];
    }
    const f32x4* img = reinterpret_cast<const f32x4*>(lds + (kc & 1) * kImg);
#pragma unroll
    for (int s4 = 0; s4 < 2; ++s4) {
#pragma unroll
      for (int t = 0; t < COT; ++t) {
        const f32x4 a4 = img[(t * 2 + s4) * 64 + lane];
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s], bv[4 * s4 + s], acc[t], 0, 0, 0);
      }
    }
    if (more) {
      f32x4* dst = reinterpret_cast<f32x4*>(lds + ((kc + 1) & 1) * kImg);
#pragma unroll
      for (int j = 0; j < kWst; ++j) dst[j * kPwBlock + tid] = wst[j];
    }
    __syncthreads();
  }

  if (!live) return;
  const int co0 = rb * 32 * COT + 4 * h;
  float* yo = y + (size_t)b * c * hw + p;
  const float* ao = EPI == 1 ? aux + (size_t)b * c * hw + p : nullptr;
#pragma unroll
  for (int t = 0; t < COT; ++t) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = co0 + 32 * t + (e & 3) + 8 * (e >> 2);
      float v = acc[t][e];
      if (EPI == 0) v += bias[co];
      if (EPI == 1) {
        const float m = fmaf(aux_scsh[co], ao[(size_t)co * hw], aux_scsh[c + co]);
        v = m > 0.f ? v : 0.f;
      }
      yo[(size_t)co * hw] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same GEMMs on the bf16 MFMA with a three-way split of every float32 operand ("bf16x6").
//
// A float32 x is cut by truncation into x = h + m + l, each part carrying 8 significand bits, i.e.
// each exactly a bfloat16 (h = x & 0xffff0000, m = (x - h) & 0xffff0000, l = x - h - m; the
// subtractions are exact).  Then a*b = ah*bh + (ah*bm + am*bh) + (ah*bl + al*bh + am*bm) + O(2^-25 |ab|):
// six bf16 products, each exact in float32, accumulated in float32 by v_mfma_f32_32x32x16_bf16.  That is
// float32-level accuracy (the dropped terms are below half an ulp of the product) at 6/16 of the
// f32-MFMA cost, which moves these K = C = 256 GEMMs from MFMA-bound to HBM-bound.
// NaN/Inf inputs propagate as NaN (Inf - Inf in the split) rather than Inf.
// ------------------------------------------------------------------------------------------------

// (vector types, split2 / split2_hm / mfma_bf16: sfa_mfma.h)

// Weight (rows x k; or its transpose) -> LDS images of the MFMA B operand, one image per 16 input
// channels, three bf16 terms:  packed16[(((rb*KC + kc)*COT + t)*3 + term)*64 + lane] (16-byte units) holds
//   term(M[rb*32*COT + 32 t + (lane&31)][16 kc + 8 (lane>>5) + j]),  j = 0..7
__global__ __launch_bounds__(kEwBlock) void pack_weight6_kernel(const float* __restrict__ w, int transpose, u32x4* __restrict__ packed,
                                                                int c, int cot) {
  const int idx = blockIdx.x * kEwBlock + threadIdx.x;  // (rb, kc, t, lane)
  const int kcn = c / 16;
  if (idx >= (c / 32) * kcn * 64) return;
  int q = idx;
  const int lane = q & 63; q >>= 6;
  const int t = q % cot; q /= cot;
  const int kc = q % kcn;
  const int rb = q / kcn;
  const int row = rb * 32 * cot + 32 * t + (lane & 31);
  const int k0 = 16 * kc + 8 * (lane >> 5);
  u32x4 h, m, l;
#pragma unroll
  for (int jp = 0; jp < 4; ++jp) {
    const int k = k0 + 2 * jp;
    const float a = transpose ? w[(size_t)k * c + row] : w[(size_t)row * c + k];
    const float b = transpose ? w[(size_t)(k + 1) * c + row] : w[(size_t)row * c + k + 1];
    unsigned hh, mm, ll;
    split2(a, b, hh, mm, ll);
    h[jp] = hh; m[jp] = mm; l[jp] = ll;
  }
  u32x4* dst = packed + ((size_t)((rb * kcn + kc) * cot + t) * 3) * 64 + lane;
  dst[0] = h;
  dst[64] = m;
  dst[128] = l;
}

// y[b, co, p] = sum_ci W[co, ci] * act(c0[b,ci]*in0[b,ci,p] + c1[b,ci]*in1[b,ci,p] + c2[b,ci])  (+ epilogue),
// EPI as in pw_gemm_kernel.  MFMA orientation D[pixel][channel]: the activation is the A operand (lane =
// pixel, 8 consecutive channels per half-wave, loaded as 8 dwords from NCHW rows: two 128-byte segments
// per load) and the weights the B operand, so a lane ends up with 4 consecutive pixels of one output
// channel per accumulator quad -> 16-byte stores along the pixel axis.
template <int COT, bool TWO_IN, bool RELU, int EPI>
__global__ __launch_bounds__(kPwBlock, 2) void pw_gemm6_kernel(const float* __restrict__ in0, const float* __restrict__ in1,
                                                               size_t in_bstride, unsigned in_bytes, const float* __restrict__ coef,
                                                               const u32x4* __restrict__ wp, const float* __restrict__ bias,
                                                               unsigned* __restrict__ relu_mask, float* __restrict__ stat_part,
                                                               float* __restrict__ y, int c, int hw, int tile0, int tile_end,
                                                               int borrowed) {
  constexpr int kImg = COT * 3 * 64;  // 16-byte units per weight image
  constexpr int kWst = kImg / kPwBlock;
  static_assert(kImg % kPwBlock == 0, "image must split evenly over the block");
  extern __shared__ u32x4 lds6[];     // 2 images | coefficient table (3c floats)
  float* cf = reinterpret_cast<float*>(lds6 + 2 * kImg);
  const int nrb = c / (32 * COT);
  const int b = blockIdx.y, rb = (blockIdx.x >> 3) % nrb;
  const int tile = tile0 + (blockIdx.x / (8 * nrb)) * 8 + (blockIdx.x & 7);  // this launch covers tiles [tile0, tile_end)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  if (tile >= tile_end || tile * (kPwBlock / DHD_WAVE) * 32 >= hw) return;  // padding tile of the last group (block-uniform)
  const int wt = tile * (kPwBlock / DHD_WAVE) + wv;     // 32-pixel wave tile
  const int nwt = (hw + 31) >> 5;
  const int p0 = wt * 32;
  const int pc = min(p0 + r, hw - 1);
  const int kcn = c / 16;

  for (int i = tid; i < 3 * c; i += kPwBlock) cf[i] = coef[(size_t)b * 3 * c + i];

  // `borrowed`: the weights were packed for row blocks of twice this kernel's COT (the tail launch reads the
  // main launch's images): this block's half of image (rb / 2, kc) starts kImg units in, images are 2 kImg apart
  const int wk = borrowed ? 2 * kImg : kImg;
  const u32x4* wsrc = borrowed ? wp + ((size_t)(rb >> 1) * kcn * 2 + (rb & 1)) * kImg : wp + (size_t)rb * kcn * kImg;
  u32x4 wst[kWst];
#pragma unroll
  for (int j = 0; j < kWst; ++j) wst[j] = wsrc[j * kPwBlock + tid];
#pragma unroll
  for (int j = 0; j < kWst; ++j) lds6[j * kPwBlock + tid] = wst[j];

  // activation rows through buffer loads: per-lane byte offset (pixel, half-wave's first channel) in a
  // VGPR, the channel row offset in an SGPR.  Loads run TWO steps ahead of their use (three register
  // sets): at ~2 us of loaded HBM latency one step of lookahead keeps only ~2 TB/s in flight.
  const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in0 + (size_t)b * in_bstride), 0, in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((TWO_IN ? in1 : in0) + (size_t)b * in_bstride), 0, in_bytes, 0x00020000);
  const int voff = (pc + 8 * h * hw) * 4;
  const int row_bytes = hw * 4;
  // Two register sets; the loads of step k+2 are issued into a set right after the prologue of step k has
  // consumed it.  Everything in the loop is unconditional (addresses clamped, kcn even, unrolled by two): a
  // conditional load or step makes the compiler copy loaded registers at the control-flow merge, and a copy
  // waits for its load -- that silently shortened the lookahead of an earlier three-set version to one step.
  float raw0[2][8], raw1[2][8];
  auto issue = [&](auto set, int kc) {
    constexpr int S = decltype(set)::value;
    const int so = 16 * kc * row_bytes;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // aux 2 = nt: the activations are streamed once per launch; keeping them out of the way of the weight
      // images in L2 is worth ~6 us per GEMM (stage 1.692 -> 1.669 ms)
      raw0[S][j] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r0, voff, so + j * row_bytes, 2));
      if (TWO_IN) raw1[S][j] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, voff, so + j * row_bytes, 2));
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  issue(I0{}, 0);
  issue(I1{}, 1);
  __syncthreads();

  f32x16 acc[COT];
#pragma unroll
  for (int t = 0; t < COT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  auto step = [&](auto cset, int kc) {
    constexpr int CS = decltype(cset)::value;
    u32x4 ah, am, al;
    {
      const int ci = 16 * kc + 8 * h;
      const f32x4* c0 = reinterpret_cast<const f32x4*>(cf + ci);
      const f32x4* c1 = reinterpret_cast<const f32x4*>(cf + c + ci);
      const f32x4* c2 = reinterpret_cast<const f32x4*>(cf + 2 * c + ci);
      float v[8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 k0 = c0[q], k2 = c2[q];
        f32x4 k1;
        if (TWO_IN) k1 = c1[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = fmaf(k0[e], raw0[CS][4 * q + e], k2[e]);
          if (TWO_IN) t = fmaf(k1[e], raw1[CS][4 * q + e], t);
          v[4 * q + e] = RELU ? fmaxf(t, 0.f) : t;
        }
      }
      if (RELU && relu_mask != nullptr && rb == 0 && wt < nwt) {  // wave-uniform
        // bit p of word (channel, wave tile) = this pixel's activation passed the ReLU; a ballot gives
        // the words of channels 16kc + j (low half-wave) and 16kc + 8 + j (high half-wave)
        unsigned word = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned long long bal = __ballot(v[j] > 0.f);
          if (lane == j) word = (unsigned)bal;
          if (lane == 8 + j) word = (unsigned)(bal >> 32);
        }
        if (lane < 16) relu_mask[((size_t)b * nwt + wt) * c + 16 * kc + lane] = word;   // [sample][wave tile][channel]: 64 contiguous bytes
      }
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        unsigned hh, mm, ll;
        split2(v[2 * jp], v[2 * jp + 1], hh, mm, ll);
        ah[jp] = hh; am[jp] = mm; al[jp] = ll;
      }
    }
    // next weight image (after the last step: a harmless reload that nobody reads)
#pragma unroll
    for (int j = 0; j < kWst; ++j) wst[j] = wsrc[(size_t)min(kc + 1, kcn - 1) * wk + j * kPwBlock + tid];
    issue(cset, min(kc + 2, kcn - 1));  // past the end: a harmless reload of the last step
    const u32x4* img = lds6 + (kc & 1) * kImg + lane;
#pragma unroll
    for (int t = 0; t < COT; t += 2) {
      const u32x4 bh0 = img[(t * 3 + 0) * 64], bm0 = img[(t * 3 + 1) * 64], bl0 = img[(t * 3 + 2) * 64];
      const u32x4 bh1 = img[(t * 3 + 3) * 64], bm1 = img[(t * 3 + 4) * 64], bl1 = img[(t * 3 + 5) * 64];
      // smallest terms first; two accumulators alternate so that no MFMA waits on its predecessor
      acc[t] = mfma_bf16(al, bh0, acc[t]);
      acc[t + 1] = mfma_bf16(al, bh1, acc[t + 1]);
      acc[t] = mfma_bf16(ah, bl0, acc[t]);
      acc[t + 1] = mfma_bf16(ah, bl1, acc[t + 1]);
      acc[t] = mfma_bf16(am, bm0, acc[t]);
      acc[t + 1] = mfma_bf16(am, bm1, acc[t + 1]);
      acc[t] = mfma_bf16(am, bh0, acc[t]);
      acc[t + 1] = mfma_bf16(am, bh1, acc[t + 1]);
      acc[t] = mfma_bf16(ah, bm0, acc[t]);
      acc[t + 1] = mfma_bf16(ah, bm1, acc[t + 1]);
      acc[t] = mfma_bf16(ah, bh0, acc[t]);
      acc[t + 1] = mfma_bf16(ah, bh1, acc[t + 1]);
    }
    {
      u32x4* dst = lds6 + ((kc + 1) & 1) * kImg;
#pragma unroll
      for (int j = 0; j < kWst; ++j) dst[j * kPwBlock + tid] = wst[j];
    }
    __syncthreads();
  };
  for (int kc = 0; kc < kcn; kc += 2) {
    step(I0{}, kc);
    step(I1{}, kc + 1);
  }

  // acc[t][4q + e] = pixel p0 + 8q + 4h + e, channel rb*32*COT + 32t + r
  const int co0 = rb * 32 * COT + r;
#pragma unroll
  for (int t = 0; t < COT; ++t) {
    const int co = co0 + 32 * t;
    const size_t row = ((size_t)b * c + co) * hw;
    float bs = 0.f, s1 = 0.f, s2 = 0.f;
    unsigned word = 0;
    if (EPI == 0) bs = bias[co];
    if (EPI == 1 && wt < nwt) word = relu_mask[((size_t)b * nwt + wt) * c + co] >> (4 * h);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = p0 + 8 * q + 4 * h;
      if (p >= hw) continue;
      f32x4 v = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
      if (EPI == 0) {
        // BatchNorm batch statistics of this output, shifted by the bias (the raw accumulator)
        s1 += (v.x + v.y) + (v.z + v.w);
        s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        v.x += bs; v.y += bs; v.z += bs; v.w += bs;
      }
      if (EPI == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ((word >> (8 * q + e)) & 1u) ? v[e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(y + row + p) = v;
    }
    if (EPI == 0 && stat_part != nullptr) {  // block-uniform
      s1 += __shfl_xor(s1, 32, DHD_WAVE);
      s2 += __shfl_xor(s2, 32, DHD_WAVE);
      if (h == 0 && wt < nwt) {
        float* q = stat_part + ((size_t)(b * nwt + wt) * 2) * c;  // [(sample, wave tile)][2][c]
        q[co] = s1;
        q[c + co] = s2;
      }
    }
  }
}

// Weight gradient: G[co][ci] = sum_{b,p} A(co,p) * B(ci,p), A/B with the affine prologues above.
// Block = 8 waves, output tile OT x OT (wave: OT/2 x OT/4), loops over 32-pixel chunks
// worker, worker + n_workers, ...; per-worker partial matrices in `partial` [worker][c][c].
template <int OT, bool A_TWO, bool B_TWO, bool B_RELU>
__global__ __launch_bounds__(kWgBlock) void pw_wgrad_kernel(const float* __restrict__ a0, const float* __restrict__ a1,
                                                            const float* __restrict__ acoef, size_t a_bstride,
                                                            const float* __restrict__ b0, const float* __restrict__ b1,
                                                            const float* __restrict__ bcoef, size_t b_bstride,
                                                            float* __restrict__ partial, int c, int hw, int nb, int n_workers) {
  constexpr int S = kWgStride;
  constexpr int kTile = OT * S;       // floats per staged operand tile
  constexpr int RPT = OT / 64;        // rows per thread per operand
  constexpr int TA = OT / 64, TB = OT / 128;
  extern __shared__ float lds[];      // [buf 2][operand 2][OT][S]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nob = c / OT;
  const int ob_co = (blockIdx.y / nob) * OT, ob_ci = (blockIdx.y % nob) * OT;
  const int wco = (wv >> 2) * (OT / 2), wci = (wv & 3) * (OT / 4);
  const int cps = (hw + 31) >> 5;     // chunks per sample
  const int n_chunks = nb * cps;
  const int srow = tid >> 3, sq = tid & 7;  // staging: row srow + 64 j, float4 column sq

  f32x16 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  f32x4 ra0[RPT], ra1[RPT], rb0[RPT], rb1[RPT];
  auto fetch = [&](int chunk) {
    const int b = chunk / cps, p = (chunk % cps) * 32 + 4 * sq;
    const bool in = p < hw;
    const size_t off = (size_t)(in ? p : 0);
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const size_t ra = (size_t)b * a_bstride + (size_t)(ob_co + srow + 64 * j) * hw + off;
      const size_t rbo = (size_t)b * b_bstride + (size_t)(ob_ci + srow + 64 * j) * hw + off;
      ra0[j] = *reinterpret_cast<const f32x4*>(a0 + ra);
      if (A_TWO) ra1[j] = *reinterpret_cast<const f32x4*>(a1 + ra);
      rb0[j] = *reinterpret_cast<const f32x4*>(b0 + rbo);
      if (B_TWO) rb1[j] = *reinterpret_cast<const f32x4*>(b1 + rbo);
    }
  };
  auto stage = [&](int chunk, int buf) {
    const int b = chunk / cps, p = (chunk % cps) * 32 + 4 * sq;
    const bool in = p < hw;
    float* ta = lds + (buf * 2) * kTile;
    float* tb = ta + kTile;
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const int row = srow + 64 * j;
      const float* ca = acoef + (size_t)b * 3 * c + ob_co + row;
      const float* cb = bcoef + (size_t)b * 3 * c + ob_ci + row;
      const float a_c0 = ca[0], a_c1 = ca[c], a_c2 = ca[2 * c];
      const float b_c0 = cb[0], b_c1 = cb[c], b_c2 = cb[2 * c];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float va = fmaf(a_c0, ra0[j][e], a_c2);
        if (A_TWO) va = fmaf(a_c1, ra1[j][e], va);
        float vb = fmaf(b_c0, rb0[j][e], b_c2);
        if (B_TWO) vb = fmaf(b_c1, rb1[j][e], vb);
        if (B_RELU) vb = fmaxf(vb, 0.f);
        ta[row * S + 4 * sq + e] = in ? va : 0.f;
        tb[row * S + 4 * sq + e] = in ? vb : 0.f;
      }
    }
  };

  int chunk = blockIdx.x;
  int buf = 0;
  if (chunk < n_chunks) {
    fetch(chunk);
    stage(chunk, 0);
  }
  __syncthreads();
  for (; chunk < n_chunks; chunk += n_workers) {
    const int next = chunk + n_workers;
    if (next < n_chunks) fetch(next);
    const float* ta = lds + (buf * 2) * kTile;
    const float* tb = ta + kTile;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
      float fa[TA], fb[TB];
#pragma unroll
      for (int i = 0; i < TA; ++i) fa[i] = ta[(wco + 32 * i + r) * S + 2 * k2 + h];
#pragma unroll
      for (int j = 0; j < TB; ++j) fb[j] = tb[(wci + 32 * j + r) * S + 2 * k2 + h];
#pragma unroll
      for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (next < n_chunks) stage(next, buf ^ 1);
    buf ^= 1;
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

// Weight gradient on the bf16 MFMA (bf16x6 split, see above): G[co][ci] = sum_{b,p} A(co,p) * B(ci,p).
// Pixels are the MFMA k dimension, 16 per step.  An "item" is 8 consecutive pixels of one channel row:
// exactly one lane's operand fragment, and 32 contiguous bytes of NCHW memory per input.  The thread
// that loads an item applies the affine prologue, splits it into the three bf16 terms ONCE and writes
// them to LDS in MFMA fragment order (lane-linear ds_write_b128 / ds_read_b128, no bank conflicts, no
// per-wave re-splitting).  Block = 8 waves, output tile OT x OT (wave: OT/2 x OT/4), double-buffered
// LDS, one barrier per step; the staging VALU work of step s+1 sits between the MFMAs of step s.
// Workers own contiguous step ranges; per-worker partial matrices are reduced by wgrad_reduce_kernel.
template <int OT, bool A_TWO, bool B_TWO, bool B_RELU>
__global__ __launch_bounds__(kWgBlock, 2) void pw_wgrad6_kernel(const float* __restrict__ a0, const float* __restrict__ a1,
                                                                const float* __restrict__ acoef, size_t a_bstride,
                                                                const float* __restrict__ b0, const float* __restrict__ b1,
                                                                const float* __restrict__ bcoef, size_t b_bstride,
                                                                float* __restrict__ partial, int c, int hw, int nb, int n_workers) {
  constexpr int TA = OT / 64, TB = OT / 128;   // 32x32 tiles per wave
  constexpr int kOp = (OT / 32) * 3 * 64;      // 16-byte units of one staged operand
  constexpr int kItems = OT == 256 ? 2 : 1;    // items per thread per step: OT = 256 one of A and one of B
  extern __shared__ u32x4 ldsw[];              // [buf 2][operand 2][tile][term][lane]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nob = c / OT;
  const int ob_co = (blockIdx.y / nob) * OT, ob_ci = (blockIdx.y % nob) * OT;
  const int wco = (wv >> 2) * (OT / 2), wci = (wv & 3) * (OT / 4);
  const int sps = (hw + 15) >> 4;              // steps per sample
  const long n_steps = (long)nb * sps;
  const int s0 = (int)(n_steps * blockIdx.x / n_workers), s1 = (int)(n_steps * (blockIdx.x + 1) / n_workers);

  // this thread's items: (operand, row, half).  OT = 256 (round 2): line-coalesced loads as in pw_wgrad3_kernel -- load
  // instruction j reads rows 128 j + 16 wv + (lane >> 2), four pixels 4 (lane & 3): 4 adjacent lanes = the row's 64 bytes of
  // this step, 16 rows per instruction -- and the 8-pixel item of row 128 (lane & 1) + 16 wv + (lane >> 2), half
  // (lane & 3) >> 1 is assembled with a lane-pair exchange.  The items and their arithmetic are the same as before
  // (bit-identical results); OT = 128 keeps one thread = 32 contiguous bytes.
  constexpr bool kCoal = OT == 256;
  const int ld_row = 16 * wv + (lane >> 2), ld_px = 4 * (lane & 3), odd = lane & 1;
  const int it_row = kCoal ? 128 * odd + ld_row : ((tid & 255) >> 1);
  const int it_h = kCoal ? ((lane & 3) >> 1) : (tid & 1);
  const bool second_is_b = true;               // OT = 256: item 0 = A, item 1 = B
  const bool single_is_b = tid >= 256;         // OT = 128: waves 0-3 stage A, waves 4-7 stage B (wave-uniform)
  const int it_slot = (it_row >> 5) * 192 + (it_row & 31) + 32 * it_h;  // + term * 64

  f32x16 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // raw registers of the step being fetched: [item][input][2 x float4]
  f32x4 raw[kItems][2][2];
  float cfa[3], cfb[3];
  int cur_b = -1;
  auto load_coefs = [&](int b) {
    const float* ca = acoef + (size_t)b * 3 * c + ob_co + it_row;
    const float* cb = bcoef + (size_t)b * 3 * c + ob_ci + it_row;
#pragma unroll
    for (int q = 0; q < 3; ++q) { cfa[q] = ca[q * c]; cfb[q] = cb[q * c]; }
    cur_b = b;
  };
  auto swap1 = [](float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true)); };
  auto fetch = [&](int s) {
    if (kCoal) {
      const int b = s / sps, p = (s % sps) * 16 + ld_px;
      const size_t off = p < hw ? p : 0;
#pragma unroll
      for (int it = 0; it < kItems; ++it) {
        const float* src0 = it ? b0 : a0;
        const float* src1 = it ? b1 : a1;
        const bool two = it ? B_TWO : A_TWO;
        const size_t base = (size_t)b * (it ? b_bstride : a_bstride) + (size_t)((it ? ob_ci : ob_co) + ld_row) * hw + off;
#pragma unroll
        for (int q = 0; q < 2; ++q) {   // q = load instruction: rows +0 / +128
          raw[it][0][q] = *reinterpret_cast<const f32x4*>(src0 + base + (size_t)(128 * q) * hw);
          if (two) raw[it][1][q] = *reinterpret_cast<const f32x4*>(src1 + base + (size_t)(128 * q) * hw);
        }
      }
      return;
    }
    const int b = s / sps, p = (s % sps) * 16 + 8 * it_h;
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      const bool is_b = OT == 256 ? (it == 1 && second_is_b) : single_is_b;
      const float* src0 = is_b ? b0 : a0;
      const float* src1 = is_b ? b1 : a1;
      const bool two = is_b ? B_TWO : A_TWO;
      const size_t base = (size_t)b * (is_b ? b_bstride : a_bstride) + (size_t)((is_b ? ob_ci : ob_co) + it_row) * hw;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int pq = p + 4 * q;
        const size_t off = base + (pq < hw ? pq : 0);
        raw[it][0][q] = *reinterpret_cast<const f32x4*>(src0 + off);
        if (two) raw[it][1][q] = *reinterpret_cast<const f32x4*>(src1 + off);
      }
    }
  };
  auto stage = [&](int s, int buf) {
    const int b = s / sps, p = (s % sps) * 16 + 8 * it_h;
    if (b != cur_b) load_coefs(b);  // block-uniform, a few times per worker
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      const bool is_b = OT == 256 ? (it == 1 && second_is_b) : single_is_b;
      const bool two = is_b ? B_TWO : A_TWO;
      const float k0 = is_b ? cfb[0] : cfa[0], k1 = is_b ? cfb[1] : cfa[1], k2 = is_b ? cfb[2] : cfa[2];
      if (kCoal) {   // assemble [lower 4 pixels | upper 4 pixels] of this thread's row: the even lane keeps its piece of the
                     // first instruction's row and takes its neighbour's, the odd lane the same for the second instruction's
#pragma unroll
        for (int in2 = 0; in2 < 2; ++in2) {
          if (in2 == 1 && !two) continue;
          const f32x4 r0 = raw[it][in2][0], r1 = raw[it][in2][1];
          f32x4 lo, hi;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float keep = odd ? r1[e] : r0[e];
            const float recv = swap1(odd ? r0[e] : r1[e]);
            lo[e] = odd ? recv : keep;
            hi[e] = odd ? keep : recv;
          }
          raw[it][in2][0] = lo;
          raw[it][in2][1] = hi;
        }
      }
      float v[8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const bool in = p + 4 * q < hw;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = fmaf(k0, raw[it][0][q][e], k2);
          if (two) t = fmaf(k1, raw[it][1][q][e], t);
          if (is_b && B_RELU) t = fmaxf(t, 0.f);
          v[4 * q + e] = in ? t : 0.f;
        }
      }
      u32x4 th, tm, tl;
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        unsigned hh, mm, ll;
        split2(v[2 * jp], v[2 * jp + 1], hh, mm, ll);
        th[jp] = hh; tm[jp] = mm; tl[jp] = ll;
      }
      u32x4* dst = ldsw + (buf * 2 + (is_b ? 1 : 0)) * kOp + it_slot;
      dst[0] = th;
      dst[64] = tm;
      dst[128] = tl;
    }
  };

  if (s0 < s1) {
    fetch(s0);
    stage(s0, 0);
    fetch(min(s0 + 1, s1 - 1));
  }
  __syncthreads();
  for (int s = s0; s < s1; ++s) {
    const int buf = (s - s0) & 1;
    // unconditional (indices clamped to the last step, whose re-staged copy nobody reads): a conditional fetch
    // would make the compiler copy the loaded registers at the merge point, and such a copy waits for the load
    stage(min(s + 1, s1 - 1), buf ^ 1);
    fetch(min(s + 2, s1 - 1));
    const u32x4* ta = ldsw + (buf * 2) * kOp + lane;
    const u32x4* tb = ta + kOp;
    u32x4 fb[TB][3];
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int t = 0; t < 3; ++t) fb[j][t] = tb[(((wci >> 5) + j) * 3 + t) * 64];
#pragma unroll
    for (int i = 0; i < TA; ++i) {
      u32x4 fa[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) fa[t] = ta[(((wco >> 5) + i) * 3 + t) * 64];
      // terms: 0 = high, 1 = mid, 2 = low; smallest products first, accumulators alternate
#pragma unroll
      for (int j = 0; j < TB; ++j) acc[i][j] = mfma_bf16(fa[2], fb[j][0], acc[i][j]);
#pragma unroll
      for (int j = 0; j < TB; ++j) acc[i][j] = mfma_bf16(fa[0], fb[j][2], acc[i][j]);
#pragma unroll
      for (int j = 0; j < TB; ++j) acc[i][j] = mfma_bf16(fa[1], fb[j][1], acc[i][j]);
#pragma unroll
      for (int j = 0; j < TB; ++j) acc[i][j] = mfma_bf16(fa[1], fb[j][0], acc[i][j]);
#pragma unroll
      for (int j = 0; j < TB; ++j) acc[i][j] = mfma_bf16(fa[0], fb[j][1], acc[i][j]);
#pragma unroll
      for (int j = 0; j < TB; ++j) acc[i][j] = mfma_bf16(fa[0], fb[j][0], acc[i][j]);
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

