// SFA stage, 1x1 convolutions: the LDS-RESIDENT GEMM family of rounds 2-3 (a persistent workgroup keeps the bf16 fragments of
// 32*COB output channels in LDS; teams of workgroups on one XCD share every pixel tile).  Included by sfa_stage.hip inside its
// unnamed namespace.  Reference: models/necks/mix.py:51 and its backward.
// Which calls still reach it (launch_pw_gemm_res in sfa_stage.hip):
//   pw_gemm_res_kernel<NT = 2>            gemm = bf16x3 (default) at C = 512 (DHD-M: SFA(1024, 512)); C = 128 / 256 run
//                                         pw_gemm_cu_kernel (sfa_gemm_cu.h) instead
//   pw_gemm_res_kernel<NT = 3>            gemm = DHD_SFA_GEMM_BF16X6 at C = 128 / 256
// ------------------------------------------------------------------------------------------------
// The same GEMM with the weights RESIDENT in LDS ("res" kernels, gemm modes 1 and 3).
//
// pw_gemm6_kernel above streams the whole packed weight set (393 KB at C = 256) through LDS once per 128-pixel
// tile: ~2 GB of L2 -> LDS traffic per GEMM at B = 4, one workgroup barrier per 16-channel step, all four waves of
// a workgroup in lockstep.  Measured: 41 % MFMA utilisation inside a round of tiles, 155-175 us per GEMM against
// 51 us of MFMA time and ~80 us of HBM time.
//
// Here a workgroup is persistent and owns 32*COB output channels for its whole life: their weight fragments
// (all K) are copied into LDS once (96-128 KB -> one workgroup per CU), the per-(sample, channel) prologue
// coefficients of every sample go next to them, and after that single barrier the waves never synchronise again.
// Each wave walks its own list of 32-pixel tiles; the activation rows of a tile are fetched D = 4 steps ahead
// (straight across tile boundaries) with raw buffer loads, so ~16 KB per wave are always in flight.  The C / (32 COB)
// workgroups that need the same pixels ("team") sit on the same XCD (blocks i, i+8, ... share an L2) and take the
// same tiles, so the activation is read from HBM once and from L2 by the other members; their prologue / split
// work is redundant VALU time that runs under the other wave's MFMAs.
//
// NT = number of bf16 terms kept per operand:
//   3  -> six products per a*b ("bf16x6", float32-level accuracy, as pw_gemm6);       COB = 2 at C = 256
//   2  -> three products  ah*bh + ah*bm + am*bh  ("bf16x3", relative error ~2^-16);   COB = 4 at C = 256
// ------------------------------------------------------------------------------------------------

// Weight (rows x k; or its transpose) -> per-team-member slices of MFMA B fragments:
//   packed16[(((g*KC + kc)*COB + t)*NT + term)*64 + lane] = term(M[g*32*COB + 32 t + (lane&31)][16 kc + 8 (lane>>5) + j]), j = 0..7
__device__ __forceinline__ void pack_weight_res_block(const float* __restrict__ w, int transpose, u32x4* __restrict__ packed, int c,
                                                      int cob, int nt, int block) {
  const int idx = block * kEwBlock + threadIdx.x;  // (g, kc, t, lane)
  const int kcn = c / 16;
  if (idx >= (c / 32) * kcn * 64) return;
  int q = idx;
  const int lane = q & 63; q >>= 6;
  const int t = q % cob; q /= cob;
  const int kc = q % kcn;
  const int g = q / kcn;
  const int row = g * 32 * cob + 32 * t + (lane & 31);
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
  u32x4* dst = packed + ((size_t)((g * kcn + kc) * cob + t) * nt) * 64 + lane;
  dst[0] = h;
  dst[64] = m;
  if (nt == 3) dst[128] = l;
}

__global__ __launch_bounds__(kEwBlock) void pack_weight_res_kernel(const float* __restrict__ w, const float* __restrict__ w_second,
                                                                   int transpose, u32x4* __restrict__ packed,
                                                                   u32x4* __restrict__ packed_second, int c, int cob, int nt) {
  // both convolutions' weights in one launch
  pack_weight_res_block(blockIdx.y == 1 ? w_second : w, transpose, blockIdx.y == 1 ? packed_second : packed, c, cob, nt, blockIdx.x);
}

constexpr int kResTrPitch = 36;
template <int NT, int COB, int KCN, bool TWO_IN, bool RELU, int EPI, int WAVES, int AUX, int DPF>
__global__ __launch_bounds__(WAVES * 64, 1) void pw_gemm_res_kernel(const float* __restrict__ in0, const float* __restrict__ in1,
                                                                    size_t in_bstride, unsigned in_bytes, const float* __restrict__ coef,
                                                                    const u32x4* __restrict__ wp, const float* __restrict__ bias,
                                                                    unsigned* __restrict__ relu_mask, float* __restrict__ stat_part,
                                                                    float* __restrict__ y, int c, int hw, int nb, int groups, int nteams) {
  // prefetch distance in 16-channel steps (register sets of 8 loads per input in flight).  Measured with DPF = 8 (one-input
  // kernel 193 VGPRs, two-input 240-256): 101.6 vs 99.3 us and 105-113 vs 108-113 us -- the loads are not what a wave
  // waits for (phase clocks, experiments/res_timeline.py: 2 % of a wave's time), see DESIGN.md
  constexpr int D = DPF;
  constexpr int kThreads = WAVES * 64;
  extern __shared__ u32x4 ldsr[];      // weight fragments (KCN * COB * NT KB) | coefficient tables [nb][3][c]
  static_assert(KCN % D == 0 && KCN >= 2 * D, "K steps: a multiple of the prefetch distance, at least two rounds");
  constexpr int nfrag = KCN * COB * NT;   // c == 16 * KCN
  float* cf = reinterpret_cast<float*>(ldsr + (size_t)nfrag * 64);
  constexpr int kTrPitch = kResTrPitch;    // floats per row of the store patch: 32 pixels + 4 (conflict-free b128 rows)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  float* tr = cf + ((nb * 3 * c + 3) & ~3) + wv * (16 * kTrPitch);   // wave-private: 16 channel rows x 32 pixels
  // blocks i, i + 8, ... run on one XCD: `groups` consecutive ones of them form a team (same pixels, different channels)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int team = xcd + 8 * (slot / groups), g = slot % groups;

  {
    const u32x4* wsrc = wp + (size_t)g * nfrag * 64;
    const int n16 = nfrag * 64;
    int i = tid;
    for (; i + 3 * kThreads < n16; i += 4 * kThreads) {
      const u32x4 a = wsrc[i], b = wsrc[i + kThreads], cc = wsrc[i + 2 * kThreads], d = wsrc[i + 3 * kThreads];
      ldsr[i] = a; ldsr[i + kThreads] = b; ldsr[i + 2 * kThreads] = cc; ldsr[i + 3 * kThreads] = d;
    }
    for (; i < n16; i += kThreads) ldsr[i] = wsrc[i];
    for (int k = tid; k < nb * 3 * c; k += kThreads) cf[k] = coef[k];
  }
  __syncthreads();

  const int nwt = (hw + 31) >> 5;                 // 32-pixel wave tiles per sample
  const int total = nb * nwt;
  const int stride = nteams * WAVES;
  const int row_bytes = hw * 4;

  struct Tile {
    __amdgpu_buffer_rsrc_t r0, r1;
    int voff, b, wt;
  };
  auto make_tile = [&](int wtg) {
    Tile t;
    wtg = min(wtg, total - 1);                    // past the end: a harmless re-read of the last tile
    t.b = wtg / nwt;
    t.wt = wtg - t.b * nwt;
    const int pc = min(t.wt * 32 + r, hw - 1);
    t.voff = (pc + 8 * h * hw) * 4;
    t.r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in0 + (size_t)t.b * in_bstride), 0, in_bytes, 0x00020000);
    t.r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((TWO_IN ? in1 : in0) + (size_t)t.b * in_bstride), 0, in_bytes, 0x00020000);
    return t;
  };

  float raw0[D][8], raw1[D][8];
  auto issue = [&](auto set, const Tile& t, int kc) {
    constexpr int S = decltype(set)::value;
    const int so = 16 * kc * row_bytes;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // default cache policy (AUX = 0): the other members of the team read the same rows through this XCD's L2
      raw0[S][j] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(t.r0, t.voff, so + j * row_bytes, AUX));
      if (TWO_IN) raw1[S][j] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(t.r1, t.voff, so + j * row_bytes, AUX));
    }
  };
  f32x16 acc[COB];

  // one 16-channel step: consume register set CS (loaded for (cur, kc)), refill it for (pf, kpf), then the MFMAs
  auto step = [&](auto cset, auto first_tag, const Tile& cur, int kc, const Tile& pf, int kpf) {
    constexpr int CS = decltype(cset)::value;
    constexpr bool FIRST = decltype(first_tag)::value;   // first step of a tile: the accumulators start from zero
    u32x4 at[NT];
    {
      const int ci = 16 * kc + 8 * h;
      const float* cb = cf + (size_t)cur.b * 3 * c + ci;
      const f32x4* c0 = reinterpret_cast<const f32x4*>(cb);
      const f32x4* c1 = reinterpret_cast<const f32x4*>(cb + c);
      const f32x4* c2 = reinterpret_cast<const f32x4*>(cb + 2 * c);
      float v[8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 k0 = c0[q], k2 = c2[q];
        f32x4 k1;
        if (TWO_IN) k1 = c1[q];
#pragma unroll
        for (int e = 0; e < 4; e += 2) {   // two values per v_pk_fma_f32
          const f32x2 a0 = {k0[e], k0[e + 1]}, a2 = {k2[e], k2[e + 1]};
          const f32x2 x0 = {raw0[CS][4 * q + e], raw0[CS][4 * q + e + 1]};
          f32x2 t = __builtin_elementwise_fma(a0, x0, a2);
          if (TWO_IN) {
            const f32x2 a1 = {k1[e], k1[e + 1]};
            const f32x2 x1 = {raw1[CS][4 * q + e], raw1[CS][4 * q + e + 1]};
            t = __builtin_elementwise_fma(a1, x1, t);
          }
          v[4 * q + e] = RELU ? fmaxf(t.x, 0.f) : t.x;
          v[4 * q + e + 1] = RELU ? fmaxf(t.y, 0.f) : t.y;
        }
      }
      // the pass bits are recorded once per team: its members (same pixels, redundant prologues) take the steps in turn
      if (RELU && relu_mask != nullptr && (kc % groups) == g) {  // wave-uniform
        int word = 0;
        static_for<8>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          const unsigned long long bal = __ballot(v[j] > 0.f);
          word = write_lane<j>(word, (int)(unsigned)bal);               // lane j: channel 16 kc + j
          word = write_lane<8 + j>(word, (int)(unsigned)(bal >> 32));   // lane 8 + j: channel 16 kc + 8 + j
        });
        if (lane < 16) relu_mask[((size_t)cur.b * nwt + cur.wt) * c + 16 * kc + lane] = (unsigned)word;   // [sample][wave tile][channel]
      }
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        unsigned hh, mm, ll;
        if (NT == 3) {
          split2(v[2 * jp], v[2 * jp + 1], hh, mm, ll);
          at[NT - 1][jp] = ll;
        } else {
          split2_hm(v[2 * jp], v[2 * jp + 1], hh, mm);
        }
        at[0][jp] = hh;
        at[1][jp] = mm;
      }
    }
    // The machine scheduler would otherwise sink these loads below the MFMAs of all four unrolled steps and hoist the
    // four prologues to the top of the loop body -- i.e. consume every register set right after it was requested.
    // Scheduling barriers pin the order  prologue(kc) -> loads(kc + D) -> MFMAs(kc).
    __builtin_amdgcn_sched_barrier(0);
    issue(cset, pf, kpf);
    __builtin_amdgcn_sched_barrier(0);
    const u32x4* img = ldsr + kc * (COB * NT * 64) + lane;
    // output tiles in pairs: the fragments of two tiles are live at a time; consecutive MFMAs alternate between
    // the two accumulators (no MFMA waits on its predecessor); smallest terms first
    constexpr int TP = COB >= 2 ? 2 : 1;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t0 = 0; t0 < COB; t0 += TP) {
      u32x4 bf[TP][NT];
#pragma unroll
      for (int u = 0; u < TP; ++u)
#pragma unroll
        for (int m = 0; m < NT; ++m) bf[u][m] = img[((t0 + u) * NT + m) * 64];
      // product order (smallest first): NT = 3: l*h, h*l, m*m, m*h, h*m, h*h;  NT = 2: m*h, h*m, h*h
      constexpr int kProd = NT == 3 ? 6 : 3;
      constexpr int pa3[6] = {2, 0, 1, 1, 0, 0}, pb3[6] = {0, 2, 1, 0, 1, 0};
      constexpr int pa2[3] = {1, 0, 0}, pb2[3] = {0, 1, 0};
#pragma unroll
      for (int q = 0; q < kProd; ++q) {
        const int ia = NT == 3 ? pa3[q] : pa2[q], ib = NT == 3 ? pb3[q] : pb2[q];
#pragma unroll
        for (int u = 0; u < TP; ++u)
          acc[t0 + u] = mfma_bf16(at[ia], bf[u][ib], (FIRST && q == 0) ? zero : acc[t0 + u]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // bf16x3: a wave takes a CONTIGUOUS range of tiles (its team partner's waves the same one); bf16x6 every stride-th tile, as
  // the streamed kernels whose results it reproduces.  Measured, three alternating runs each (experiments/ab/run.sh): stage
  // forward 0.387 -> 0.383 ms, backward 0.928 -> 0.920 ms with contiguous ranges in bf16x3; the bf16x6 step 1.950 -> 1.960 ms.
  constexpr bool kContig = NT == 2;
  const int per_wave = (total + stride - 1) / stride;
  int wtg = kContig ? (team * WAVES + wv) * per_wave : team * WAVES + wv;
  const int wtg_end = kContig ? min(total, wtg + per_wave) : total;
  const int wtg_step = kContig ? 1 : stride;
  // BatchNorm statistics of the output (EPI 0): bf16x3 keeps them per lane over all of the wave's tiles; at the end the
  // workgroup's eight waves meet in LDS and ONE row per workgroup is written, stat_part[team][2][c] (each member of a team
  // its own channels): 128 rows at B = 4, where round 2 wrote a row per wave (1 024) and round 1 one per (sample, wave
  // tile) (5 000) -- the finalize kernel that reads them is a latency chain on the stage's critical path.  bf16x6 writes a
  // row per (sample, wave tile) like the streamed kernels, whose results it reproduces bit for bit.
  constexpr bool kWaveStats = EPI == 0 && NT == 2;
  float ws1[COB], ws2[COB];
#pragma unroll
  for (int t = 0; t < COB; ++t) ws1[t] = ws2[t] = 0.f;
  auto flush_stats = [&]() {   // every wave of the workgroup calls this exactly once (it contains a barrier)
    if (!kWaveStats || stat_part == nullptr) return;
#pragma unroll
    for (int t = 0; t < COB; ++t) {
      const float s1 = ws1[t] + __shfl_xor(ws1[t], 32, DHD_WAVE), s2 = ws2[t] + __shfl_xor(ws2[t], 32, DHD_WAVE);
      if (h == 0) {
        tr[32 * t + r] = s1;                       // the wave's store patch: [2][COB * 32] floats
        tr[COB * 32 + 32 * t + r] = s2;
      }
    }
    __syncthreads();
    if (wv == 0) {
      const float* all = cf + ((nb * 3 * c + 3) & ~3);
      float* q = stat_part + ((size_t)team * 2) * c + g * 32 * COB;
      for (int i = lane; i < 2 * COB * 32; i += DHD_WAVE) {
        float v = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < WAVES; ++w8) v += all[w8 * (16 * kTrPitch) + i];
        q[(i / (COB * 32)) * c + (i % (COB * 32))] = v;
      }
    }
  };
  if (wtg >= wtg_end) {                            // wave-uniform
    flush_stats();                                 // zeros into the workgroup's sum
    return;
  }
  // epilogue operands are requested long before they are used: a load issued inside the epilogue is waited for at once,
  // behind the stores of the previous half tiles (measured: 2.4 k clocks per 32-channel tile, a third of the wave's time)
  float bias_r[COB];                               // this lane's output channels are the same for every tile
#pragma unroll
  for (int t = 0; t < COB; ++t) bias_r[t] = EPI == 0 ? bias[g * 32 * COB + 32 * t + r] : 0.f;
  Tile cur = make_tile(wtg);
  static_for<D>([&](auto sc) { issue(sc, cur, decltype(sc)::value); });
  for (; wtg < wtg_end; wtg += wtg_step) {
    const Tile nxt = make_tile(wtg + wtg_step < wtg_end ? wtg + wtg_step : wtg);   // past the wave's range: a harmless re-read
    // The K loop is fully unrolled (straight-line code per tile): with an inner loop the register allocator
    // copied every prefetch register and every accumulator at the loop header (and a copy of a loaded register
    // waits for its load: no lookahead left).  Steps kc >= KCN - D prefetch the first steps of the next tile.
    int mask_r[COB];                               // EPI 1: the ReLU pass bits of this tile, requested before the K loop
#pragma unroll
    for (int t = 0; t < COB; ++t)
      mask_r[t] = EPI == 1 ? (int)relu_mask[((size_t)cur.b * nwt + cur.wt) * c + g * 32 * COB + 32 * t + r] : 0;
    static_for<KCN>([&](auto kcc) {
      constexpr int kc = decltype(kcc)::value;
      step(std::integral_constant<int, kc % D>{}, std::integral_constant<bool, kc == 0>{}, cur, kc, kc + D < KCN ? cur : nxt,
           (kc + D) % KCN);
    });

    // acc[t][4q + e] = pixel p0 + 8q + 4h + e, channel g*32*COB + 32t + r.  Stored straight from this layout a
    // store instruction would write 64 separate 16-byte pieces (adjacent lanes = different channel rows): measured 40 us
    // of a 125 us GEMM.  Each half tile (16 channels x 32 pixels) goes through a wave-private LDS patch instead and is
    // written row-wise: 8 adjacent lanes = one whole 128-byte line, 8 lines per store instruction.
    const int p0 = cur.wt * 32;
    const int co0 = g * 32 * COB + r;
    const bool full = p0 + 32 <= hw;               // wave-uniform; always true when hw % 32 == 0
    // stores: one buffer resource per sample, the row of a store as a scalar byte offset, the lane's place inside a
    // 16-row half tile as a 32-bit vector offset (was: a 64-bit multiply per lane and store)
    const __amdgpu_buffer_rsrc_t ry =
        __builtin_amdgcn_make_buffer_rsrc(y + (size_t)cur.b * c * hw, 0, (unsigned)((size_t)c * hw * sizeof(float)), 0x00020000);
    const int pst = p0 + 4 * (lane & 7);
    const int vst = ((lane >> 3) * hw + pst) * 4;
    const bool st_ok = full || pst < hw;
#pragma unroll
    for (int t = 0; t < COB; ++t) {
      const int co = co0 + 32 * t;
      float bs = 0.f, s1 = 0.f, s2 = 0.f;
      int word = 0;
      if (EPI == 0) bs = bias_r[t];
      if (EPI == 1) word = (int)((unsigned)mask_r[t] >> (4 * h));
      f32x4 vq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
        if (EPI == 0) {
          if (full || p0 + 8 * q + 4 * h < hw) {   // hw % 4 == 0: a 4-pixel group is inside or outside as a whole
            s1 += (v.x + v.y) + (v.z + v.w);
            s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
          }
          v.x += bs; v.y += bs; v.z += bs; v.w += bs;
        }
        if (EPI == 1) {   // pass bit -> all-ones / zero with one signed bit-field extract, then AND
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __int_as_float(__float_as_int(v[e]) & __builtin_amdgcn_sbfe(word, 8 * q + e, 1));
        }
        vq[q] = v;
      }
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        if ((r >> 4) == ph) {
#pragma unroll
          for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(tr + (r & 15) * kTrPitch + 8 * q + 4 * h) = vq[q];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(tr + ((lane >> 3) + 8 * k) * kTrPitch + 4 * (lane & 7));
          const int srow = (g * 32 * COB + 32 * t + 16 * ph + 8 * k) * row_bytes;   // scalar
          if (st_ok) store_b128_guarded<0>(__builtin_bit_cast(u32x4, w), ry, vst, srow);
        }
      }
      if (kWaveStats) {
        ws1[t] += s1;
        ws2[t] += s2;
      } else if (EPI == 0 && stat_part != nullptr) {  // block-uniform
        s1 += __shfl_xor(s1, 32, DHD_WAVE);
        s2 += __shfl_xor(s2, 32, DHD_WAVE);
        if (h == 0) {
          float* q = stat_part + ((size_t)(cur.b * nwt + cur.wt) * 2) * c;  // [(sample, wave tile)][2][c]
          q[co] = s1;
          q[c + co] = s2;
        }
      }
    }
    cur = nxt;
  }
  flush_stats();
}


