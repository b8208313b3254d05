// MGHS pooling forward / backward for gfx950 (MI355X).  Replaces 4x bev_pool_v2 + permute + cat
// (ops/bev_pool_v2/bev_pool.py:86-106, models/necks/lss_heightmap.py:298-299) and
// QuickCumsumCuda.backward (bev_pool.py:44-83) of the reference, for all grids in one call, on the
// grouping produced by dhd_mghs_prepare.
//
// The outputs are ~176 MB of dense (B, nz*C, ny, nx) tensors per sample of which ~97% are zeros;
// only ~50 000 voxels per sample receive any point.  Compact path (C == 64):
//
//   forward   1. mghs_gather_sums : one wave per 64 grouped entries, perfectly balanced; lane = channel;
//                                   per-voxel sums -> vsum[slot][64]   (~12 MB per sample, L2/MALL resident)
//             2. mghs_stream_fwd  : one workgroup per segment of 4 output rows (3200 contiguous,
//                                   cache-line aligned bytes per channel): streams zero-filled 16-byte
//                                   vectors, patching in the rows of vsum that fall into the segment
//   backward  1. mghs_stream_bwd  : same segments; streams out_grad once and extracts the rows of the
//                                   non-empty voxels -> vsum[slot][64]
//             2. mghs_pixel_bwd   : one wave per feature pixel: the pixel's <=2D entries are found through
//                                   the per-point slots; <g, feat> by DPP wave reduction -> depth_grad,
//                                   sum g * depth -> feat_grad; one writer per element, no atomics
//
// Every kernel is HBM/L2-bound integer-and-fp32 streaming; no MFMA.  The streaming kernels have no
// data-dependent loops: the latency-bound gather lives in the balanced per-entry kernels.
// A generic dense-row path (any C, any grid shape) is kept for configurations the compact path does
// not cover.
#include "lift_device.h"
#include <stdlib.h>
#include "mghs_layout.h"

namespace dhd {
namespace {

#ifndef DHD_STREAM_BLOCK
#define DHD_STREAM_BLOCK 512
#endif
constexpr int kStreamBlock = DHD_STREAM_BLOCK;
constexpr int kStreamWaves = kStreamBlock / DHD_WAVE;
#ifndef DHD_TABLE_FLOATS
#define DHD_TABLE_FLOATS 8192
#endif
constexpr int kTableFloats = DHD_TABLE_FLOATS;  // LDS patch table: 128 voxels x 64 channels, or more voxels x fewer channels per pass

// ---------------------------------------------------------------------------------------
// Segments
// ---------------------------------------------------------------------------------------
struct Segment {
  int g, b, z, y0, nx, ny, nz;
  int v0;    // first voxel id
  int nvox;  // kSegRows * nx
};

__device__ __forceinline__ bool decode_segment(const Layout& L, int s, Segment* sg) {
  if (s >= L.n_segs) return false;
  int g = 0;
#pragma unroll
  for (int k = 1; k < DHD_MAX_GRIDS; ++k)
    if (k < L.G && s >= L.seg_base[k]) g = k;
  const dhd_grid& gr = L.grid[g];
  const int local = s - L.seg_base[g];
  const int per_plane = gr.n[1] / kSegRows;
  sg->g = g; sg->nx = gr.n[0]; sg->ny = gr.n[1]; sg->nz = gr.n[2];
  sg->y0 = (local % per_plane) * kSegRows;
  const int bz = local / per_plane;
  sg->z = bz % gr.n[2];
  sg->b = bz / gr.n[2];
  sg->nvox = kSegRows * gr.n[0];
  sg->v0 = L.vox_base[g] + local * sg->nvox;
  return true;
}

// The LDS patch table of the segment writer / reader holds table[cc * pitch + j] = channel c_lo + cc of the segment's j-th
// non-empty voxel, pitch = nnz | 1.  (Until round 4 it was [j][cc]: the lanes of a wave work on ONE channel run, i.e. the same cc
// and different j, so their table accesses met in one LDS bank -- nothing on the band grids, where 3 % of the voxels hold a point,
// but the full-height grid is half occupied at DHD-S and 90 % at the DHD-L geometry: 30- to 60-way conflicts on its segments.)
// Neighbouring occupied voxels have neighbouring slots -> neighbouring banks; the fill / write-out side walks cc fastest, an odd
// pitch spreads it over the banks.
// channels handled per pass so that pitch * cp <= kTableFloats (cp in {64,32,...,4}; kTableFloats >= (kSegMaxVox + 1) * 4)
__device__ __forceinline__ int channels_per_pass(int nnz) {
  int cp = kTileC;
  while (cp > 4 && (nnz | 1) * cp > kTableFloats) cp >>= 1;
  return cp;
}

// ---------------------------------------------------------------------------------------
// 1. forward gather: per-voxel sums.  Wave w covers entries [64w, 64w+64) and owns the voxels whose
// FIRST entry lies there (it runs past the end of its slice to finish its last voxel), so every
// slot has exactly one writer.  The kernel is instruction-bound (measured: SALU 680 + VALU 335
// instructions per wave before this form), so the 64 entries of a batch are processed by fully
// unrolled code with compile-time lane numbers: per entry one v_readlane + one v_lshl_add_u32 + one
// global_load (SGPR base + 32-bit VGPR offset), one v_readlane + v_fmac, one s_bitcmp + branch.
// Entries outside the owned range are neutralised without per-entry range checks: garbage
// accumulated before the first owned voxel goes to a scratch row, and a terminator bit makes the
// run end on the scratch row as well, so nothing outside the owned voxels reaches a real slot.
// ---------------------------------------------------------------------------------------
// (Round 2, tried and reverted: 16 lanes x 16 bytes per feature row = four entries per load instruction with one partial
// sum per entry slot, as mghs_pixel_bwd now does.  Every voxel boundary then needs a cross-slot reduction (8 lane
// exchanges) where this form needs none, and voxels are short (8-17 entries): DHD-S 0.394 -> 0.440 ms per step, DHD-L
// geometry 1.09 -> 1.28 ms.)
template <int J>
struct GatherStep {
  static __device__ __forceinline__ void load(float (&f)[DHD_WAVE], __amdgpu_buffer_rsrc_t feat_rsrc, int lane4, int pix,
                                              int jmax) {
    if ((J & 7) == 0 && J > jmax) return;  // wave-uniform early exit, checked every 8 positions
    // buffer load: per-lane byte offset in a VGPR (constant), row offset (pixel * 256 B) in an SGPR:
    // one v_readlane + one s_lshl + one buffer_load per entry, no vector address arithmetic
    f[J] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(feat_rsrc, lane4, lane_i(pix, J) << 8, 0));
    GatherStep<J + 1>::load(f, feat_rsrc, lane4, pix, jmax);
  }
  static __device__ __forceinline__ void run(const float (&f)[DHD_WAVE], unsigned long long firsts, int term, int slot,
                                             float dv, float* vrow, int scratch_row, int jmax, int& cur, float& acc) {
    if ((J & 7) == 0 && J > jmax) return;
    if ((firsts >> J) & 1ull) {
      vrow[(size_t)cur * kTileC] = acc;  // cur == scratch_row while nothing is owned
      cur = rfl((J == term) ? scratch_row : lane_i(slot, J));
      acc = 0.f;
    }
    acc = fmaf(lane_f(dv, J), f[J], acc);
    GatherStep<J + 1>::run(f, firsts, term, slot, dv, vrow, scratch_row, jmax, cur, acc);
  }
};
template <>
struct GatherStep<DHD_WAVE> {
  static __device__ __forceinline__ void load(float (&)[DHD_WAVE], __amdgpu_buffer_rsrc_t, int, int, int) {}
  static __device__ __forceinline__ void run(const float (&)[DHD_WAVE], unsigned long long, int, int, float, float*, int, int,
                                             int&, float&) {}
};

// BANDS_ONLY: the entries of the full-height grid 0 (the first offset[vox_base[1]] of the list) are left to mghs_col_sums.
template <bool BANDS_ONLY>
__device__ __forceinline__ void gather_sums_body(const Layout& L, const float* __restrict__ depth, const float* __restrict__ feat,
                                                 const int block) {
  const int lane = threadIdx.x & 63;
  const int T0 = BANDS_ONLY ? L.offset[L.vox_base[1]] : 0;
  const int T = L.offset[L.V];  // total entries (device-side value, scalar load)
  // everything that steers control flow is forced into SGPRs: the compiler cannot see that
  // threadIdx.x >> 6 is wave-uniform and would otherwise predicate every branch through EXEC
  // XCD x (workgroups x, x+8, ...) takes the x-th eighth of the entries actually present: entries are
  // sorted by voxel, so one XCD's waves gather a compact part of the feature map through their L2
  const int per_xcd = ((T - T0 + kBlock - 1) / kBlock + 7) >> 3;
  if ((block >> 3) >= per_xcd) return;
  const int wg = (block & 7) * per_xcd + (block >> 3);
  const int a = rfl(T0 + (wg * (kBlock / DHD_WAVE) + (threadIdx.x >> 6)) * DHD_WAVE);
  if (a >= T) return;
  const int b = min(T, a + DHD_WAVE);
  const __amdgpu_buffer_rsrc_t feat_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(feat), 0, L.B * L.N * L.hw * kTileC * 4, 0x00020000);
  const int lane4 = 4 * lane;
  const int scratch_row = L.n_slots_max;
  float* vrow = L.vsum + lane;
  bool started = false;
  int cur = rfl(scratch_row);
  float acc = 0.f;
  int idx = a + lane;
  int slot_n = -1, prev_n = -2, pix_n = 0, pid_n = 0;
  float dv_n = 0.f;
  if (idx < T) {
    const int4 en = L.s_ent[idx];
    slot_n = en.z;
    pix_n = en.y;
    pid_n = en.x;
    if (idx > T0) prev_n = L.s_ent[idx - 1].z;   // (column form: grid 0's entries before T0 are not written)
  }
  if (idx < T) dv_n = depth[pid_n];
  for (int a0 = a; a0 < T; a0 += DHD_WAVE) {
    const int nb = min(DHD_WAVE, T - a0);
    const int slot = slot_n, pix = pix_n;   // lanes >= nb hold slot -1, pixel 0, depth 0
    const float dv = dv_n;
    unsigned long long firsts = __ballot(lane < nb && slot != prev_n);
    int lo = 0;
    if (!started) {
      if (firsts == 0ull) lo = nb;                  // still inside a voxel owned by an earlier wave
      else lo = __builtin_ctzll(firsts);
      if (a0 + lo >= b) break;                      // the first voxel starting here belongs to the next wave
      started = lo < nb;
    }
    // terminator: the first voxel boundary at or after b (that voxel belongs to the next wave), or
    // the end of the entry list; 64 = none in this batch
    int term = DHD_WAVE;
    {
      const int s = min(max(b - a0, lo + 1), DHD_WAVE);
      const unsigned long long m = s >= DHD_WAVE ? 0ull : (firsts >> s) << s;
      if (m != 0ull) term = __builtin_ctzll(m);
      else if (nb < DHD_WAVE) term = nb;
    }
    const bool last = term < DHD_WAVE;
    if (last) firsts = (firsts & ((1ull << term) - 1ull)) | (1ull << term);  // nothing is flushed after the terminator
    if (lo >= nb) firsts = 0ull;
    idx = a0 + DHD_WAVE + lane;
    slot_n = -1; pix_n = 0; pid_n = 0;
    if (!last && idx < T) {  // next batch: its index words travel while this batch is processed
      const int4 en = L.s_ent[idx];
      slot_n = en.z;
      pix_n = en.y;
      pid_n = en.x;
      prev_n = L.s_ent[idx - 1].z;
    }
    if (lo < nb) {
      const int jmax = last ? term : DHD_WAVE - 1;
      float f[DHD_WAVE];
      GatherStep<0>::load(f, feat_rsrc, lane4, pix, jmax);
      GatherStep<0>::run(f, firsts, term, slot, dv, vrow, scratch_row, jmax, cur, acc);
    }
    if (last) { cur = rfl(scratch_row); break; }
    dv_n = 0.f;
    if (idx < T) dv_n = depth[pid_n];
  }
  vrow[(size_t)cur * kTileC] = acc;
}

template <bool BANDS_ONLY>
__global__ __launch_bounds__(kBlock) void mghs_gather_sums(Layout L, const float* __restrict__ depth,
                                                            const float* __restrict__ feat) {
  gather_sums_body<BANDS_ONLY>(L, depth, feat, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------
// 1b. forward sums of the FULL-HEIGHT grid 0 by pixel column ("column form").
// Grid 0 pools every pixel and has one z cell, so the fH rows of a pixel column at one depth bin mostly fall into the same
// voxel: measured on the benchmark's rigs, runs of equal keys along a column are 6.3 rows long at the DHD-S geometry and 11.9
// at DHD-L, and grid 0 holds 80 % of all entries.  mghs_gather_sums reads one 256-byte context row per ENTRY (D = 44 / 88 times
// per row: 1.5 GB of L2 gathers per step at the DHD-L geometry, B = 2).  Here a wave owns a pixel column (camera, w) and
// kDepthChunk depth bins: the column's fH context rows are loaded ONCE into registers (lane = channel), the keys / depth
// values / slots of a depth bin are fetched by lanes 0..fH-1, and the rows are walked with compile-time register numbers --
// one v_readlane + v_fmac per row, a flush of the running sum at every key change: one atomic add per RUN and channel into
// vsum[slot].  The band grids (z cells of 0.4 m: runs of 1.1-1.2 rows) stay with mghs_gather_sums<true>.
// Summation order inside a voxel is the arrival order of the atomics (the float32 sum differs in its last bits from run to
// run, as with the default grouping); DHD_MGHS_DETERMINISTIC keeps the sorted gather for all grids.
// ---------------------------------------------------------------------------------------
constexpr int kDepthChunk = 8;

__global__ __launch_bounds__(kBlock) void mghs_zero_grid0_rows(Layout L) {
  const int n0 = L.nzoff[L.vox_base[1]];                       // non-empty voxels of grid 0 = its slots [0, n0)
  vfloat4* v = reinterpret_cast<vfloat4*>(L.vsum);
  const size_t n4 = (size_t)n0 * (kTileC / 4);
  const vfloat4 z = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (size_t)gridDim.x * kBlock) v[i] = z;
}

template <int FH, int H>
struct ColStep {
  static __device__ __forceinline__ void run(const float (&f)[FH], unsigned long long firsts, int slot, float dv, float* vrow,
                                             int& cur, float& acc) {
    if ((firsts >> H) & 1ull) {                                  // wave-uniform: a new run starts at row H
      if (cur >= 0) atomicAdd(vrow + (size_t)cur * kTileC, acc);
      cur = rfl(lane_i(slot, H));
      acc = 0.f;
    }
    acc = fmaf(lane_f(dv, H), f[H], acc);                        // dv = 0 for dropped points
    ColStep<FH, H + 1>::run(f, firsts, slot, dv, vrow, cur, acc);
  }
};
template <int FH>
struct ColStep<FH, FH> {
  static __device__ __forceinline__ void run(const float (&)[FH], unsigned long long, int, float, float*, int&, float&) {}
};

template <int FH>
__device__ __forceinline__ void col_sums_body(const Layout& L, const float* __restrict__ depth, const float* __restrict__ feat,
                                              const int block) {
  const int lane = threadIdx.x & 63;
  const int n_chunks = (L.D + kDepthChunk - 1) / kDepthChunk;
  const int wave = rfl((int)(block * (kBlock / DHD_WAVE) + (threadIdx.x >> 6)));
  const int n_cols = L.B * L.N * L.fw;
  if (wave >= n_cols * n_chunks) return;
  // consecutive waves = consecutive columns of one camera and depth chunk (neighbouring columns hit neighbouring voxels)
  const int chunk = wave / n_cols, col = wave - chunk * n_cols;
  const int bn = col / L.fw, w = col - bn * L.fw;
  float f[FH];
#pragma unroll
  for (int h = 0; h < FH; ++h) f[h] = feat[((size_t)(bn * FH + h) * L.fw + w) * kTileC + lane];
  float* vrow = L.vsum + lane;
  const int d0 = chunk * kDepthChunk;
  const int hl = lane < FH ? lane : FH - 1;                      // lanes >= FH repeat the last row (never a run start)
  int key[kDepthChunk], slot[kDepthChunk];
  float dv[kDepthChunk];
#pragma unroll
  for (int k = 0; k < kDepthChunk; ++k) {
    const int d = min(d0 + k, L.D - 1);
    const int p = ((bn * L.D + d) * FH + hl) * L.fw + w;
    key[k] = d0 + k < L.D ? L.key[pt_index(L, bn, d, hl, w)] : -1;   // (column-major: the lanes of a column read adjacent words)
    dv[k] = depth[p];
  }
#pragma unroll
  for (int k = 0; k < kDepthChunk; ++k) slot[k] = key[k] >= 0 ? L.nzoff[key[k]] : -1;
#pragma unroll
  for (int k = 0; k < kDepthChunk; ++k) {
    const int prev = __shfl_up(slot[k], 1, DHD_WAVE);
    // run starts: row 0 and every change of slot (dropped rows form runs of slot -1, whose sums are discarded)
    const unsigned long long firsts = __ballot(lane < FH && (lane == 0 || slot[k] != prev));
    const float dvk = key[k] >= 0 ? dv[k] : 0.f;
    int cur = -1;
    float acc = 0.f;
    ColStep<FH, 0>::run(f, firsts, slot[k], dvk, vrow, cur, acc);
    if (cur >= 0) atomicAdd(vrow + (size_t)cur * kTileC, acc);
  }
}

template <int FH>
__global__ __launch_bounds__(kBlock) void mghs_col_sums(Layout L, const float* __restrict__ depth, const float* __restrict__ feat) {
  col_sums_body<FH>(L, depth, feat, (int)blockIdx.x);
}

// Round 6: the column sums of grid 0 and the sorted gather of the band grids touch disjoint slots
// of vsum, so they can share one launch instead of running back to back (56 + 30 us at the DHD-L geometry, B = 2).  Groups of 8
// consecutive workgroups (one per XCD: both bodies place their work by blockIdx & 7) alternate 2 : 1 between the two roles, so
// that both kinds are resident together from start to end.  The merged kernel carries the gather's 84 registers.
template <int FH>
__global__ __launch_bounds__(kBlock) void mghs_sums_cols_bands(Layout L, const float* __restrict__ depth, const float* __restrict__ feat,
                                                               int n_col_groups, int n_gather_groups, int kc) {
  const int grp = rfl((int)(blockIdx.x >> 3)), x = (int)(blockIdx.x & 7);
  const int period = grp / (kc + 1), r = grp - (kc + 1) * period;
  if (r < kc) {
    const int cg = kc * period + r;
    if (cg < n_col_groups) col_sums_body<FH>(L, depth, feat, (cg << 3) | x);
  } else if (period < n_gather_groups) {
    gather_sums_body<true>(L, depth, feat, (period << 3) | x);
  }
}

// ---------------------------------------------------------------------------------------
// 2. forward stream.  LDS: table[kTableFloats] + slot_of[kSegMaxVox] (uint16: 1 + index of the
// voxel's row in this segment's slot range, 0 = empty) + 2 ints.
// ---------------------------------------------------------------------------------------
constexpr size_t kStreamLds = (size_t)kTableFloats * 4 + (size_t)kSegMaxVox * 2 + 16;

// Every segment is written by `split` workgroups, a contiguous 1/split of the kTileC channels each (workgroup = (segment,
// part)).  Measured on the writer alone, three alternating runs per variant on one box (experiments/ab/run_split.sh):
//   DHD-S, B = 4 (3 400 segments):  1 / 2 / 4 / 8 / 16 parts: 139.7 / 137.9 / 130.8 / 144.4 / 237 us
//   DHD-S, B = 1 / 2 / 8:            44.0 -> 37.1 (4 parts) / 66 -> 66 / 254 -> 247 us
//   DHD-L geometry, B = 2:           86.3 / 77.9 / 87.7 us  (1 / 2 / 4 parts);  DHD-M, B = 3: 95.0 / 94.8 / 99.5 us
// More, shorter workgroups fill the chip more evenly (four 512-thread workgroups fit a CU: the 3 400 whole segments of a
// DHD-S batch of four are 3.3 rounds of 1 024) and their prologues (slot map, table) overlap other workgroups' streaming;
// the price is that the slot map of a segment is built once per part.  Where segments are sparse that is cheap and 16
// channels per workgroup are best; where they are dense (D = 88: twice the points, or the 32 x 88 feature maps of
// DHD-L) the table holds 32 channels per pass anyway and halves are best.  The host picks by points per segment
// (stream_split).  mghs_stream_bwd at DHD-S B = 4 gains nothing from 2 parts and loses with 4 (backward 131.5 / 132.2 / 147 us) and
// stays whole there; at the dense geometries two parts pay (round 5, A/B of three builds on one box, MGHS-only step: DHD-L
// B = 2 0.4693 -> 0.459 ms, DHD-M B = 3 0.3398 -> 0.3325), so it takes 2 where the writer takes 2.
// Element types of the dense tensors (dhd_tensor_view.dtype): a lane always moves 16 bytes = kVox<T> voxels of one channel run.
// Half types: float32 sums from the table, rounded to nearest even on the way out (what `.half()` / `.bfloat16()` of the
// float32 tensor would give); gradients are widened exactly.
template <class T> struct VoxVec { static constexpr int n = 16 / sizeof(T); };
template <class T> __device__ __forceinline__ vfloat4 pack_vox(const float* f);
template <> __device__ __forceinline__ vfloat4 pack_vox<float>(const float* f) { return vfloat4{f[0], f[1], f[2], f[3]}; }
template <> __device__ __forceinline__ vfloat4 pack_vox<_Float16>(const float* f) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  const h8 v = {(_Float16)f[0], (_Float16)f[1], (_Float16)f[2], (_Float16)f[3], (_Float16)f[4], (_Float16)f[5], (_Float16)f[6], (_Float16)f[7]};
  return __builtin_bit_cast(vfloat4, v);
}
template <> __device__ __forceinline__ vfloat4 pack_vox<__bf16>(const float* f) {
  typedef __bf16 b8 __attribute__((ext_vector_type(8)));
  const b8 v = {(__bf16)f[0], (__bf16)f[1], (__bf16)f[2], (__bf16)f[3], (__bf16)f[4], (__bf16)f[5], (__bf16)f[6], (__bf16)f[7]};
  return __builtin_bit_cast(vfloat4, v);
}
template <class T> __device__ __forceinline__ void unpack_vox(vfloat4 v, float* f);
template <> __device__ __forceinline__ void unpack_vox<float>(vfloat4 v, float* f) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
template <> __device__ __forceinline__ void unpack_vox<_Float16>(vfloat4 v, float* f) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  const h8 h = __builtin_bit_cast(h8, v);
#pragma unroll
  for (int k = 0; k < 8; ++k) f[k] = (float)h[k];
}
template <> __device__ __forceinline__ void unpack_vox<__bf16>(vfloat4 v, float* f) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const u4 w = __builtin_bit_cast(u4, v);
#pragma unroll
  for (int k = 0; k < 4; ++k) { f[2 * k] = __uint_as_float(w[k] << 16); f[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
}

// OP = 1: the same code instantiated once more for the single-grid operator (dhd_bev_pool_v2_fused_*), so that its launches are
// their own rows in a kernel profile instead of being averaged into the hot path's.
template <class T, int OP = 0>
__global__ __launch_bounds__(kStreamBlock) void mghs_stream_fwd(Layout L, OutPtrs out, int split) {
  constexpr int VPL = VoxVec<T>::n;                // voxels per lane and store: 4 (float32) or 8 (half types)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* table = reinterpret_cast<float*>(smem);
  unsigned short* slot_of = reinterpret_cast<unsigned short*>(smem + (size_t)kTableFloats * 4);
  int* ctl = reinterpret_cast<int*>(smem + (size_t)kTableFloats * 4 + (size_t)kSegMaxVox * 2);

  const int seg = blockIdx.x / split, pw = kTileC / split;
  const int c_begin = (blockIdx.x % split) * pw, c_end = c_begin + pw;
  Segment sg;
  if (!decode_segment(L, seg, &sg)) return;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t == 0) ctl[0] = L.nzoff[sg.v0];
  if (t == 1) ctl[1] = L.nzoff[sg.v0 + sg.nvox];
  for (int i = t; i < kSegMaxVox / 2; i += kStreamBlock) reinterpret_cast<unsigned*>(slot_of)[i] = 0u;
  __syncthreads();
  const int k0 = rfl(ctl[0]), nnz = rfl(ctl[1]) - k0;
  for (int j = t; j < nnz; j += kStreamBlock) slot_of[L.nzvox[k0 + j] - sg.v0] = (unsigned short)(j + 1);
  const int cp = min(channels_per_pass(nnz), c_end - c_begin), pitch = nnz | 1;
  const int nvec = sg.nvox / VPL;
  T* og = reinterpret_cast<T*>(out.p[sg.g]);
  const long sb = out.sb[sg.g], sz = out.sz[sg.g], sc = out.sc[sg.g];
  for (int c_lo = c_begin; c_lo < c_end; c_lo += cp) {
    __syncthreads();  // slot_of complete / previous pass done with the table
    // table[cc*pitch + j] = vsum[(k0+j)*64 + c_lo + cc]: runs of cp floats, coalesced
    for (int i = t; i < nnz * cp; i += kStreamBlock) {
      const int j = i / cp, cc = i % cp;
      table[cc * pitch + j] = L.vsum[(size_t)(k0 + j) * kTileC + c_lo + cc];
    }
    __syncthreads();
    // The cp channel runs of this pass (nvec 16-byte vectors each) are one flat index space: iteration `it` of wave
    // `wv` stores vectors [(it * waves + wv) * 64, + 64).  With a wave per channel run, 200 vectors were 3 full
    // stores + one 8-lane store (32 store instructions per wave and pass for 25 stores' worth of bytes); flat, every
    // store but the last of the pass is a full 1 KB and the workgroup writes 8 KB contiguous per iteration.
    T* base = og + (size_t)sg.b * sb + (size_t)sg.z * sz + (size_t)c_lo * sc + (size_t)sg.y0 * sg.nx;
    const int total = cp * nvec;
    constexpr int kStride = kStreamWaves * DHD_WAVE;
    const int q_step = kStride / nvec, r_step = kStride % nvec;
    int idx = wv * DHD_WAVE + lane;
    int cc = idx / nvec, i = idx % nvec;
    for (; idx < total; idx += kStride) {
      float f[VPL];
#pragma unroll
      for (int k = 0; k < VPL; ++k) f[k] = 0.f;
      unsigned sl[VPL / 2];                        // two 16-bit slot words per 32-bit word
      if (VPL == 4) {
        const uint2 w = *reinterpret_cast<const uint2*>(slot_of + 4 * i);
        sl[0] = w.x; sl[1] = w.y;
      } else {
        const uint4 w = *reinterpret_cast<const uint4*>(slot_of + 8 * i);
        sl[0] = w.x; sl[1] = w.y; sl[VPL / 2 - 2] = w.z; sl[VPL / 2 - 1] = w.w;
      }
      unsigned any = 0;
#pragma unroll
      for (int k = 0; k < VPL / 2; ++k) any |= sl[k];
      if (any) {
#pragma unroll
        for (int k = 0; k < VPL / 2; ++k) {
          const unsigned s0 = sl[k] & 0xffffu, s1 = sl[k] >> 16;
          if (s0) f[2 * k] = table[cc * pitch + (s0 - 1)];
          if (s1) f[2 * k + 1] = table[cc * pitch + (s1 - 1)];
        }
      }
      vfloat4* dst = reinterpret_cast<vfloat4*>(base + (size_t)cc * sc) + i;
      // streamed once, not re-read here: non-temporal, so the output stream does not evict vsum from L2
      __builtin_nontemporal_store(pack_vox<T>(f), dst);
      cc += q_step;
      i += r_step;
      if (i >= nvec) { i -= nvec; ++cc; }
    }
  }
}

// ---------------------------------------------------------------------------------------
// backward 1: stream out_grad, extract the rows of the non-empty voxels into vsum[slot][64].
// Segments without any point are skipped: their out_grad is never needed.
// ---------------------------------------------------------------------------------------
template <class T, int OP = 0>
__global__ __launch_bounds__(kStreamBlock) void mghs_stream_bwd(Layout L, InPtrs og_in, int split) {
  constexpr int VPL = VoxVec<T>::n;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* table = reinterpret_cast<float*>(smem);
  unsigned short* slot_of = reinterpret_cast<unsigned short*>(smem + (size_t)kTableFloats * 4);
  int* ctl = reinterpret_cast<int*>(smem + (size_t)kTableFloats * 4 + (size_t)kSegMaxVox * 2);

  // `split` channel parts per segment (1 on the MGHS path, see above; the single-grid operator, whose whole tensor is 200
  // segments, takes 4 so that every CU has work)
  const int seg = blockIdx.x / split, pw = kTileC / split;
  const int c_begin = (blockIdx.x % split) * pw, c_end = c_begin + pw;
  Segment sg;
  if (!decode_segment(L, seg, &sg)) return;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t == 0) ctl[0] = L.nzoff[sg.v0];
  if (t == 1) ctl[1] = L.nzoff[sg.v0 + sg.nvox];
  for (int i = t; i < kSegMaxVox / 2; i += kStreamBlock) reinterpret_cast<unsigned*>(slot_of)[i] = 0u;
  __syncthreads();
  const int k0 = rfl(ctl[0]), nnz = rfl(ctl[1]) - k0;
  if (nnz == 0) return;
  for (int j = t; j < nnz; j += kStreamBlock) slot_of[L.nzvox[k0 + j] - sg.v0] = (unsigned short)(j + 1);
  const int cp = min(channels_per_pass(nnz), c_end - c_begin), pitch = nnz | 1;
  const int nvec = sg.nvox / VPL;
  const T* og = reinterpret_cast<const T*>(og_in.p[sg.g]);
  const long sb = og_in.sb[sg.g], sz = og_in.sz[sg.g], sc = og_in.sc[sg.g];
  // Only ~7 % of the voxels (~56 % of the 128-byte lines of out_grad) hold a point, and the gradient of every other
  // voxel is never needed: a lane reads its 16 bytes only if one of its four voxels does.  The loads stay
  // unconditional instructions (lanes without work re-read one resident dummy line), so that the four of a channel
  // run are in flight together instead of each waiting behind a branch.
  // As in the forward writer the cp channel runs of a pass form one flat (run, vector) index space (a wave per 200-vector
  // run issued 4 loads of which the last served 8 lanes: 32 load instructions per wave for 25 loads' worth); five
  // consecutive positions of a wave are requested together.
  constexpr int kStride = kStreamWaves * DHD_WAVE, kBatch = 5;
  const int q_step = kStride / nvec, r_step = kStride % nvec;
  const T* gbase = og + (size_t)sg.b * sb + (size_t)sg.z * sz + (size_t)sg.y0 * sg.nx;
  __syncthreads();  // slot_of complete
  for (int c_lo = c_begin; c_lo < c_end; c_lo += cp) {
    __syncthreads();
    const int total = cp * nvec;
    int idx = wv * DHD_WAVE + lane;
    int cc = idx / nvec, i = idx % nvec;
    for (; idx - lane < total; ) {                 // wave-uniform condition
      unsigned sl[kBatch][VPL / 2], any[kBatch];
      vfloat4 v[kBatch];
      int ccs[kBatch];
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        const bool in = idx < total;
        if (VPL == 4) {
          const uint2 w = in ? *reinterpret_cast<const uint2*>(slot_of + 4 * i) : make_uint2(0u, 0u);
          sl[k][0] = w.x; sl[k][1] = w.y;
        } else {
          const uint4 w = in ? *reinterpret_cast<const uint4*>(slot_of + 8 * i) : make_uint4(0u, 0u, 0u, 0u);
          sl[k][0] = w.x; sl[k][1] = w.y; sl[k][VPL / 2 - 2] = w.z; sl[k][VPL / 2 - 1] = w.w;
        }
        any[k] = 0;
#pragma unroll
        for (int u = 0; u < VPL / 2; ++u) any[k] |= sl[k][u];
        ccs[k] = cc;
        const vfloat4* p = any[k] ? reinterpret_cast<const vfloat4*>(gbase + (size_t)(c_lo + cc) * sc) + i
                                  : reinterpret_cast<const vfloat4*>(L.vsum);
        v[k] = __builtin_nontemporal_load(p);
        idx += kStride;
        cc += q_step;
        i += r_step;
        if (i >= nvec) { i -= nvec; ++cc; }
      }
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        if (any[k]) {
          float f[VPL];
          unpack_vox<T>(v[k], f);
#pragma unroll
          for (int u = 0; u < VPL / 2; ++u) {
            const unsigned s0 = sl[k][u] & 0xffffu, s1 = sl[k][u] >> 16;
            if (s0) table[ccs[k] * pitch + (s0 - 1)] = f[2 * u];
            if (s1) table[ccs[k] * pitch + (s1 - 1)] = f[2 * u + 1];
          }
        }
      }
    }
    __syncthreads();
    for (int i = t; i < nnz * cp; i += kStreamBlock) {
      const int j = i / cp, cc = i % cp;
      L.vsum[(size_t)(k0 + j) * kTileC + c_lo + cc] = table[cc * pitch + j];
    }
  }
}

// ---------------------------------------------------------------------------------------
// backward 2: one wave per feature pixel (b, n, h, w); lane = channel.  The pixel's D frustum points
// are entries of at most two voxels each (grid 0 and the pixel's band grid); their slots were
// recorded per point by prepare (p_slot).  With g = extracted out_grad row of a voxel:
//   depth_grad[p]     = <g_full(p), feat[q,:]> + <g_band(p), feat[q,:]>     (DPP wave reduction)
//   feat_grad[q,:]    = sum over the pixel's entries of g * depth[p]         (register accumulation)
// Every output element has exactly one writer: no atomics, no memset, deterministic.
// ---------------------------------------------------------------------------------------
// sum over the 16 lanes of a DPP row (row_ror 8, 4, 2, 1): every lane of the row ends up with the total
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));
  return v;
}

// Round 2 form: a table row (64 channels = 256 bytes) is read by 16 lanes x 16 bytes, so one load instruction gathers the
// rows of FOUR candidates (lane = (candidate slot e = lane / 16, channel quad cq = lane % 16)), 32 rows in flight per wave
// instead of 16, a quarter of the load instructions; <g, f> is a 16-lane DPP row sum.  Candidate i = 4 u + e of a batch
// of 64 is handled in round u by slot e; lane (e, cq == u) keeps its point's total, so that the depth gradients of a
// batch leave in one store instruction as before.
__global__ __launch_bounds__(kBlock) void mghs_pixel_bwd(Layout L, const float* __restrict__ depth,
                                                          const float* __restrict__ feat, float* __restrict__ depth_grad,
                                                          float* __restrict__ feat_grad) {
  const int lane = threadIdx.x & 63;
  const int es = lane >> 4, cq = lane & 15;
  // Workgroups go round-robin over the 8 XCDs: XCD x takes the x-th eighth of the pixels, so that the pixels
  // that share voxels (neighbours in the image, the same camera) gather their rows through one L2.
  const int per_xcd = gridDim.x >> 3;
  const int wg = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int q = rfl(wg * (kBlock / DHD_WAVE) + (threadIdx.x >> 6));  // pixel id = row of feat_nhwc (forced scalar)
  if (q >= L.B * L.N * L.hw) return;
  const int bn = q / L.hw, pl = q % L.hw;
  const int p0 = bn * L.dhw + pl;  // point id of depth bin 0; bin d is p0 + d*hw
  const vfloat4 f = reinterpret_cast<const vfloat4*>(feat)[(size_t)q * (kTileC / 4) + cq];
  const vfloat4* gc = reinterpret_cast<const vfloat4*>(L.vsum) + cq;
  vfloat4 fg = {0.f, 0.f, 0.f, 0.f};
  // 2*D (depth bin, grid class) candidates, 64 at a time: lane c loads the slot / depth of candidate e0 + c = (bin, class)
  const int n_cand = 2 * L.D;
  constexpr int kRounds = DHD_WAVE / 4, kU = 8;   // 16 rounds of 4 candidates, 8 rounds (32 rows) in flight
  for (int e0 = 0; e0 < n_cand; e0 += DHD_WAVE) {
    const int nb = min(DHD_WAVE, n_cand - e0);
    int slot = -1;
    float dv = 0.f;
    if (lane < nb) {
      const int e = e0 + lane;
      const int pid = p0 + (e >> 1) * L.hw;
      slot = L.p_slot[(e & 1) * L.P + pid];
      dv = depth[pid];
    }
    float mine = 0.f;
#pragma unroll
    for (int u0 = 0; u0 < kRounds; u0 += kU) {
      if (4 * u0 >= nb) break;  // wave-uniform
      vfloat4 g[kU];
      float d[kU];
#pragma unroll
      for (int j = 0; j < kU; ++j) {
        const int i = 4 * (u0 + j) + es;          // this slot's candidate in round u0 + j
        const int sl = __shfl(slot, i, DHD_WAVE);
        d[j] = __shfl(dv, i, DHD_WAVE);
        g[j] = sl >= 0 ? gc[(size_t)sl * (kTileC / 4)] : fg * 0.f;   // dead candidates: g = 0
      }
#pragma unroll
      for (int j = 0; j < kU; ++j) {
        fg += g[j] * d[j];
        const float part = row16_sum((g[j].x * f.x + g[j].y * f.y) + (g[j].z * f.z + g[j].w * f.w));
        // the two classes of a point are candidates 2 m, 2 m + 1 = slots (0, 1) or (2, 3) of one round
        const float both = part + __shfl_xor(part, 16, DHD_WAVE);
        if (cq == u0 + j) mine = both;
      }
    }
    // lane (es, cq) holds the total of candidate 4 cq + es; even slots write their point
    const int cand = 4 * cq + es;
    if (!(es & 1) && cand < nb) depth_grad[p0 + ((e0 + cand) >> 1) * L.hw] = mine;
  }
  // feature gradient: sum of the four slots' accumulators
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float v = fg[c];
    v += __shfl_xor(v, 16, DHD_WAVE);
    v += __shfl_xor(v, 32, DHD_WAVE);
    fg[c] = v;
  }
  if (es == 0) {
    if (L.flags & DHD_MGHS_FEAT_GRAD_NCHW) {
      // (B*N, C, fH, fW): channel 4 cq + c of pixel pl; the 64 four-byte stores of a pixel meet those of its row
      // neighbours (the next waves of this XCD's pixel range) in L2 before the lines leave it
      float* dst = feat_grad + ((size_t)bn * kTileC + 4 * cq) * L.hw + pl;
#pragma unroll
      for (int c = 0; c < 4; ++c) dst[(size_t)c * L.hw] = fg[c];
    } else {
      reinterpret_cast<vfloat4*>(feat_grad)[(size_t)q * (kTileC / 4) + cq] = fg;
    }
  }
}

// generic path with DHD_MGHS_FEAT_GRAD_NCHW: (bn, hw, C) -> (bn, C, hw)
__global__ __launch_bounds__(kLiftBlock) void mghs_fg_to_nchw(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
  __shared__ float tile[64][65];
  transpose_tile(tile, src, dst, rows, cols, blockIdx.z, blockIdx.y * 64, blockIdx.x * 64);
}

// depth_grad = part(grid 0) + part(band grid)
__global__ __launch_bounds__(kBlock) void mghs_sum_parts(const float* __restrict__ part, int P, float* __restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < P) out[i] = part[i] + part[P + i];
}

// =======================================================================================
// Generic dense-row path (any channel count, any grid shape).  One workgroup per output row tile:
// [<=64 channels][<=256 voxels] in LDS, gathered voxel by voxel (CL lanes cover the channels, 64/CL
// points in flight per wave), streamed to/from HBM as whole rows.
// =======================================================================================
constexpr int kRowBlock = 512;
constexpr int kRowWaves = kRowBlock / DHD_WAVE;

struct RowTile {
  int g, b, z, y, nx, ny, nz;
  int vrow;  // voxel id of (b,z,y,x=0)
};

__device__ __forceinline__ bool decode_row(const Layout& L, int r, RowTile* t) {
  if (r >= L.R) return false;
  int g = 0;
#pragma unroll
  for (int k = 1; k < DHD_MAX_GRIDS; ++k)
    if (k < L.G && r >= L.row_base[k]) g = k;
  const dhd_grid& gr = L.grid[g];
  int local = r - L.row_base[g];
  t->g = g; t->nx = gr.n[0]; t->ny = gr.n[1]; t->nz = gr.n[2];
  t->y = local % gr.n[1];
  int bz = local / gr.n[1];
  t->z = bz % gr.n[2];
  t->b = bz / gr.n[2];
  t->vrow = L.vox_base[g] + local * gr.n[0];
  return true;
}

// blockIdx -> (output row, x chunk): XCD-grouped (block b runs on XCD b % 8; all chunks of kRowGroup
// rows stay on one XCD, back to back), heavy (grid 0) row groups front-loaded 1:k with an odd period.
__device__ __forceinline__ int scheduled_tile(const Layout& L, int block, int* xchunk) {
  const int pos = xcd_grouped_tile(block, kRowGroup * L.nxc);
  *xchunk = pos % L.nxc;
  const int rpos = pos / L.nxc;
  int grp = rpos / kRowGroup;
  const int within = rpos % kRowGroup;
  if (L.sched_heavy > 0) {
    const int k1 = L.sched_ratio + 1;
    if (grp < L.sched_heavy * k1) {
      const int q = grp / k1, r = grp % k1;
      grp = (r == 0) ? q : L.sched_heavy + q * L.sched_ratio + (r - 1);
    }
  }
  return grp * kRowGroup + within;
}

__global__ __launch_bounds__(kRowBlock) void mghs_rows_fwd(Layout L, const float* __restrict__ depth,
                                                           const float* __restrict__ feat, OutPtrs out, int tile_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = reinterpret_cast<float*>(smem);                                 // [kTileC][tile_stride]
  int* offs = reinterpret_cast<int*>(smem + (size_t)kTileC * tile_stride * 4);  // [kMaxTileX + 1]
  RowTile rt;
  int xchunk;
  if (!decode_row(L, scheduled_tile(L, blockIdx.x, &xchunk), &rt)) return;
  const int c0 = blockIdx.y * kTileC;
  const int cn = min(kTileC, L.C - c0);
  const int x0 = xchunk * kMaxTileX;
  if (x0 >= rt.nx) return;
  const int xn = min(kMaxTileX, rt.nx - x0);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;

  for (int i = t; i <= xn; i += kRowBlock) offs[i] = L.offset[rt.vrow + x0 + i];
  for (int i = t; i < cn * tile_stride; i += kRowBlock) tile[i] = 0.f;
  __syncthreads();

  if (offs[xn] != offs[0]) {
    const int CL = next_pow2(cn);
    const int nsub = DHD_WAVE / CL;
    const int c = lane % CL, sub = lane / CL;
    const bool c_ok = c < cn;
    const float* featc = feat + c0 + c;
    for (int x = wv; x < xn; x += kRowWaves) {
      const int s = offs[x], e = offs[x + 1];
      if (e == s) continue;
      float acc = 0.f;
      for (int s0 = s; s0 < e; s0 += DHD_WAVE) {
        const int nb = min(DHD_WAVE, e - s0);
        int pix = 0;
        float dv = 0.f;
        if (lane < nb) {
          const int4 en = L.s_ent[s0 + lane];
          pix = en.y;
          dv = depth[en.x];
        }
        // uniform trip count: the cross-lane reads below must be executed by every lane
        const int steps = (nb + nsub - 1) / nsub;
        for (int k = 0; k < steps; ++k) {
          const int i = k * nsub + sub;
          const bool live = i < nb;
          int q = __shfl(pix, live ? i : 0, DHD_WAVE);
          float d = __shfl(dv, live ? i : 0, DHD_WAVE);
          float f = (live && c_ok) ? featc[(size_t)q * L.C] : 0.f;
          acc = fmaf(d, f, acc);
        }
      }
      for (int m = CL; m < DHD_WAVE; m <<= 1) acc += __shfl_xor(acc, m, DHD_WAVE);
      if (sub == 0 && c_ok) tile[c * tile_stride + x] = acc;
    }
    __syncthreads();
  }

  float* og = out.p[rt.g];
  const bool vec = ((rt.nx & 3) == 0) && ((xn & 3) == 0) && ((x0 & 3) == 0);
  for (int cc = wv; cc < cn; cc += kRowWaves) {
    size_t row = (size_t)rt.b * out.sb[rt.g] + (size_t)rt.z * out.sz[rt.g] + (size_t)(c0 + cc) * out.sc[rt.g] + (size_t)rt.y * rt.nx + x0;
    const float* src = tile + cc * tile_stride;
    if (vec) {
      vfloat4* dst = reinterpret_cast<vfloat4*>(og + row);
      for (int i = lane; i < xn / 4; i += DHD_WAVE) {
        vfloat4 v = {src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]};
        __builtin_nontemporal_store(v, dst + i);
      }
    } else {
      for (int i = lane; i < xn; i += DHD_WAVE) __builtin_nontemporal_store(src[i], og + row + i);
    }
  }
}

__global__ __launch_bounds__(kRowBlock) void mghs_rows_bwd(Layout L, const float* __restrict__ depth,
                                                           const float* __restrict__ feat, InPtrs og,
                                                           float* __restrict__ feat_grad, int tile_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = reinterpret_cast<float*>(smem);
  int* offs = reinterpret_cast<int*>(smem + (size_t)kTileC * tile_stride * 4);
  RowTile rt;
  int xchunk;
  if (!decode_row(L, scheduled_tile(L, blockIdx.x, &xchunk), &rt)) return;
  const int c0 = blockIdx.y * kTileC;
  const int cn = min(kTileC, L.C - c0);
  const int x0 = xchunk * kMaxTileX;
  if (x0 >= rt.nx) return;
  const int xn = min(kMaxTileX, rt.nx - x0);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;

  for (int i = t; i <= xn; i += kRowBlock) offs[i] = L.offset[rt.vrow + x0 + i];
  __syncthreads();
  if (offs[xn] == offs[0]) return;  // no point lands in this row: its out_grad is never read

  const float* gsrc = og.p[rt.g];
  const bool vec = ((rt.nx & 3) == 0) && ((xn & 3) == 0) && ((x0 & 3) == 0);
  for (int cc = wv; cc < cn; cc += kRowWaves) {
    size_t row = (size_t)rt.b * og.sb[rt.g] + (size_t)rt.z * og.sz[rt.g] + (size_t)(c0 + cc) * og.sc[rt.g] + (size_t)rt.y * rt.nx + x0;
    float* dst = tile + cc * tile_stride;
    if (vec) {
      const float4* src = reinterpret_cast<const float4*>(gsrc + row);
      for (int i = lane; i < xn / 4; i += DHD_WAVE) {
        float4 v = src[i];
        dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
      }
    } else {
      for (int i = lane; i < xn; i += DHD_WAVE) dst[i] = gsrc[row + i];
    }
  }
  __syncthreads();

  // depth-gradient parts: with one channel tile every (point, grid class) has one writer (plain
  // store); with several channel tiles (C > 64) the tiles' partial dot products need atomics
  float* dgp = L.dg_part + (rt.g == 0 ? 0 : L.P);
  const bool dg_atomic = L.C > kTileC;
  const int CL = next_pow2(cn);
  const int nsub = DHD_WAVE / CL;
  const int c = lane % CL, sub = lane / CL;
  const bool c_ok = c < cn;
  const float* featc = feat + c0 + c;
  float* fgc = feat_grad + c0 + c;
  for (int x = wv; x < xn; x += kRowWaves) {
    const int s = offs[x], e = offs[x + 1];
    if (e == s) continue;
    const float g = c_ok ? tile[c * tile_stride + x] : 0.f;
    for (int s0 = s; s0 < e; s0 += DHD_WAVE) {
      const int nb = min(DHD_WAVE, e - s0);
      int pix = 0, pid = 0;
      float dv = 0.f;
      if (lane < nb) {
        const int4 en = L.s_ent[s0 + lane];
        pix = en.y;
        pid = en.x;
        dv = depth[pid];
      }
      float mine = 0.f;  // depth-gradient contribution of the point this lane loaded
      const int steps = (nb + nsub - 1) / nsub;
      for (int k = 0; k < steps; ++k) {
        const int i = k * nsub + sub;
        const bool live = i < nb;
        int q = __shfl(pix, live ? i : 0, DHD_WAVE);
        float d = __shfl(dv, live ? i : 0, DHD_WAVE);
        float prod = 0.f;
        if (live && c_ok) {
          float f = featc[(size_t)q * L.C];
          unsafeAtomicAdd(fgc + (size_t)q * L.C, g * d);
          prod = g * f;
        }
        float tot = group_sum(prod, CL);
        // hand the sum of point (k*nsub + j) to lane (k*nsub + j): it sits in every lane of sub-slot j
        int owner_sub = lane - k * nsub;
        float got = __shfl(tot, (owner_sub >= 0 && owner_sub < nsub) ? owner_sub * CL : 0, DHD_WAVE);
        if (owner_sub >= 0 && owner_sub < nsub) mine = got;
      }
      if (lane < nb) {
        if (dg_atomic) unsafeAtomicAdd(dgp + pid, mine); else dgp[pid] = mine;
      }
    }
  }
}

void rows_launch_shape(const Layout& L, int* stride, size_t* smem, dim3* grid) {
  int nx_max = 0;
  for (int g = 0; g < L.G; ++g) nx_max = nx_max > L.grid[g].n[0] ? nx_max : L.grid[g].n[0];
  int xt = nx_max < kMaxTileX ? nx_max : kMaxTileX;
  *stride = xt | 1;  // odd row stride: the transposed (lane = channel) LDS accesses hit 32 distinct banks
  *smem = (size_t)kTileC * (*stride) * 4 + (size_t)(kMaxTileX + 1) * 4;
  *grid = dim3(xcd_grouped_blocks(L.R * L.nxc, kRowGroup * L.nxc), dhd_cdiv(L.C, kTileC), 1);
}

}  // namespace

int launch_stream_fwd(const Layout& L, const OutPtrs& o, int split, hipStream_t st, bool op) {
  if (!L.compact || L.n_segs <= 0) return DHD_EINVAL;
  if (op) {
    if (o.dtype != DHD_F32) return DHD_EUNSUPPORTED;
    hipLaunchKernelGGL((mghs_stream_fwd<float, 1>), dim3(L.n_segs * split), dim3(kStreamBlock), kStreamLds, st, L, o, split);
    DHD_LAUNCH_CHECK();
    return DHD_OK;
  }
  if (o.dtype == DHD_F16) hipLaunchKernelGGL(mghs_stream_fwd<_Float16>, dim3(L.n_segs * split), dim3(kStreamBlock), kStreamLds, st, L, o, split);
  else if (o.dtype == DHD_BF16) hipLaunchKernelGGL(mghs_stream_fwd<__bf16>, dim3(L.n_segs * split), dim3(kStreamBlock), kStreamLds, st, L, o, split);
  else hipLaunchKernelGGL(mghs_stream_fwd<float>, dim3(L.n_segs * split), dim3(kStreamBlock), kStreamLds, st, L, o, split);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

int launch_stream_bwd(const Layout& L, const InPtrs& in, int split, hipStream_t st, bool op) {
  if (!L.compact || L.n_segs <= 0) return DHD_EINVAL;
  if (op) {
    if (in.dtype != DHD_F32) return DHD_EUNSUPPORTED;
    hipLaunchKernelGGL((mghs_stream_bwd<float, 1>), dim3(L.n_segs * split), dim3(kStreamBlock), kStreamLds, st, L, in, split);
    DHD_LAUNCH_CHECK();
    return DHD_OK;
  }
  if (in.dtype == DHD_F16) hipLaunchKernelGGL(mghs_stream_bwd<_Float16>, dim3(L.n_segs * split), dim3(kStreamBlock), kStreamLds, st, L, in, split);
  else if (in.dtype == DHD_BF16) hipLaunchKernelGGL(mghs_stream_bwd<__bf16>, dim3(L.n_segs * split), dim3(kStreamBlock), kStreamLds, st, L, in, split);
  else hipLaunchKernelGGL(mghs_stream_bwd<float>, dim3(L.n_segs * split), dim3(kStreamBlock), kStreamLds, st, L, in, split);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

}  // namespace dhd

using namespace dhd;

// Channel parts per segment of the streaming writer (see mghs_stream_fwd): 4 where a segment holds few points (DHD-S: 218
// per segment), 2 where it holds many (D = 88: 437; the DHD-L feature maps: 1 750).
static int stream_split(const Layout& L) {
  return (long)L.P < 300L * L.n_segs ? 4 : 2;
}

extern "C" {

int dhd_mghs_forward_gather(const dhd_mghs_desc* desc, const float* depth, const float* feat_nhwc, const dhd_mghs_workspace* workspace,
                            void* stream) {
  Layout L;
  int rc = make_layout(desc, workspace, &L);
  if (rc) return rc;
  if (!workspace || !depth || !feat_nhwc) return DHD_EINVAL;
  if (!L.compact) return DHD_OK;  // the generic path gathers inside its row kernel
  hipStream_t st = dhd_stream(stream);
  // column form for the full-height grid (Layout::columns), sorted gather for the rest
  if (L.columns) {
    hipLaunchKernelGGL(mghs_zero_grid0_rows, dim3(1024), dim3(kBlock), 0, st, L);
    const int waves = L.B * L.N * L.fw * dhd_cdiv(L.D, kDepthChunk);
    const dim3 grid(dhd_cdiv(waves, kBlock / DHD_WAVE));
    const int n_gather = dhd_cdiv((long)L.P, 8 * kBlock) * 8;   // the band grids' entries: at most one per point
    // one launch for both (round 6; experiments/ab/run_merged_sums.sh, DHD-L geometry B = 2, MGHS-only step, alternating runs on one
    // box: 0.479 / 0.475 / 0.483 ms separate -> 0.464 / 0.463 / 0.469 ms merged; interleave 1:1 and 2:1 equal, 3:1 and beyond worse).
    // DHD_MGHS_SEPARATE_SUMS=1 keeps the two launches (A/B switch, read once)
    static const bool merged = getenv("DHD_MGHS_SEPARATE_SUMS") == nullptr;
    constexpr int kc = 2;
    if (merged) {
      const int ncg = dhd_cdiv(grid.x, 8), ngg = n_gather / 8;
      const int periods = ((ncg + kc - 1) / kc) > ngg ? (ncg + kc - 1) / kc : ngg;
      hipLaunchKernelGGL(mghs_sums_cols_bands<32>, dim3(periods * (kc + 1) * 8), dim3(kBlock), 0, st, L, depth, feat_nhwc, ncg, ngg, kc);
    } else {
      hipLaunchKernelGGL(mghs_col_sums<32>, grid, dim3(kBlock), 0, st, L, depth, feat_nhwc);
      hipLaunchKernelGGL(mghs_gather_sums<true>, dim3(n_gather), dim3(kBlock), 0, st, L, depth, feat_nhwc);
    }
  } else {
    // 2P is an upper bound of the entry count; waves past the real count exit at once
    hipLaunchKernelGGL(mghs_gather_sums<false>, dim3(dhd_cdiv(2L * L.P, 8 * kBlock) * 8), dim3(kBlock), 0, st, L, depth, feat_nhwc);
  }
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

static int forward_stream_impl(const dhd_mghs_desc* desc, const float* depth, const float* feat_nhwc,
                               float* const out[DHD_MAX_GRIDS], const dhd_tensor_view* views, const dhd_mghs_workspace* workspace,
                               void* stream) {
  Layout L;
  int rc = make_layout(desc, workspace, &L);
  if (rc) return rc;
  if (!workspace || !depth || !feat_nhwc || (!out && !views)) return DHD_EINVAL;
  OutPtrs o;
  if ((rc = make_views<OutPtrs, float>(L, out, views, &o))) return rc;
  hipStream_t st = dhd_stream(stream);
  if (L.compact) {
    // half types: a channel run is half as many bytes, so a workgroup takes twice the channels (measured, writer alone: DHD-S B = 4
    // 88.3 / 70.3 / 75.3 us with 4 / 2 / 1 parts, DHD-L geometry B = 2 71.8 / 63.7 / 66.6)
    if ((rc = launch_stream_fwd(L, o, o.dtype == DHD_F32 ? stream_split(L) : 2, st))) return rc;
  } else {
    int stride; size_t smem; dim3 grid;
    rows_launch_shape(L, &stride, &smem, &grid);
    hipLaunchKernelGGL(mghs_rows_fwd, grid, dim3(kRowBlock), smem, st, L, depth, feat_nhwc, o, stride);
    DHD_LAUNCH_CHECK();
  }
  return DHD_OK;
}

int dhd_mghs_forward_stream(const dhd_mghs_desc* desc, const float* depth, const float* feat_nhwc,
                            float* const out[DHD_MAX_GRIDS], const dhd_mghs_workspace* workspace, void* stream) {
  return forward_stream_impl(desc, depth, feat_nhwc, out, nullptr, workspace, stream);
}

int dhd_mghs_forward_stream_views(const dhd_mghs_desc* desc, const float* depth, const float* feat_nhwc,
                                  const dhd_tensor_view out[DHD_MAX_GRIDS], const dhd_mghs_workspace* workspace, void* stream) {
  if (!out) return DHD_EINVAL;
  return forward_stream_impl(desc, depth, feat_nhwc, nullptr, out, workspace, stream);
}

int dhd_mghs_forward(const dhd_mghs_desc* desc, const float* depth, const float* feat_nhwc,
                     float* const out[DHD_MAX_GRIDS], const dhd_mghs_workspace* workspace, void* stream) {
  int rc = dhd_mghs_forward_gather(desc, depth, feat_nhwc, workspace, stream);
  if (rc) return rc;
  return forward_stream_impl(desc, depth, feat_nhwc, out, nullptr, workspace, stream);
}

int dhd_mghs_forward_views(const dhd_mghs_desc* desc, const float* depth, const float* feat_nhwc,
                           const dhd_tensor_view out[DHD_MAX_GRIDS], const dhd_mghs_workspace* workspace, void* stream) {
  if (!out) return DHD_EINVAL;
  int rc = dhd_mghs_forward_gather(desc, depth, feat_nhwc, workspace, stream);
  if (rc) return rc;
  return forward_stream_impl(desc, depth, feat_nhwc, nullptr, out, workspace, stream);
}

static int backward_impl(const dhd_mghs_desc* desc, const float* depth, const float* feat_nhwc,
                         const float* const out_grad[DHD_MAX_GRIDS], const dhd_tensor_view* views, float* depth_grad,
                         float* feat_grad_nhwc, const dhd_mghs_workspace* workspace, void* stream) {
  Layout L;
  int rc = make_layout(desc, workspace, &L);
  if (rc) return rc;
  if (!workspace || !depth || !feat_nhwc || (!out_grad && !views) || !depth_grad || !feat_grad_nhwc) return DHD_EINVAL;
  InPtrs in;
  if ((rc = make_views<InPtrs, const float>(L, out_grad, views, &in))) return rc;
  hipStream_t st = dhd_stream(stream);
  if (L.compact) {
    // two channel parts per segment where segments are dense (the D = 88 geometries), whole segments otherwise: see stream_split
    if ((rc = launch_stream_bwd(L, in, stream_split(L) == 2 ? 2 : 1, st))) return rc;
    hipLaunchKernelGGL(mghs_pixel_bwd, dim3(dhd_cdiv((long)L.B * L.N * L.hw, 8 * (kBlock / DHD_WAVE)) * 8), dim3(kBlock), 0, st, L,
                       depth, feat_nhwc, depth_grad, feat_grad_nhwc);
    DHD_LAUNCH_CHECK();
    return DHD_OK;
  }
  const bool to_nchw = (L.flags & DHD_MGHS_FEAT_GRAD_NCHW) != 0;
  float* fg = to_nchw ? L.fg_stage : feat_grad_nhwc;
  DHD_HIP(hipMemsetAsync(L.dg_part, 0, 2 * (size_t)L.P * 4, st));
  DHD_HIP(hipMemsetAsync(fg, 0, (size_t)L.B * L.N * L.hw * L.C * 4, st));
  int stride; size_t smem; dim3 grid;
  rows_launch_shape(L, &stride, &smem, &grid);
  hipLaunchKernelGGL(mghs_rows_bwd, grid, dim3(kRowBlock), smem, st, L, depth, feat_nhwc, in, fg, stride);
  DHD_LAUNCH_CHECK();
  if (to_nchw) {
    hipLaunchKernelGGL(mghs_fg_to_nchw, dim3(dhd_cdiv(L.C, 64), dhd_cdiv(L.hw, 64), L.B * L.N), dim3(kLiftBlock), 0, st, fg,
                       feat_grad_nhwc, L.hw, L.C);
    DHD_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(mghs_sum_parts, dim3(dhd_cdiv(L.P, kBlock)), dim3(kBlock), 0, st, L.dg_part, L.P, depth_grad);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

int dhd_mghs_backward(const dhd_mghs_desc* desc, const float* depth, const float* feat_nhwc,
                      const float* const out_grad[DHD_MAX_GRIDS], float* depth_grad, float* feat_grad_nhwc,
                      const dhd_mghs_workspace* workspace, void* stream) {
  return backward_impl(desc, depth, feat_nhwc, out_grad, nullptr, depth_grad, feat_grad_nhwc, workspace, stream);
}

int dhd_mghs_backward_views(const dhd_mghs_desc* desc, const float* depth, const float* feat_nhwc,
                            const dhd_tensor_view out_grad[DHD_MAX_GRIDS], float* depth_grad, float* feat_grad_nhwc,
                            const dhd_mghs_workspace* workspace, void* stream) {
  if (!out_grad) return DHD_EINVAL;
  return backward_impl(desc, depth, feat_nhwc, nullptr, out_grad, depth_grad, feat_grad_nhwc, workspace, stream);
}

}  // extern "C"
