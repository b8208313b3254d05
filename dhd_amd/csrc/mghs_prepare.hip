// MGHS index preparation for gfx950 (MI355X): geometry -> voxel index -> device counting sort.
// Replaces 4x MGHS.get_ego_coor (models/necks/lss_heightmap.py:179-231) and 4x
// voxel_pooling_prepare_v2 (:303-371) of the reference with one geometry pass and one grouping
// shared by all grids.
//
// Output (in the workspace, see mghs_layout.h): the kept (point, grid) pairs ("entries") grouped
// by voxel, each with its point id, pixel id and slot (ordinal of its voxel among the non-empty
// voxels); per voxel the entry prefix `offset` and the slot prefix `nzoff`; per slot its voxel id.
//
// File:line citations are into /root/reference/projects/mmdet3d_plugin/.
#include <stdlib.h>

#include "lift_device.h"
#include "mghs_layout.h"

namespace dhd {
namespace {

// ---------------------------------------------------------------------------------------
// Geometry.  The operation order, the absence of FMA contraction and the IEEE division are
// part of the contract: voxel indices must be bit-identical to the reference's float32 chain
// (lss_heightmap.py:206-230 and :331-333).  torch's CPU bmm accumulates acc = 0; acc += a*b
// with separately rounded products and sums; the explicit _rn intrinsics below are never
// contracted by the compiler.
// ---------------------------------------------------------------------------------------

struct CamMats {
  float ipr[9];    // inverse(post_rot)
  float comb[9];   // sensor2ego[:3,:3] @ inverse(intrin)
  float trans[3];  // sensor2ego[:3,3]
  float ptran[3];  // post_tran
  float bda[9];
};

__device__ __forceinline__ float dot3_seq(const float* m, float x, float y, float z) {
  float acc = __fadd_rn(0.0f, __fmul_rn(m[0], x));
  acc = __fadd_rn(acc, __fmul_rn(m[1], y));
  acc = __fadd_rn(acc, __fmul_rn(m[2], z));
  return acc;
}

// LU with partial pivoting + substitution on the permuted identity, one thread, float32,
// every operation rounded separately (LAPACK sgetf2 + strsm with IEEE division; this is what
// torch.inverse reaches, lss_heightmap.py:209,220).
__device__ void inv3x3_lu(const float* src, float* dst) {
  float a[3][3];
  int perm[3] = {0, 1, 2};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) a[i][j] = src[i * 3 + j];
  for (int j = 0; j < 3; ++j) {
    int p = j;
    for (int i = j + 1; i < 3; ++i)
      if (fabsf(a[i][j]) > fabsf(a[p][j])) p = i;
    if (p != j) {
      for (int k = 0; k < 3; ++k) { float t = a[j][k]; a[j][k] = a[p][k]; a[p][k] = t; }
      int t = perm[j]; perm[j] = perm[p]; perm[p] = t;
    }
    for (int i = j + 1; i < 3; ++i) {
      a[i][j] = __fdiv_rn(a[i][j], a[j][j]);
      for (int k = j + 1; k < 3; ++k) a[i][k] = __fsub_rn(a[i][k], __fmul_rn(a[i][j], a[j][k]));
    }
  }
  for (int c = 0; c < 3; ++c) {
    float b[3];
    for (int i = 0; i < 3; ++i) b[i] = (perm[i] == c) ? 1.0f : 0.0f;
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < i; ++k) b[i] = __fsub_rn(b[i], __fmul_rn(a[i][k], b[k]));
    for (int i = 2; i >= 0; --i) {
      for (int k = i + 1; k < 3; ++k) b[i] = __fsub_rn(b[i], __fmul_rn(a[i][k], b[k]));
      b[i] = __fdiv_rn(b[i], a[i][i]);
    }
    for (int i = 0; i < 3; ++i) dst[i * 3 + c] = b[i];
  }
}

__device__ void load_camera(const dhd_calib& cal, int bn, int b, CamMats* m) {
  const float* s2e = cal.sensor2ego + (size_t)bn * 16;
  if (cal.inv_post_rot) {
    for (int i = 0; i < 9; ++i) m->ipr[i] = cal.inv_post_rot[(size_t)bn * 9 + i];
  } else {
    inv3x3_lu(cal.post_rot + (size_t)bn * 9, m->ipr);
  }
  if (cal.combine) {
    for (int i = 0; i < 9; ++i) m->comb[i] = cal.combine[(size_t)bn * 9 + i];
  } else {
    float ik[9];
    inv3x3_lu(cal.intrin + (size_t)bn * 9, ik);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        float acc = __fadd_rn(0.0f, __fmul_rn(s2e[i * 4 + 0], ik[0 * 3 + j]));
        acc = __fadd_rn(acc, __fmul_rn(s2e[i * 4 + 1], ik[1 * 3 + j]));
        acc = __fadd_rn(acc, __fmul_rn(s2e[i * 4 + 2], ik[2 * 3 + j]));
        m->comb[i * 3 + j] = acc;
      }
  }
  for (int i = 0; i < 3; ++i) {
    m->trans[i] = s2e[i * 4 + 3];
    m->ptran[i] = cal.post_tran[(size_t)bn * 3 + i];
  }
  for (int i = 0; i < 9; ++i) m->bda[i] = cal.bda[(size_t)b * 9 + i];
}

// MGHS.get_ego_coor for one frustum point (u, v, d).
__device__ __forceinline__ void frustum_to_ego(const CamMats& m, float u, float v, float d, float* e) {
  float px = __fsub_rn(u, m.ptran[0]);
  float py = __fsub_rn(v, m.ptran[1]);
  float pz = __fsub_rn(d, m.ptran[2]);
  float qx = dot3_seq(m.ipr + 0, px, py, pz);
  float qy = dot3_seq(m.ipr + 3, px, py, pz);
  float qz = dot3_seq(m.ipr + 6, px, py, pz);
  float rx = __fmul_rn(qx, qz);
  float ry = __fmul_rn(qy, qz);
  float cx = __fadd_rn(dot3_seq(m.comb + 0, rx, ry, qz), m.trans[0]);
  float cy = __fadd_rn(dot3_seq(m.comb + 3, rx, ry, qz), m.trans[1]);
  float cz = __fadd_rn(dot3_seq(m.comb + 6, rx, ry, qz), m.trans[2]);
  e[0] = dot3_seq(m.bda + 0, cx, cy, cz);
  e[1] = dot3_seq(m.bda + 3, cx, cy, cz);
  e[2] = dot3_seq(m.bda + 6, cx, cy, cz);
}

// voxel_pooling_prepare_v2's index rule (:331-342): idx = trunc_toward_zero((p - lower) / interval),
// kept iff 0 <= idx and float(idx) < size on all three axes.  Returns the voxel index inside the
// grid, ((b*nz + z)*ny + y)*nx + x, or -1.
__device__ __forceinline__ int voxel_of(const dhd_grid& g, const float* e, int b) {
  int idx[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float t = __fdiv_rn(__fsub_rn(e[a], g.lower[a]), g.interval[a]);
    float tt = truncf(t);
    if (!(tt >= 0.0f && tt < g.size[a])) return -1;  // also rejects NaN
    int ii = (int)tt;
    if (ii >= g.n[a]) return -1;
    idx[a] = ii;
  }
  return ((b * g.n[2] + idx[2]) * g.n[1] + idx[1]) * g.n[0] + idx[0];
}

// One thread per camera: the two 3x3 inverses are ~25 serial IEEE divisions, far too slow to
// repeat in the prologue of every geometry workgroup.
__device__ __forceinline__ void camera_block(const Layout& L, const dhd_calib& cal, int block) {
  const int bn = block * kLiftBlock + threadIdx.x;
  if (bn >= L.B * L.N) return;
  CamMats m;
  load_camera(cal, bn, bn / L.N, &m);
  float* dst = L.cam + (size_t)bn * kCamFloats;
  const float* src = reinterpret_cast<const float*>(&m);
  for (int i = 0; i < (int)(sizeof(CamMats) / 4); ++i) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------
// Prologue: everything of a lift that depends on nothing computed before it, as the roles of ONE launch (a block
// takes the role its index falls into): zero-fill of the counters and of the scan's chunk aggregates, the per-camera
// matrices, height argmax -> band id, NCHW -> NHWC of the context features.  (Round 2 issued them as a memset and
// four kernels of 5-8 us each, none of which fills the chip.)
// ---------------------------------------------------------------------------------------
struct PrologueArgs {
  int n_zero, n_cam, n_band, n_tr;   // blocks per role
  // zero-fill: [zero_ptr, zero_ptr + zero_bytes), 256-byte aligned
  char* zero_ptr;
  size_t zero_bytes;
  // band
  const float* height;
  int n_height;
  uint8_t* band;
  BandLut lut;
  // transposition (bn, C, hw) -> (bn, hw, C)
  const float* feat_nchw;
  float* feat_nhwc;
  int tr_cols_tiles, tr_rows_tiles, tr_vec;
};
constexpr int kZeroBytesPerBlock = kLiftBlock * 16 * 8;   // 32 KB per block: eight 16-byte stores per thread

__global__ __launch_bounds__(kLiftBlock) void mghs_prologue(Layout L, dhd_calib cal, PrologueArgs a) {
  __shared__ float tile[64][65];
  int blk = blockIdx.x;
  if (blk < a.n_zero) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i z = {0, 0, 0, 0};
    const size_t lo = (size_t)blk * kZeroBytesPerBlock;
    const size_t hi = lo + kZeroBytesPerBlock < a.zero_bytes ? lo + kZeroBytesPerBlock : a.zero_bytes;
    for (size_t o = lo + (size_t)threadIdx.x * 16; o < hi; o += (size_t)kLiftBlock * 16) *reinterpret_cast<v4i*>(a.zero_ptr + o) = z;
    return;
  }
  blk -= a.n_zero;
  if (blk < a.n_cam) { camera_block(L, cal, blk); return; }
  blk -= a.n_cam;
  if (blk < a.n_band) { height_band_block(blk, a.height, L.B * L.N * L.hw, a.n_height, L.hw, a.lut, a.band); return; }
  blk -= a.n_band;
  if (blk < a.n_tr) {
    const int per_b = a.tr_cols_tiles * a.tr_rows_tiles;
    const int b = blk / per_b, rem = blk % per_b;
    const int r0 = (rem / a.tr_cols_tiles) * 64, c0 = (rem % a.tr_cols_tiles) * 64;
    if (a.tr_vec) transpose4_tile(tile, a.feat_nchw, a.feat_nhwc, L.C, L.hw, b, r0, c0);
    else transpose_tile(tile, a.feat_nchw, a.feat_nhwc, L.C, L.hw, b, r0, c0);
  }
}

// Counting with run aggregation.  A wave holds whole pixel COLUMNS of one depth plane: lane =
// (column cc, row hh) with HP = next_pow2(fH) lanes per column.  Rows of one column at one depth
// differ only vertically, so in the full-height grid 0 they almost always share one voxel (this is
// where its ~17 entries per voxel come from), and in the band grids neighbouring rows often share a
// z bin.  Runs of equal keys along hh are counted with ONE returning atomic by the run's first lane
// (+= run length); the members take base + offset.  Measured: the per-entry device-scope atomics
// were 49 of the 108 us of prepare at B=4.
__device__ __forceinline__ int count_runs(int* __restrict__ count, int key, int hh, int lane) {
  const bool valid = key >= 0;
  const int prev = __shfl_up(key, 1, DHD_WAVE);
  const bool first = valid && (hh == 0 || prev != key);
  const unsigned long long F = __ballot(first);
  const unsigned long long V = __ballot(valid);
  // leader of this lane's run: nearest first at or below the lane
  const unsigned long long below = F & ((2ull << lane) - 1ull);
  const int leader = valid ? 63 - __builtin_clzll(below | 1ull) : lane;
  int base = 0;
  if (first) {
    // run end: next first, or next invalid lane, above the leader
    const unsigned long long stop = (F | ~V) & ~((2ull << lane) - 1ull);
    const int end = stop ? __builtin_ctzll(stop) : DHD_WAVE;
    base = atomicAdd(&count[key], end - lane);
  }
  base = __shfl(base, leader, DHD_WAVE);
  return valid ? base + (lane - leader) : 0;
}

// BAND_ONLY (static rig, dhd_mghs_lift_static): grid 0's keys, ranks and counters are those of the earlier full prepare
// and are left alone; only the band grid's entry of every point is recomputed and counted.
template <bool BAND_ONLY>
__global__ __launch_bounds__(kBlock) void mghs_geom_count(Layout L, dhd_calib cal, const uint8_t* __restrict__ band) {
  __shared__ CamMats cam;
  const int bn = blockIdx.y;
  const int b = bn / L.N;
  if (threadIdx.x < sizeof(CamMats) / 4)
    reinterpret_cast<float*>(&cam)[threadIdx.x] = L.cam[(size_t)bn * kCamFloats + threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int HP = next_pow2(L.fh);           // lanes per column (fh <= 64 checked on the host)
  const int hh = lane & (HP - 1), cc = lane / HP;
  const int wave = blockIdx.x * (kBlock / DHD_WAVE) + (threadIdx.x >> 6);
  const int col = wave * (DHD_WAVE / HP) + cc;  // column index inside the camera: d * fW + w
  const bool in_range = hh < L.fh && col < L.D * L.fw;
  const int w = in_range ? col % L.fw : 0, d = in_range ? col / L.fw : 0, h = in_range ? hh : 0;
  float e[3];
  frustum_to_ego(cam, cal.frustum_u[w], cal.frustum_v[h], cal.frustum_d[d], e);
  const int pid = pt_index(L, bn, d, h, w);   // where this point's key / rank go (column-major: the lanes of a column are adjacent)
  int k0 = -1, k1 = -1;
  if (in_range) {
    if (!BAND_ONLY) {
      const int v0 = voxel_of(L.grid[0], e, b);
      if (v0 >= 0) k0 = L.vox_base[0] + v0;
    }
    if (L.G > 1) {
      const int g = (int)band[bn * L.hw + h * L.fw + w] + 1;
      if (g < L.G) {
        const int v1 = voxel_of(L.grid[g], e, b);
        if (v1 >= 0) k1 = L.vox_base[g] + v1;
      }
    }
  }
  int r0 = 0, r1 = 0;
  if (!BAND_ONLY) r0 = count_runs(L.count, k0, hh, lane);
  if (L.G > 1) r1 = count_runs(L.count, k1, hh, lane);
  if (in_range) {
    if (!BAND_ONLY) { L.key[pid] = k0; L.rnk[pid] = r0; }
    L.key[L.P + pid] = k1; L.rnk[L.P + pid] = r1;
  }
}

// Introspection twin of the kernel above: one grid, band-independent, optional ego output.
__global__ __launch_bounds__(kBlock) void mghs_voxel_index_kernel(Layout L, dhd_calib cal, int g, int* __restrict__ rank_map,
                                                                 float* __restrict__ ego) {
  __shared__ CamMats cam;
  const int bn = blockIdx.y;
  const int b = bn / L.N;
  if (threadIdx.x == 0) load_camera(cal, bn, b, &cam);
  __syncthreads();
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= L.dhw) return;
  const int w = i % L.fw;
  const int h = (i / L.fw) % L.fh;
  const int d = i / L.hw;
  float e[3];
  frustum_to_ego(cam, cal.frustum_u[w], cal.frustum_v[h], cal.frustum_d[d], e);
  const size_t pid = (size_t)bn * L.dhw + i;
  rank_map[pid] = voxel_of(L.grid[g], e, b);
  if (ego) { ego[pid * 3 + 0] = e[0]; ego[pid * 3 + 1] = e[1]; ego[pid * 3 + 2] = e[2]; }
}

// ---------------------------------------------------------------------------------------
// Exclusive scans over the per-voxel counters in ONE pass: `offset` = prefix of count (entry index), `nzoff` = prefix of
// (count > 0) (slot index).  Block i takes chunk i of kChunk counters, publishes its chunk aggregate as one 64-bit word
// [valid:1 | entries:31 | slots:32], and then adds up the published words of ALL its predecessors, its 256 threads polling
// 256 words at a time.  No block waits for another block's PREFIX, only for aggregates, which every block publishes before
// it waits for anything: the chain of a decoupled look-back (a block adopts the inclusive prefix of a predecessor, which had
// to wait for its own predecessors ...) does not exist, and a block only ever waits for lower-numbered blocks, which the
// dispatcher starts first.  Measured at B = 4 (1 328 chunks): 6 + 11 us as a chunk-sum pass and a scan pass (round 2);
// 30 us with chunks handed out by an atomic ticket (1 328 returning atomics on one address), with or without a look-back
// chain; 14 us as below.  Value and flag share one word, written and read with relaxed device-scope atomics: no fences.
// ---------------------------------------------------------------------------------------
constexpr int kScanSpinLimit = 1 << 16;   // polls of one predecessor word (~1 us each under load) before self-service

__device__ __forceinline__ unsigned long long scan_word(int entries, int slots) {
  return (1ull << 63) | ((unsigned long long)(unsigned)entries << 32) | (unsigned long long)(unsigned)slots;
}

__global__ __launch_bounds__(kBlock) void mghs_scan(Layout L) {
  __shared__ int ws[2][kBlock / DHD_WAVE];
  __shared__ int ws2[2][kBlock / DHD_WAVE];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int V = L.V;
  // chunk = workgroup index: a block waits only for lower-numbered blocks, which the dispatcher starts first (1-D grid,
  // in-order dispatch) and which wait for nobody before publishing.  (A ticket counter here cost 20 us: 1 328 returning
  // atomics on one address.)
  const int chunk = blockIdx.x;
  const int first = chunk * kChunk + t * kScanItems;
  int v[kScanItems];
  int tsum = 0, tz = 0;
  // a thread owns kScanItems = 8 consecutive counters: two 16-byte loads when the chunk lies inside [0, V)
  // (the arrays of the workspace are 256-byte aligned), scalar loads at the ragged end
  const bool whole = first + kScanItems <= V;
  if (whole) {
    const int4 a = *reinterpret_cast<const int4*>(L.count + first), b = *reinterpret_cast<const int4*>(L.count + first + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) v[k] = (first + k < V) ? L.count[first + k] : 0;
  }
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    tsum += v[k];
    tz += v[k] > 0;
  }
  int incl = tsum, inclz = tz;
  for (int d = 1; d < 64; d <<= 1) {
    int o = __shfl_up(incl, d, DHD_WAVE), oz = __shfl_up(inclz, d, DHD_WAVE);
    if (lane >= d) { incl += o; inclz += oz; }
  }
  if (lane == 63) { ws2[0][wv] = incl; ws2[1][wv] = inclz; }
  __syncthreads();
  if (t == 0)
    __hip_atomic_store(L.scan_state + chunk,
                       scan_word(ws2[0][0] + ws2[0][1] + ws2[0][2] + ws2[0][3], ws2[1][0] + ws2[1][1] + ws2[1][2] + ws2[1][3]),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // the aggregates of all predecessor chunks
  int part = 0, partz = 0;
  // Waiting only for lower-numbered workgroups is safe while a 1-D grid is dispatched in index order -- what the hardware does,
  // not something HIP promises.  So the wait is BOUNDED: a predecessor that has not published after kScanSpinLimit polls (it may
  // not have been started yet while this workgroup occupies its slot) is not waited for any longer -- its counters are final
  // (the counting kernel finished before this launch), so the thread sums that chunk itself.  Slow, never taken in practice,
  // and exact; DHD_MGHS_DEBUG_SCAN_SELF_SERVE (tests) sets the limit to zero so that every aggregate goes this way.
  const int spin_limit = (L.flags & DHD_MGHS_DEBUG_SCAN_SELF_SERVE) ? 0 : kScanSpinLimit;
  for (int j = t; j < chunk; j += kBlock) {
    unsigned long long w = 0;
    for (int spins = 0; spins < spin_limit; ++spins) {
      w = __hip_atomic_load(L.scan_state + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (w >> 63) break;
    }
    if (w >> 63) {
      part += (int)((w >> 32) & 0x7fffffffu);
      partz += (int)(w & 0xffffffffu);
    } else {
      const int lo = j * kChunk, hi = min(V, lo + kChunk);
      for (int i = lo; i < hi; ++i) {
        const int cnt = L.count[i];
        part += cnt;
        partz += cnt > 0;
      }
    }
  }
  part = wave_sum_i(part);
  partz = wave_sum_i(partz);
  if (lane == 0) { ws[0][wv] = part; ws[1][wv] = partz; }
  __syncthreads();
  int run = ws[0][0] + ws[0][1] + ws[0][2] + ws[0][3];
  int runz = ws[1][0] + ws[1][1] + ws[1][2] + ws[1][3];
  for (int k = 0; k < wv; ++k) { run += ws2[0][k]; runz += ws2[1][k]; }
  run += incl - tsum;
  runz += inclz - tz;
  int off[kScanItems], nzo[kScanItems];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    off[k] = run;
    nzo[k] = runz;
    if (first + k < V && v[k] > 0) L.nzvox[runz] = first + k;
    run += v[k];
    runz += v[k] > 0;
    if (first + k == V - 1) { L.offset[V] = run; L.nzoff[V] = runz; }
  }
  if (whole) {
    *reinterpret_cast<int4*>(L.offset + first) = make_int4(off[0], off[1], off[2], off[3]);
    *reinterpret_cast<int4*>(L.offset + first + 4) = make_int4(off[4], off[5], off[6], off[7]);
    *reinterpret_cast<int4*>(L.nzoff + first) = make_int4(nzo[0], nzo[1], nzo[2], nzo[3]);
    *reinterpret_cast<int4*>(L.nzoff + first + 4) = make_int4(nzo[4], nzo[5], nzo[6], nzo[7]);
  } else {
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
      if (first + k < V) { L.offset[first + k] = off[k]; L.nzoff[first + k] = nzo[k]; }
  }
}

// J0 = 1: only the band grid's entries (static rig: grid 0's part of the sorted lists is already in place)
// Element-wise form: one thread per point in the natural order (key / rnk in the natural order too: -DDHD_KEYS_ROWMAJOR, and depth
// planes too large for the LDS turn of mghs_scatter_planes).
template <int J0>
__global__ __launch_bounds__(kBlock) void mghs_scatter(Layout L, int blocks_per_bn) {
  // workgroups go round-robin over the XCDs: XCD x takes the x-th eighth of the points, so that the 4-byte
  // scatters into one cache line of the sorted arrays mostly come from one L2
  const int per_xcd = gridDim.x >> 3;
  const int wg = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int bn = wg / blocks_per_bn;
  if (bn >= L.B * L.N) return;
  const int i = (wg % blocks_per_bn) * kBlock + threadIdx.x;
  if (i >= L.dhw) return;
  const int pid = bn * L.dhw + i;
  const int pix = bn * L.hw + (i % L.hw);
  const int d = i / L.hw, r = i - d * L.hw, h = r / L.fw, w = r - h * L.fw;
  const int cix = pt_index(L, bn, d, h, w);
#pragma unroll
  for (int j = J0; j < 2; ++j) {
    int k = L.key[j * L.P + cix];
    int slot = -1;
    if (k >= 0) {
      slot = L.nzoff[k];
      if (j == 1 || !L.columns)   // column form: nobody reads grid 0's sorted entries (forward by column, backward by p_slot)
        L.s_ent[L.offset[k] + L.rnk[j * L.P + cix]] = make_int4(pid, pix, slot, 0);
    }
    L.p_slot[j * L.P + pid] = slot;
  }
}

// The same work per (camera, depth bin) PLANE: the plane's keys / ranks are read in their column-major order (coalesced), the
// entries scattered, the slots turned through LDS and written to p_slot in the natural order (coalesced).  Dynamic LDS: 2 hw ints.
template <int J0>
__global__ __launch_bounds__(kBlock) void mghs_scatter_planes(Layout L) {
  extern __shared__ int sl[];   // [class 2][h * fw + w]
  const int per_xcd = gridDim.x >> 3;
  const int plane = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);   // XCD x takes the x-th eighth of the planes (see above)
  if (plane >= L.B * L.N * L.D) return;
  const int bn = plane / L.D, d = plane - bn * L.D;
  const int base = bn * L.dhw + d * L.hw;          // the plane's first point in either order
  for (int i = threadIdx.x; i < L.hw; i += kBlock) {
    const int w = i / L.fh, h = i - w * L.fh;       // i-th word of the plane in key / rnk: (d * fw + w) * fh + h
    const int nat = h * L.fw + w;
    const int pid = base + nat, pix = bn * L.hw + nat;
#pragma unroll
    for (int j = J0; j < 2; ++j) {
      const int k = L.key[j * L.P + base + i];
      int slot = -1;
      if (k >= 0) {
        slot = L.nzoff[k];
        if (j == 1 || !L.columns) L.s_ent[L.offset[k] + L.rnk[j * L.P + base + i]] = make_int4(pid, pix, slot, 0);
      }
      sl[j * L.hw + nat] = slot;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < L.hw; i += kBlock) {
#pragma unroll
    for (int j = J0; j < 2; ++j) L.p_slot[j * L.P + base + i] = sl[j * L.hw + i];
  }
}

// Deterministic mode: the position of an entry inside its voxel is the number of the voxel's entries with a smaller
// point id, instead of the arrival order of the counting atomics.  One thread per point reads the (atomic-order)
// entry list of its voxel -- 6 to 17 entries on average, at most a few hundred -- and rewrites its rank; the scatter
// then runs a second time.  The per-voxel sums of the forward are then accumulated in ascending point order and two
// runs are bit-identical.
__global__ __launch_bounds__(kBlock) void mghs_rank_by_pid(Layout L, int t0) {
  const int t = t0 + blockIdx.x * kBlock + threadIdx.x;
  if (t >= 2 * L.P) return;
  const int k = L.key[t];
  if (k < 0) return;
  const int cix = t < L.P ? t : t - L.P;
#ifdef DHD_KEYS_ROWMAJOR
  const int pid = cix;
#else
  const int bn = cix / L.dhw, i = cix - bn * L.dhw;
  const int colm = i / L.fh, h = i - colm * L.fh, d = colm / L.fw, w = colm - d * L.fw;
  const int pid = pt_natural(L, bn, d, h, w);
#endif
  const int lo = L.offset[k], hi = L.offset[k + 1];
  int r = 0;
  for (int e = lo; e < hi; ++e) r += L.s_ent[e].x < pid;
  L.rnk[t] = r;
}

// parity hook (dhd_mghs_debug_keys): the two key rows in the natural point order
__global__ __launch_bounds__(kBlock) void mghs_keys_natural(Layout L, int* __restrict__ out) {
  const int pid = blockIdx.x * kBlock + threadIdx.x;
  if (pid >= L.P) return;
  const int bn = pid / L.dhw, i = pid - bn * L.dhw;
  const int d = i / L.hw, r = i - d * L.hw, h = r / L.fw, w = r - h * L.fw;
  const int cix = pt_index(L, bn, d, h, w);
  out[pid] = L.key[cix];
  out[L.P + pid] = L.G > 1 ? L.key[L.P + cix] : -1;
}

}  // namespace

int launch_scan(const Layout& L, hipStream_t st) {
  hipLaunchKernelGGL(mghs_scan, dim3(L.n_chunks), dim3(kBlock), 0, st, L);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

}  // namespace dhd

using namespace dhd;

namespace {

int check_calib(const dhd_calib* c) {
  if (!c || !c->sensor2ego || !c->post_tran || !c->bda || !c->frustum_u || !c->frustum_v || !c->frustum_d)
    return DHD_EINVAL;
  if (!c->inv_post_rot && !c->post_rot) return DHD_EINVAL;
  if (!c->combine && !c->intrin) return DHD_EINVAL;
  return DHD_OK;
}

// prologue (zero-fill [+ cameras] [+ band + transposition]) -> geometry + counting -> scan -> scatter [-> ranking -> scatter]
int lift_impl(const dhd_mghs_desc* desc, const dhd_calib* calib, const float* height, int n_height, const float* height_range,
              const float* mask_range, const float* feat_nchw, uint8_t* band, float* feat_nhwc, const dhd_mghs_workspace* ws,
              bool fused_lift, bool static_rig, void* stream) {
  Layout L;
  if (!ws) return DHD_EINVAL;
  int rc = make_layout(desc, ws, &L);
  if (rc) return rc;
  if ((rc = check_calib(calib))) return rc;
  if (L.G > 1 && !band) return DHD_EINVAL;
  int hp = 1;
  while (hp < L.fh) hp <<= 1;
  if (hp > DHD_WAVE) return DHD_EUNSUPPORTED;  // feature maps taller than 64 rows
  PrologueArgs a = {};
  if (fused_lift) {
    if (!feat_nchw || !feat_nhwc) return DHD_EINVAL;
    if (L.G > 1) {
      if (!height || !height_range || !mask_range) return DHD_EINVAL;
      if (n_height <= 0 || n_height > kMaxHeightBins) return DHD_EUNSUPPORTED;
      make_band_lut(height_range, n_height, mask_range, &a.lut);
      a.height = height; a.n_height = n_height; a.band = band;
      a.n_band = dhd_cdiv((long)L.B * L.N * L.hw * kBandLanes, kLiftBlock);
    }
    a.feat_nchw = feat_nchw; a.feat_nhwc = feat_nhwc;
    a.tr_cols_tiles = dhd_cdiv(L.hw, 64); a.tr_rows_tiles = dhd_cdiv(L.C, 64);
    a.tr_vec = transpose_vectorisable(feat_nchw, feat_nhwc, L.C, L.hw) ? 1 : 0;
    a.n_tr = a.tr_cols_tiles * a.tr_rows_tiles * L.B * L.N;
  }
  // static rig: grid 0's counters stay, the band grids' counters and the scan's chunk aggregates are cleared.  The zero-fill
  // works in 256-byte blocks, so the B * nz0 * ny0 * nx0 counters of grid 0 must end on a 256-byte boundary; where they do
  // not (e.g. a 100 x 100 x 1 grid at B = 1) the call runs as a full lift, whose results are identical by contract.
  size_t keep = 0;
  if (static_rig && (((size_t)L.vox_base[1] * 4) & 255)) static_rig = false;
  if (static_rig) keep = (size_t)L.vox_base[1] * 4;
  a.zero_ptr = reinterpret_cast<char*>(L.count) + keep;
  a.zero_bytes = L.zero_bytes - keep;
  a.n_zero = dhd_cdiv((long)a.zero_bytes, kZeroBytesPerBlock);
  a.n_cam = static_rig ? 0 : dhd_cdiv(L.B * L.N, kLiftBlock);
  hipStream_t st = dhd_stream(stream);
  hipLaunchKernelGGL(mghs_prologue, dim3(a.n_zero + a.n_cam + a.n_band + a.n_tr), dim3(kLiftBlock), 0, st, L, *calib, a);
  DHD_LAUNCH_CHECK();
  const dim3 gp(dhd_cdiv(L.dhw, kBlock), L.B * L.N);
  const long cols_per_block = (long)(kBlock / DHD_WAVE) * (DHD_WAVE / hp);
  const dim3 gc(dhd_cdiv((long)L.D * L.fw, cols_per_block), L.B * L.N);
  if (static_rig) hipLaunchKernelGGL(mghs_geom_count<true>, gc, dim3(kBlock), 0, st, L, *calib, band);
  else hipLaunchKernelGGL(mghs_geom_count<false>, gc, dim3(kBlock), 0, st, L, *calib, band);
  DHD_LAUNCH_CHECK();
  if ((rc = launch_scan(L, st))) return rc;
  const dim3 gs(dhd_cdiv((long)gp.x * gp.y, 8) * 8);
  const dim3 gpl(dhd_cdiv((long)L.B * L.N * L.D, 8) * 8);
  const size_t plane_lds = (size_t)2 * L.hw * sizeof(int);
#ifdef DHD_KEYS_ROWMAJOR
  const bool by_plane = false;
#else
  const bool by_plane = plane_lds <= 48 * 1024;
#endif
  auto scatter = [&]() {
    if (by_plane) {
      if (static_rig) hipLaunchKernelGGL(mghs_scatter_planes<1>, gpl, dim3(kBlock), plane_lds, st, L);
      else hipLaunchKernelGGL(mghs_scatter_planes<0>, gpl, dim3(kBlock), plane_lds, st, L);
    } else {
      if (static_rig) hipLaunchKernelGGL(mghs_scatter<1>, gs, dim3(kBlock), 0, st, L, (int)gp.x);
      else hipLaunchKernelGGL(mghs_scatter<0>, gs, dim3(kBlock), 0, st, L, (int)gp.x);
    }
  };
  scatter();
  DHD_LAUNCH_CHECK();
  if (L.flags & DHD_MGHS_DETERMINISTIC) {
    const int t0 = static_rig ? L.P : 0;
    hipLaunchKernelGGL(mghs_rank_by_pid, dim3(dhd_cdiv(2L * L.P - t0, kBlock)), dim3(kBlock), 0, st, L, t0);
    scatter();
    DHD_LAUNCH_CHECK();
  }
  return DHD_OK;
}

}  // namespace

extern "C" {

int dhd_abi_version(void) { return DHD_ABI_VERSION; }

int dhd_mghs_workspace_bytes(const dhd_mghs_desc* desc, size_t* state_bytes, size_t* scratch_bytes) {
  if (!state_bytes || !scratch_bytes) return DHD_EINVAL;
  Layout L;
  return make_layout(desc, nullptr, &L, state_bytes, scratch_bytes);
}

int dhd_mghs_prepare(const dhd_mghs_desc* desc, const dhd_calib* calib, const uint8_t* band, const dhd_mghs_workspace* ws,
                     void* stream) {
  return lift_impl(desc, calib, nullptr, 0, nullptr, nullptr, nullptr, const_cast<uint8_t*>(band), nullptr, ws, false, false, stream);
}

int dhd_mghs_lift(const dhd_mghs_desc* desc, const dhd_calib* calib, const float* height, int n_height, const float* height_range,
                  const float* mask_range, const float* feat_nchw, uint8_t* band, float* feat_nhwc, const dhd_mghs_workspace* ws,
                  void* stream) {
  return lift_impl(desc, calib, height, n_height, height_range, mask_range, feat_nchw, band, feat_nhwc, ws, true, false, stream);
}

int dhd_mghs_lift_static(const dhd_mghs_desc* desc, const dhd_calib* calib, const float* height, int n_height,
                         const float* height_range, const float* mask_range, const float* feat_nchw, uint8_t* band,
                         float* feat_nhwc, const dhd_mghs_workspace* ws, void* stream) {
  if (!desc || desc->n_grids < 2) return DHD_EINVAL;   // with one grid nothing changes from frame to frame: reuse the workspace
  return lift_impl(desc, calib, height, n_height, height_range, mask_range, feat_nchw, band, feat_nhwc, ws, true, true, stream);
}

int dhd_mghs_voxel_index(const dhd_mghs_desc* desc, const dhd_calib* calib, int grid_index, int32_t* rank_map,
                         float* ego, void* stream) {
  Layout L;
  int rc = make_layout(desc, nullptr, &L);
  if (rc) return rc;
  if ((rc = check_calib(calib))) return rc;
  if (!rank_map || grid_index < 0 || grid_index >= L.G) return DHD_EINVAL;
  dim3 gp(dhd_cdiv(L.dhw, kBlock), L.B * L.N);
  hipLaunchKernelGGL(mghs_voxel_index_kernel, gp, dim3(kBlock), 0, dhd_stream(stream), L, *calib, grid_index,
                     rank_map, ego);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

int dhd_mghs_debug_keys(const dhd_mghs_desc* desc, const dhd_mghs_workspace* ws, int32_t* keys, void* stream) {
  Layout L;
  if (!ws || !keys) return DHD_EINVAL;
  int rc = make_layout(desc, ws, &L);
  if (rc) return rc;
  // keys[] is in the natural point order (the reference's ranks_depth); the product keeps its per-point arrays in pt_index order
  // a single-grid plan has no band grids: the counting kernel never writes row 1, so the hook reports "no key" there instead of
  // whatever the scratch held
  hipLaunchKernelGGL(mghs_keys_natural, dim3(dhd_cdiv(L.P, kBlock)), dim3(kBlock), 0, dhd_stream(stream), L, keys);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

int dhd_mghs_stats(const dhd_mghs_desc* desc, const dhd_mghs_workspace* ws, int32_t n_kept[DHD_MAX_GRIDS],
                   int32_t n_intervals[DHD_MAX_GRIDS], void* stream) {
  Layout L;
  if (!ws || !n_kept || !n_intervals) return DHD_EINVAL;
  int rc = make_layout(desc, ws, &L);
  if (rc) return rc;
  hipStream_t st = dhd_stream(stream);
  DHD_HIP(hipStreamSynchronize(st));
  // not a hot path: read the two prefix arrays at the grid boundaries
  for (int g = 0; g < DHD_MAX_GRIDS; ++g) {
    int lo[2], hi[2];
    DHD_HIP(hipMemcpy(&lo[0], L.offset + L.vox_base[g], 4, hipMemcpyDeviceToHost));
    DHD_HIP(hipMemcpy(&hi[0], L.offset + L.vox_base[g + 1], 4, hipMemcpyDeviceToHost));
    DHD_HIP(hipMemcpy(&lo[1], L.nzoff + L.vox_base[g], 4, hipMemcpyDeviceToHost));
    DHD_HIP(hipMemcpy(&hi[1], L.nzoff + L.vox_base[g + 1], 4, hipMemcpyDeviceToHost));
    n_kept[g] = hi[0] - lo[0];
    n_intervals[g] = hi[1] - lo[1];
  }
  return DHD_OK;
}

}  // extern "C"
