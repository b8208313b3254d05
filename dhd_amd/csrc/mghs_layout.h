// Workspace layout and shared device helpers of the fused MGHS kernels (mghs_prepare.hip,
// mghs_pool.hip).
#pragma once
#include "common.h"

namespace dhd {

constexpr int kBlock = 256;                  // geometry / scan / scatter / per-entry kernels
constexpr int kScanItems = 8;
constexpr int kChunk = kBlock * kScanItems;  // counters per scan block
constexpr int kTileC = 64;                   // channels handled by one wave (lane == channel)
constexpr int kCamFloats = 36;               // sizeof(CamMats)/4 = 33, padded
#ifndef DHD_SEG_ROWS
#define DHD_SEG_ROWS 4
#endif
constexpr int kSegRows = DHD_SEG_ROWS;       // output rows per streaming segment (4 rows = 3200 contiguous bytes per channel at nx=200)
constexpr int kSegMaxVox = 256 * kSegRows;   // voxels per segment (kSegRows * nx, nx <= 256)
constexpr int kMaxTileX = 256;               // generic dense-row path: voxels along x per tile
constexpr int kRowGroup = 4;                 // generic dense-row path: rows handed to one XCD at a time

// Host-derived description, passed to kernels by value.
struct Layout {
  int B, N, D, fh, fw, C, G;
  int dhw;       // D*fh*fw points per camera
  int hw;        // fh*fw pixels per camera
  int P;         // B*N*dhw points
  int V;         // total voxels over all grids
  int R;         // total output rows (b, z, y) over all grids
  int n_chunks;  // scan blocks
  int vox_base[DHD_MAX_GRIDS + 1];
  int row_base[DHD_MAX_GRIDS + 1];
  dhd_grid grid[DHD_MAX_GRIDS];
  // compact path (C == 64, every ny a multiple of kSegRows, nx a multiple of 4): the output is cut
  // into `n_segs` segments of kSegRows rows; seg_base[g] = first segment of grid g
  int compact;
  int n_slots_max;  // capacity of vsum in rows; row n_slots_max is a scratch row
  int n_segs;
  int seg_base[DHD_MAX_GRIDS + 1];
  // generic dense-row path
  int nxc;                        // x chunks per row
  int sched_heavy, sched_ratio;   // grid-0 row groups front-loaded 1:ratio among the others
  int flags;       // dhd_mghs_desc.flags
  int columns;     // the full-height grid's forward sums by pixel column (mghs_col_sums): its sorted entry list is not built
  // scratch carve (device pointers): valid from prepare to the forward after it (offset / s_ent: state carve when
  // !compact, see make_layout)
  int* count;      // [V]     entries per voxel                      } one contiguous, zero-filled range per prepare:
  unsigned long long* scan_state;  // [n_chunks] chunk aggregates      } count | scan_state
  int* offset;     // [V+1]   exclusive prefix of count            (entry index space)
  int* key;        // [2P]    voxel id of point p in grid 0 ([p]) and in its band grid ([P+p]); -1 = dropped
  int* rnk;        // [2P]    arrival rank of the point inside its voxel
  // [2P] entries grouped by voxel, ONE 16-byte record each: x = point id (index into depth), y = pixel id (row of
  // feat_nhwc), z = slot (non-empty voxel ordinal, non-decreasing), w unused.  (Three int arrays before: the scatter wrote
  // three 4-byte words per entry into three different cache lines.)
  int4* s_ent;
  float* cam;      // [B*N*kCamFloats] per-camera matrices
  float* dg_part;  // [2P]    backward scratch of the generic path: depth-gradient parts of grid 0 / band grid
  float* fg_stage; // [B*N*hw*C] generic path with DHD_MGHS_FEAT_GRAD_NCHW: the (B*N,fH,fW,C) gradient before its transposition
  float* vsum;     // [(min(2P, V) + 1) * kTileC] compact per-voxel rows: forward sums / backward extracted gradients
                   //            (only when `compact`)
  // state carve: what the backward pass needs from prepare
  int* nzoff;      // [V+1]   exclusive prefix of (count > 0)      (non-empty voxel ordinal, "slot")
  int* nzvox;      // [min(2P, V)] voxel id of every slot
  int* p_slot;     // [2P]    slot of point p in grid 0 ([p]) and in its band grid ([P+p]); -1 = dropped
  size_t zero_bytes;  // bytes of the count | scan_state range
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Index of frustum point (camera bn, depth bin d, row h, column w) in the scratch arrays `key` and `rnk`.  Since round 5
// COLUMN-major inside a depth plane (h fastest): the counting kernel's lanes run along h (a wave holds whole pixel columns so that
// a run of equal keys shares one atomic), and with the natural order (w fastest, the reference's `ranks_depth`) each of its four
// stores per point touched 64 lines per wave -- 40 of mghs_geom_count's 70 us at the DHD-L geometry together with the returning
// atomics (round 4's ablation).  mghs_scatter_planes and mghs_col_sums read the arrays in the same order (coalesced too).  `p_slot`
// (state, read by mghs_pixel_bwd next to depth / depth_grad) stays in the natural order pt_natural: the scatter turns a depth plane
// through LDS.  (Measured with p_slot column-major as well: the pixel kernel loses 9 of the 14 us the lift gains -- its waves walk
// the pixels w-fastest and then no longer share p_slot lines.)  -DDHD_KEYS_ROWMAJOR: the round-4 order, for A/B builds.
__host__ __device__ inline int pt_natural(const Layout& L, int bn, int d, int h, int w) { return bn * L.dhw + (d * L.fh + h) * L.fw + w; }
__host__ __device__ inline int pt_index(const Layout& L, int bn, int d, int h, int w) {
#ifdef DHD_KEYS_ROWMAJOR
  return pt_natural(L, bn, d, h, w);
#else
  return bn * L.dhw + (d * L.fw + w) * L.fh + h;
#endif
}

// Fills *L from the description; with `ws` also carves the device pointers and checks the sizes.  state_bytes /
// scratch_bytes (optional) receive the sizes needed.
inline int make_layout(const dhd_mghs_desc* d, const dhd_mghs_workspace* ws, Layout* L, size_t* state_bytes = nullptr,
                       size_t* scratch_bytes = nullptr) {
  if (!d) return DHD_EINVAL;
  if (d->batch <= 0 || d->n_cams <= 0 || d->n_depth <= 0 || d->fh <= 0 || d->fw <= 0 || d->channels <= 0)
    return DHD_EINVAL;
  if (d->n_grids < 1 || d->n_grids > DHD_MAX_GRIDS) return DHD_EINVAL;
  if (d->flags & ~(DHD_MGHS_DETERMINISTIC | DHD_MGHS_FEAT_GRAD_NCHW | DHD_MGHS_DEBUG_SCAN_SELF_SERVE)) return DHD_EINVAL;
  L->flags = d->flags;
  L->B = d->batch; L->N = d->n_cams; L->D = d->n_depth; L->fh = d->fh; L->fw = d->fw;
  L->C = d->channels; L->G = d->n_grids;
  L->hw = d->fh * d->fw;
  long dhw = (long)d->n_depth * L->hw;
  long P = (long)d->batch * d->n_cams * dhw;
  if (P > (1L << 29)) return DHD_EUNSUPPORTED;
  L->dhw = (int)dhw; L->P = (int)P;
  long v = 0, r = 0;
  bool compact = d->channels == kTileC;
  int nx_max = 0;
  for (int g = 0; g < DHD_MAX_GRIDS; ++g) {
    L->vox_base[g] = (int)v; L->row_base[g] = (int)r; L->seg_base[g] = (int)(r / kSegRows);
    if (g < d->n_grids) {
      const dhd_grid& gr = d->grid[g];
      if (gr.n[0] <= 0 || gr.n[1] <= 0 || gr.n[2] <= 0) return DHD_EINVAL;
      L->grid[g] = gr;
      v += (long)d->batch * gr.n[2] * gr.n[1] * gr.n[0];
      r += (long)d->batch * gr.n[2] * gr.n[1];
      if (v > (1L << 30)) return DHD_EUNSUPPORTED;
      compact = compact && gr.n[1] % kSegRows == 0 && gr.n[0] % 4 == 0 && gr.n[0] * kSegRows <= kSegMaxVox;
      nx_max = nx_max > gr.n[0] ? nx_max : gr.n[0];
    } else {
      L->grid[g] = d->grid[0];
    }
  }
  L->vox_base[DHD_MAX_GRIDS] = (int)v; L->row_base[DHD_MAX_GRIDS] = (int)r;
  for (int g = d->n_grids; g < DHD_MAX_GRIDS; ++g) { L->vox_base[g] = (int)v; L->row_base[g] = (int)r; }
  L->V = (int)v; L->R = (int)r;
  L->compact = compact ? 1 : 0;
  // Column form (mghs_pool.hip, mghs_col_sums): pays where the runs of equal keys along a pixel column are long and every context
  // row is re-read many times -- the 32-row feature maps of the DHD-L geometry (measured: gather 134 -> 99 us at B = 2; at the
  // 16-row maps of DHD-S it loses, 39 -> 47 us); needs the compact path and the default (unordered) summation
  L->columns = (compact && d->n_grids > 1 && !(d->flags & DHD_MGHS_DETERMINISTIC) && d->fh == 32) ? 1 : 0;
  L->n_segs = compact ? (int)(r / kSegRows) : 0;
  for (int g = d->n_grids; g <= DHD_MAX_GRIDS; ++g) L->seg_base[g] = (int)(r / kSegRows);
  L->n_chunks = dhd_cdiv(v, kChunk);
  L->nxc = (nx_max + kMaxTileX - 1) / kMaxTileX;
  L->sched_heavy = 0; L->sched_ratio = 0;
  if (L->G > 1 && L->row_base[1] % kRowGroup == 0) {
    int heavy = L->row_base[1] / kRowGroup, light = (L->R - L->row_base[1]) / kRowGroup;
    int k = heavy > 0 ? light / heavy : 0;
    // the period k+1 must be odd: row groups go round-robin over the 8 XCDs, an even period would
    // put every heavy group on the same few XCDs (measured: XCDs {0,4} only)
    if (k > 4) k = 4;
    if (k == 3) k = 2;
    if (k == 1) k = 0;
    if (k >= 2) { L->sched_heavy = heavy; L->sched_ratio = k; }
  }
  if (ws) {
    if (!ws->state || !ws->scratch) return DHD_EINVAL;
    // 16-byte vector access to the carved arrays
    if ((reinterpret_cast<uintptr_t>(ws->state) | reinterpret_cast<uintptr_t>(ws->scratch)) & 255) return DHD_EINVAL;
  }
  const size_t P2 = 2 * (size_t)L->P;
  // non-empty voxels ("slots"): grid 0 receives at most one entry per point, the band grids TOGETHER at most one per point
  // (a pixel belongs to one band), and no grid more slots than it has voxels
  const size_t v0 = (size_t)L->vox_base[1], vb = (size_t)L->V - v0, pp = (size_t)L->P;
  const size_t max_slots = (v0 < pp ? v0 : pp) + (vb < pp ? vb : pp);
  L->n_slots_max = (int)max_slots;
  size_t off = 0;
  char* base = ws ? static_cast<char*>(ws->scratch) : nullptr;
  auto carve = [&](size_t n_words) { int* p = reinterpret_cast<int*>(base + off); off = align_up(off + n_words * 4, 256); return p; };
  L->count = carve((size_t)L->V);
  L->scan_state = reinterpret_cast<unsigned long long*>(carve(2 * (size_t)L->n_chunks));
  L->zero_bytes = off;
  // generic (non-compact) path: its backward walks the grouped entry lists again (mghs_rows_bwd), so `offset` and
  // `s_ent` belong to the STATE there (carved below); the compact backward needs neither
  if (compact) L->offset = carve((size_t)L->V + 1);
  L->key = carve(P2);
  L->rnk = carve(P2);
  if (compact) L->s_ent = reinterpret_cast<int4*>(carve(4 * P2));
  L->cam = reinterpret_cast<float*>(carve((size_t)L->B * L->N * kCamFloats));
  L->dg_part = reinterpret_cast<float*>(carve(compact ? 0 : P2));
  L->fg_stage = reinterpret_cast<float*>(carve(compact ? 0 : (size_t)L->B * L->N * L->hw * L->C));
  L->vsum = reinterpret_cast<float*>(carve(compact ? (max_slots + 1) * kTileC : 0));  // +1: scratch row for discarded stores
  const size_t scratch_need = off;
  off = 0;
  base = ws ? static_cast<char*>(ws->state) : nullptr;
  L->nzoff = carve((size_t)L->V + 1);
  L->nzvox = carve(max_slots);
  L->p_slot = carve(P2);
  if (!compact) {
    L->offset = carve((size_t)L->V + 1);
    L->s_ent = reinterpret_cast<int4*>(carve(4 * P2));
  }
  const size_t state_need = off;
  if (state_bytes) *state_bytes = state_need;
  if (scratch_bytes) *scratch_bytes = scratch_need;
  if (ws && (ws->state_bytes < state_need || ws->scratch_bytes < scratch_need)) return DHD_ENOSPACE;
  return DHD_OK;
}

// ---- device helpers ------------------------------------------------------------------------
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int lane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float lane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ int wave_sum_i(int v) {
  for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, DHD_WAVE);
  return v;
}

typedef float vfloat4 __attribute__((ext_vector_type(4)));  // native vector: accepted by the nontemporal builtins

// Where grid g's dense tensor lives: element (b, z, c, y, x) is at
//   p[g] + b*sb[g] + z*sz[g] + c*sc[g] + y*nx + x   (floats).
// Default (reference layout after permute + collapse_z, (B, nz*C, ny, nx)): sb = nz*C*plane,
// sz = C*plane, sc = plane.  The un-collapsed (B, C, nz_total, ny, nx) tensor of MGHS_Depth is the
// same data with sb = C*nz_total*plane, sc = nz_total*plane, sz = plane and p offset by the grid's
// first z slice.
// (p is a float* for DHD_F32 and points to 2-byte elements for DHD_F16 / DHD_BF16: the half kernels cast it; strides in elements)
struct OutPtrs { float* p[DHD_MAX_GRIDS]; long sb[DHD_MAX_GRIDS], sz[DHD_MAX_GRIDS], sc[DHD_MAX_GRIDS]; int dtype; };
struct InPtrs { const float* p[DHD_MAX_GRIDS]; long sb[DHD_MAX_GRIDS], sz[DHD_MAX_GRIDS], sc[DHD_MAX_GRIDS]; int dtype; };

template <class Ptrs, class T>
inline int make_views(const Layout& L, T* const bases[DHD_MAX_GRIDS], const dhd_tensor_view* views, Ptrs* o) {
  o->dtype = DHD_F32;
  if (views) {
    o->dtype = views[0].dtype;
    if (o->dtype != DHD_F32 && o->dtype != DHD_F16 && o->dtype != DHD_BF16) return DHD_EINVAL;
    if (o->dtype != DHD_F32 && !L.compact) return DHD_EUNSUPPORTED;   // half tensors: the segment writer / reader only
  }
  const long vec = o->dtype == DHD_F32 ? 4 : 8;                       // elements per 16-byte access
  for (int g = 0; g < DHD_MAX_GRIDS; ++g) {
    o->p[g] = nullptr; o->sb[g] = o->sz[g] = o->sc[g] = 0;
    if (g >= L.G) continue;
    const long plane = (long)L.grid[g].n[1] * L.grid[g].n[0];
    if (views) {
      if (!views[g].ptr) return DHD_EINVAL;
      o->p[g] = static_cast<T*>(const_cast<void*>(static_cast<const void*>(views[g].ptr)));
      o->sb[g] = views[g].batch_stride; o->sz[g] = views[g].z_stride; o->sc[g] = views[g].channel_stride;
      // the 16-byte vector path needs aligned rows
      if (views[g].dtype != o->dtype) return DHD_EINVAL;
      if (((o->sb[g] | o->sz[g] | o->sc[g]) & (vec - 1)) || (reinterpret_cast<uintptr_t>(views[g].ptr) & 15)) return DHD_EINVAL;
      if (o->dtype != DHD_F32 && (L.grid[g].n[0] & 1)) return DHD_EUNSUPPORTED;   // 4 rows of nx voxels in 8-voxel vectors
    } else {
      if (!bases || !bases[g]) return DHD_EINVAL;
      o->p[g] = bases[g];
      o->sb[g] = (long)L.grid[g].n[2] * L.C * plane; o->sz[g] = (long)L.C * plane; o->sc[g] = plane;
    }
  }
  return DHD_OK;
}

// ---- seams between the translation units of the library (host functions, not part of the C ABI) ----
// mghs_prepare.hip: count[V] (+ zeroed scan_state) -> offset[V+1], nzoff[V+1], nzvox[slots]
int launch_scan(const Layout& L, hipStream_t st);
// mghs_pool.hip: the segment writer / reader of the compact path (vsum[slot][64] <-> dense tensors), `split` channel parts per segment
int launch_stream_fwd(const Layout& L, const OutPtrs& o, int split, hipStream_t st, bool op = false);
int launch_stream_bwd(const Layout& L, const InPtrs& in, int split, hipStream_t st, bool op = false);

}  // namespace dhd
