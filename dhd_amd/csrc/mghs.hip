// Fused MGHS view transform for gfx950 (MI355X): geometry -> voxel index -> counting sort ->
// 4-grid pooling forward / backward.  Replaces the chain
//   4x MGHS.get_ego_coor            (models/necks/lss_heightmap.py:179-231)
//   4x voxel_pooling_prepare_v2     (:303-371)
//   4x bev_pool_v2 + permute + cat  (ops/bev_pool_v2/bev_pool.py:86-106, lss_heightmap.py:298-299)
// of the reference with one geometry pass, one device counting sort shared by all grids, and one
// pooling launch that writes every output tensor once, in its final (B, nz*C, ny, nx) layout.
//
// All of this is HBM-bound byte shuffling (about 176 MB of dense output per sample, 97% of it
// zeros): the kernels are organised around coalesced 16-byte row stores/loads staged through
// LDS; no MFMA is involved.
//
// File:line citations are into /root/reference/projects/mmdet3d_plugin/.
#include "common.h"

namespace {

constexpr int kBlock = 256;      // geometry / scan / scatter
#ifndef DHD_POOL_BLOCK
#define DHD_POOL_BLOCK 512
#endif
#ifndef DHD_TILE_X
#define DHD_TILE_X 256
#endif
constexpr int kPoolBlock = DHD_POOL_BLOCK;  // pooling: waves per output tile
constexpr int kPoolWaves = kPoolBlock / DHD_WAVE;
constexpr int kScanItems = 8;
constexpr int kChunk = kBlock * kScanItems;  // counters per scan block
constexpr int kMaxTileX = DHD_TILE_X;        // voxels along x per pooling tile
constexpr int kTileC = 64;                   // channels per pooling tile (one wave lane each)
constexpr int kCamFloats = 36;               // sizeof(CamMats)/4 = 33, padded
constexpr int kRowGroup = 4;                 // consecutive dense rows handed to one XCD (25 whole cache lines at nx=200)
constexpr int kGroupRows = 4;                // rows per sparse group
constexpr int kGroupMaxVox = 1024;           // voxels per sparse group (kGroupRows * nx)
constexpr int kGroupSlots = 128;             // non-empty voxels a sparse group can hold before it falls back to dense rows
constexpr int kGroupMaxPts = 2048;           // points above which a group goes straight to the dense path

// Host-derived layout, passed to kernels by value.
struct Layout {
  int B, N, D, fh, fw, C, G;
  int dhw;       // D*fh*fw points per camera
  int hw;        // fh*fw pixels per camera
  int P;         // B*N*dhw points
  int V;         // total voxels over all grids
  int R;         // total output rows (b, z, y) over all grids
  int n_chunks;  // scan blocks
  int nxc;          // x chunks per dense row
  // pooling work list: `n_dense_rows` rows are processed one row per workgroup through a dense LDS
  // tile (grid 0, or every row when grouping is off); the remaining rows are processed as
  // `n_groups` groups of kGroupRows consecutive rows through the sparse path.
  int n_dense_rows, n_groups;
  int sched_cycles, sched_light;  // per XCD: `sched_cycles` x (kRowGroup dense rows + sched_light groups), then groups only
  int per_xcd;                    // work items per XCD
  int sched_heavy, sched_ratio;   // row-only schedule (backward): grid-0 row groups front-loaded 1:ratio
  int vox_base[DHD_MAX_GRIDS + 1];
  int row_base[DHD_MAX_GRIDS + 1];
  dhd_grid grid[DHD_MAX_GRIDS];
  // workspace carve (device pointers)
  int* count;      // [V]     points per voxel
  int* offset;     // [V+1]   exclusive prefix of count
  int* chunk_sum;  // [n_chunks]
  int* key;        // [2P]    voxel id of point p in grid 0 ([p]) and in its band grid ([P+p]); -1 = dropped
  int* rnk;        // [2P]    arrival rank of the point inside its voxel
  int* s_pid;      // [2P]    point ids grouped by voxel (index into depth)
  int* s_pix;      // [2P]    pixel ids grouped by voxel (row of feat_nhwc)
  int* s_vox;      // [2P]    voxel id of every grouped entry (non-decreasing)
  float* cam;      // [B*N*kCamFloats] per-camera matrices (CamMats), written by mghs_camera
  float* dg_part;  // [2P] backward scratch: depth-gradient parts from grid 0 ([p]) and the band grid ([P+p])
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int make_layout(const dhd_mghs_desc* d, void* ws, Layout* L, size_t* bytes) {
  if (!d) return DHD_EINVAL;
  if (d->batch <= 0 || d->n_cams <= 0 || d->n_depth <= 0 || d->fh <= 0 || d->fw <= 0 || d->channels <= 0)
    return DHD_EINVAL;
  if (d->n_grids < 1 || d->n_grids > DHD_MAX_GRIDS) return DHD_EINVAL;
  L->B = d->batch; L->N = d->n_cams; L->D = d->n_depth; L->fh = d->fh; L->fw = d->fw;
  L->C = d->channels; L->G = d->n_grids;
  L->hw = d->fh * d->fw;
  long dhw = (long)d->n_depth * L->hw;
  long P = (long)d->batch * d->n_cams * dhw;
  if (P > (1L << 30)) return DHD_EUNSUPPORTED;
  L->dhw = (int)dhw; L->P = (int)P;
  long v = 0, r = 0;
  for (int g = 0; g < DHD_MAX_GRIDS; ++g) {
    L->vox_base[g] = (int)v; L->row_base[g] = (int)r;
    if (g < d->n_grids) {
      const dhd_grid& gr = d->grid[g];
      if (gr.n[0] <= 0 || gr.n[1] <= 0 || gr.n[2] <= 0) return DHD_EINVAL;
      L->grid[g] = gr;
      v += (long)d->batch * gr.n[2] * gr.n[1] * gr.n[0];
      r += (long)d->batch * gr.n[2] * gr.n[1];
      if (v > (1L << 30)) return DHD_EUNSUPPORTED;
    } else {
      L->grid[g] = d->grid[0];
    }
  }
  L->vox_base[DHD_MAX_GRIDS] = (int)v; L->row_base[DHD_MAX_GRIDS] = (int)r;
  for (int g = d->n_grids; g < DHD_MAX_GRIDS; ++g) { L->vox_base[g] = (int)v; L->row_base[g] = (int)r; }
  L->V = (int)v; L->R = (int)r;
  L->n_chunks = dhd_cdiv(v, kChunk);
  size_t off = 0;
  char* base = static_cast<char*>(ws);
  auto carve = [&](size_t n_ints) { int* p = reinterpret_cast<int*>(base + off); off = align_up(off + n_ints * 4, 256); return p; };
  L->count = carve((size_t)L->V);
  L->offset = carve((size_t)L->V + 1);
  L->chunk_sum = carve((size_t)L->n_chunks);
  L->key = carve(2 * (size_t)L->P);
  L->rnk = carve(2 * (size_t)L->P);
  L->s_pid = carve(2 * (size_t)L->P);
  L->s_pix = carve(2 * (size_t)L->P);
  L->s_vox = carve(2 * (size_t)L->P);
  L->cam = reinterpret_cast<float*>(carve((size_t)L->B * L->N * kCamFloats));
  L->dg_part = reinterpret_cast<float*>(carve(2 * (size_t)L->P));
  {
    int nx_max = 0;
    for (int g = 0; g < L->G; ++g) nx_max = nx_max > L->grid[g].n[0] ? nx_max : L->grid[g].n[0];
    L->nxc = (nx_max + kMaxTileX - 1) / kMaxTileX;
  }
  // Work list and launch order.  Grid 0 pools every pixel over the whole height: its rows carry
  // ~10x the points of a band-grid row and go through the dense row tile.  Band-grid rows hold a
  // handful of points each: they are processed four rows at a time (3200 contiguous, line-aligned
  // bytes per channel at nx=200) by the sparse path.  Dense rows are latency-bound; dispatched back
  // to back they would occupy every workgroup slot while HBM idles, so each XCD alternates
  // kRowGroup dense rows with `sched_light` groups until the dense rows are used up.
  bool grouping = L->G > 1 && L->C == kTileC && L->nxc == 1 && L->row_base[1] % kRowGroup == 0;
  for (int g = 1; g < L->G && grouping; ++g)
    grouping = L->grid[g].n[1] % kGroupRows == 0 && L->grid[g].n[0] % 4 == 0 && L->grid[g].n[0] * kGroupRows <= kGroupMaxVox;
  L->n_dense_rows = grouping ? L->row_base[1] : L->R;
  L->n_groups = grouping ? (L->R - L->row_base[1]) / kGroupRows : 0;
  {
    const int dense_tiles = L->n_dense_rows * L->nxc;
    const int unit = kRowGroup * L->nxc;                      // dense tiles handed to an XCD at a time
    const int dense_units = (dense_tiles + unit - 1) / unit;
    L->sched_cycles = (dense_units + 7) / 8;
    const int groups_per_xcd = (L->n_groups + 7) / 8;
    L->sched_light = L->sched_cycles > 0 && groups_per_xcd >= 2 * L->sched_cycles ? 2 : 0;
    int rest = groups_per_xcd - L->sched_cycles * L->sched_light;
    if (rest < 0) rest = 0;
    L->per_xcd = L->sched_cycles * (unit + L->sched_light) + rest;
  }
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
  if (bytes) *bytes = off;
  return DHD_OK;
}

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
__global__ __launch_bounds__(64) void mghs_camera(Layout L, dhd_calib cal) {
  const int bn = blockIdx.x * 64 + threadIdx.x;
  if (bn >= L.B * L.N) return;
  CamMats m;
  load_camera(cal, bn, bn / L.N, &m);
  float* dst = L.cam + (size_t)bn * kCamFloats;
  const float* src = reinterpret_cast<const float*>(&m);
  for (int i = 0; i < (int)(sizeof(CamMats) / 4); ++i) dst[i] = src[i];
}

__global__ __launch_bounds__(kBlock) void mghs_geom_count(Layout L, dhd_calib cal, const uint8_t* __restrict__ band) {
  __shared__ CamMats cam;
  const int bn = blockIdx.y;
  const int b = bn / L.N;
  if (threadIdx.x < sizeof(CamMats) / 4)
    reinterpret_cast<float*>(&cam)[threadIdx.x] = L.cam[(size_t)bn * kCamFloats + threadIdx.x];
  __syncthreads();
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= L.dhw) return;
  const int w = i % L.fw;
  const int h = (i / L.fw) % L.fh;
  const int d = i / L.hw;
  float e[3];
  frustum_to_ego(cam, cal.frustum_u[w], cal.frustum_v[h], cal.frustum_d[d], e);
  const int pid = bn * L.dhw + i;
  int k0 = -1, r0 = 0, k1 = -1, r1 = 0;
  int v0 = voxel_of(L.grid[0], e, b);
  if (v0 >= 0) {
    k0 = L.vox_base[0] + v0;
    r0 = atomicAdd(&L.count[k0], 1);
  }
  if (L.G > 1) {
    int g = (int)band[bn * L.hw + (i % L.hw)] + 1;
    if (g < L.G) {
      int v1 = voxel_of(L.grid[g], e, b);
      if (v1 >= 0) {
        k1 = L.vox_base[g] + v1;
        r1 = atomicAdd(&L.count[k1], 1);
      }
    }
  }
  L.key[pid] = k0; L.rnk[pid] = r0;
  L.key[L.P + pid] = k1; L.rnk[L.P + pid] = r1;
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
// Exclusive scan of the per-voxel counters (two launches: block sums, then scan + carry-in).
// ---------------------------------------------------------------------------------------

__device__ __forceinline__ int wave_sum_i(int v) {
  for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, DHD_WAVE);
  return v;
}

__global__ __launch_bounds__(kBlock) void mghs_chunk_sum(const int* __restrict__ count, int V, int* __restrict__ chunk_sum) {
  __shared__ int ws[kBlock / DHD_WAVE];
  const int base = blockIdx.x * kChunk;
  int s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    int i = base + k * kBlock + threadIdx.x;
    if (i < V) s += count[i];
  }
  s = wave_sum_i(s);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) chunk_sum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(kBlock) void mghs_scan(const int* __restrict__ count, int V, const int* __restrict__ chunk_sum,
                                                     int* __restrict__ offset) {
  __shared__ int ws[kBlock / DHD_WAVE];
  __shared__ int ws2[kBlock / DHD_WAVE];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  int part = 0;
  for (int j = t; j < (int)blockIdx.x; j += kBlock) part += chunk_sum[j];
  part = wave_sum_i(part);
  if (lane == 0) ws[wv] = part;
  const int first = blockIdx.x * kChunk + t * kScanItems;
  int v[kScanItems];
  int tsum = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    v[k] = (first + k < V) ? count[first + k] : 0;
    tsum += v[k];
  }
  int incl = tsum;
  for (int d = 1; d < 64; d <<= 1) {
    int o = __shfl_up(incl, d, DHD_WAVE);
    if (lane >= d) incl += o;
  }
  if (lane == 63) ws2[wv] = incl;
  __syncthreads();
  int run = ws[0] + ws[1] + ws[2] + ws[3];
  for (int k = 0; k < wv; ++k) run += ws2[k];
  run += incl - tsum;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (first + k < V) offset[first + k] = run;
    run += v[k];
    if (first + k == V - 1) offset[V] = run;
  }
}

__global__ __launch_bounds__(kBlock) void mghs_scatter(Layout L) {
  const int bn = blockIdx.y;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= L.dhw) return;
  const int pid = bn * L.dhw + i;
  const int pix = bn * L.hw + (i % L.hw);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int k = L.key[j * L.P + pid];
    if (k >= 0) {
      int pos = L.offset[k] + L.rnk[j * L.P + pid];
      L.s_pid[pos] = pid;
      L.s_pix[pos] = pix;
      L.s_vox[pos] = k;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Pooling.  One workgroup per output row tile: grid g, batch b, slice z, row y, all x (<=256)
// and up to 64 channels.  The tile lives in LDS as tile[c][x] so that it is gathered
// voxel-by-voxel (lane = channel, transposed LDS access) and streamed to/from HBM as whole
// 16-byte-vectorised rows of the (B, nz*C, ny, nx) tensor.
// ---------------------------------------------------------------------------------------

struct OutPtrs { float* p[DHD_MAX_GRIDS]; };
struct InPtrs { const float* p[DHD_MAX_GRIDS]; };

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

#ifdef DHD_ABLATION
// Experiment-only build (make ablate): phases of the pooling kernels can be switched off to
// price them.  Never defined in libdhd_amd.so.
__device__ int g_ablate = 0;
__device__ long long* g_trace = nullptr;  // 8 x int64 per block: tile, t0..t3 (100 MHz wall clock), xcc, n_points, 0
#define ABL(bit) ((g_ablate & (bit)) != 0)
#define TRACE(slot, val)                                                             \
  do {                                                                               \
    if (g_trace && threadIdx.x == 0 && blockIdx.y == 0)           \
      g_trace[(size_t)blockIdx.x * 8 + (slot)] = (long long)(val);                   \
  } while (0)
#define NOW() wall_clock64()
#else
#define ABL(bit) false
#define TRACE(slot, val) do {} while (0)
#define NOW() 0
#endif

typedef float vfloat4 __attribute__((ext_vector_type(4)));  // native vector: accepted by the nontemporal builtins

constexpr int kGatherUnroll = 16;  // feature rows in flight per wave

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int lane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float lane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// blockIdx -> (output row, x chunk) for the row-only schedule: XCD-grouped (all chunks of kRowGroup
// rows stay on one XCD, back to back), then heavy (grid 0) row groups front-loaded 1:k.
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

// blockIdx -> work item of the forward work list (see make_layout): returns 0 = nothing,
// 1 = dense row tile (*row, *xchunk), 2 = sparse group (*row = first row of the group).
__device__ __forceinline__ int scheduled_work(const Layout& L, int block, int* row, int* xchunk) {
  const int xcd = block & 7, i = block >> 3;
  const int unit = kRowGroup * L.nxc, period = unit + L.sched_light;
  int group;
  if (i < L.sched_cycles * period) {
    const int cycle = i / period, r = i % period;
    if (r < unit) {
      const int t = (cycle * 8 + xcd) * unit + r;
      *row = t / L.nxc;
      *xchunk = t % L.nxc;
      return *row < L.n_dense_rows ? 1 : 0;
    }
    group = (cycle * L.sched_light + (r - unit)) * 8 + xcd;
  } else {
    group = (L.sched_cycles * L.sched_light + (i - L.sched_cycles * period)) * 8 + xcd;
  }
  *row = L.n_dense_rows + group * kGroupRows;
  *xchunk = 0;
  return group < L.n_groups ? 2 : 0;
}

// voxel (column of the tile) that sorted point `idx` belongs to: the x with offs[x] <= idx < offs[x+1]
__device__ __forceinline__ int column_of(const int* offs, int xn, int idx) {
  int lo = 0, hi = xn;  // invariant: offs[lo] <= idx < offs[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (offs[mid] <= idx) lo = mid; else hi = mid;
  }
  return lo;
}

// The sorted points of a row are split evenly over the waves of the workgroup (multiples of
// kGatherUnroll), independent of how they cluster into voxels.
__device__ __forceinline__ void wave_point_range(const int* offs, int xn, int wv, int* a, int* b) {
  const int s = rfl(offs[0]), n = rfl(offs[xn]) - s;
  int per = (n + kPoolWaves - 1) / kPoolWaves;
  per = (per + kGatherUnroll - 1) / kGatherUnroll * kGatherUnroll;
  *a = min(s + n, s + wv * per);
  *b = min(s + n, *a + per);
}

// ---------------------------------------------------------------------------------------
// Forward gather.  The sorted entries [S, E) of a row (or row group) are a non-decreasing stream of
// voxel ids.  Wave `wv` owns the voxels whose FIRST entry lies in its slice [a, b) of the stream
// (so every voxel has exactly one owner and no atomics are needed), accumulates
// sum depth * feat[pixel, lane] in a register (lane = channel) and hands each finished voxel to
// `flush(voxel, acc)`.  Entry indices are wave-uniform and travel through SGPRs (v_readlane);
// kGatherUnroll feature rows are in flight per wave; the next batch's index words and depth
// values are fetched while the current batch is processed.
// ---------------------------------------------------------------------------------------
template <class Flush>
__device__ __forceinline__ void gather_owned(const Layout& L, const float* __restrict__ depth,
                                             const float* __restrict__ featc, int S, int E, int a, int b, int lane,
                                             Flush&& flush) {
  if (a >= b) return;
  const int C = L.C;
  bool active = false, done = false;
  int cur = -1;
  float acc = 0.f;
  int idx = a + lane;
  int vox_n = -1, prev_n = -2, pix_n = 0, pid_n = 0;
  float dv_n = 0.f;
  if (idx < E) {
    vox_n = L.s_vox[idx];
    pix_n = L.s_pix[idx];
    pid_n = L.s_pid[idx];
    if (idx > S) prev_n = L.s_vox[idx - 1];
  }
  if (idx < E) dv_n = depth[pid_n];
  for (int a0 = a; a0 < E && !done; a0 += DHD_WAVE) {
    const int nb = min(DHD_WAVE, E - a0);
    const int vox = vox_n, pix = pix_n;
    const float dv = dv_n;
    const unsigned long long firsts = __ballot(lane < nb && vox != prev_n);
    idx = a0 + DHD_WAVE + lane;
    if (idx < E) {  // next batch (it may lie past b: the last owned voxel can run over the slice end)
      vox_n = L.s_vox[idx];
      pix_n = L.s_pix[idx];
      pid_n = L.s_pid[idx];
      prev_n = L.s_vox[idx - 1];
    }
    for (int i0 = 0; i0 < nb && !done; i0 += kGatherUnroll) {
      float f[kGatherUnroll];
#pragma unroll
      for (int j = 0; j < kGatherUnroll; ++j) f[j] = featc[(size_t)lane_i(pix, min(i0 + j, nb - 1)) * C];
#pragma unroll
      for (int j = 0; j < kGatherUnroll; ++j) {
        const int i = i0 + j;
        if (i < nb && !done) {
          if ((firsts >> i) & 1ull) {
            if (active) flush(cur, acc);
            active = a0 + i < b;
            done = !active;
            cur = lane_i(vox, i);
            acc = 0.f;
          }
          if (active) acc = fmaf(lane_f(dv, i), f[j], acc);
        }
      }
    }
    if (idx < E) dv_n = depth[pid_n];
  }
  if (active) flush(cur, acc);
}

__device__ __forceinline__ void wave_slice(int S, int E, int wv, int* a, int* b) {
  const int n = E - S;
  int per = (n + kPoolWaves - 1) / kPoolWaves;
  per = (per + kGatherUnroll - 1) / kGatherUnroll * kGatherUnroll;
  *a = min(E, S + wv * per);
  *b = min(E, *a + per);
}

// Dense row tile: [cn channels][xn voxels] in LDS, zero-filled, gathered, streamed out as cn rows of
// xn floats.  Called by every thread of the workgroup (contains barriers).
template <bool FULL>
__device__ __forceinline__ void fwd_dense_row(const Layout& L, const RowTile& rt, int c0, int cn, int x0, int xn,
                                              float* tile, int* offs, int tile_stride, const float* __restrict__ depth,
                                              const float* __restrict__ feat, float* __restrict__ og) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  __syncthreads();  // the LDS buffers may still be in use by a previous call
  for (int i = t; i < cn * tile_stride; i += kPoolBlock) tile[i] = 0.f;
  if (FULL) {
    if (t < 2) offs[t] = L.offset[rt.vrow + x0 + (t ? xn : 0)];
  } else {
    for (int i = t; i <= xn; i += kPoolBlock) offs[i] = L.offset[rt.vrow + x0 + i];
  }
  __syncthreads();
  TRACE(2, NOW());
  const int S = FULL ? rfl(offs[0]) : offs[0], E = FULL ? rfl(offs[1]) : offs[xn];
  TRACE(6, E - S);

  if (E != S && !ABL(8)) {  // rows without any point skip the gather entirely
    if (FULL) {
      float* trow = tile + lane * tile_stride;
      const int vbase = rt.vrow + x0;
      int a, b;
      wave_slice(S, E, wv, &a, &b);
      gather_owned(L, depth, feat + c0 + lane, S, E, a, b, lane, [&](int vox, float acc) { trow[vox - vbase] = acc; });
    } else {
      // lane -> (sub-slot, channel): CL lanes cover the channels, 64/CL points are in flight per wave
      const int CL = next_pow2(cn);
      const int nsub = DHD_WAVE / CL;
      const int c = lane % CL, sub = lane / CL;
      const bool c_ok = c < cn;
      const float* featc = feat + c0 + c;
      for (int x = wv; x < xn; x += kPoolWaves) {
        const int s = offs[x], e = offs[x + 1];
        if (e == s) continue;
        float acc = 0.f;
        for (int s0 = s; s0 < e; s0 += DHD_WAVE) {
          const int nb = min(DHD_WAVE, e - s0);
          int pix = 0;
          float dv = 0.f;
          if (lane < nb) {
            pix = L.s_pix[s0 + lane];
            dv = depth[L.s_pid[s0 + lane]];
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
    }
    __syncthreads();
  }
  TRACE(3, NOW());
  if (ABL(16)) return;

  // stream the tile out: one (channel) row of xn floats per wave iteration, 16 bytes per lane,
  // non-temporal so the 700 MB output stream does not evict the gather's working set from L2
  const bool vec = ((rt.nx & 3) == 0) && ((xn & 3) == 0) && ((x0 & 3) == 0);
  for (int cc = wv; cc < cn; cc += kPoolWaves) {
    size_t row = ((((size_t)rt.b * rt.nz + rt.z) * L.C + c0 + cc) * rt.ny + rt.y) * rt.nx + x0;
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

// Sparse group: kGroupRows consecutive rows of one (grid, batch, slice), all 64 channels.  Per channel
// the group is one contiguous, cache-line aligned run of kGroupRows*nx floats in the output, almost
// all zeros.  The few non-empty voxels are summed into a compact LDS table comp[slot][channel]
// (slot_of[position] = slot + 1), then every wave streams zero-filled 16-byte vectors for its
// channels, patching in the table values.  Returns false (uniformly) when the group does not fit
// the table; the caller then processes its rows through the dense tile.
struct GroupLds {
  float* comp;              // [kGroupSlots][kTileC]
  unsigned short* slot_of;  // [kGroupMaxVox]
  int* ctl;                 // [0] slots used, [1] overflow, [2] S, [3] E
};

__device__ __forceinline__ GroupLds group_lds(char* smem) {
  GroupLds g;
  g.comp = reinterpret_cast<float*>(smem);
  g.slot_of = reinterpret_cast<unsigned short*>(smem + (size_t)kGroupSlots * kTileC * 4);
  g.ctl = reinterpret_cast<int*>(smem + (size_t)kGroupSlots * kTileC * 4 + (size_t)kGroupMaxVox * 2);
  return g;
}
constexpr size_t kGroupLdsBytes = (size_t)kGroupSlots * kTileC * 4 + (size_t)kGroupMaxVox * 2 + 16;

__device__ __forceinline__ bool fwd_sparse_group(const Layout& L, const RowTile& rt, char* smem,
                                                 const float* __restrict__ depth, const float* __restrict__ feat,
                                                 float* __restrict__ og) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  GroupLds G = group_lds(smem);
  const int nvox = kGroupRows * rt.nx;
  __syncthreads();
  for (int i = t; i < kGroupMaxVox / 2; i += kPoolBlock) reinterpret_cast<unsigned*>(G.slot_of)[i] = 0u;
  if (t < 2) G.ctl[t] = 0;
  if (t == 2) G.ctl[2] = L.offset[rt.vrow];
  if (t == 3) G.ctl[3] = L.offset[rt.vrow + nvox];
  __syncthreads();
  TRACE(2, NOW());
  const int S = rfl(G.ctl[2]), E = rfl(G.ctl[3]);
  TRACE(6, E - S);
  if (E - S > kGroupMaxPts) return false;
  if (E != S && !ABL(8)) {
    int a, b;
    wave_slice(S, E, wv, &a, &b);
    const int vbase = rt.vrow;
    gather_owned(L, depth, feat + lane, S, E, a, b, lane, [&](int vox, float acc) {
      int slot = 0;
      if (lane == 0) slot = atomicAdd(&G.ctl[0], 1);
      slot = rfl(slot);
      if (slot < kGroupSlots) {
        G.comp[slot * kTileC + lane] = acc;
        if (lane == 0) G.slot_of[vox - vbase] = (unsigned short)(slot + 1);
      } else if (lane == 0) {
        G.ctl[1] = 1;
      }
    });
    __syncthreads();
    if (rfl(G.ctl[1])) return false;
  }
  TRACE(3, NOW());
  if (ABL(16)) return true;

  // stream out: channel cc is one run of nvox floats starting at row y0 of its plane
  const int nvec = nvox / 4;
  for (int cc = wv; cc < kTileC; cc += kPoolWaves) {
    vfloat4* dst = reinterpret_cast<vfloat4*>(og + ((((size_t)rt.b * rt.nz + rt.z) * L.C + cc) * rt.ny + rt.y) * rt.nx);
    for (int i = lane; i < nvec; i += DHD_WAVE) {
      vfloat4 v = {0.f, 0.f, 0.f, 0.f};
      const uint2 sl = *reinterpret_cast<const uint2*>(G.slot_of + 4 * i);
      if (sl.x | sl.y) {
        const unsigned s0 = sl.x & 0xffffu, s1 = sl.x >> 16, s2 = sl.y & 0xffffu, s3 = sl.y >> 16;
        if (s0) v.x = G.comp[(s0 - 1) * kTileC + cc];
        if (s1) v.y = G.comp[(s1 - 1) * kTileC + cc];
        if (s2) v.z = G.comp[(s2 - 1) * kTileC + cc];
        if (s3) v.w = G.comp[(s3 - 1) * kTileC + cc];
      }
      __builtin_nontemporal_store(v, dst + i);
    }
  }
  return true;
}

template <bool FULL>
__global__ __launch_bounds__(kPoolBlock) void mghs_pool_fwd(Layout L, const float* __restrict__ depth,
                                                            const float* __restrict__ feat, OutPtrs out, int tile_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = reinterpret_cast<float*>(smem);                                 // dense: [kTileC][tile_stride]
  int* offs = reinterpret_cast<int*>(smem + (size_t)kTileC * tile_stride * 4);  // dense: [kMaxTileX + 1]

  int row, xchunk;
  const int kind = scheduled_work(L, blockIdx.x, &row, &xchunk);
  if (kind == 0) return;
  RowTile rt;
  if (!decode_row(L, row, &rt)) return;
  TRACE(0, rt.vrow); TRACE(1, NOW());
  TRACE(5, __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11)));  // HW_REG_XCC_ID, bits 3:0
  const int c0 = blockIdx.y * kTileC;
  const int cn = FULL ? kTileC : min(kTileC, L.C - c0);
  if (kind == 1) {
    const int x0 = xchunk * kMaxTileX;
    if (x0 < rt.nx)
      fwd_dense_row<FULL>(L, rt, c0, cn, x0, min(kMaxTileX, rt.nx - x0), tile, offs, tile_stride, depth, feat, out.p[rt.g]);
  } else {
    // groups exist only when FULL, one x chunk and one channel tile (make_layout)
    if (!fwd_sparse_group(L, rt, smem, depth, feat, out.p[rt.g])) {
      for (int r = 0; r < kGroupRows; ++r) {
        RowTile r1 = rt;
        r1.y = rt.y + r;
        r1.vrow = rt.vrow + r * rt.nx;
        fwd_dense_row<FULL>(L, r1, c0, cn, 0, rt.nx, tile, offs, tile_stride, depth, feat, out.p[rt.g]);
      }
    }
  }
  TRACE(4, NOW());
}

template <bool FULL>
__global__ __launch_bounds__(kPoolBlock) void mghs_pool_bwd(Layout L, const float* __restrict__ depth,
                                                            const float* __restrict__ feat, InPtrs og,
                                                            float* __restrict__ feat_grad, int tile_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = reinterpret_cast<float*>(smem);
  int* offs = reinterpret_cast<int*>(smem + (size_t)kTileC * tile_stride * 4);

  RowTile rt;
  int xchunk;
  if (!decode_row(L, scheduled_tile(L, blockIdx.x, &xchunk), &rt)) return;
  const int c0 = blockIdx.y * kTileC;
  const int cn = FULL ? kTileC : min(kTileC, L.C - c0);
  const int x0 = xchunk * kMaxTileX;
  if (x0 >= rt.nx) return;
  const int xn = min(kMaxTileX, rt.nx - x0);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;

  for (int i = t; i <= xn; i += kPoolBlock) offs[i] = L.offset[rt.vrow + x0 + i];
  __syncthreads();
  if (offs[xn] == offs[0]) return;  // no point lands in this row: its out_grad is never read

  const float* gsrc = og.p[rt.g];
  const bool vec = ((rt.nx & 3) == 0) && ((xn & 3) == 0);
  for (int cc = wv; cc < cn; cc += kPoolWaves) {
    size_t row = ((((size_t)rt.b * rt.nz + rt.z) * L.C + c0 + cc) * rt.ny + rt.y) * rt.nx + x0;
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
  if (ABL(8)) return;

  // depth-gradient parts: a point appears at most once in grid 0 and once in its band grid, so
  // each part has exactly one writer (plain stores); dhd_mghs_backward sums the two parts.
  // With several channel tiles (C > 64) the parts need atomics again.
  float* dgp = L.dg_part + (rt.g == 0 ? 0 : L.P);
  const bool dg_atomic = L.C > kTileC;

  if (FULL) {
    const float* featc = feat + c0 + lane;
    float* fgc = feat_grad + c0 + lane;
    const float* trow = tile + lane * tile_stride;
    int a, b;
    wave_point_range(offs, xn, wv, &a, &b);
    int pix_n = 0, pid_n = 0;
    float dv_n = 0.f;
    if (a + lane < b) { pix_n = L.s_pix[a + lane]; pid_n = L.s_pid[a + lane]; }
    if (a + lane < b) dv_n = depth[pid_n];
    for (int a0 = a; a0 < b; a0 += DHD_WAVE) {
      const int nb = min(DHD_WAVE, b - a0);
      const int pix = pix_n, pid = pid_n;
      const float dv = dv_n;
      const int col = lane < nb ? column_of(offs, xn, a0 + lane) : 0;
      const int nxt = a0 + DHD_WAVE + lane;
      if (nxt < b) { pix_n = L.s_pix[nxt]; pid_n = L.s_pid[nxt]; }
      float mine = 0.f;  // <out_grad[voxel], feat[pixel]> of the point this lane loaded
      int i = 0;
      for (; i + kGatherUnroll <= nb; i += kGatherUnroll) {
        float f[kGatherUnroll], g[kGatherUnroll];
        size_t off[kGatherUnroll];
#pragma unroll
        for (int j = 0; j < kGatherUnroll; ++j) {
          off[j] = (size_t)lane_i(pix, i + j) * L.C;
          f[j] = featc[off[j]];
          g[j] = trow[lane_i(col, i + j)];
        }
#pragma unroll
        for (int j = 0; j < kGatherUnroll; ++j) {
          if (!ABL(1)) unsafeAtomicAdd(fgc + off[j], g[j] * lane_f(dv, i + j));
          float tot = ABL(2) ? g[j] * f[j] : wave_sum_bcast(g[j] * f[j]);
          if (lane == i + j) mine = tot;
        }
      }
      for (; i < nb; ++i) {
        const size_t off = (size_t)lane_i(pix, i) * L.C;
        const float f = featc[off], g = trow[lane_i(col, i)];
        if (!ABL(1)) unsafeAtomicAdd(fgc + off, g * lane_f(dv, i));
        float tot = ABL(2) ? g * f : wave_sum_bcast(g * f);
        if (lane == i) mine = tot;
      }
      if (lane < nb && !ABL(4)) {
        if (dg_atomic) unsafeAtomicAdd(dgp + pid, mine); else dgp[pid] = mine;
      }
      if (nxt < b) dv_n = depth[pid_n];
    }
    return;
  }

  const int CL = next_pow2(cn);
  const int nsub = DHD_WAVE / CL;
  const int c = lane % CL, sub = lane / CL;
  const bool c_ok = c < cn;
  const float* featc = feat + c0 + c;
  float* fgc = feat_grad + c0 + c;

  for (int x = wv; x < xn; x += kPoolWaves) {
    const int s = offs[x], e = offs[x + 1];
    if (e == s) continue;
    const float g = c_ok ? tile[c * tile_stride + x] : 0.f;
    for (int s0 = s; s0 < e; s0 += DHD_WAVE) {
      const int nb = min(DHD_WAVE, e - s0);
      int pix = 0, pid = 0;
      float dv = 0.f;
      if (lane < nb) {
        pix = L.s_pix[s0 + lane];
        pid = L.s_pid[s0 + lane];
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

// depth_grad = part(grid 0) + part(band grid)
__global__ __launch_bounds__(kBlock) void mghs_sum_parts(const float* __restrict__ part, int P, float* __restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < P) out[i] = part[i] + part[P + i];
}

int pool_smem_and_stride(const Layout& L, int* stride, size_t* smem) {
  int nx_max = 0;
  for (int g = 0; g < L.G; ++g) nx_max = nx_max > L.grid[g].n[0] ? nx_max : L.grid[g].n[0];
  int xt = nx_max < kMaxTileX ? nx_max : kMaxTileX;
  int st = xt | 1;  // odd row stride: the transposed (lane = channel) LDS accesses hit 32 distinct banks
  *stride = st;
  *smem = (size_t)kTileC * st * 4 + (size_t)(kMaxTileX + 1) * 4;
  if (*smem < kGroupLdsBytes) *smem = kGroupLdsBytes;
  return nx_max;
}

}  // namespace

extern "C" {

#ifdef DHD_ABLATION
int dhd_debug_set_ablation(int mask) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ablate), &mask, sizeof(int));
}
int dhd_debug_set_trace(void* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &buf, sizeof(void*));
}
#endif

int dhd_abi_version(void) { return DHD_ABI_VERSION; }

int dhd_mghs_workspace_bytes(const dhd_mghs_desc* desc, size_t* bytes) {
  if (!bytes) return DHD_EINVAL;
  Layout L;
  return make_layout(desc, nullptr, &L, bytes);
}

static int check_calib(const dhd_calib* c) {
  if (!c || !c->sensor2ego || !c->post_tran || !c->bda || !c->frustum_u || !c->frustum_v || !c->frustum_d)
    return DHD_EINVAL;
  if (!c->inv_post_rot && !c->post_rot) return DHD_EINVAL;
  if (!c->combine && !c->intrin) return DHD_EINVAL;
  return DHD_OK;
}

int dhd_mghs_prepare(const dhd_mghs_desc* desc, const dhd_calib* calib, const uint8_t* band, void* workspace,
                     size_t workspace_bytes, void* stream) {
  Layout L;
  size_t need = 0;
  int rc = make_layout(desc, workspace, &L, &need);
  if (rc) return rc;
  if (!workspace) return DHD_EINVAL;
  if (workspace_bytes < need) return DHD_ENOSPACE;
  if ((rc = check_calib(calib))) return rc;
  if (L.G > 1 && !band) return DHD_EINVAL;
  hipStream_t st = dhd_stream(stream);
  DHD_HIP(hipMemsetAsync(L.count, 0, (size_t)L.V * 4, st));
  hipLaunchKernelGGL(mghs_camera, dim3(dhd_cdiv(L.B * L.N, 64)), dim3(64), 0, st, L, *calib);
  DHD_LAUNCH_CHECK();
  dim3 gp(dhd_cdiv(L.dhw, kBlock), L.B * L.N);
  hipLaunchKernelGGL(mghs_geom_count, gp, dim3(kBlock), 0, st, L, *calib, band);
  DHD_LAUNCH_CHECK();
  hipLaunchKernelGGL(mghs_chunk_sum, dim3(L.n_chunks), dim3(kBlock), 0, st, L.count, L.V, L.chunk_sum);
  DHD_LAUNCH_CHECK();
  hipLaunchKernelGGL(mghs_scan, dim3(L.n_chunks), dim3(kBlock), 0, st, L.count, L.V, L.chunk_sum, L.offset);
  DHD_LAUNCH_CHECK();
  hipLaunchKernelGGL(mghs_scatter, gp, dim3(kBlock), 0, st, L);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

int dhd_mghs_forward(const dhd_mghs_desc* desc, const float* depth, const float* feat_nhwc,
                     float* const out[DHD_MAX_GRIDS], const void* workspace, void* stream) {
  Layout L;
  int rc = make_layout(desc, const_cast<void*>(workspace), &L, nullptr);
  if (rc) return rc;
  if (!workspace || !depth || !feat_nhwc || !out) return DHD_EINVAL;
  OutPtrs o;
  for (int g = 0; g < DHD_MAX_GRIDS; ++g) {
    o.p[g] = g < L.G ? out[g] : nullptr;
    if (g < L.G && !out[g]) return DHD_EINVAL;
  }
  int stride; size_t smem;
  int nx_max = pool_smem_and_stride(L, &stride, &smem);
  (void)nx_max;
  dim3 grid(8 * L.per_xcd, dhd_cdiv(L.C, kTileC), 1);
  if (L.C % kTileC == 0)
    hipLaunchKernelGGL(mghs_pool_fwd<true>, grid, dim3(kPoolBlock), smem, dhd_stream(stream), L, depth, feat_nhwc, o, stride);
  else
    hipLaunchKernelGGL(mghs_pool_fwd<false>, grid, dim3(kPoolBlock), smem, dhd_stream(stream), L, depth, feat_nhwc, o, stride);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

int dhd_mghs_backward(const dhd_mghs_desc* desc, const float* depth, const float* feat_nhwc,
                      const float* const out_grad[DHD_MAX_GRIDS], float* depth_grad, float* feat_grad_nhwc,
                      void* workspace, void* stream) {
  Layout L;
  int rc = make_layout(desc, workspace, &L, nullptr);
  if (rc) return rc;
  if (!workspace || !depth || !feat_nhwc || !out_grad || !depth_grad || !feat_grad_nhwc) return DHD_EINVAL;
  InPtrs in;
  for (int g = 0; g < DHD_MAX_GRIDS; ++g) {
    in.p[g] = g < L.G ? out_grad[g] : nullptr;
    if (g < L.G && !out_grad[g]) return DHD_EINVAL;
  }
  hipStream_t st = dhd_stream(stream);
  DHD_HIP(hipMemsetAsync(L.dg_part, 0, 2 * (size_t)L.P * 4, st));
  DHD_HIP(hipMemsetAsync(feat_grad_nhwc, 0, (size_t)L.B * L.N * L.hw * L.C * 4, st));
  int stride; size_t smem;
  int nx_max = pool_smem_and_stride(L, &stride, &smem);
  (void)nx_max;
  dim3 grid(xcd_grouped_blocks(L.R * L.nxc, kRowGroup * L.nxc), dhd_cdiv(L.C, kTileC), 1);
  if (L.C % kTileC == 0)
    hipLaunchKernelGGL(mghs_pool_bwd<true>, grid, dim3(kPoolBlock), smem, st, L, depth, feat_nhwc, in, feat_grad_nhwc,
                       stride);
  else
    hipLaunchKernelGGL(mghs_pool_bwd<false>, grid, dim3(kPoolBlock), smem, st, L, depth, feat_nhwc, in, feat_grad_nhwc,
                       stride);
  DHD_LAUNCH_CHECK();
  hipLaunchKernelGGL(mghs_sum_parts, dim3(dhd_cdiv(L.P, kBlock)), dim3(kBlock), 0, st, L.dg_part, L.P, depth_grad);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

int dhd_mghs_voxel_index(const dhd_mghs_desc* desc, const dhd_calib* calib, int grid_index, int32_t* rank_map,
                         float* ego, void* stream) {
  Layout L;
  int rc = make_layout(desc, nullptr, &L, nullptr);
  if (rc) return rc;
  if ((rc = check_calib(calib))) return rc;
  if (!rank_map || grid_index < 0 || grid_index >= L.G) return DHD_EINVAL;
  dim3 gp(dhd_cdiv(L.dhw, kBlock), L.B * L.N);
  hipLaunchKernelGGL(mghs_voxel_index_kernel, gp, dim3(kBlock), 0, dhd_stream(stream), L, *calib, grid_index,
                     rank_map, ego);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

int dhd_mghs_stats(const dhd_mghs_desc* desc, const void* workspace, int32_t n_kept[DHD_MAX_GRIDS],
                   int32_t n_intervals[DHD_MAX_GRIDS], void* stream) {
  Layout L;
  int rc = make_layout(desc, const_cast<void*>(workspace), &L, nullptr);
  if (rc) return rc;
  if (!workspace || !n_kept || !n_intervals) return DHD_EINVAL;
  hipStream_t st = dhd_stream(stream);
  DHD_HIP(hipStreamSynchronize(st));
  // not a hot path: copy the counters back and reduce on the host
  int* h = static_cast<int*>(malloc((size_t)L.V * 4));
  if (!h) return DHD_ENOSPACE;
  hipError_t e = hipMemcpy(h, L.count, (size_t)L.V * 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { free(h); return (int)e; }
  for (int g = 0; g < DHD_MAX_GRIDS; ++g) {
    long k = 0, iv = 0;
    for (int v = L.vox_base[g]; v < L.vox_base[g + 1]; ++v) { k += h[v]; iv += h[v] > 0; }
    n_kept[g] = (int32_t)k; n_intervals[g] = (int32_t)iv;
  }
  free(h);
  return DHD_OK;
}

}  // extern "C"
