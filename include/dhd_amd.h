/*
 * dhd_amd.h -- C ABI of libdhd_amd.so: the MI355X (gfx950) implementation of DHD's
 * height-decoupled LSS view transform (MGHS) and SFA attention stage.
 *
 * Conventions
 *   - Every pointer marked [dev] is device (HBM) memory owned by the caller; the library
 *     never allocates, frees or copies.  Scratch lives in one caller-provided workspace.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream, which is
 *     what the reference's launches use: ops/bev_pool_v2/src/bev_pool_cuda.cu:129,138).
 *     All work is enqueued asynchronously on it; nothing synchronises.
 *   - Return value: 0 on success, a positive hipError_t if a launch failed, or one of the
 *     negative DHD_E* codes for argument errors.  (The reference validates nothing and
 *     returns void: ops/bev_pool_v2/src/bev_pool.cpp:30-57.)
 *   - Floating point data is float32 and indices int32, exactly as in the reference
 *     (ops/bev_pool_v2/bev_pool.py:19-25), unless an argument says otherwise: since ABI 3 the large tensors at the
 *     edges of the MGHS and SFA operators may be float16 / bfloat16 (dhd_tensor_view.dtype, dhd_sfa_weights.io_dtype),
 *     since ABI 4 every tensor of the SFA stage may be (dhd_sfa_weights.storage_dtype) -- what a caller inside an
 *     autocast region holds.  Arithmetic and accumulation are float32 in every case.
 *
 * File:line citations are into /root/reference/projects/mmdet3d_plugin/.
 */
#ifndef DHD_AMD_H
#define DHD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DHD_OK 0
#define DHD_EINVAL (-1)     /* null pointer / non-positive size / inconsistent sizes */
#define DHD_ENOSPACE (-2)   /* workspace too small */
#define DHD_EUNSUPPORTED (-3)

#define DHD_ABI_VERSION 5
int dhd_abi_version(void);

/* ------------------------------------------------------------------------------------ *
 * 1. Operator-level drop-in: the two entry points of the reference's pybind extension.
 * ------------------------------------------------------------------------------------ */

/* Replaces bev_pool_v2_forward (ops/bev_pool_v2/src/bev_pool.cpp:30-57 -> kernel
 * bev_pool_cuda.cu:21-50).  Same argument meaning and order of the index arrays
 * (interval_lengths BEFORE interval_starts).  `out` is (B,Dz,Dy,Dx,C), must be pre-zeroed by
 * the caller (bev_pool.py:27) and is written in place: for every interval i,
 *   out[ranks_bev[start_i]*c + ch] = sum_k feat[ranks_feat[start_i+k]*c + ch] * depth[ranks_depth[start_i+k]]. */
int dhd_bev_pool_v2_forward(const float* depth,            /* [dev] (B,N,D,fH,fW)     */
                            const float* feat,             /* [dev] (B,N,fH,fW,C)     */
                            float* out,                    /* [dev] (B,Dz,Dy,Dx,C)    */
                            const int32_t* ranks_depth,    /* [dev] (n_points)        */
                            const int32_t* ranks_feat,     /* [dev] (n_points)        */
                            const int32_t* ranks_bev,      /* [dev] (n_points)        */
                            const int32_t* interval_lengths, /* [dev] (n_intervals)   */
                            const int32_t* interval_starts,  /* [dev] (n_intervals)   */
                            int c, int n_intervals, void* stream);

/* Replaces bev_pool_v2_backward (bev_pool.cpp:74-104 -> kernel bev_pool_cuda.cu:69-123).
 * Intervals are those of the points re-grouped by ranks_feat (bev_pool.py:47-57); depth_grad
 * and feat_grad are pre-zeroed by the caller (bev_pool.py:67-68).  out_grad is (B,Dz,Dy,Dx,C). */
int dhd_bev_pool_v2_backward(const float* out_grad, float* depth_grad, float* feat_grad,
                             const float* depth, const float* feat,
                             const int32_t* ranks_depth, const int32_t* ranks_feat,
                             const int32_t* ranks_bev,
                             const int32_t* interval_lengths_bp, const int32_t* interval_starts_bp,
                             int c, int n_intervals_bp, void* stream);

/* The re-grouping QuickCumsumCuda.backward performs on every call (bev_pool.py:47-57: argsort by ranks_feat, three
 * gathers, run-length scan) as a device counting sort without host synchronisation: the point lists ordered by feature
 * pixel (inside a pixel by ascending ranks_depth: deterministic) and ONE interval per pixel, empty pixels included with
 * length 0 -- pass n_intervals_bp = n_pixels to dhd_bev_pool_v2_backward, which skips empty intervals.  n_pixels =
 * B*N*fH*fW (rows of feat); points whose ranks_feat is outside [0, n_pixels) are dropped.  The *_bp lists have n_points
 * entries (only the first sum(lengths) are written), interval_*_bp n_pixels.  scratch: dhd_bev_pool_v2_regroup_scratch_bytes. */
size_t dhd_bev_pool_v2_regroup_scratch_bytes(int n_points, int n_pixels);
int dhd_bev_pool_v2_regroup(const int32_t* ranks_depth, const int32_t* ranks_feat, const int32_t* ranks_bev,
                            int n_points, int n_pixels, int32_t* ranks_depth_bp, int32_t* ranks_feat_bp,
                            int32_t* ranks_bev_bp, int32_t* interval_starts_bp, int32_t* interval_lengths_bp,
                            void* scratch /*[dev]*/, size_t scratch_bytes, void* stream);

/* Fused form of the operator AS THE REFERENCE'S WRAPPER USES IT (bev_pool.py:27,86-106: zero-filled (B,Dz,Dy,Dx,C) tensor ->
 * kernel -> `permute(0, 4, 1, 2, 3).contiguous()`): `out` is the FINAL (B,C,Dz,Dy,Dx) tensor, written exactly once, zeros
 * included (no pre-zeroing, no permute copy); the backward takes out_grad in that same layout.  Same index lists as above
 * (intervals = runs of equal ranks_bev, as voxel_pooling_prepare_v2 builds them, lss_heightmap.py:303-371; any interval
 * order; voxels outside [0, B*Dz*Dy*Dx) are dropped).  Shapes: c == 64, Dy % 4 == 0, Dx % 4 == 0, Dx <= 256, else
 * DHD_EUNSUPPORTED (use the entry points above).  `state` (voxel -> row map, kept from the forward to its backward) and
 * `scratch` (reusable between calls on a stream) are 256-byte aligned device buffers of the sizes
 * dhd_bev_pool_v2_fused_workspace_bytes returns; depth_grad / feat_grad are pre-zeroed by the caller as above; the *_bp lists
 * are those of dhd_bev_pool_v2_regroup.  Values are bit-identical to the unfused entry points (same kernels, same order). */
int dhd_bev_pool_v2_fused_workspace_bytes(int c, int batch, int dz, int dy, int dx, int n_intervals, size_t* state_bytes,
                                          size_t* scratch_bytes);
int dhd_bev_pool_v2_fused_forward(const float* depth, const float* feat, float* out /* [dev] (B,C,Dz,Dy,Dx) */,
                                  const int32_t* ranks_depth, const int32_t* ranks_feat, const int32_t* ranks_bev,
                                  const int32_t* interval_lengths, const int32_t* interval_starts, int c, int n_intervals,
                                  int batch, int dz, int dy, int dx, void* state, size_t state_bytes,
                                  int state_valid /* 1: `state` was filled by an earlier forward with the SAME ranks_bev /
                                                     interval lists (a static rig) and is reused as it is */,
                                  void* scratch, size_t scratch_bytes, void* stream);
int dhd_bev_pool_v2_fused_backward(const float* out_grad /* [dev] (B,C,Dz,Dy,Dx) */, float* depth_grad, float* feat_grad,
                                   const float* depth, const float* feat, const int32_t* ranks_depth_bp,
                                   const int32_t* ranks_feat_bp, const int32_t* ranks_bev_bp,
                                   const int32_t* interval_lengths_bp, const int32_t* interval_starts_bp, int c,
                                   int n_intervals_bp, int n_intervals, int batch, int dz, int dy, int dx, void* state,
                                   size_t state_bytes, void* scratch, size_t scratch_bytes, void* stream);

/* ------------------------------------------------------------------------------------ *
 * 2. Fused MGHS view transform (replaces the 4x get_ego_coor + 4x voxel_pooling_prepare_v2
 *    + 4x bev_pool_v2 + permute + cat chain of models/necks/lss_heightmap.py:380-459).
 * ------------------------------------------------------------------------------------ */

#define DHD_MAX_GRIDS 4

/* One voxel grid (MGHS.create_grid_infos, lss_heightmap.py:86-102): float32 lower bound,
 * interval and size per axis (x,y,z), computed by the caller exactly as the reference does
 * (python double arithmetic, then float32), plus the integer extents n = int(size). */
typedef struct dhd_grid {
  float lower[3];
  float interval[3];
  float size[3];
  int32_t n[3]; /* nx, ny, nz */
} dhd_grid;

/* Problem description shared by prepare / forward / backward. */
typedef struct dhd_mghs_desc {
  int32_t batch;    /* B                                   */
  int32_t n_cams;   /* N                                   */
  int32_t n_depth;  /* D  (frustum depth bins)             */
  int32_t fh, fw;   /* feature map                          */
  int32_t channels; /* C  (context channels)               */
  int32_t n_grids;  /* 1..4; grid 0 pools every pixel, grid k>=1 pools pixels of band k-1 */
  dhd_grid grid[DHD_MAX_GRIDS];
  int32_t flags;    /* DHD_MGHS_* bits below; per call, nothing about them is process-wide */
} dhd_mghs_desc;

/* dhd_mghs_desc.flags
 *   DHD_MGHS_DETERMINISTIC   reproducible forward sums.  By default the order of the entries inside a voxel is the
 *       arrival order of the counting atomics of dhd_mghs_prepare, so the float32 sum of a voxel differs in its last bits
 *       from run to run (the reference's order is unspecified as well: unstable argsort, lss_heightmap.py:355).  With the
 *       bit set, prepare orders the entries of every voxel by point id (one extra ranking pass + a second scatter):
 *       dhd_mghs_forward is then bit-identical from run to run, and equal to a sum in ascending ranks_depth order.  The
 *       backward is deterministic either way.
 *   DHD_MGHS_FEAT_GRAD_NCHW  dhd_mghs_backward* write the context gradient as (B*N, C, fH, fW) -- the layout of the
 *       reference's tran_feat -- instead of (B*N, fH, fW, C), so that the caller needs no transposition pass. */
#define DHD_MGHS_DETERMINISTIC 1
#define DHD_MGHS_FEAT_GRAD_NCHW 2
#define DHD_MGHS_DEBUG_SCAN_SELF_SERVE 4 /* tests only: the single-pass scan never waits for another workgroup (its bounded-spin fallback for every chunk) */

/* Camera calibration, all [dev] float32, laid out as the reference's input list
 * (lss_heightmap.py:384-390).  inv_post_rot / combine are OPTIONAL (may be NULL): when given they
 * are used instead of the library's own 3x3 inverse (so a caller can inject the matrices
 * torch.inverse produced, lss_heightmap.py:209,220); when NULL they are derived on the device
 * with LU + partial pivoting in float32 (the LAPACK algorithm the reference reaches through
 * torch.inverse). */
typedef struct dhd_calib {
  const float* sensor2ego;   /* (B,N,4,4) */
  const float* intrin;       /* (B,N,3,3) */
  const float* post_rot;     /* (B,N,3,3) */
  const float* post_tran;    /* (B,N,3)   */
  const float* bda;          /* (B,3,3)   */
  const float* inv_post_rot; /* (B,N,3,3) or NULL */
  const float* combine;      /* (B,N,3,3) or NULL: sensor2ego[:3,:3] @ inv(intrin) */
  const float* frustum_u;    /* (fW)  MGHS.create_frustum axes, lss_heightmap.py:105-134 */
  const float* frustum_v;    /* (fH) */
  const float* frustum_d;    /* (D)  */
} dhd_calib;

/* Device memory of a view transform, in two caller-owned parts (both 256-byte aligned, DHD_EINVAL otherwise):
 *   state   : what dhd_mghs_backward needs from dhd_mghs_prepare -- per-voxel slot prefix, voxel id of every slot, per-point
 *             slots (about 28 MB at DHD-S, B = 4); for layouts off the compact path (C != 64 or grid shapes the segment
 *             writer does not cover) also the grouped entry lists and their per-voxel offsets, which that backward walks
 *             again.  One per prepare whose backward is still to come.
 *   scratch : everything else (counters, sort keys, the grouped entry lists, the compact per-voxel table: about 0.45 GB at
 *             DHD-S, B = 4).  Valid from a prepare to the forward that follows it; dhd_mghs_backward uses it as plain scratch.
 *             Calls that share a scratch must be ordered on one stream; any number of states may share it. */
int dhd_mghs_workspace_bytes(const dhd_mghs_desc* desc, size_t* state_bytes, size_t* scratch_bytes);

typedef struct dhd_mghs_workspace {
  void* state;
  size_t state_bytes;
  void* scratch;
  size_t scratch_bytes;
} dhd_mghs_workspace;

/* Height argmax -> band id per pixel (height_feature_to_height_map + create_mask_3,
 * lss_heightmap.py:528-564).  height is (B*N, H, fH, fW) (probabilities or logits: only the
 * argmax matters, first index on ties as torch.argmax); height_range is H host floats;
 * mask_range = {h_min, thr1, thr2, h_max} host floats.  band[p] = 0/1/2, or 255 if the pixel's
 * height is in no band. */
int dhd_height_band(const float* height /*[dev]*/, int bn, int n_height, int fh, int fw,
                    const float* height_range /*host*/, const float* mask_range /*host*/,
                    uint8_t* band /*[dev] (B*N,fH,fW)*/, void* stream);

/* (B*N, C, fH, fW) -> (B*N, fH, fW, C) and back (the reference's feat.permute(0,1,3,4,2) at
 * lss_heightmap.py:290, made contiguous at bev_pool.py:21). */
int dhd_feat_nchw_to_nhwc(const float* src, float* dst, int bn, int c, int hw, void* stream);
int dhd_feat_nhwc_to_nchw(const float* src, float* dst, int bn, int c, int hw, void* stream);

/* Geometry + index preparation (get_ego_coor lss_heightmap.py:179-231 and
 * voxel_pooling_prepare_v2 :303-371 for all grids at once): computes every frustum point's
 * voxel in each grid with the reference's float32 operation order and truncation rule, and
 * groups the kept points by voxel (device counting sort; order inside a voxel is unspecified,
 * as with the reference's unstable argsort, :355).  `band` is the per-pixel band id from
 * dhd_height_band (ignored when n_grids == 1; may then be NULL).  The result stays in
 * `ws` and is consumed by dhd_mghs_forward / dhd_mghs_backward until the next prepare. */
int dhd_mghs_prepare(const dhd_mghs_desc* desc, const dhd_calib* calib, const uint8_t* band,
                     const dhd_mghs_workspace* ws, void* stream);

/* The lift side of MGHS.view_transform in one call and four launches (lss_heightmap.py:434-442 + :290 + the
 * preparation above): dhd_height_band (height, height_range, mask_range -> band), dhd_feat_nchw_to_nhwc (tran_feat
 * (B*N,C,fH,fW) -> feat_nhwc) and dhd_mghs_prepare.  band (B*N,fH,fW) and feat_nhwc (B*N,fH,fW,C) are outputs the
 * caller owns (feat_nhwc is an input of dhd_mghs_forward / backward).  n_grids == 1: height / band may be NULL. */
int dhd_mghs_lift(const dhd_mghs_desc* desc, const dhd_calib* calib, const float* height, int n_height,
                  const float* height_range /*host*/, const float* mask_range /*host*/, const float* feat_nchw,
                  uint8_t* band, float* feat_nhwc, const dhd_mghs_workspace* ws, void* stream);

/* Static rig (the reference's dormant accelerate / pre_compute idea, lss_heightmap.py:234-258,374-378): with the same
 * calibration from frame to frame only the height bands change, and the full-height grid 0 pools every pixel whatever
 * its band, so its whole grouping is frame-independent.  dhd_mghs_lift_static is dhd_mghs_lift for a `ws` that holds
 * the result of an earlier dhd_mghs_prepare / dhd_mghs_lift / dhd_mghs_lift_static of the SAME desc and calibration
 * (the caller guarantees that): camera matrices and the grid-0 part of the grouping are reused, only the band grids'
 * entries are counted, scanned and scattered again.  Results are identical to a full dhd_mghs_lift. */
int dhd_mghs_lift_static(const dhd_mghs_desc* desc, const dhd_calib* calib, const float* height, int n_height,
                         const float* height_range /*host*/, const float* mask_range /*host*/, const float* feat_nchw,
                         uint8_t* band, float* feat_nhwc, const dhd_mghs_workspace* ws, void* stream);

/* Pooling forward for all grids.  depth (B*N,D,fH,fW); feat_nhwc (B*N,fH,fW,C).
 * out[g] is the FINAL reference layout (B, nz_g*C, ny_g, nx_g) with channel = z*C + c, i.e. the
 * result of bev_pool.py:105 (permute) followed by lss_heightmap.py:298-299 (collapse_z); the same
 * memory viewed as (B, nz_g, C, ny_g, nx_g) serves collapse_z=False.  Every element is written
 * (zeros included); no pre-zeroing needed.  Uses the scratch part of `ws` (per-voxel sums). */
int dhd_mghs_forward(const dhd_mghs_desc* desc, const float* depth, const float* feat_nhwc,
                     float* const out[DHD_MAX_GRIDS], const dhd_mghs_workspace* ws, void* stream);

/* The two phases of dhd_mghs_forward, for callers that want to time or overlap them:
 *   gather : per-voxel sums of depth * context into the scratch's compact table (balanced over the
 *            grouped entries; instruction-bound)
 *   stream : the dense writer -- every output tensor once, zero-filled, with the table rows patched in
 *            (HBM-bound; the dominant kernel of the forward pass)
 * dhd_mghs_forward == gather then stream on the same stream. */
int dhd_mghs_forward_gather(const dhd_mghs_desc* desc, const float* depth, const float* feat_nhwc,
                            const dhd_mghs_workspace* ws, void* stream);
int dhd_mghs_forward_stream(const dhd_mghs_desc* desc, const float* depth, const float* feat_nhwc,
                            float* const out[DHD_MAX_GRIDS], const dhd_mghs_workspace* ws, void* stream);

/* Strided placement of one grid's dense tensor: element (b, z, c, y, x) lives at
 *   ptr + b*batch_stride + z*z_stride + c*channel_stride + y*nx + x      (strides in elements, multiples of 4).
 * The default layout of out[g] above is {nz*C*ny*nx, C*ny*nx, ny*nx}.  MGHS_Depth's un-collapsed
 * (B, C, 16, ny, nx) tensor (lss_heightmap.py:845, bev_feat_w_z) is written in place by giving every band
 * grid the view {C*16*ny*nx, ny*nx, 16*ny*nx} with ptr advanced to the band's first z slice: no
 * permute / cat copies. */
/* Element type of a dense pooled tensor / its gradient.  The reference's operator computes and returns float32
 * (bev_pool.py:20-21); under autocast (DHD-S.py:281, the fp16 / bf16 configurations) the first convolution behind it casts
 * that tensor to half at once.  With DHD_F16 / DHD_BF16 the writer emits what that cast would produce -- float32 sums in
 * the per-voxel table, rounded to nearest even when patched in: bit-identical to `float32 result -> .half()` -- at half
 * the bytes, and the backward reads half gradients (float32 arithmetic inside).  All views of a call share one dtype; half
 * types need the compact path (C = 64, ny % 4 == 0, nx % 4 == 0), strides that are multiples of 8 elements and 16-byte
 * aligned pointers (DHD_EUNSUPPORTED / DHD_EINVAL otherwise). */
#define DHD_F32 0
#define DHD_F16 1
#define DHD_BF16 2

typedef struct dhd_tensor_view {
  const void* ptr; /* [dev]; written through by dhd_mghs_forward_views */
  int64_t batch_stride, z_stride, channel_stride; /* in elements */
  int32_t dtype;   /* DHD_F32 / DHD_F16 / DHD_BF16 */
} dhd_tensor_view;

int dhd_mghs_forward_views(const dhd_mghs_desc* desc, const float* depth, const float* feat_nhwc,
                           const dhd_tensor_view out[DHD_MAX_GRIDS], const dhd_mghs_workspace* ws, void* stream);
/* the stream phase alone (after dhd_mghs_forward_gather), as dhd_mghs_forward_stream but through views */
int dhd_mghs_forward_stream_views(const dhd_mghs_desc* desc, const float* depth, const float* feat_nhwc,
                                  const dhd_tensor_view out[DHD_MAX_GRIDS], const dhd_mghs_workspace* ws, void* stream);
int dhd_mghs_backward_views(const dhd_mghs_desc* desc, const float* depth, const float* feat_nhwc,
                            const dhd_tensor_view out_grad[DHD_MAX_GRIDS], float* depth_grad,
                            float* feat_grad, const dhd_mghs_workspace* ws, void* stream);

/* Pooling backward.  out_grad[g] has the layout of out[g].  depth_grad (B*N,D,fH,fW) and
 * feat_grad ((B*N,fH,fW,C), or (B*N,C,fH,fW) with DHD_MGHS_FEAT_GRAD_NCHW) are fully overwritten.  Pixels outside
 * a band contribute nothing to that band's grid, matching d(tran_feat * mask), :436-442.
 * Reads the state part of `ws`, uses its scratch part as scratch (the state is not modified). */
int dhd_mghs_backward(const dhd_mghs_desc* desc, const float* depth, const float* feat_nhwc,
                      const float* const out_grad[DHD_MAX_GRIDS], float* depth_grad,
                      float* feat_grad, const dhd_mghs_workspace* ws, void* stream);

/* Introspection for parity tests and for the voxel_pooling_prepare_v2 mirror: per-point voxel
 * rank of ONE grid for every frustum point, -1 if dropped (band-independent), i.e. the map
 * point -> ranks_bev of lss_heightmap.py:329-354.  rank_map is [dev] (B*N*D*fH*fW) int32.
 * ego (optional, may be NULL) receives the (B,N,D,fH,fW,3) coordinates of get_ego_coor. */
int dhd_mghs_voxel_index(const dhd_mghs_desc* desc, const dhd_calib* calib, int grid_index,
                         int32_t* rank_map, float* ego, void* stream);

/* TEST-ONLY entry point (as the DHD_MGHS_DEBUG_SCAN_SELF_SERVE flag: part of the ABI so that the parity tests reach the product's
 * own kernels through it, of no use to a caller and free to change with the kernels).
 * Parity hook: the voxel keys the PRODUCT kernels (mghs_geom_count of dhd_mghs_prepare / dhd_mghs_lift) computed in the
 * last prepare on `ws`: keys[p] = global voxel id of point p in grid 0, keys[P + p] = in the grid of its pixel's height
 * band (P = B*N*D*fH*fW), -1 = dropped; the global id of voxel r of grid g is r + sum over g' < g of B*nz*ny*nx.  These are
 * the words the grouping is built from; dhd_mghs_voxel_index is a separate kernel over the same device functions.
 * Valid from a prepare to the next use of the scratch.  keys is [dev] 2P int32; with a single-grid plan (no band grids) the second
 * row reads -1 everywhere. */
int dhd_mghs_debug_keys(const dhd_mghs_desc* desc, const dhd_mghs_workspace* ws, int32_t* keys, void* stream);

/* Number of pooled (point, grid) pairs of the last prepare, per grid: n_kept[g] points,
 * n_intervals[g] non-empty voxels.  Reads back from `ws` right after a prepare: synchronises `stream`. */
int dhd_mghs_stats(const dhd_mghs_desc* desc, const dhd_mghs_workspace* ws, int32_t n_kept[DHD_MAX_GRIDS],
                   int32_t n_intervals[DHD_MAX_GRIDS], void* stream);

/* HBM calibration streams for roofline reporting (bench.py): move `bytes` (a multiple of 16, buf 16-byte aligned)
 * once with   pattern 0: hipMemsetAsync,   1: a linear grid-stride fill with 16-byte non-temporal stores,
 * 2: a linear grid-stride read with 16-byte non-temporal loads (one float per workgroup is written to buf's first
 * page so that the loads stay live).  The rate a measured kernel is compared against on the SAME box and run. */
int dhd_hbm_calibrate(void* buf /*[dev]*/, size_t bytes, int pattern, void* stream);

/* ------------------------------------------------------------------------------------ *
 * 3. SFA channel/spatial attention stage (models/necks/mix.py:37-59), memory-bound parts.
 *    x is (B, 2C, H, W): channels [0,C) = x_bev, [C,2C) = x_voxel.
 * ------------------------------------------------------------------------------------ */

/* fea_S = x.mean(-1).mean(-1)  (mix.py:41) -> s (B, 2C). */
int dhd_sfa_channel_mean(const float* x, float* s, int b, int c2, int hw, void* stream);

/* fea_U_1 = a1*x_bev + (1-a1)*x_voxel  (mix.py:46-50); a1 is (B,C) post-sigmoid. */
int dhd_sfa_blend1(const float* x, const float* a1, float* u, int b, int c, int hw, void* stream);

/* x_fuse = sigmoid(s2)*(a1*x_bev) + (1-sigmoid(s2))*((1-a1)*x_voxel)  (mix.py:53-58);
 * s2 = spacial_leanring(fea_U_1) pre-sigmoid, (B,C,H,W). */
int dhd_sfa_blend2(const float* x, const float* a1, const float* s2, float* out, int b, int c,
                   int hw, void* stream);

/* Backward of blend2 and blend1 together.  Inputs: go = dL/dx_fuse, gu = dL/dfea_U_1 (from the
 * 1x1-conv branch).  Outputs: gx (B,2C,H,W) = dL/dx through both blends (the mean path is added
 * by dhd_sfa_mean_backward), gs2 (B,C,H,W) = dL/ds2, ga1 (B,C) = dL/da1.
 * Two entry points because gs2 is needed before gu exists:
 *   dhd_sfa_blend2_backward: go -> gs2, and the blend2 part of gx / ga1 (ga1 zero-initialised here)
 *   dhd_sfa_blend1_backward: gu -> accumulates into gx / ga1. */
int dhd_sfa_blend2_backward(const float* x, const float* a1, const float* s2, const float* go,
                            float* gx, float* gs2, float* ga1, int b, int c, int hw, void* stream);
int dhd_sfa_blend1_backward(const float* x, const float* a1, const float* gu, float* gx,
                            float* ga1, int b, int c, int hw, void* stream);

/* gx[b,ch,:] += gs[b,ch] / hw   (backward of the channel mean). */
int dhd_sfa_mean_backward(const float* gs, float* gx, int b, int c2, int hw, void* stream);

/* ------------------------------------------------------------------------------------ *
 * 4. The whole SFA attention stage as one operator (models/necks/mix.py:8-59,
 *    channel_spatial_stage): channel mean -> fc -> blend1 -> conv1x1 -> BatchNorm -> ReLU ->
 *    conv1x1 -> BatchNorm -> sigmoid -> blend2, forward and backward.  The two 1x1 convolutions
 *    run on the matrix cores (bf16 MFMA on exact bf16 splits of the float32 operands by default,
 *    dhd_sfa_weights.gemm; single half products under dhd_sfa_weights.storage_dtype) with the
 *    blends / BatchNorm / ReLU fused into their operand paths (csrc/sfa_stage.hip, sfa_gemm_cu.h,
 *    sfa_half.h); only the conv outputs y1, y2 and one ReLU pass bit per activation are kept for backward.
 *    Supported: C == 128 or C % 256 == 0, hw % 4 == 0, 2*C*hw*4 bytes < 4 GiB (dhd_sfa_stage_supported); other shapes
 *    return DHD_EUNSUPPORTED and callers use the section-3 kernels around library convolutions.
 * ------------------------------------------------------------------------------------ */

typedef struct dhd_sfa_weights { /* [dev] float32 */
  const float* fc1_w;   /* (hidden, 2C)  channel_spatial_stage.fc[0]  (mix.py:14-19) */
  const float* fc1_b;   /* (hidden) */
  const float* fc2_w;   /* (C, hidden)   fc[2] */
  const float* fc2_b;   /* (C) */
  const float* conv1_w; /* (C, C)        spacial_leanring[0]  (mix.py:21-33) */
  const float* conv1_b; /* (C) */
  const float* bn1_w;   /* (C)           spacial_leanring[1] */
  const float* bn1_b;
  float* bn1_mean;      /* running statistics; updated in place when training, may be NULL then */
  float* bn1_var;
  const float* conv2_w; /* spacial_leanring[3] */
  const float* conv2_b;
  const float* bn2_w;   /* spacial_leanring[4] */
  const float* bn2_b;
  float* bn2_mean;
  float* bn2_var;
  int32_t hidden;       /* 2C / 16 in the reference */
  int32_t training;     /* 1: batch statistics (nn.Module.train()), 0: running statistics */
  float eps1, eps2;
  float momentum1, momentum2; /* update factor of the running statistics */
  int32_t gemm;         /* DHD_SFA_GEMM_* below */
  int64_t* bn1_batches; /* [dev] nn.BatchNorm2d.num_batches_tracked of the two layers, or NULL: incremented by one in a */
  int64_t* bn2_batches; /*       training-mode forward (instead of two one-element launches by the caller)              */
  int32_t io_dtype;     /* DHD_F32 / DHD_F16 / DHD_BF16: element type of `out`, `gout` and `gx` (ABI 3).  x, the parameters and every
                           tensor the operator keeps are float32.  Half types are for a caller inside an autocast region: it hands a
                           half x (which it widens once for the operator), gets the stage's result rounded to nearest even -- what the
                           next convolution's cast of a float32 result would produce -- and passes the half gradient straight back */
  int32_t storage_dtype; /* ABI 4.  DHD_F32 (0): as above.  DHD_F16 / DHD_BF16 (must equal io_dtype; C == 128 or 256 and hw % 8 == 0,
                           dhd_sfa_stage_half_storage_supported): HALF STORAGE -- x is read in that type (no widened copy), and every
                           (B,C,H,W) tensor the stage keeps or passes between its kernels (y1, y2, g2, g1, du) is stored in it, as the
                           reference's own formulation does under autocast (mix.py:37-59 with DHD-S.py:281); each 1x1 convolution is ONE
                           half product per a*b accumulated in float32 (`gemm` is ignored); prologues, BatchNorm statistics (of the
                           stored values), coefficient tables, parameters and parameter gradients stay float32.  `saved` / `scratch`
                           sizes: dhd_sfa_stage_workspace_bytes */
} dhd_sfa_weights;

typedef struct dhd_sfa_grads { /* [dev] float32 outputs, shapes as in dhd_sfa_weights, overwritten */
  float *fc1_w, *fc1_b, *fc2_w, *fc2_b;
  float *conv1_w, *conv1_b, *bn1_w, *bn1_b;
  float *conv2_w, *conv2_b, *bn2_w, *bn2_b;
} dhd_sfa_grads;

int dhd_sfa_stage_supported(int c, int hw);
int dhd_sfa_stage_half_storage_supported(int c, int hw);   /* storage_dtype = DHD_F16 / DHD_BF16 (ABI 4) */
/* dhd_sfa_weights.gemm: how the stage's C x C GEMMs are computed in THIS call (forward and backward of one stage must
 * pass the same value; nothing is process-wide).  Every float32 operand is cut into bfloat16 parts (round-to-nearest-even,
 * exact: h + m + l == x) for the bf16 MFMA:
 *   DHD_SFA_GEMM_DEFAULT (0)  = DHD_SFA_GEMM_BF16X3
 *   DHD_SFA_GEMM_BF16X6 (1)   three parts, six products -- float32-level accuracy (dropped terms < 2^-25 |ab|)
 *   DHD_SFA_GEMM_F32 (2)      f32 MFMA (v_mfma_f32_32x32x2_f32), a plain float32 fma chain
 *   DHD_SFA_GEMM_BF16X3 (3)   two parts per operand, three products ah*bh + ah*bm + am*bh per a*b (error <= 3 * 2^-18
 *                             |ab| per product; stage output within ~2e-5 of float64 at full size, inside the path's 1e-3 bar)
 * Kernel forms (resident weights in LDS, or weights streamed through LDS per pixel tile for channel counts the resident form
 * does not cover) are chosen by the library and do not change results within a mode.  The weight-gradient GEMMs follow the
 * same precision. */
#define DHD_SFA_GEMM_DEFAULT 0
#define DHD_SFA_GEMM_BF16X6 1
#define DHD_SFA_GEMM_F32 2
#define DHD_SFA_GEMM_BF16X3 3
/* `saved` carries forward state to backward (a1, BatchNorm batch statistics, y1, y2);
 * `scratch` is reusable between calls on one stream.  0 if the shape is unsupported. */
size_t dhd_sfa_stage_saved_bytes(int b, int c, int hw, int hidden);
size_t dhd_sfa_stage_scratch_bytes(int b, int c, int hw, int hidden);
/* The same two sizes for a given dhd_sfa_weights.storage_dtype (ABI 4; DHD_F32 gives the values above). */
int dhd_sfa_stage_workspace_bytes(int b, int c, int hw, int hidden, int storage_dtype, size_t* saved_bytes, size_t* scratch_bytes);

/* x (B,2C,H,W) -> out (B,C,H,W) = x_fuse of mix.py:58.  x is float32, or w->storage_dtype when that is a half type. */
int dhd_sfa_stage_forward(const void* x, const dhd_sfa_weights* w, void* out /* w->io_dtype */, void* saved,
                          void* scratch, int b, int c, int hw, void* stream);
/* gout (B,C,H,W) -> gx (B,2C,H,W) and every parameter gradient. */
int dhd_sfa_stage_backward(const void* x, const dhd_sfa_weights* w, const void* saved,
                           const void* gout, void* gx /* both w->io_dtype */, const dhd_sfa_grads* grads, void* scratch,
                           int b, int c, int hw, void* stream);

/* The same operator under nn.SyncBatchNorm (core/hook/syncbncontrol.py:18-32 converts the stage's two BatchNorms when
 * DHD-L.py:308-311 turns SyncbnControlHook on): forward and backward are cut at the points where BatchNorm needs sums over
 * the whole (cross-rank) batch.  Call phase 0, 1, 2 in order; between two phases all-reduce (SUM) the (2C + 1) float64 values
 * of sync_sums over the ranks -- a phase that ends at a statistics point writes this rank's [sum][C] | [second sum][C] |
 * count there, the next phase reads the reduced vector from the same place:
 *   forward  0: channel mean, fc, conv1            -> sums of (y1 - bias), (y1 - bias)^2
 *            1: BatchNorm-1 statistics, ReLU, conv2 -> sums of (y2 - bias), (y2 - bias)^2
 *            2: BatchNorm-2 statistics, sigmoid, blend -> out          (`out` may be NULL in phases 0 and 1)
 *   backward 0: blend backward                     -> sums of g2, g2 (y2 - mean2)
 *            1: BatchNorm-2 backward, dW2, dgrad 2  -> sums of g1, g1 (y1 - mean1)
 *            2: BatchNorm-1 backward, dW1, dgrad 1, fc backward -> gx  (`gx` may be NULL in phases 0 and 1)
 * Running statistics are updated from the global sums (identical on every rank); dgamma / dbeta / the convolution-bias
 * gradients are this rank's contributions (the caller's DDP averages parameter gradients), exactly as torch's
 * SyncBatchNorm.  Training mode and the bf16 GEMM precisions only (DHD_EUNSUPPORTED otherwise).  With one rank and no
 * all-reduce the three phases reproduce dhd_sfa_stage_forward / backward. */
int dhd_sfa_stage_forward_phase(const void* x, const dhd_sfa_weights* w, void* out, void* saved, void* scratch,
                                int b, int c, int hw, int phase, double* sync_sums /*[dev] 2C+1*/, void* stream);
int dhd_sfa_stage_backward_phase(const void* x, const dhd_sfa_weights* w, const void* saved, const void* gout,
                                 void* gx, const dhd_sfa_grads* grads, void* scratch, int b, int c, int hw,
                                 int phase, double* sync_sums /*[dev] 2C+1*/, void* stream);

/* ------------------------------------------------------------------------------------ *
 * 5. Occupancy-head losses (models/dense_heads/occ_head.py:102-139, predictor.loss): the
 *    class-balanced camera-masked cross entropy (models/losses/cross_entropy_loss.py:12-63,
 *    weight = mask_camera, avg_factor = sum_i #valid(i) * w_i), sem_scal_loss_with_mask
 *    (models/losses/semkitti_loss.py:171-226) and geo_scal_loss_with_mask (:136-169) in two
 *    streaming passes over the (n_voxels, 18) logits -- the caller row after the hot path.
 *    labels: uint8, ignore_index (255) = unknown; mask: uint8 camera visibility; class_weight (18).
 *    losses / grad_losses: [dev] float[3] = {cross entropy, sem scal, geo scal}, before the
 *    head's weight_ce / weight_sem / weight_geo factors.  n_classes must be 18.
 * ------------------------------------------------------------------------------------ */
size_t dhd_occ_loss_workspace_bytes(void);
/* workspace carries the global sums from forward to backward. */
int dhd_occ_loss_forward(const float* logits, const uint8_t* labels, const uint8_t* mask,
                         const float* class_weight, int64_t n_voxels, int n_classes, int ignore_index,
                         int non_empty_idx, float* losses, void* workspace, void* stream);
/* grad_logits (n_voxels, 18) = d(sum_k grad_losses[k] * losses[k]) / d logits, overwritten. */
int dhd_occ_loss_backward(const float* logits, const uint8_t* labels, const uint8_t* mask,
                          const float* class_weight, int64_t n_voxels, int n_classes, int ignore_index,
                          int non_empty_idx, const float* grad_losses, const void* workspace,
                          float* grad_logits, void* stream);

/* Evaluation side: pred[v] = argmax_k logits[v,k] (predictor.get_occ, occ_head.py:141-153) and
 * hist[t*18 + pred] += 1 for voxels with mask != 0 (mask may be NULL) and label t < 18
 * (Metric_mIoU.hist_info, core/evaluation/occ_metrics.py:79-104).  hist is [dev] int64[324],
 * ACCUMULATED into (zero it once per evaluation); pred or hist may be NULL. */
int dhd_occ_argmax_hist(const float* logits, const uint8_t* labels, const uint8_t* mask,
                        int64_t n_voxels, int n_classes, uint8_t* pred, int64_t* hist, void* stream);

/* ------------------------------------------------------------------------------------ *
 * 6. Height / depth supervision of the view transformer (row a16):
 *    MGHS.get_height_loss (models/necks/lss_heightmap.py:595-622), its label builders
 *    get_downsampled_gt_depth / _height (:625-667, :670-701; non-SID binning) and
 *    MGHS_Depth.get_depth_and_height_loss (:859-897).
 * ------------------------------------------------------------------------------------ */

/* gt_depth, gt_height: (bn, fh*downsample, fw*downsample) sparse maps (0 = no LiDAR return).
 * Per feature pixel: m = min over the window of the non-zero values (1e5 if none),
 * g = (m - offset) / step in float32, bin = trunc(g) if 0 <= g < bins + 1 else 0.  Bin 0 means
 * "no label"; bin k >= 1 is channel k-1 of the reference's one-hot matrix.  Outputs (bn*fh*fw) int16.
 * offset/step: depth (d0 - dstep, dstep) of grid_config['depth'] (:649-651), height
 * (height_range[0], height_interval) (:691). */
int dhd_sparse_bin_labels(const float* gt_depth, const float* gt_height, int bn, int fh, int fw,
                          int downsample, float depth_offset, float depth_step, int depth_bins,
                          float height_offset, float height_step, int height_bins,
                          int16_t* depth_bin, int16_t* height_bin, void* stream);

/* The same with spacing-increasing depth bins (MGHS(sid=True), lss_heightmap.py:655-660):
 *   g = (log(m) - log_d0) * (depth_bins - 1) / log_ratio + 1,   log_d0 = log(d0), log_ratio = log((d1 - 1) / d0),
 * both as the float32 values torch computes on the host; the height labels are binned linearly as above. */
int dhd_sparse_bin_labels_sid(const float* gt_depth, const float* gt_height, int bn, int fh, int fw,
                              int downsample, float log_d0, float log_ratio, int depth_bins,
                              float height_offset, float height_step, int height_bins,
                              int16_t* depth_bin, int16_t* height_bin, void* stream);

size_t dhd_bin_bce_workspace_bytes(void);
/* loss[0] = weight * sum_{pixels with fg_bin > 0} sum_c BCE(pred[b,c,pixel], [c == bin-1]) / max(1, n_fg)
 * (F.binary_cross_entropy, logs clamped at -100; :612-622).  pred is (bn, c, hw), the softmax map in
 * its native layout.  workspace carries n_fg to backward. */
int dhd_bin_bce_forward(const float* pred, const int16_t* bin, const int16_t* fg_bin, int bn, int c,
                        int hw, float weight, float* loss, void* workspace, void* stream);
/* grad_pred (bn, c, hw) = grad_loss[0] * d loss / d pred, overwritten (zeros outside the foreground). */
int dhd_bin_bce_backward(const float* pred, const int16_t* bin, const int16_t* fg_bin, int bn, int c,
                         int hw, float weight, const float* grad_loss, const void* workspace,
                         float* grad_pred, void* stream);

/* ------------------------------------------------------------------------------------ *
 * 7. Deformable-convolution sampling for HeightNet / DepthNet's DCN layer
 *    (models/necks/depthnet.py:225-236, :466-477 -> mmcv-full 1.5.3 DeformConv2dPack, v1,
 *    stride 1, one deformable group).  x (B,C,H,W), offset (B, 2*k*k, H, W) with channel 2t =
 *    dy and 2t+1 = dx of tap t = ky*k + kx;  col (B, C*k*k, H*W), row c*k*k + t: the im2col
 *    matrix of the sampled values, to be multiplied by the layer's weight.
 * ------------------------------------------------------------------------------------ */
int dhd_deform_im2col(const float* x, const float* offset, float* col, int b, int c, int h, int w,
                      int k, int pad, int dil, void* stream);
/* dcol -> dx (B,C,H,W) and doffset (B, 2*k*k, H, W), both overwritten.  H*W*4 must be <= 48 KiB. */
int dhd_deform_col2im(const float* dcol, const float* x, const float* offset, float* dx,
                      float* doffset, int b, int c, int h, int w, int k, int pad, int dil,
                      void* stream);
/* ABI 5.  The same two operators with the column matrix in `col_dtype` (DHD_F32 / DHD_F16 / DHD_BF16: under autocast the GEMM
 * behind the sampling runs in half, so the columns are written, and their gradient read, in half; offset, doffset and the
 * arithmetic stay float32), x read -- and dx written -- in `x_dtype`: float32 or col_dtype (dense NCHW: the sampling kernels put
 * consecutive cells on consecutive lanes; a channels_last x measured 5-7x slower and is converted by the caller); and col2im in
 * its GATHER form: the bilinear corner entries of an image are grouped by the cell they
 * land in (they are shared by all channels), then every cell sums its own list from LDS-staged dcol rows -- no atomics
 * (csrc/deform.hip).  `workspace`: dhd_deform_col2im_workspace_bytes(b, h, w, k) bytes of device scratch, 16-byte aligned,
 * owned by the caller.  Shapes the gather form does not take (dhd_deform_col2im_gather_supported == 0: k*k*h*w elements of
 * `col_dtype` must fit 144 KiB of LDS) return DHD_EUNSUPPORTED; dhd_deform_col2im covers them in float32. */
int dhd_deform_im2col_t(const void* x, int x_dtype, const float* offset, void* col, int col_dtype,
                        int b, int c, int h, int w, int k, int pad, int dil, void* stream);
size_t dhd_deform_col2im_workspace_bytes(int b, int h, int w, int k);
int dhd_deform_col2im_gather_supported(int col_dtype, int h, int w, int k);
int dhd_deform_col2im_t(const void* dcol, int col_dtype, const void* x, int x_dtype, const float* offset,
                        void* dx, float* doffset, int b, int c, int h, int w, int k, int pad, int dil,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ABI 5.  The element-wise head of MGHS.forward / MGHS_Depth.forward (lss_heightmap.py:484-489, :829-834) in one launch each
 * way (csrc/mghs_softmax.hip):
 *   xd  (bn, ct >= d + c, hw)  depth_net's output in `xd_dtype`, dense NCHW (xd_nhwc = 0) or channels_last (1)
 *   hl  (bn, ht >= h_bins, hw) height_net's output, likewise; NULL: no height branch (then height / band are not written)
 *   -> depth (bn, d, hw) = softmax over xd's first d channels, feat (bn, c, hw) = its next c channels, height (bn, h_bins, hw)
 *      = softmax over hl's first h_bins channels, all float32 dense NCHW; band (bn, hw) uint8 = the height band of
 *      argmax(height) (0 / 1 / 2, 255 none; NULL: not wanted), from the float32 height_range[h_bins] / mask_range[4] as
 *      dhd_height_band.  The softmax is torch's for this shape, operation for operation (max, sum of expf, correctly rounded
 *      division, bins in order): the same bits as `x.float().softmax(1)` on the GPU.
 * backward: g_depth / g_feat / g_height (float32 NCHW, any of them NULL = zero) -> g_xd (bn, ct, hw) in xd's dtype and layout
 * (softmax Jacobian on the first d channels, g_feat on the next c, zeros beyond) and g_hl (bn, ht, hw) likewise; either output
 * may be NULL.  Nothing is allocated; the caller's stream. */
int dhd_mghs_softmax_forward(const void* xd, int xd_dtype, int xd_nhwc, int ct, const void* hl, int hl_dtype,
                             int hl_nhwc, int ht, int bn, int hw, int d, int c, int h_bins,
                             const float* height_range, const float* mask_range, float* depth, float* feat,
                             float* height, uint8_t* band, void* stream);
int dhd_mghs_softmax_backward(const float* g_depth, const float* g_feat, const float* g_height,
                              const float* depth, const float* height, int bn, int hw, int d, int c,
                              int h_bins, void* g_xd, int xd_dtype, int xd_nhwc, int ct, void* g_hl,
                              int hl_dtype, int hl_nhwc, int ht, void* stream);

/* LiDAR points -> sparse depth / height maps, all cameras of a sample at once
 * (datasets/pipelines/loading_new.py:35-99, PointToMultiViewDepthandHeight.points2depthmap /
 * points2heightmap).  points: (n_cams, n_points, 4) = (u, v, d, h) in augmented image coordinates;
 * maps: (n_cams, height/downsample, width/downsample), 0 where no point lands.  A pixel keeps the
 * point with the smallest float32 key  pixel_rank + d/100  (ties: lowest index).  zbuffer: scratch of
 * n_cams * h * w * 8 bytes.  Points outside the image or outside [depth_lo, depth_hi) are ignored. */
int dhd_points_to_maps(const float* points, int n_cams, int n_points, int height, int width,
                       int downsample, float depth_lo, float depth_hi, float* depth_map,
                       float* height_map, void* zbuffer, void* stream);

/* Temporal-stereo cost volume of DepthNet (models/necks/depthnet.py:307-361): prev / curr are the
 * stereo features of the adjacent and the current frame in NHWC (bn, h, w, c); grid (bn, n_depth*h, w, 2)
 * holds the normalised sampling positions of F.grid_sample(align_corners=True, padding zeros), as the
 * reference's gen_grid produces them.  cost = sum_c |curr - sample(prev)|, + bias where the sample of
 * channel flag_channel (the reference's last group, C-4) is exactly 0; out (bn, n_depth, h, w) =
 * softmax over the depth hypotheses of -cost.  c % 4 == 0, c <= 1024, n_depth <= 256.  No gradient. */
int dhd_stereo_cost_volume(const float* prev_nhwc, const float* curr_nhwc, const float* grid, int bn,
                           int c, int h, int w, int n_depth, float bias, int flag_channel, float* out,
                           void* stream);

/* ------------------------------------------------------------------------------------ *
 * 8. Weight EMA of the training loop (projects/mmdet3d_plugin/core/hook/ema.py:48-59,
 *    ModelEMA.update: for every floating-point entry of the state dict  v *= d; v += (1-d)*m).
 *    One launch over a chunk table: chunk i updates len[i] float32 values at device address
 *    ema_addr[i] from those at model_addr[i]  (ema = fl(fl(ema*decay) + fl(one_minus_decay*model)),
 *    the reference's two roundings).  The three tables live in device memory; the caller splits its
 *    tensors into chunks (any length > 0; 16-byte aligned starts take the vector path) and passes
 *    one_minus_decay = (float)(1.0 - (double)decay) as the reference computes it.
 * ------------------------------------------------------------------------------------ */
int dhd_ema_update(const uint64_t* ema_addr, const uint64_t* model_addr, const int* len, int n_chunks,
                   float decay, float one_minus_decay, void* stream);
/* The same with the two factors in device memory, decay_pair = {decay, one_minus_decay}: for a launch that is captured into a
 * HIP graph (kernel arguments are frozen at capture; the reference's decay ramps with the update count, ema.py:29,55), the
 * host refreshes the pair before every replay. */
int dhd_ema_update_dev(const uint64_t* ema_addr, const uint64_t* model_addr, const int* len, int n_chunks,
                       const float* decay_pair /*[dev] float[2]*/, void* stream);

/* ------------------------------------------------------------------------------------ *
 * 9. Training-mode BatchNorm2d of the dense callers (torch.nn.BatchNorm2d semantics: biased batch
 *    variance for the normalisation, unbiased for running_var, running = (1-factor)*running +
 *    factor*batch).  x, y, grad_y, grad_x: (n, c, hw) NCHW in `dtype` (0 float32, 1 float16,
 *    2 bfloat16); parameters, statistics and gradients of the parameters float32; gamma / beta /
 *    running_* / dgamma / dbeta may be NULL.  hw must be a multiple of 4 (float32) or 8 elements and
 *    n*c <= 65535 (dhd_bn_supported).  save_mean / save_rstd carry the batch statistics to backward.
 * ------------------------------------------------------------------------------------ */
int dhd_bn_supported(int dtype, int n, int c, int hw);
size_t dhd_bn_workspace_bytes(int n, int c, int hw);
int dhd_bn_train_forward(const void* x, int dtype, int n, int c, int hw, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, float factor,
                         float eps, void* y, float* save_mean, float* save_rstd, void* workspace,
                         void* stream);
int dhd_bn_train_backward(const void* x, const void* grad_y, int dtype, int n, int c, int hw,
                          const float* gamma, const float* save_mean, const float* save_rstd,
                          void* grad_x, float* dgamma, float* dbeta, void* workspace, void* stream);

/* 9b. The same for channels_last activations: x, y, residual and the gradients are row-major matrices (rows = n*hw, c) --
 *     what torch.channels_last tensors are in memory -- with c a multiple of 4 (float32) or 8 elements.  `flags` fuses the
 *     element-wise operators that follow the normalisation in the dense callers (resnet.py's Bottleneck, mmcv's ConvModule):
 *       DHD_BN_RELU  y = max(0, bn(x));   DHD_BN_ADD  y = max(0, bn(x) + residual)   (ADD implies the ReLU).
 *     Backward takes the gradient with respect to that y: its ReLU mask is recomputed from x and save_affine (= the forward's
 *     per-channel scale | shift, [2*c]; RELU) or read from the saved output y (ADD), and with ADD grad_residual (may be NULL)
 *     receives the masked gradient.  Without flags y / residual / grad_residual may be NULL in backward.
 *     Workspace: dhd_bn_nhwc_workspace_bytes. */
#define DHD_BN_RELU 1
#define DHD_BN_ADD 2
int dhd_bn_nhwc_supported(int dtype, long rows, int c);
size_t dhd_bn_nhwc_workspace_bytes(long rows, int c);
int dhd_bn_nhwc_train_forward(const void* x, const void* residual, int dtype, long rows, int c, int flags,
                              const float* gamma, const float* beta, float* running_mean, float* running_var,
                              float factor, float eps, void* y, float* save_mean, float* save_rstd,
                              float* save_affine, void* workspace, void* stream);
int dhd_bn_nhwc_train_backward(const void* x, const void* y, const void* grad_y, int dtype, long rows, int c, int flags,
                               const float* gamma, const float* save_mean, const float* save_rstd,
                               const float* save_affine, void* grad_x, void* grad_residual, float* dgamma,
                               float* dbeta, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------ *
 * 10. Bilinear up-sampling with align_corners = True (nn.Upsample of necks/lss_fpn.py:27,43 and
 *     backbones/unet.py:86), forward and backward, for the dense callers.  layout 0: (n, c, h, w) NCHW,
 *     1: (n, h, w, c) channels_last with c a multiple of 4 (float32) / 8 elements; dtype as in section 9;
 *     hout >= hin, wout >= win, factors up to 8.  Index arithmetic as torch's UpSample.cuh (float32);
 *     backward is a gather: one writer per element, float32 accumulation, deterministic.
 * ------------------------------------------------------------------------------------ */
int dhd_upsample_bilinear_supported(int dtype, int layout, int n, int c, int hin, int win, int hout, int wout);
int dhd_upsample_bilinear_forward(const void* x, int dtype, int layout, int n, int c, int hin, int win,
                                  int hout, int wout, void* y, void* stream);
int dhd_upsample_bilinear_backward(const void* grad_y, int dtype, int layout, int n, int c, int hin, int win,
                                   int hout, int wout, void* grad_x, void* stream);

/* ------------------------------------------------------------------------------------ *
 * 11. Layout conversion at the boundary between the NCHW operators above and dense stacks that run in
 *     channels_last: batched transpose in[b][rows][cols] -> out[b][cols][rows] of 2- or 4-byte elements
 *     (NCHW -> NHWC: rows = c, cols = h*w; NHWC -> NCHW: rows = h*w, cols = c).  in and out must not overlap.
 * ------------------------------------------------------------------------------------ */
int dhd_transpose_batched(const void* in, void* out, int elem_bytes, long batch, int rows, int cols, void* stream);

/* ------------------------------------------------------------------------------------ *
 * 12. Shifted-window partition / reverse of the Swin backbone (DHD-L; backbones/swin.py:448-513: F.pad + torch.roll +
 *     permute-reshape, and their inverses) as one row gather each way.  Token map (b, h, w, c) <-> windows
 *     (b, nh*nw, window*window, c) with nh = ceil(h / window), nw = ceil(w / window); reverse = 0: partition (padding rows
 *     are zeros), 1: reverse (padding rows dropped).  c a multiple of 8; in / out dtypes as in section 9 and may differ
 *     (float32 LayerNorm output -> autocast dtype; half gradients -> float32).  Each direction is the other's transpose.
 * ------------------------------------------------------------------------------------ */
int dhd_window_rows(const void* in, void* out, int in_dtype, int out_dtype, int b, int h, int w, int c, int window,
                    int shift, int reverse, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DHD_AMD_H */
