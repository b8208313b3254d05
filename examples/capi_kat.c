/* The operator seam from plain C: no Python, no torch types -- the caller owns HIP memory and the stream, exactly what a
 * binding written in the reference's own extension style (ops/bev_pool_v2/src/bev_pool.cpp:30-57,74-104) would pass.
 * Runs the reference's in-file known-answer test (ops/bev_pool_v2/bev_pool.py:163-194): loss = sum(out) = 4.4,
 * depth.grad = [2,2,0,0,2,0,2,0], feat.grad = [1,1,.4,.4,.8,.8,0,0].
 *
 *   hipcc examples/capi_kat.c -Iinclude -Ldhd_amd/csrc -ldhd_amd -Wl,-rpath,$PWD/dhd_amd/csrc -o /tmp/capi_kat && /tmp/capi_kat
 * (hipcc only for the HIP runtime's include and link paths; there is no device code in this file.)
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dhd_amd.h"

#define CHECK(call)                                                        \
  do {                                                                     \
    hipError_t e_ = (call);                                                \
    if (e_ != hipSuccess) {                                                \
      fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));           \
      return 2;                                                            \
    }                                                                      \
  } while (0)

static void* to_dev(const void* host, size_t bytes) {
  void* d = NULL;
  if (hipMalloc(&d, bytes ? bytes : 4) != hipSuccess) return NULL;
  if (bytes && hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}

int main(void) {
  if (dhd_abi_version() != DHD_ABI_VERSION) {
    fprintf(stderr, "libdhd_amd.so ABI %d, header %d\n", dhd_abi_version(), DHD_ABI_VERSION);
    return 2;
  }
  /* depth (B,N,D,fH,fW) = (1,1,2,2,2), feat (B,N,fH,fW,C) = (1,1,2,2,2), out (B,Dz,Dy,Dx,C) = (1,1,2,2,2) */
  const float depth[8] = {0.3f, 0.4f, 0.2f, 0.1f, 0.7f, 0.6f, 0.8f, 0.9f};
  float feat[8];
  for (int i = 0; i < 8; ++i) feat[i] = 1.0f;
  const int32_t ranks_depth[4] = {0, 4, 1, 6}, ranks_feat[4] = {0, 0, 1, 2}, ranks_bev[4] = {0, 0, 1, 1};
  const int32_t starts[2] = {0, 2}, lengths[2] = {2, 2};
  const int c = 2, n_pixels = 4;
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  float *d_depth = to_dev(depth, sizeof depth), *d_feat = to_dev(feat, sizeof feat);
  int32_t *d_rd = to_dev(ranks_depth, sizeof ranks_depth), *d_rf = to_dev(ranks_feat, sizeof ranks_feat);
  int32_t *d_rb = to_dev(ranks_bev, sizeof ranks_bev), *d_st = to_dev(starts, sizeof starts), *d_ln = to_dev(lengths, sizeof lengths);
  float *d_out = NULL, *d_og = NULL, *d_dg = NULL, *d_fg = NULL;
  CHECK(hipMalloc((void**)&d_out, 8 * 4)); CHECK(hipMalloc((void**)&d_og, 8 * 4));
  CHECK(hipMalloc((void**)&d_dg, 8 * 4)); CHECK(hipMalloc((void**)&d_fg, 8 * 4));
  if (!d_depth || !d_feat || !d_rd || !d_rf || !d_rb || !d_st || !d_ln) return 2;

  /* forward: the caller zeroes `out` (bev_pool.py:27); note lengths BEFORE starts, as in bev_pool.cpp:30-39 */
  CHECK(hipMemsetAsync(d_out, 0, 8 * 4, st));
  int rc = dhd_bev_pool_v2_forward(d_depth, d_feat, d_out, d_rd, d_rf, d_rb, d_ln, d_st, c, 2, st);
  if (rc) { fprintf(stderr, "dhd_bev_pool_v2_forward: %d\n", rc); return 1; }
  float out[8];
  CHECK(hipMemcpyAsync(out, d_out, sizeof out, hipMemcpyDeviceToHost, st));
  CHECK(hipStreamSynchronize(st));
  float loss = 0.f;
  for (int i = 0; i < 8; ++i) loss += out[i];

  /* backward of loss = sum(out): out_grad = 1; the point lists regrouped by feature pixel on the device */
  float ones[8];
  for (int i = 0; i < 8; ++i) ones[i] = 1.0f;
  CHECK(hipMemcpyAsync(d_og, ones, sizeof ones, hipMemcpyHostToDevice, st));
  int32_t *d_rd2, *d_rf2, *d_rb2, *d_st2, *d_ln2;
  void* d_scratch;
  const size_t sb = dhd_bev_pool_v2_regroup_scratch_bytes(4, n_pixels);
  CHECK(hipMalloc((void**)&d_rd2, 16)); CHECK(hipMalloc((void**)&d_rf2, 16)); CHECK(hipMalloc((void**)&d_rb2, 16));
  CHECK(hipMalloc((void**)&d_st2, n_pixels * 4)); CHECK(hipMalloc((void**)&d_ln2, n_pixels * 4)); CHECK(hipMalloc(&d_scratch, sb));
  rc = dhd_bev_pool_v2_regroup(d_rd, d_rf, d_rb, 4, n_pixels, d_rd2, d_rf2, d_rb2, d_st2, d_ln2, d_scratch, sb, st);
  if (rc) { fprintf(stderr, "dhd_bev_pool_v2_regroup: %d\n", rc); return 1; }
  CHECK(hipMemsetAsync(d_dg, 0, 8 * 4, st)); CHECK(hipMemsetAsync(d_fg, 0, 8 * 4, st));
  rc = dhd_bev_pool_v2_backward(d_og, d_dg, d_fg, d_depth, d_feat, d_rd2, d_rf2, d_rb2, d_ln2, d_st2, c, n_pixels, st);
  if (rc) { fprintf(stderr, "dhd_bev_pool_v2_backward: %d\n", rc); return 1; }
  float dg[8], fg[8];
  CHECK(hipMemcpyAsync(dg, d_dg, sizeof dg, hipMemcpyDeviceToHost, st));
  CHECK(hipMemcpyAsync(fg, d_fg, sizeof fg, hipMemcpyDeviceToHost, st));
  CHECK(hipStreamSynchronize(st));

  const float dg_ref[8] = {2, 2, 0, 0, 2, 0, 2, 0}, fg_ref[8] = {1, 1, .4f, .4f, .8f, .8f, 0, 0};
  int bad = fabsf(loss - 4.4f) > 1e-6f;
  for (int i = 0; i < 8; ++i) bad |= fabsf(dg[i] - dg_ref[i]) > 1e-6f || fabsf(fg[i] - fg_ref[i]) > 1e-6f;
  /* argument errors come back as codes, before any launch */
  bad |= dhd_bev_pool_v2_forward(NULL, d_feat, d_out, d_rd, d_rf, d_rb, d_ln, d_st, c, 2, st) != DHD_EINVAL;
  bad |= dhd_bev_pool_v2_forward(d_depth, d_feat, d_out, d_rd, d_rf, d_rb, d_ln, d_st, 0, 2, st) != DHD_EINVAL;
  printf("loss %.7f depth_grad %g %g %g %g %g %g %g %g feat_grad %g %g %g %g %g %g %g %g -> %s\n", loss, dg[0], dg[1], dg[2], dg[3],
         dg[4], dg[5], dg[6], dg[7], fg[0], fg[1], fg[2], fg[3], fg[4], fg[5], fg[6], fg[7], bad ? "MISMATCH" : "KAT ok");
  return bad ? 1 : 0;
}
