// Stand-alone check + timing of the half-storage GEMM kernels (dhd_amd/csrc/sfa_half.h: pw_gemm_cuh_kernel, pw_wgrad_h_kernel) at
// the SFA stage's size (B = 4, C = 256, HW = 200 x 200): results against a float64-accumulating reference on the same rounded
// operands, time per launch by HIP events, and the time of a blend pass with the same HBM traffic (the floor).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off experiments/gemm_cuh_bench.hip -o experiments/build/gemm_cuh_bench
// run  : experiments/build/gemm_cuh_bench [B] [HW] [C = 256 | 128]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../dhd_amd/csrc/sfa_stage.hip"   // the product translation unit (launchers live in its unnamed namespace)

using namespace dhd_sfa;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <class TS> __device__ float to_f(TS v) { return (float)v; }

template <class TS>
__global__ void fill_kernel(TS* p, size_t n, unsigned seed, float scale, float shift) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    p[i] = (TS)(((float)(h & 0xffffff) / 16777216.f * 2.f - 1.f) * scale + shift);
  }
}

template <class TS> __global__ void pack_kernel(const float* w, int transpose, u32x4* wp, int c) {
  cuh_pack_weight<TS>(w, transpose, wp, c, blockIdx.x * blockDim.x + threadIdx.x);
}

// y[b][co][p] = sum_k TS(W[co][k]) TS(act(c0 in0 + c1 in1 + c2))  (double accumulation), channels [co0, co0 + nco)
template <class TS>
__global__ void ref_kernel(const TS* in0, const TS* in1, size_t bs, const float* coef, const float* w, int transpose, const float* bias, int relu,
                           double* y, int c, int hw, int co0, int nco) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (p >= hw) return;
  for (int co = co0; co < co0 + nco; ++co) {
    double acc = 0.0;
    for (int k = 0; k < c; ++k) {
      float v = fmaf(coef[(size_t)b * 3 * c + k], (float)in0[(size_t)b * bs + (size_t)k * hw + p], coef[(size_t)b * 3 * c + 2 * c + k]);
      if (in1) v = fmaf(coef[(size_t)b * 3 * c + c + k], (float)in1[(size_t)b * bs + (size_t)k * hw + p], v);
      if (relu) v = fmaxf(v, 0.f);
      const float wv = transpose ? w[(size_t)k * c + co] : w[(size_t)co * c + k];
      acc += (double)(float)(TS)wv * (double)(float)(TS)v;
    }
    y[((size_t)b * nco + (co - co0)) * hw + p] = acc + (bias ? (double)bias[co] : 0.0);
  }
}

// G[co][ci] for co in [co0, co0 + nco): sum_{b,p} TS(A(co,p)) TS(B(ci,p))
template <class TS>
__global__ void ref_wgrad_kernel(const TS* a0, const TS* a1, const float* acoef, size_t abs_, const TS* b0, const TS* b1, const float* bcoef,
                                 size_t bbs, int brelu, double* g, int c, int hw, int nb, int co0) {
  const int ci = blockIdx.x * blockDim.x + threadIdx.x, co = co0 + blockIdx.y;
  if (ci >= c) return;
  double acc = 0.0;
  for (int b = 0; b < nb; ++b)
    for (int p = 0; p < hw; ++p) {
      float va = fmaf(acoef[(size_t)b * 3 * c + co], (float)a0[(size_t)b * abs_ + (size_t)co * hw + p], acoef[(size_t)b * 3 * c + 2 * c + co]);
      va = fmaf(acoef[(size_t)b * 3 * c + c + co], (float)a1[(size_t)b * abs_ + (size_t)co * hw + p], va);
      float vb = fmaf(bcoef[(size_t)b * 3 * c + ci], (float)b0[(size_t)b * bbs + (size_t)ci * hw + p], bcoef[(size_t)b * 3 * c + 2 * c + ci]);
      if (b1) vb = fmaf(bcoef[(size_t)b * 3 * c + c + ci], (float)b1[(size_t)b * bbs + (size_t)ci * hw + p], vb);
      if (brelu) vb = fmaxf(vb, 0.f);
      acc += (double)(float)(TS)va * (double)(float)(TS)vb;
    }
  g[(size_t)blockIdx.y * c + ci] = acc;
}

__global__ void blend_kernel(const u32x4* in0, const u32x4* in1, u32x4* out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const u32x4 a = __builtin_nontemporal_load(in0 + i), b = __builtin_nontemporal_load(in1 + i);
    __builtin_nontemporal_store(a ^ b, out + i);
  }
}

static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

template <class K>
static double time_kernel(K launch, int reps = 15) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); launch();
  CK(hipDeviceSynchronize());
  std::vector<float> ms(reps);
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms[r], e0, e1));
  }
  std::sort(ms.begin(), ms.end());
  return ms[reps / 2] * 1e3;
}

template <class TS>
static void run(const char* tname, int B, int HW, int C) {
  const size_t plane = (size_t)C * HW;
  TS *x, *y, *g;
  float *w, *coef, *bias, *stat, *partial, *gw;
  u32x4 *wp, *wpt;
  unsigned* mask;
  CK(hipMalloc(&x, (size_t)B * 2 * plane * 2)); CK(hipMalloc(&y, (size_t)B * plane * 2)); CK(hipMalloc(&g, (size_t)B * plane * 2));
  CK(hipMalloc(&w, (size_t)C * C * 4)); CK(hipMalloc(&coef, (size_t)B * 3 * C * 4)); CK(hipMalloc(&bias, C * 4));
  CK(hipMalloc(&stat, (size_t)2048 * 2 * C * 4)); CK(hipMalloc(&wp, (size_t)C * C * 2)); CK(hipMalloc(&wpt, (size_t)C * C * 2));
  CK(hipMalloc(&mask, cuh_mask_words(B, C, HW) * 4)); CK(hipMalloc(&partial, (size_t)256 * C * C * 4)); CK(hipMalloc(&gw, (size_t)C * C * 4));
  std::vector<float> hw_((size_t)C * C), hcoef((size_t)B * 3 * C), hbias(C);
  srand(1);
  for (auto& v : hw_) v = frand() * 0.0625f;
  for (auto& v : hcoef) v = frand();
  for (auto& v : hbias) v = frand();
  CK(hipMemcpy(w, hw_.data(), hw_.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(coef, hcoef.data(), hcoef.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, hbias.data(), C * 4, hipMemcpyHostToDevice));
  fill_kernel<TS><<<1024, 256>>>(x, (size_t)B * 2 * plane, 17u, 1.f, 0.1f);
  fill_kernel<TS><<<1024, 256>>>(g, (size_t)B * plane, 99u, 1.f, 0.f);
  pack_kernel<TS><<<(C / 32) * (C / 16) * 64 / 256, 256>>>(w, 0, wp, C);
  pack_kernel<TS><<<(C / 32) * (C / 16) * 64 / 256, 256>>>(w, 1, wpt, C);
  CK(hipDeviceSynchronize());

  const int nco = 8;
  double* yref;
  CK(hipMalloc(&yref, (size_t)B * nco * HW * 8));
  std::vector<double> href((size_t)B * nco * HW);
  std::vector<TS> hy((size_t)B * plane);
  auto check = [&](const char* what, const TS* in0, const TS* in1, size_t bs, int relu, int transpose, const float* bs_, int co0, bool masked) {
    ref_kernel<TS><<<dim3((HW + 255) / 256, B), 256>>>(in0, in1, bs, coef, w, transpose, bs_, relu, yref, C, HW, co0, nco);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(href.data(), yref, href.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hy.data(), y, hy.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    size_t n_zero = 0, n_bad = 0;
    for (int b = 0; b < B; ++b)
      for (int co = 0; co < nco; ++co)
        for (int p = 0; p < HW; ++p) {
          const double r = href[((size_t)b * nco + co) * HW + p];
          const double v = (double)(float)hy[((size_t)b * C + co0 + co) * HW + p];
          if (masked && v == 0.0) { ++n_zero; continue; }
          if (!(std::fabs(v - r) < 0.05)) {
            if (n_bad < 12 && (p & 1) == 0) {
              unsigned short lo16, hi16;
              memcpy(&lo16, &hy[((size_t)b * C + co0 + co) * HW + p], 2);
              memcpy(&hi16, &hy[((size_t)b * C + co0 + co) * HW + p + 1], 2);
              const unsigned word = (unsigned)lo16 | ((unsigned)hi16 << 16);
              float as_f;
              memcpy(&as_f, &word, 4);
              printf("    bad: b %d co %d p %d (tile %d, px %d)  got %g want %g | word as float %g; wants of the oct:", b, co0 + co, p, p / 64, p % 64, v, r, as_f);
              for (int e = 0; e < 8; ++e) printf(" %.4f", href[((size_t)b * nco + co) * HW + (p & ~7) + e]);
              printf("\n");
            }
            ++n_bad;
            continue;
          }
          worst = std::max(worst, std::fabs(v - r));
          scale = std::max(scale, std::fabs(r));
        }
    printf("  %-28s max |err| %.3e (scale %.3f)%s  bad %zu\n", what, worst, scale, masked ? " [masked zeros skipped]" : "", n_bad);
    if (masked) printf("    masked-out fraction %.3f\n", (double)n_zero / ((double)B * nco * HW));
  };
  hipStream_t st = 0;
  int rows = 0;
  printf("%s  B %d  HW %d  C %d\n", tname, B, HW, C);
  // conv1: two inputs, bias + statistics
  CK(hipMemset(y, 0xff, (size_t)B * plane * 2));
  if (launch_pw_gemm_cuh<TS>(x, x + plane, 2 * plane, C, coef, false, wp, bias, nullptr, stat, y, 0, B, C, HW, st, &rows)) printf("launch failed\n");
  CK(hipDeviceSynchronize());
  check("conv1 (two in, EPI 0)", x, x + plane, 2 * plane, 0, 0, bias, 0, false);
  check("conv1 channels 120..127", x, x + plane, 2 * plane, 0, 0, bias, C - 8, false);
  // conv2: one input, relu, record
  CK(hipMemset(y, 0xff, (size_t)B * plane * 2));
  if (launch_pw_gemm_cuh<TS>(g, (const TS*)nullptr, plane, C, coef, true, wp, bias, mask, stat, y, 0, B, C, HW, st, &rows)) printf("launch failed\n");
  CK(hipDeviceSynchronize());
  check("conv2 (relu, record, EPI 0)", g, nullptr, plane, 1, 0, bias, 40, false);
  // dgrad with mask
  CK(hipMemset(y, 0xff, (size_t)B * plane * 2));
  if (launch_pw_gemm_cuh<TS>(x, x + plane, 2 * plane, C, coef, false, wpt, nullptr, mask, nullptr, y, 1, B, C, HW, st, nullptr)) printf("launch failed\n");
  CK(hipDeviceSynchronize());
  check("dgrad 2 (EPI 1, W^T)", x, x + plane, 2 * plane, 0, 1, nullptr, 64, true);
  CK(hipMemset(y, 0xff, (size_t)B * plane * 2));
  if (launch_pw_gemm_cuh<TS>(x, x + plane, 2 * plane, C, coef, false, wpt, nullptr, nullptr, nullptr, y, 2, B, C, HW, st, nullptr)) printf("launch failed\n");
  CK(hipDeviceSynchronize());
  check("dgrad 1 (EPI 2, W^T)", x, x + plane, 2 * plane, 0, 1, nullptr, 96, false);

  // weight gradients (8 rows checked)
  double* gref;
  CK(hipMalloc(&gref, (size_t)nco * C * 8));
  std::vector<double> hg((size_t)nco * C);
  std::vector<float> hgw((size_t)C * C);
  auto check_w = [&](const char* what, const TS* b0, const TS* b1, size_t bbs, int brelu, int co0) {
    ref_wgrad_kernel<TS><<<dim3((C + 63) / 64, nco), 64>>>(x, x + plane, coef, 2 * plane, b0, b1, coef, bbs, brelu, gref, C, HW, B, co0);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hg.data(), gref, hg.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hgw.data(), gw, hgw.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (int co = 0; co < nco; ++co)
      for (int ci = 0; ci < C; ++ci) {
        worst = std::max(worst, std::fabs((double)hgw[(size_t)(co0 + co) * C + ci] - hg[(size_t)co * C + ci]));
        scale = std::max(scale, std::fabs(hg[(size_t)co * C + ci]));
      }
    printf("  %-28s max |err| %.3e (scale %.1f)\n", what, worst, scale);
  };
  if (launch_pw_wgrad_h<TS>(x, x + plane, coef, 2 * plane, g, (const TS*)nullptr, coef, plane, true, partial, gw, B, C, HW, st)) printf("launch failed\n");
  CK(hipDeviceSynchronize());
  check_w("wgrad (B relu)", g, nullptr, plane, 1, 16);
  if (launch_pw_wgrad_h<TS>(x, x + plane, coef, 2 * plane, x, x + plane, coef, 2 * plane, false, partial, gw, B, C, HW, st)) printf("launch failed\n");
  CK(hipDeviceSynchronize());
  check_w("wgrad (B two in)", x, x + plane, 2 * plane, 0, C - 8);

  // timings
  const size_t n16 = (size_t)B * plane * 2 / 16;
  const double t_blend = time_kernel([&] { blend_kernel<<<2048, 256>>>((const u32x4*)x, (const u32x4*)g, (u32x4*)y, n16); });
  printf("  blend floor (read 2, write 1 planes of %zu MB): %.1f us = %.2f TB/s\n", (size_t)B * plane * 2 >> 20, t_blend, 3.0 * B * plane * 2 / t_blend / 1e6);
  printf("  conv1  %.1f us\n", time_kernel([&] { launch_pw_gemm_cuh<TS>(x, x + plane, 2 * plane, C, coef, false, wp, bias, nullptr, stat, y, 0, B, C, HW, st, &rows); }));
  printf("  conv2  %.1f us\n", time_kernel([&] { launch_pw_gemm_cuh<TS>(g, (const TS*)nullptr, plane, C, coef, true, wp, bias, mask, stat, y, 0, B, C, HW, st, &rows); }));
  printf("  dgrad2 %.1f us\n", time_kernel([&] { launch_pw_gemm_cuh<TS>(x, x + plane, 2 * plane, C, coef, false, wpt, nullptr, mask, nullptr, y, 1, B, C, HW, st, nullptr); }));
  printf("  dgrad1 %.1f us\n", time_kernel([&] { launch_pw_gemm_cuh<TS>(x, x + plane, 2 * plane, C, coef, false, wpt, nullptr, nullptr, nullptr, y, 2, B, C, HW, st, nullptr); }));
  printf("  wgrad2 (+reduce) %.1f us\n", time_kernel([&] { launch_pw_wgrad_h<TS>(x, x + plane, coef, 2 * plane, g, (const TS*)nullptr, coef, plane, true, partial, gw, B, C, HW, st); }));
  printf("  wgrad1 (+reduce) %.1f us\n", time_kernel([&] { launch_pw_wgrad_h<TS>(x, x + plane, coef, 2 * plane, x, x + plane, coef, 2 * plane, false, partial, gw, B, C, HW, st); }));
  CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(g)); CK(hipFree(yref)); CK(hipFree(gref)); CK(hipFree(partial));
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4, HW = argc > 2 ? atoi(argv[2]) : 40000, C = argc > 3 ? atoi(argv[3]) : 256;
  run<_Float16>("fp16", B, HW, C);
  run<__bf16>("bf16", B, HW, C);
  return 0;
}
