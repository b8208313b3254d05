// Stand-alone check + timing of pw_gemm_cu_kernel (dhd_amd/csrc/sfa_gemm_cu.h) at the SFA stage's size (B = 4, C = 256,
// HW = 200 x 200): result against a float64-accumulating reference kernel, time per launch by HIP events for several variants
// (tile order, cache policy of the loads, waves per workgroup, staging interleaved with the MFMAs or after them), and the time
// of a pure blend pass with the same HBM traffic (the floor of this GEMM: read 2 x 164 MB, write 164 MB).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off experiments/gemm_cu_bench.hip -o experiments/build/gemm_cu_bench
// run  : experiments/build/gemm_cu_bench [B] [HW]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../dhd_amd/csrc/sfa_stage.hip"   // the whole product translation unit: launch_pw_gemm_res (anonymous namespace) for the A/B
#include "../dhd_amd/csrc/sfa_gemm_cu.h"

using namespace dhd_sfa;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void pack_kernel(const float* w, int transpose, u32x4* wp, int c) { cu_pack_weight(w, transpose, wp, c, blockIdx.x * blockDim.x + threadIdx.x); }

// y[b][co][p] = sum_k W[co][k] act(c0 in0 + c1 in1 + c2)   (double accumulation)
__global__ void ref_kernel(const float* in0, const float* in1, size_t bs, const float* coef, const float* w, const float* bias, int relu,
                           double* y, int c, int hw, int nb, int co0, int nco) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (p >= hw) return;
  for (int co = co0; co < co0 + nco; ++co) {
    double acc = 0.0;
    for (int k = 0; k < c; ++k) {
      float v = fmaf(coef[(size_t)b * 3 * c + k], in0[(size_t)b * bs + (size_t)k * hw + p], coef[(size_t)b * 3 * c + 2 * c + k]);
      if (in1) v = fmaf(coef[(size_t)b * 3 * c + c + k], in1[(size_t)b * bs + (size_t)k * hw + p], v);
      if (relu) v = fmaxf(v, 0.f);
      acc += (double)w[(size_t)co * c + k] * (double)v;
    }
    y[((size_t)b * nco + (co - co0)) * hw + p] = acc + (bias ? (double)bias[co] : 0.0);
  }
}

__global__ void blend_kernel(const f32x4* in0, const f32x4* in1, f32x4* out, size_t n4, float c0, float c1) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 a = __builtin_nontemporal_load(in0 + i), b = __builtin_nontemporal_load(in1 + i);
    out[i] = a * c0 + b * c1;
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

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4, HW = argc > 2 ? atoi(argv[2]) : 40000;
  constexpr int C = 256, KCN = 16;
  const size_t plane = (size_t)C * HW;
  srand(1);
  std::vector<float> hx((size_t)B * 2 * plane), hw_((size_t)C * C), hcoef((size_t)B * 3 * C), hbias(C);
  for (auto& v : hx) v = frand();
  for (auto& v : hw_) v = frand() * 0.0625f;
  for (auto& v : hcoef) v = frand();
  for (auto& v : hbias) v = frand();
  float *x, *w, *coef, *bias, *y, *stat;
  u32x4* wp;
  unsigned* mask;
  CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&w, hw_.size() * 4)); CK(hipMalloc(&coef, hcoef.size() * 4)); CK(hipMalloc(&bias, C * 4));
  CK(hipMalloc(&y, (size_t)B * plane * 4)); CK(hipMalloc(&stat, (size_t)512 * 2 * C * 4)); CK(hipMalloc(&wp, (size_t)C * C * 4));
  CK(hipMalloc(&mask, cu_mask_words(B, C, HW) * 4));
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw_.data(), hw_.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(coef, hcoef.data(), hcoef.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, hbias.data(), C * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(pack_kernel, dim3((C / 32) * KCN * 64 / 256), dim3(256), 0, 0, w, 0, wp, C);
  CK(hipDeviceSynchronize());
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  printf("B=%d HW=%d CUs=%d\n", B, HW, cus);

  // reference for 8 output channels spread over the waves' ranges, all pixels
  const int nchk = 2;
  const int chk0[2] = {0, 200};
  const int nco = 4;
  double* yref;
  CK(hipMalloc(&yref, (size_t)B * nco * HW * 8));
  std::vector<double> href((size_t)B * nco * HW);
  std::vector<float> hy((size_t)B * plane);

  auto check = [&](const char* name, bool two, bool relu, bool with_bias) {
    CK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (int c = 0; c < nchk; ++c) {
      hipLaunchKernelGGL(ref_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, 0, x, two ? x + plane : nullptr, 2 * plane, coef, w,
                         with_bias ? bias : nullptr, relu ? 1 : 0, yref, C, HW, B, chk0[c], nco);
      CK(hipMemcpy(href.data(), yref, href.size() * 8, hipMemcpyDeviceToHost));
      for (int b = 0; b < B; ++b)
        for (int co = 0; co < nco; ++co)
          for (int p = 0; p < HW; ++p) {
            const double r = href[((size_t)b * nco + co) * HW + p], v = hy[((size_t)b * C + chk0[c] + co) * HW + p];
            worst = std::max(worst, std::fabs(r - v));
            scale = std::max(scale, std::fabs(r));
          }
    }
    printf("  %-40s max |err| %.3e (max |ref| %.3f)\n", name, worst, scale);
  };

#define RUN(WAVES, TWO, RELU, EPI, REC, AUX, RR, NACC, contig, label) RUNA(WAVES, TWO, RELU, EPI, REC, AUX, RR, NACC, 0, contig, label)
#define RUNA(WAVES, TWO, RELU, EPI, REC, AUX, RR, NACC, ABL, contig, label) RUNS(WAVES, TWO, RELU, EPI, REC, AUX, RR, NACC, ABL, 0, contig, label)
#define RUNS(WAVES, TWO, RELU, EPI, REC, AUX, RR, NACC, ABL, SAUX, contig, label) RUNP(WAVES, TWO, RELU, EPI, REC, AUX, RR, NACC, ABL, SAUX, false, contig, label)
#define RUNP(WAVES, TWO, RELU, EPI, REC, AUX, RR, NACC, ABL, SAUX, PP, contig, label)                                                         \
  do {                                                                                                                     \
    auto kern = pw_gemm_cu_kernel<KCN, WAVES, TWO, RELU, EPI, REC, AUX, RR, NACC, ABL, SAUX, PP>;                                            \
    const size_t lds = cu_lds_bytes(C, WAVES, B);                                                                             \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));    \
    CK(hipMemset(y, 0xff, (size_t)B* plane * 4));                                                                          \
    auto launch = [&] {                                                                                                    \
      hipLaunchKernelGGL(kern, dim3(cus), dim3(WAVES * 64), lds, 0, x, TWO ? x + plane : nullptr, 2 * plane,               \
                         (unsigned)(plane * 4), coef, wp, bias, mask, stat, y, HW, B, contig);                             \
    };                                                                                                                     \
    const double us = time_kernel(launch);                                                                                 \
    const double bytes = (double)B * plane * 4 * ((TWO ? 2 : 1) + 1);                                                      \
    printf("%-58s %7.1f us  %5.2f TB/s\n", label, us, bytes / us / 1e6);                                                   \
    CK(hipGetLastError());                                                                                                 \
    if (ABL == 0) check(label, TWO, RELU, EPI == 0);                                                                       \
    if (ABL & 256) {                                                                                                       \
      std::vector<unsigned> pc((size_t)cus * WAVES * 8);                                                                   \
      CK(hipMemcpy(pc.data(), reinterpret_cast<unsigned*>(stat) + 1024, pc.size() * 4, hipMemcpyDeviceToHost));            \
      const char* nm[6] = {"mfma", "wait+stage", "issue", "epilogue", "-", "barrier"};                                     \
      for (int wsel = 0; wsel < WAVES; wsel += WAVES - 1) {                                                                \
        double sum[6] = {0, 0, 0, 0, 0, 0}; double nt = 0;                                                                 \
        for (int wg = 0; wg < cus; ++wg) { const unsigned* o = &pc[((size_t)wg * WAVES + wsel) * 8]; for (int i = 0; i < 6; ++i) sum[i] += o[i]; nt += o[6]; } \
        printf("    wave %d clocks per tile:", wsel);                                                                      \
        for (int i = 0; i < 6; ++i) if (i != 4) printf("  %s %.0f", nm[i], sum[i] / nt);                                   \
        printf("\n");                                                                                                      \
      }                                                                                                                    \
    }                                                                                                                      \
    if (ABL & 64) {                                                                                                        \
      long long c[256];                                                                                                    \
      CK(hipMemcpy(c, stat, sizeof(c), hipMemcpyDeviceToHost));                                                            \
      std::sort(c, c + 256);                                                                                               \
      printf("    shader clocks per workgroup: median %lld, max %lld -> %.0f MHz average\n", c[128], c[255], c[255] / us); \
    }                                                                                                                      \
  } while (0)

  {
    const size_t n4 = (size_t)B * plane / 4;
    f32x4 *a = (f32x4*)x, *b = (f32x4*)(x + (size_t)B * plane), *o = (f32x4*)y;
    const double us = time_kernel([&] { hipLaunchKernelGGL(blend_kernel, dim3(cus * 8), dim3(512), 0, 0, a, b, o, n4, 0.5f, 0.25f); });
    printf("%-58s %7.1f us  %5.2f TB/s\n", "floor: blend pass (2 reads + 1 write of B*C*HW floats)", us, (double)B * plane * 12 / us / 1e6);
  }
  RUNP(8, true, false, 2, false, 2, 1, 1, 0, 0, true, 1, "8w two-in plain, R=1, PING-PONG");
  RUNP(8, true, false, 2, false, 2, 1, 1, 64 + 256, 0, true, 1, "8w two-in plain, R=1, PING-PONG, phase clocks");
  RUNP(8, true, false, 0, false, 2, 1, 1, 0, 0, true, 1, "8w conv1 (bias+stats), PING-PONG");
  RUNP(8, false, true, 0, true, 2, 1, 1, 0, 0, true, 1, "8w conv2 (relu+record), PING-PONG");
  RUNP(8, true, false, 1, false, 2, 1, 1, 0, 0, true, 1, "8w dgrad2 (mask), PING-PONG");
  // ---- A/B against the resident-weights kernel of the product (teams of two CUs), alternating, same process ---------------
  {
    float* wpres; float* statres;
    CK(hipMalloc(&wpres, (size_t)2 * C * C * 4)); CK(hipMalloc(&statres, (size_t)B * (HW / 32 + 64) * 2 * C * 4));
    g_gemm_mode = 3;
    hipLaunchKernelGGL(pack_weight_res_kernel, dim3(dhd_cdiv((C / 32) * (C / 16) * 64, kEwBlock), 1), dim3(kEwBlock), 0, 0, w, nullptr, 0,
                       reinterpret_cast<u32x4*>(wpres), nullptr, C, res_cob(C, 2), 2);
    CK(hipDeviceSynchronize());
    auto old_conv1 = [&] { int rows; launch_pw_gemm_res(x, x + plane, 2 * plane, C, coef, false, wpres, bias, nullptr, statres, y, 0, B, C, HW, 0, &rows); };
    auto old_dgrad = [&] { int rows; launch_pw_gemm_res(x, x + plane, 2 * plane, C, coef, false, wpres, nullptr, nullptr, nullptr, y, 2, B, C, HW, 0, &rows); };
    auto old_conv2 = [&] { int rows; launch_pw_gemm_res(x, nullptr, 2 * plane, C, coef, true, wpres, bias, mask, statres, y, 0, B, C, HW, 0, &rows); };
#define NEWK(WAVES, TWO, RELU, EPI, REC, AUX, RR, NACC, contig) NEWP(WAVES, TWO, RELU, EPI, REC, AUX, RR, NACC, false, contig)
#define NEWP(WAVES, TWO, RELU, EPI, REC, AUX, RR, NACC, PP, contig) NEWQ(WAVES, TWO, RELU, EPI, REC, AUX, RR, NACC, PP, 0, 0, contig)
#define NEWQ(WAVES, TWO, RELU, EPI, REC, AUX, RR, NACC, PP, BPF, EORD, contig)                                                            \
    [&] {                                                                                                                  \
      auto kern = pw_gemm_cu_kernel<KCN, WAVES, TWO, RELU, EPI, REC, AUX, RR, NACC, 0, 0, PP, BPF, EORD>;                                 \
      const size_t lds = cu_lds_bytes(C, WAVES, B);                                                                        \
      static bool once = false;                                                                                            \
      if (!once) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; } \
      hipLaunchKernelGGL(kern, dim3(cus), dim3(WAVES * 64), lds, 0, x, TWO ? x + plane : nullptr, 2 * plane,               \
                         (unsigned)(plane * 4), coef, wp, bias, mask, stat, y, HW, B, contig);                             \
    }
    auto c1_0 = NEWQ(8, true, false, 0, false, 2, 1, 1, true, 0, 1, 1);
    auto c1_1 = NEWQ(8, true, false, 0, false, 2, 1, 1, true, 1, 1, 1);
    auto dg_0 = NEWQ(8, true, false, 2, false, 2, 1, 1, true, 0, 1, 1);
    auto dg_1 = NEWQ(8, true, false, 2, false, 2, 1, 1, true, 1, 1, 1);
    auto dg_00 = NEWQ(8, true, false, 2, false, 2, 1, 1, true, 0, 0, 1);
    auto c2_0 = NEWQ(8, false, true, 0, true, 2, 1, 1, true, 0, 1, 1);
    auto c2_1 = NEWQ(8, false, true, 0, true, 2, 1, 1, true, 1, 1, 1);
    auto c2_00 = NEWQ(8, false, true, 0, true, 2, 1, 1, true, 0, 0, 1);
    auto d2_0 = NEWQ(8, true, false, 1, false, 2, 1, 1, true, 0, 1, 1);
    auto blend = [&] { hipLaunchKernelGGL(blend_kernel, dim3(cus * 8), dim3(512), 0, 0, (f32x4*)x, (f32x4*)(x + (size_t)B * plane), (f32x4*)y, (size_t)B * plane / 4, 0.5f, 0.25f); };
    for (int round = 0; round < 4; ++round) {
      printf("round %d:  blend %.1f | conv1 old %.1f epi-first bpf0 %.1f bpf1 %.1f | dgrad old %.1f epi-first bpf0 %.1f bpf1 %.1f epi-last bpf0 %.1f dgrad2 %.1f | conv2 old %.1f epi-first bpf0 %.1f bpf1 %.1f epi-last bpf0 %.1f  (us)\n", round,
             time_kernel(blend, 7), time_kernel(old_conv1, 7), time_kernel(c1_0, 7), time_kernel(c1_1, 7),
             time_kernel(old_dgrad, 7), time_kernel(dg_0, 7), time_kernel(dg_1, 7), time_kernel(dg_00, 7), time_kernel(d2_0, 7),
             time_kernel(old_conv2, 7), time_kernel(c2_0, 7), time_kernel(c2_1, 7), time_kernel(c2_00, 7));
      CK(hipGetLastError());
    }
  }
  return 0;
}
