"""Patches the COPY of sfa_half.h / gemm_cuh_bench.hip made by make.sh (the product sources stay untouched): with -DDHD_EXP_PAIR the
data-gradient GEMM (EPI = 1) also loads the y tile of the rows it stores and accumulates pair_sums_h's two sums per channel."""
import sys
d = sys.argv[1]
p = d + '/dhd_amd/csrc/sfa_half.h'
s = open(p).read()

def rep(old, new, count=1):
    global s
    assert s.count(old) >= 1, old[:60]
    s = s.replace(old, new, count)

rep("// y[b, co, p] = TS( sum_ci TS(W[co, ci]) * TS(act(c0[b,ci]*in0[b,ci,p] + c1[b,ci]*in1[b,ci,p] + c2[b,ci])) (+ epilogue) )",
    """#ifdef DHD_EXP_PAIR
__device__ const void* g_exp_y = nullptr;       // y1: (B, C, HW) of TS
__device__ const float* g_exp_mean = nullptr;   // [C]
__device__ float* g_exp_stat = nullptr;         // [grid][2][C]
__global__ void exp_set_kernel(const void* y, const float* m, float* st_) { g_exp_y = y; g_exp_mean = m; g_exp_stat = st_; }
#endif
// y[b, co, p] = TS( sum_ci TS(W[co, ci]) * TS(act(c0[b,ci]*in0[b,ci,p] + c1[b,ci]*in1[b,ci,p] + c2[b,ci])) (+ epilogue) )""")
rep("""  if (EPI == 0)
    for (int i = tid; i < C; i += WAVES * 64) bias_lds[i] = bias[i];
""", """  if (EPI == 0)
    for (int i = tid; i < C; i += WAVES * 64) bias_lds[i] = bias[i];
#ifdef DHD_EXP_PAIR
  if (EPI == 1)
    for (int i = tid; i < C; i += WAVES * 64) bias_lds[i] = g_exp_mean[i];
#endif
""")
rep("""    // ---- MFMA phase: D[channel][pixel] over all K, two 32-pixel halves -------------------------------------------------
""", """#ifdef DHD_EXP_PAIR
    u32x4 yv[4];
    if (EPI == 1) {   // the y tile of the rows this lane stores, requested before the MFMA phase
      const __amdgpu_buffer_rsrc_t ryy = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<TS*>(static_cast<const TS*>(g_exp_y)) + (size_t)b * C * hw, 0, (unsigned)((size_t)C * hw * sizeof(TS)), 0x00020000);
      const int vo = (p0 + 8 * q < hw) ? st_voff : st_voff - 16 * q;
#pragma unroll
      for (int k = 0; k < 4; ++k) yv[k] = __builtin_amdgcn_raw_buffer_load_b128(ryy, vo, 8 * k * row_bytes + p0 * 2, 2);
    }
#endif
    // ---- MFMA phase: D[channel][pixel] over all K, two 32-pixel halves -------------------------------------------------
""")
rep("""        } else {
          pk = narrow8<TS>(o);
        }
""", """        } else {
          pk = narrow8<TS>(o);
#ifdef DHD_EXP_PAIR
          if (EPI == 1) {
            float r[8], yy[8];
            widen8<TS>(pk, r);
            widen8<TS>(yv[k], yy);
            const float mu = bias_lds[kbase + g + 8 * k];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1 += r[e]; s2 = fmaf(r[e], yy[e] - mu, s2); }
            ws1[k] += oct_ok ? s1 : 0.f;
            ws2[k] += oct_ok ? s2 : 0.f;
          }
#endif
        }
""")
rep("""  if (EPI == 0 && stat_part != nullptr) {                            // one row [2][C] per workgroup""",
    """#ifdef DHD_EXP_PAIR
  if (EPI == 1) stat_part = g_exp_stat;
  if ((EPI == 0 || EPI == 1) && stat_part != nullptr) {
#else
  if (EPI == 0 && stat_part != nullptr) {                            // one row [2][C] per workgroup
#endif""")
open(p, 'w').write(s)

p = d + '/experiments/gemm_cuh_bench.hip'
s = open(p).read()
rep('#include "../dhd_amd/csrc/sfa_stage.hip"', '#include "../dhd_amd/csrc/sfa_stage.hip"')
rep("""  if (launch_pw_gemm_cuh<TS>(x, x + plane, 2 * plane, C, coef, false, wp, bias, nullptr, stat, y, 0, B, C, HW, st, &rows)) printf("launch failed\\n");
""", """#ifdef DHD_EXP_PAIR
  {
    const void* yp = g; const float* mp = coef; float* sp = stat;
    hipLaunchKernelGGL(exp_set_kernel, dim3(1), dim3(1), 0, 0, yp, mp, sp);
    CK(hipDeviceSynchronize());
    printf("  [DHD_EXP_PAIR: dgrad2's epilogue also loads the y tile and accumulates pair_sums' two sums]\\n");
  }
#endif
  if (launch_pw_gemm_cuh<TS>(x, x + plane, 2 * plane, C, coef, false, wp, bias, nullptr, stat, y, 0, B, C, HW, st, &rows)) printf("launch failed\\n");
""")
open(p, 'w').write(s)
print('patched')
