// Shared device helpers of the SFA stage's GEMM kernels (sfa_stage.hip, sfa_gemm_cu.h): vector types, the exact bf16 splits of a
// float32 operand, the bf16 MFMA wrapper and compile-time loops.
#pragma once
#include <type_traits>
#include <utility>

#include "common.h"

namespace dhd_sfa {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

// two floats -> packed bf16 pairs (a in the low half = lower k) of the three terms.
// Round-to-nearest-even cuts (v_cvt_pk_bf16_f32, one instruction per pair): h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m).  The residuals are exact in float32 (x - h has at most 16 significant bits, x - h - m at most
// 8, so l is exact too): h + m + l == x.  Compared with cuts by truncation the parts are up to 4x smaller
// (|x - h| <= 2^-9 |x|, |x - h - m| <= 2^-17 |x|), which matters for the three-product mode where the terms
// am*bm, al*bh, ah*bl are dropped: worst case 3 * 2^-18 |ab| per product.
__device__ __forceinline__ unsigned pack_bf16(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ f32x2 unpack_bf16(unsigned p) {
  f32x2 r = {__uint_as_float(p << 16), __uint_as_float(p & 0xffff0000u)};
  return r;
}
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  const f32x2 x = {a, b};
  h = pack_bf16(x);
  const f32x2 r1 = x - unpack_bf16(h);
  m = pack_bf16(r1);
  l = pack_bf16(r1 - unpack_bf16(m));
}
__device__ __forceinline__ void split2_hm(float a, float b, unsigned& h, unsigned& m) {
  const f32x2 x = {a, b};
  h = pack_bf16(x);
  m = pack_bf16(x - unpack_bf16(h));
}

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// word with lane L replaced by the wave-uniform value sval (v_writelane_b32: one VALU instruction).  gfx940+ needs two
// wait states between a VALU write of an SGPR (the v_cmp that made sval) and a VALU read of it; the compiler inserts
// them for its own instructions but cannot see into inline assembly (without them: stale pass bits, found by the
// full-size parity test).
template <int L>
__device__ __forceinline__ int write_lane(int word, int sval) {
  asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(word) : "s"(sval), "n"(L));
  return word;
}

// buffer_store_dwordx4 whose write-data registers stay untouched for the wait states the hardware needs.  hipcc lets a VALU
// instruction overwrite the data VGPRs of a MUBUF store of more than 8 bytes in the very next issue slot when the store's
// soffset is an SGPR (LLVM GCNHazardRecognizer::createsVALUHazard assumes the hazard away in that case); on gfx950 the store
// then sends the NEW values -- measured in round 5 with pw_gemm_cuh_kernel<_Float16> (experiments/gemm_cuh_bench.hip): 1.6 % of
// the 16-byte stores carried four bytes of the next channel row until a wait state followed the store.  The asm "uses" the data
// register, so the allocator cannot recycle it before, and spends two wait states.  experiments/lint_store_hazard.py checks the
// generated code for the pattern (tests/test_build_lint.py).
template <int AUX>
__device__ __forceinline__ void store_b128_guarded(u32x4 v, __amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voffset, soffset, AUX);
  asm volatile("s_nop 1" : "+v"(v) : : "memory");
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {   // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), straight-line
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

}  // namespace dhd_sfa
