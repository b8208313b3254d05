// Shifted-window partition / reverse of the Swin backbone (DHD-L: backbones/swin.py:448-513 in the reference's ShiftWindowMSA.forward)
// as ONE gather each way.  The reference pads the (B, H, W, C) token map to multiples of the window, rolls it by -shift, and cuts it into
// windows (three copies of the activation: F.pad, torch.roll, permute + reshape); after attention it undoes the three (three more).
// Both directions are permutations of token rows (plus zero rows for the padding), so each is a row copy through an index map:
//   partition: out[b][wy][wx][iy][ix][:] = in[b][y][x][:]  with (y, x) = ((wy ws + iy + shift) mod Hp, (wx ws + ix + shift) mod Wp), 0 outside H x W
//   reverse  : out[b][y][x][:] = win[b][wy][wx][iy][ix][:] with (wy ws + iy, wx ws + ix) = ((y - shift) mod Hp, (x - shift) mod Wp)
// Each is also the other's transpose (the gradient of one is the other applied to the gradient).  The element type may change on the
// way (float32 LayerNorm output -> the autocast dtype the qkv projection would cast to anyway; half gradients -> float32).
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "common.h"

namespace {

constexpr int kWinBlock = 256;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// eight consecutive elements <-> float registers, as 16-byte accesses
template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const f32x4 a = reinterpret_cast<const f32x4*>(p)[0], b = reinterpret_cast<const f32x4*>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    reinterpret_cast<f32x4*>(p)[0] = f32x4{v[0], v[1], v[2], v[3]};
    reinterpret_cast<f32x4*>(p)[1] = f32x4{v[4], v[5], v[6], v[7]};
  }
};
template <> struct Elem<__half> {
  static __device__ __forceinline__ void load8(const __half* p, float (&v)[8]) {
    const u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned w = t[i];
      const __half2 h = *reinterpret_cast<const __half2*>(&w);
      v[2 * i] = __low2float(h);
      v[2 * i + 1] = __high2float(h);
    }
  }
  static __device__ __forceinline__ void store8(__half* p, const float (&v)[8]) {
    u32x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const __half2 h = __floats2half2_rn(v[2 * i], v[2 * i + 1]);
      t[i] = *reinterpret_cast<const unsigned*>(&h);
    }
    *reinterpret_cast<u32x4*>(p) = t;
  }
};
template <> struct Elem<__hip_bfloat16> {
  static __device__ __forceinline__ void load8(const __hip_bfloat16* p, float (&v)[8]) {
    const u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(t[i] << 16);
      v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store8(__hip_bfloat16* p, const float (&v)[8]) {
    auto rne = [](float f) {
      const unsigned u = __float_as_uint(f);
      return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    };
    u32x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = rne(v[2 * i]) | (rne(v[2 * i + 1]) << 16);
    *reinterpret_cast<u32x4*>(p) = t;
  }
};

struct WinGeom {
  int b, h, w, c, ws, shift, hp, wp, nh, nw;
};

// one thread per (row, group of 8 channels); rows of the OUTPUT are walked in order, the source row comes from the index map
template <typename TI, typename TO, bool REVERSE>
__global__ __launch_bounds__(kWinBlock) void window_rows(const TI* __restrict__ in, TO* __restrict__ out, WinGeom g) {
  const int groups = g.c >> 3;
  const long rows = REVERSE ? (long)g.b * g.h * g.w : (long)g.b * g.hp * g.wp;
  const long total = rows * groups;
  for (long idx = (long)blockIdx.x * kWinBlock + threadIdx.x; idx < total; idx += (long)gridDim.x * kWinBlock) {
    const int cg = (int)(idx % groups);
    const long row = idx / groups;
    long src;   // source row, -1 = zeros
    if (REVERSE) {
      const int x = (int)(row % g.w), y = (int)((row / g.w) % g.h);
      const long bi = row / ((long)g.w * g.h);
      int py = y - g.shift, px = x - g.shift;
      if (py < 0) py += g.hp;
      if (px < 0) px += g.wp;
      src = (((bi * g.nh + py / g.ws) * g.nw + px / g.ws) * g.ws + py % g.ws) * g.ws + px % g.ws;
    } else {
      const int ws2 = g.ws * g.ws;
      const int i = (int)(row % ws2);
      const long win = row / ws2;
      const int wx = (int)(win % g.nw), wy = (int)((win / g.nw) % g.nh);
      const long bi = win / ((long)g.nw * g.nh);
      int y = wy * g.ws + i / g.ws + g.shift, x = wx * g.ws + i % g.ws + g.shift;
      if (y >= g.hp) y -= g.hp;
      if (x >= g.wp) x -= g.wp;
      src = (y < g.h && x < g.w) ? (bi * g.h + y) * g.w + x : -1;
    }
    float v[8];
    if (src < 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = 0.f;
    } else {
      Elem<TI>::load8(in + src * g.c + cg * 8, v);
    }
    Elem<TO>::store8(out + row * g.c + cg * 8, v);
  }
}

template <typename TI, typename TO>
int window_launch(const void* in, void* out, const WinGeom& g, int reverse, hipStream_t st) {
  const long rows = reverse ? (long)g.b * g.h * g.w : (long)g.b * g.hp * g.wp;
  long blocks = (rows * (g.c >> 3) + kWinBlock - 1) / kWinBlock;
  if (blocks > 65536) blocks = 65536;
  if (reverse) hipLaunchKernelGGL((window_rows<TI, TO, true>), dim3((unsigned)blocks), dim3(kWinBlock), 0, st, (const TI*)in, (TO*)out, g);
  else hipLaunchKernelGGL((window_rows<TI, TO, false>), dim3((unsigned)blocks), dim3(kWinBlock), 0, st, (const TI*)in, (TO*)out, g);
  DHD_LAUNCH_CHECK();
  return DHD_OK;
}

template <typename TI>
int window_out(const void* in, void* out, int out_dtype, const WinGeom& g, int reverse, hipStream_t st) {
  switch (out_dtype) {
    case 0: return window_launch<TI, float>(in, out, g, reverse, st);
    case 1: return window_launch<TI, __half>(in, out, g, reverse, st);
    default: return window_launch<TI, __hip_bfloat16>(in, out, g, reverse, st);
  }
}

}  // namespace

extern "C" int dhd_window_rows(const void* in, void* out, int in_dtype, int out_dtype, int b, int h, int w, int c, int window, int shift,
                               int reverse, void* stream) {
  if (!in || !out) return DHD_EINVAL;
  if (in_dtype < 0 || in_dtype > 2 || out_dtype < 0 || out_dtype > 2 || b <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 7) || window <= 0 ||
      shift < 0 || shift >= window)
    return DHD_EUNSUPPORTED;
  WinGeom g{b, h, w, c, window, shift, 0, 0, 0, 0};
  g.nh = (h + window - 1) / window;
  g.nw = (w + window - 1) / window;
  g.hp = g.nh * window;
  g.wp = g.nw * window;
  if ((long)b * g.hp * g.wp >= (1L << 40)) return DHD_EUNSUPPORTED;
  hipStream_t st = dhd_stream(stream);
  switch (in_dtype) {
    case 0: return window_out<float>(in, out, out_dtype, g, reverse, st);
    case 1: return window_out<__half>(in, out, out_dtype, g, reverse, st);
    default: return window_out<__hip_bfloat16>(in, out, out_dtype, g, reverse, st);
  }
}
