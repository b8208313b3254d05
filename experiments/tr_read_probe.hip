// Probe of ds_read_b64_tr_b16 (gfx950): which LDS element does lane l, element j receive, given the addresses the lanes supply?
// Every lane of a 16-lane group supplies the address of 4 contiguous 16-bit elements; the probe fills LDS with its own element
// index and prints result(l, j) for the two address patterns used by pw_gemm_t (sfa_gemm.hip):
//   A  lane s of group g: element  g*64 + (s>>2)*16 + 4*(s&3)      ([4][16] block per group, row-major)
//   B  rows at a 32-element pitch: g*16 + (s>>2)*32*... see below
// build: hipcc --offload-arch=gfx950 -O3 experiments/tr_read_probe.hip -o experiments/build/tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(short* out, int pattern) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, s = l & 15;
  int elem;
  if (pattern == 0) elem = g * 64 + (s >> 2) * 16 + 4 * (s & 3);
  else {
    // staging image [k row][32 pixels]: group g reads rows 8*(g>>1) + (s>>2), pixels 16*(g&1) + 4*(s&3)
    elem = (8 * (g >> 1) + (s >> 2)) * 32 + 16 * (g & 1) + 4 * (s & 3);
  }
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + elem));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int p = 0; p < 2; ++p) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, p);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %d\n", p);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  }
  return 0;
}
