// Store-pattern microbenchmark for mghs_stream_fwd (experiments only): how fast can 704 MB be zero-filled
//   A  linear, 16-byte non-temporal stores, grid-stride
//   B  linear, plain stores
//   C  the writer's pattern: workgroup = 64 runs of 3200 contiguous bytes at a stride of 160 000 bytes (one 4-row segment of
//      64 channel planes), flat (run, vector) index space, nt stores
//   D  as C with plain stores
//   E  workgroup = ONE contiguous 204 800-byte chunk of a plane (same bytes per workgroup as C), nt
// build: hipcc --offload-arch=gfx950 -O3 experiments/fill_patterns.hip -o experiments/build/fill_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 512;
template <bool NT> __device__ inline void st(v4* p, v4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <bool NT> __global__ __launch_bounds__(kThreads) void fill_linear(v4* out, size_t n4) {
  const v4 z = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (size_t)gridDim.x * kThreads) st<NT>(out + i, z);
}
// planes of 40000 floats; segment s of plane-group pg: rows [4 s, 4 s + 4) of the 64 planes pg*64 .. pg*64+63
template <bool NT> __global__ __launch_bounds__(kThreads) void fill_segments(float* out, int n_seg_per_group) {
  const v4 z = {0.f, 0.f, 0.f, 0.f};
  const int pg = blockIdx.x / n_seg_per_group, s = blockIdx.x % n_seg_per_group;
  float* base = out + (size_t)pg * 64 * 40000 + (size_t)s * 800;
  const int nvec = 200, total = 64 * nvec;
  for (int idx = threadIdx.x; idx < total; idx += kThreads) {
    const int cc = idx / nvec, i = idx % nvec;
    st<NT>(reinterpret_cast<v4*>(base + (size_t)cc * 40000) + i, z);
  }
}
template <bool NT> __global__ __launch_bounds__(kThreads) void fill_chunks(v4* out, int vec_per_chunk) {
  const v4 z = {0.f, 0.f, 0.f, 0.f};
  v4* base = out + (size_t)blockIdx.x * vec_per_chunk;
  for (int i = threadIdx.x; i < vec_per_chunk; i += kThreads) st<NT>(base + i, z);
}
int main() {
  const size_t planes = 4 * 17 * 64;            // B * sum(nz) * C
  const size_t bytes = planes * 40000 * 4;      // 696 MB
  float* buf; hipMalloc(&buf, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %7.1f us  %6.2f TB/s\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
  };
  const size_t n4 = bytes / 16;
  const int n_seg = 50, groups = (int)(planes / 64);
  timeit("memset (hipMemsetAsync)", [&] { hipMemsetAsync(buf, 0, bytes, 0); });
  timeit("A linear nt, 4096 blocks", [&] { fill_linear<true><<<4096, kThreads>>>((v4*)buf, n4); });
  timeit("A linear nt, 16384 blocks", [&] { fill_linear<true><<<16384, kThreads>>>((v4*)buf, n4); });
  timeit("B linear plain, 4096 blocks", [&] { fill_linear<false><<<4096, kThreads>>>((v4*)buf, n4); });
  timeit("C segments (64 runs x 3200 B) nt", [&] { fill_segments<true><<<groups * n_seg, kThreads>>>(buf, n_seg); });
  timeit("D segments plain", [&] { fill_segments<false><<<groups * n_seg, kThreads>>>(buf, n_seg); });
  timeit("E contiguous 204800-B chunks nt", [&] { fill_chunks<true><<<(int)(bytes / 204800), kThreads>>>((v4*)buf, 12800); });
  timeit("E contiguous 204800-B chunks plain", [&] { fill_chunks<false><<<(int)(bytes / 204800), kThreads>>>((v4*)buf, 12800); });
  timeit("E contiguous 40000-B chunks nt", [&] { fill_chunks<true><<<(int)(bytes / 40000), kThreads>>>((v4*)buf, 2500); });
  return 0;
}
