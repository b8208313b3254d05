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
// F  the linear sweep with patches: per 16-byte vector one nibble of a voxel bitmask (uint32 per 32 voxels + a prefix
//    count, 8 bytes per 32 voxels); a non-empty voxel takes its value from a channel-major table at slot = prefix + popcount
template <bool NT, bool ROWMAJOR> __global__ __launch_bounds__(kThreads) void sweep_patched(v4* out, size_t n4, const uint2* __restrict__ words,
                                                                              const float* __restrict__ table, int n_slots) {
  const size_t stride = (size_t)gridDim.x * kThreads;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
    const size_t plane = i / 10000;                 // 10000 vectors per 200x200 plane
    const int q = (int)(i - plane * 10000);
    const int grp = (int)(plane >> 6), c = (int)(plane & 63);   // 64 channel planes share one voxel slice
    const int v0 = grp * 40000 + 4 * q;             // first voxel of this vector
    const uint2 w = words[v0 >> 5];
    const unsigned nib = (w.x >> (v0 & 31)) & 0xFu;
    v4 v = {0.f, 0.f, 0.f, 0.f};
    if (nib) {
      const int base = w.y + __builtin_popcount(w.x & ((1u << (v0 & 31)) - 1u));
      int k = 0;
      if (ROWMAJOR) {   // table[slot][64]: a 4-byte gather per non-empty voxel at a stride of 256 bytes
        const float* t = table + c;
        if (nib & 1u) v.x = t[(size_t)(base + k++) * 64];
        if (nib & 2u) v.y = t[(size_t)(base + k++) * 64];
        if (nib & 4u) v.z = t[(size_t)(base + k++) * 64];
        if (nib & 8u) v.w = t[(size_t)(base + k++) * 64];
      } else {
        const float* t = table + (size_t)c * n_slots;
        if (nib & 1u) v.x = t[base + k++];
        if (nib & 2u) v.y = t[base + k++];
        if (nib & 4u) v.z = t[base + k++];
        if (nib & 8u) v.w = t[base + k++];
      }
    }
    st<NT>(out + i, v);
  }
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
  {
    // 68 voxel slices of 40000 voxels, ~7 % non-empty in clustered runs (like rays)
    const int n_vox = 68 * 40000, n_words = n_vox / 32;
    std::vector<uint2> words(n_words);
    unsigned cnt = 0, state = 12345u;
    for (int wi = 0; wi < n_words; ++wi) {
      unsigned bits = 0;
      for (int b = 0; b < 32; ++b) {
        state = state * 1664525u + 1013904223u;
        const bool on = ((state >> 8) % 100) < 7;
        if (on) bits |= 1u << b;
      }
      words[wi] = make_uint2(bits, cnt);
      cnt += __builtin_popcount(bits);
    }
    uint2* dwords; float* dtable;
    hipMalloc(&dwords, n_words * sizeof(uint2)); hipMemcpy(dwords, words.data(), n_words * sizeof(uint2), hipMemcpyHostToDevice);
    hipMalloc(&dtable, (size_t)64 * cnt * 4); hipMemset(dtable, 0, (size_t)64 * cnt * 4);
    printf("non-empty voxels: %u of %d\n", cnt, n_vox);
    timeit("F sweep + patches, channel-major table, 16384 blocks", [&] { sweep_patched<true, false><<<16384, kThreads>>>((v4*)buf, n4, dwords, dtable, (int)cnt); });
    timeit("F sweep + patches, channel-major table, 32768 blocks", [&] { sweep_patched<true, false><<<32768, kThreads>>>((v4*)buf, n4, dwords, dtable, (int)cnt); });
    timeit("G sweep + patches, row-major table [slot][64], 16384 blocks", [&] { sweep_patched<true, true><<<16384, kThreads>>>((v4*)buf, n4, dwords, dtable, (int)cnt); });
    timeit("A linear nt, 16384 blocks (again)", [&] { fill_linear<true><<<16384, kThreads>>>((v4*)buf, n4); });
    timeit("C segments nt (again)", [&] { fill_segments<true><<<groups * n_seg, kThreads>>>(buf, n_seg); });
  }
  return 0;
}
