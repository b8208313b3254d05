// Infinity-Cache (256 MB L3) reuse microbenchmark (experiments only): does a streaming pass run faster when it starts
// where the previous pass over the same tensor ended?  The SFA stage's backward is a chain of HBM-bound passes over
// 164-328 MB tensors; every pass walks them front to back, so whatever the L3 still holds of the previous pass
// (its tail) is the part the next pass reaches last.
//   producer: write S bytes front to back (plain or non-temporal stores)  |  or read them front to back
//   consumer: read-sum the same S bytes front to back / back to front, plain or non-temporal loads
// build: hipcc --offload-arch=gfx950 -O3 experiments/mall_reuse.hip -o experiments/build/mall_reuse
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 256, kIter = 16;   // a workgroup owns 64 KB

template <bool NT> __global__ __launch_bounds__(kThreads) void writer(v4* buf, float val) {
  v4* p = buf + (size_t)blockIdx.x * kThreads * kIter + threadIdx.x;
  const v4 v = {val, val, val, val};
#pragma unroll
  for (int k = 0; k < kIter; ++k) {
    if (NT) __builtin_nontemporal_store(v, p + k * kThreads); else p[k * kThreads] = v;
  }
}
template <bool NT, bool REV> __global__ __launch_bounds__(kThreads) void reader(const v4* buf, float* sink) {
  const size_t blk = REV ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  const v4* p = buf + blk * kThreads * kIter + threadIdx.x;
  v4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < kIter; ++k) {
    const v4 v = NT ? __builtin_nontemporal_load(p + k * kThreads) : p[k * kThreads];
    a += v;
  }
  if (a.x + a.y + a.z + a.w == 123.456f) sink[0] = a.x;
}

int main() {
  const size_t max_bytes = (size_t)1 << 30;
  v4* buf; float* sink; v4* trash;
  hipMalloc(&buf, max_bytes); hipMalloc(&sink, 4); hipMalloc(&trash, max_bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 10;
  for (size_t mb : {64, 128, 164, 256, 328, 492, 656}) {
    const size_t bytes = mb << 20;
    const int grid = (int)(bytes / (kThreads * kIter * 16));
    const int tgrid = (int)(max_bytes / (kThreads * kIter * 16));
    for (int prod = 0; prod < 4; ++prod) {      // 0 plain write, 1 nt write, 2 plain read, 3 nt read
      for (int cons = 0; cons < 4; ++cons) {    // bit0 = nt loads, bit1 = reversed
        float tot = 0.f;
        for (int r = 0; r < reps; ++r) {
          writer<true><<<tgrid, kThreads>>>(trash, 1.f);   // flush the caches with 1 GB of other lines
          switch (prod) {
            case 0: writer<false><<<grid, kThreads>>>(buf, 2.f); break;
            case 1: writer<true><<<grid, kThreads>>>(buf, 2.f); break;
            case 2: reader<false, false><<<grid, kThreads>>>(buf, sink); break;
            case 3: reader<true, false><<<grid, kThreads>>>(buf, sink); break;
          }
          hipEventRecord(e0);
          switch (cons) {
            case 0: reader<false, false><<<grid, kThreads>>>(buf, sink); break;
            case 1: reader<true, false><<<grid, kThreads>>>(buf, sink); break;
            case 2: reader<false, true><<<grid, kThreads>>>(buf, sink); break;
            case 3: reader<true, true><<<grid, kThreads>>>(buf, sink); break;
          }
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (r >= 2) tot += ms;
        }
        const float ms = tot / (reps - 2);
        static const char* pn[] = {"write", "write-nt", "read", "read-nt"};
        static const char* cn[] = {"fwd", "fwd-nt", "rev", "rev-nt"};
        printf("%4zu MB  after %-8s  consumer %-6s  %7.1f us  %6.2f TB/s\n", mb, pn[prod], cn[cons], ms * 1e3, bytes / (ms * 1e-3) / 1e12);
      }
    }
  }
  return 0;
}
