// Throughput of returning global atomics on a histogram of 8160 counters against the STRIDE between counters
// (contiguous ints, one per 64 B, per 256 B, per 4 KiB): does spreading them over more memory channels help?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <bool RET>
__global__ void k_hist(const int *__restrict__ idx, int n, int *__restrict__ counters, int stride, int *__restrict__ sink) {
  int acc = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (RET) acc += atomicAdd(counters + (size_t)idx[i] * stride, 1);
    else atomicAdd(counters + (size_t)idx[i] * stride, 1);
  }
  if (RET && acc == 0x7fffffff) sink[0] = acc;
}

int main() {
  const int n = 360000, nc = 8160;
  std::vector<int> h(n);
  unsigned s = 1;
  for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s >> 8) % nc; }
  int *idx, *cnt, *sink;
  const size_t words = (size_t)nc * 1024 + 1024;
  CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&cnt, words * 4)); CK(hipMalloc(&sink, 4));
  CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int ret = 0; ret < 2; ++ret)
    for (int stride : {1, 2, 4, 16, 32, 64, 256, 1024}) {
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemset(cnt, 0, words * 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        if (ret) hipLaunchKernelGGL(k_hist<true>, dim3(1568), dim3(256), 0, 0, idx, n, cnt, stride, sink);
        else hipLaunchKernelGGL(k_hist<false>, dim3(1568), dim3(256), 0, 0, idx, n, cnt, stride, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("%s  stride %5d ints (%6d B)  %7.1f us  -> %5.1f G atomics/s\n", ret ? "returning    " : "not returning", stride, stride * 4, best * 1e3, n / (best * 1e-3) / 1e9);
    }
  return 0;
}
