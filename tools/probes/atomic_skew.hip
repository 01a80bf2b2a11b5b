// Returning global atomics on a SKEWED histogram: 246k adds into the 20 x 17 tiles in the middle of a 120 x 68 tile grid (what
// bench.py --cloud-scale 0.2 does to the binning counters), with the counters laid out row-major (16 per 64-byte line), one per
// line, and row-major with the low four index bits moved to the top (neighbouring tiles 2 KB apart).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_hist(const int *__restrict__ idx, int n, int *__restrict__ counters, int *__restrict__ sink) {
  int acc = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += atomicAdd(counters + idx[i], 1);
  if (acc == 0x7fffffff) sink[0] = acc;
}

int main() {
  const int n = 246000, tw = 120, th = 68, M = tw * th;
  int *idx, *cnt, *sink;
  CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&cnt, (size_t)M * 16 * 4)); CK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int layout = 0; layout < 3; ++layout)
    for (int uniform = 0; uniform < 2; ++uniform) {
      std::vector<int> h(n);
      unsigned s = 1;
      for (int i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        const unsigned r = s >> 8;
        const int t = uniform ? (int)(r % M) : (25 + (int)((r >> 10) % 17)) * tw + 50 + (int)(r % 20);
        h[i] = layout == 0 ? t : (layout == 1 ? t * 16 : (t & 15) * (M >> 4) + (t >> 4));
      }
      CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemset(cnt, 0, (size_t)M * 16 * 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_hist, dim3(1568), dim3(256), 0, 0, idx, n, cnt, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("%-28s %-8s %7.1f us\n", layout == 0 ? "row-major (16 per line)" : (layout == 1 ? "one counter per line" : "low 4 bits to the top"),
             uniform ? "uniform" : "skewed", best * 1e3);
    }
  return 0;
}
