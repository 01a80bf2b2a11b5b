// Returning global atomics on a SKEWED histogram: 246k adds into the 20 x 17 tiles in the middle of a 120 x 68 tile grid (what
// bench.py --cloud-scale 0.2 does to the binning counters), against where the counters are put: row-major (16 per 64-byte
// line), one per line, the low four index bits moved to the top, 4 / 8 / 16 sub-counters per tile chosen by the low bits of
// the Gaussian's index, and the tile index multiplied by an odd constant modulo the tile count (neighbouring tiles 268 B /
// 1 KB / 4 KB apart: different memory channels), or runs of 16 / 8 / 4 neighbouring counters kept together and the runs spread.
// Third access pattern: every wave bumps an 8 x 8 block of neighbouring tiles (a large splat binned by one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_hist(const int *__restrict__ idx, int n, int *__restrict__ counters, int *__restrict__ sink) {
  int acc = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += atomicAdd(counters + idx[i], 1);
  if (acc == 0x7fffffff) sink[0] = acc;
}

static const char *kNames[] = {"row-major (16 per line)", "one counter per line", "low 4 bits to the top", "4 sub-counters per tile",
                               "8 sub-counters per tile", "16 sub-counters per tile", "t * 67 mod M", "t * 257 mod M",
                               "t * 1031 mod M", "t * 1031 mod M, 8 subs", "t * 4099 mod M", "lines of 16 * 1031", "groups of 8 * 1031", "groups of 4 * 1031"};
static int place(int layout, int t, int i, int M) {
  switch (layout) {
    case 0: return t;
    case 1: return t * 16;
    case 2: return (t & 15) * (M >> 4) + (t >> 4);
    case 3: return t * 4 + (i & 3);
    case 4: return t * 8 + (i & 7);
    case 5: return t * 16 + (i & 15);
    case 6: return (int)(((long long)t * 67) % M);
    case 7: return (int)(((long long)t * 257) % M);
    case 8: return (int)(((long long)t * 1031) % M);
    case 9: return (int)(((long long)t * 1031) % M) * 8 + (i & 7);
    case 10: return (int)(((long long)t * 4099) % M);
    case 11: return (int)(((long long)(t >> 4) * 1031) % (M >> 4)) * 16 + (t & 15);
    case 12: return (int)(((long long)(t >> 3) * 1031) % (M >> 3)) * 8 + (t & 7);
    default: return (int)(((long long)(t >> 2) * 1031) % (M >> 2)) * 4 + (t & 3);
  }
}

int main() {
  const int n = 246000, tw = 120, th = 68, M = tw * th;
  int *idx, *cnt, *sink;
  CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&cnt, (size_t)M * 16 * 4)); CK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int layout = 0; layout < 14; ++layout)
    for (int uniform = 0; uniform < 3; ++uniform) {
      std::vector<int> h(n);
      unsigned s = 1;
      for (int i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        const unsigned r = s >> 8;
        int t = uniform ? (int)(r % M) : (25 + (int)((r >> 10) % 17)) * tw + 50 + (int)(r % 20);
        if (uniform == 2) {   // what a large splat does: the 64 lanes of a wave bump an 8 x 8 block of neighbouring tiles
          static int base;
          if ((i & 63) == 0) base = (int)((r >> 7) % (th - 8)) * tw + (int)(r % (tw - 8));
          t = base + ((i & 63) >> 3) * tw + (i & 7);
        }
        h[i] = place(layout, t, i, M);
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
      printf("%-28s %-8s %7.1f us\n", kNames[layout], uniform == 2 ? "8x8 blocks" : (uniform ? "uniform" : "skewed"), best * 1e3);
    }
  return 0;
}
