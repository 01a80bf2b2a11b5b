// Returning atomics on tile counters: device scope on counters shared by the whole chip (what k_preprocess_fwd's binning does)
// against workgroup scope on a private copy of the counters per XCD (the atomic then completes in the XCD's own L2).
//   hipcc --offload-arch=gfx950 -O3 xcd_atomics.hip -o xcd_atomics && ./xcd_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

// every lane: `per` returning atomics on pseudo-random counters; MODE 0 = device scope, shared counters; 1 = device scope, per-XCD
// copy; 2 = workgroup scope, per-XCD copy
template <int MODE>
__global__ void __launch_bounds__(256) k_bin(int32_t *counters, int n_tiles, int per, int32_t *sink, unsigned *xcc_seen) {
  const unsigned x = xcc_id();
  if (threadIdx.x == 0) atomicOr(xcc_seen + (blockIdx.x & 7), 1u << x);
  int32_t *mine = MODE == 0 ? counters : counters + (size_t)x * n_tiles;
  unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  int32_t acc = 0;
  for (int i = 0; i < per; ++i) {
    s = s * 1664525u + 1013904223u;
    int32_t *p = mine + (s >> 8) % (unsigned)n_tiles;
    if (MODE == 3) {
      unsigned long long *p64 = reinterpret_cast<unsigned long long *>(counters) + ((s >> 8) % (unsigned)n_tiles) / 2;
      const unsigned long long o = __hip_atomic_fetch_add(p64, 0x100000001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc += (int32_t)o + (int32_t)(o >> 32);
      continue;
    }
    if (MODE == 2) acc += __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else acc += __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (acc == 0x7fffffff) sink[0] = acc;
}

__global__ void k_sum(const int32_t *counters, int n, unsigned long long *out) {
  unsigned long long s = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += (unsigned long long)counters[i];
  atomicAdd(out, s);
}

template <int MODE>
void run(const char *name, int n_gauss, int per, int n_tiles) {
  int32_t *c, *sink; unsigned *seen; unsigned long long *tot;
  hipMalloc(&c, (size_t)8 * n_tiles * 4); hipMalloc(&sink, 4); hipMalloc(&seen, 32); hipMalloc(&tot, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipMemset(c, 0, (size_t)8 * n_tiles * 4); hipMemset(seen, 0, 32); hipMemset(tot, 0, 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_bin<MODE>, dim3((n_gauss + 255) / 256), dim3(256), 0, 0, c, n_tiles, per, sink, seen);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  hipLaunchKernelGGL(k_sum, dim3(1), dim3(1024), 0, 0, c, 8 * n_tiles, tot);
  unsigned long long h; hipMemcpy(&h, tot, 8, hipMemcpyDeviceToHost);
  unsigned hs[8]; hipMemcpy(hs, seen, 32, hipMemcpyDeviceToHost);
  const long long total = (long long)((n_gauss + 255) / 256) * 256 * per * (MODE == 3 ? 2 : 1);
  printf("%-34s %8d lanes x %3d on %6d tiles: %8.1f us  %6.1f G atomics/s  sum %llu of %lld %s  xcc masks by blockIdx&7:", name, n_gauss, per, n_tiles,
         best * 1e3, total / (best * 1e-3) * 1e-9, h, total, h == (unsigned long long)total ? "OK" : "LOST UPDATES");
  for (int i = 0; i < 8; ++i) printf(" %02x", hs[i]);
  printf("\n");
  hipFree(c); hipFree(sink); hipFree(seen); hipFree(tot);
}

int main() {
  // few tiles (a 512 x 512 image: 1024): the atomics of a line serialise, and a private copy of the counters per XCD (= 8 replicas) spreads them
  for (auto cfg : {std::pair<int, int>{60000, 9}, {1000000, 1}}) {
    run<0>("1024 tiles, shared counters", cfg.first, cfg.second, 1024);
    run<1>("1024 tiles, 8 copies (one per XCD)", cfg.first, cfg.second, 1024);
    run<0>("2040 tiles, shared counters", cfg.first, cfg.second, 2040);
    run<1>("2040 tiles, 8 copies (one per XCD)", cfg.first, cfg.second, 2040);
  }
  for (auto cfg : {std::pair<int, int>{100000, 4}, {100000, 42}, {1700000, 3}}) {
    run<0>("device scope, shared counters", cfg.first, cfg.second, 8160);
    run<1>("device scope, per-XCD counters", cfg.first, cfg.second, 8160);
    run<2>("workgroup scope, per-XCD counters", cfg.first, cfg.second, 8160);
    run<3>("device scope, 64-bit on a pair (x2)", cfg.first, cfg.second / 2 > 0 ? cfg.second / 2 : 1, 8160);
  }
  return 0;
}
