// Throughput of global int atomics by memory scope on MI355X: 'agent' (what atomicAdd gives) against
// 'workgroup' (resolved in the XCD's own L2), on a histogram-like pattern: 600k adds over 8160 counters,
// privatised per XCD for the workgroup-scope variant (xcc id from the hardware register).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ int xcc_id() {
  // HW_REG_XCC_ID (id 20), bits [3:0]
  return __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
}

template <int SCOPE, bool RET>
__global__ void k_hist(const int *__restrict__ idx, int n, int *__restrict__ counters, int n_counters, int *__restrict__ sink,
                       int *__restrict__ xcc_seen) {
  const int x = xcc_id();
  if (threadIdx.x == 0) atomicOr(xcc_seen, 1 << x);
  int *base = SCOPE == __HIP_MEMORY_SCOPE_AGENT ? counters : counters + (size_t)x * n_counters;
  int acc = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (RET) acc += __hip_atomic_fetch_add(base + idx[i], 1, __ATOMIC_RELAXED, SCOPE);
    else (void)__hip_atomic_fetch_add(base + idx[i], 1, __ATOMIC_RELAXED, SCOPE);
  }
  if (RET && acc == 0x7fffffff) sink[0] = acc;
}

int main() {
  const int n = 600000, nc = 8160;
  std::vector<int> h(n);
  unsigned s = 1;
  for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s >> 8) % nc; }
  int *idx, *cnt, *sink, *seen;
  CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&cnt, 16 * nc * 4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&seen, 4));
  CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char *name, void (*k)(const int *, int, int *, int, int *, int *), bool priv) -> int {
    float best = 1e9f;
    long total = 0;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipMemset(cnt, 0, 16 * nc * 4)); CK(hipMemset(seen, 0, 4));
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k, dim3(1568), dim3(256), 0, 0, idx, n, cnt, nc, sink, seen);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
      std::vector<int> c(16 * nc);
      CK(hipMemcpy(c.data(), cnt, 16 * nc * 4, hipMemcpyDeviceToHost));
      total = 0;
      for (int v : c) total += v;
    }
    int hs; CK(hipMemcpy(&hs, seen, 4, hipMemcpyDeviceToHost));
    printf("%-44s %7.1f us   sum of counters %ld (expected %d)   xcc ids seen mask 0x%x\n", name, best * 1e3, total, n, hs);
    return 0;
  };
  run("agent scope, no return (atomicAdd)", k_hist<__HIP_MEMORY_SCOPE_AGENT, false>, false);
  run("agent scope, returning", k_hist<__HIP_MEMORY_SCOPE_AGENT, true>, false);
  run("workgroup scope, per-XCD copy, no return", k_hist<__HIP_MEMORY_SCOPE_WORKGROUP, false>, true);
  run("workgroup scope, per-XCD copy, returning", k_hist<__HIP_MEMORY_SCOPE_WORKGROUP, true>, true);
  return 0;
}
