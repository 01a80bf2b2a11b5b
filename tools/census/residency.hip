// How many 256-thread workgroups does a CU of the MI355X admit at once, by VGPR count and dynamic LDS size?
// Every workgroup stamps its start (s_memrealtime, 100 MHz), spins ~20 us, exits; workgroups that start late were queued.
//   hipcc --offload-arch=gfx950 -O3 residency.hip -o residency && ./residency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int REGS>
__global__ void __launch_bounds__(256) k_census(unsigned long long *starts, int spin_ticks) {
  extern __shared__ float lds[];
  const unsigned long long t0 = __builtin_readcyclecounter() * 0 + __builtin_amdgcn_s_memrealtime();
  if (REGS >= 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
  else if (REGS >= 168) asm volatile("v_mov_b32 v167, 0" ::: "v167");
  else if (REGS >= 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if (threadIdx.x == 0) { starts[blockIdx.x] = t0; lds[0] = 1.f; }
  while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < spin_ticks) __builtin_amdgcn_s_sleep(8);
}

template <int REGS>
void run(size_t lds, int grid) {
  unsigned long long *d;
  hipMalloc(&d, grid * sizeof(*d));
  if (lds > 64 * 1024) hipFuncSetAttribute((const void *)k_census<REGS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_census<REGS>, dim3(grid), dim3(256), lds, 0, d, 2000);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(grid);
  hipMemcpy(h.data(), d, grid * sizeof(*d), hipMemcpyDeviceToHost);
  const unsigned long long t0 = *std::min_element(h.begin(), h.end());
  int early = 0;
  for (auto v : h) early += (v - t0) < 500;   // within 5 us of the first
  printf("regs %3d  lds %6zu B  grid %4d: %4d workgroups started within 5 us (%.2f per CU)\n", REGS, lds, grid, early, early / 256.0);
  hipFree(d);
}

int main() {
  for (size_t lds : {(size_t)0, (size_t)16384, (size_t)32768, (size_t)46080, (size_t)65536}) {
    run<64>(lds, 2048);
    run<128>(lds, 2048);
    run<168>(lds, 2048);
    run<256>(lds, 2048);
  }
  return 0;
}
