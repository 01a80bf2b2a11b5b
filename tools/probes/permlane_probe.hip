// Probe of v_permlane16_swap / v_permlane32_swap semantics on gfx950 (prints what each lane receives).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
  const unsigned lane = threadIdx.x;
  unsigned a = lane, b = lane + 100;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[lane] = r[0]; out[64 + lane] = r[1];
  auto r2 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[128 + lane] = r2[0]; out[192 + lane] = r2[1];
  unsigned c = lane;
  auto r3 = __builtin_amdgcn_permlane16_swap(c, c, false, false);
  out[256 + lane] = r3[0]; out[320 + lane] = r3[1];
}
int main() {
  unsigned *d; hipMalloc(&d, 384 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *names[] = {"p16 r0(a=lane,b=lane+100)", "p16 r1", "p32 r0", "p32 r1", "p16 same r0", "p16 same r1"};
  for (int s = 0; s < 6; ++s) { printf("%s:\n", names[s]); for (int i = 0; i < 64; ++i) printf("%d ", h[64 * s + i]); printf("\n"); }
  return 0;
}
