// Micro-benchmark of the L1+SSIM kernels through the C ABI: the forward/backward pair and the single fused kernel
// (shapes chosen at compile time: -DSO_SSIM_THREADS / -DSO_SSIM_ROWS / -DSO_SSIM_WAVES / -DSO_FUSED_TAPGROUP).  Prints mean launch time and checksums at 1080p.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/splat_one_amd.h"


#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
  const int B = 1, H = argc > 2 ? atoi(argv[2]) : 1080, W = argc > 1 ? atoi(argv[1]) : 1920, CH = 3, iters = 50;
  const size_t n = (size_t)B * H * W * CH;
  std::vector<float> a(n), c(n);
  unsigned s = 12345u;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u; a[i] = (s >> 8) * (1.f / 16777216.f);
    s = s * 1664525u + 1013904223u; c[i] = 0.7f * a[i] + 0.3f * (s >> 8) * (1.f / 16777216.f);
  }
  float *d1, *d2, *dm, *dv, *sums, *loss;
  CK(hipMalloc(&d1, n * 4)); CK(hipMalloc(&d2, n * 4)); CK(hipMalloc(&dm, 3 * n * 4)); CK(hipMalloc(&dv, n * 4));
  CK(hipMalloc(&sums, 64)); CK(hipMalloc(&loss, 12));
  CK(hipMemcpy(d1, a.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d2, c.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  float tf = 0, tb = 0;
  for (int it = 0; it < iters + 5; ++it) {
    CK(hipMemsetAsync(sums, 0, 64, 0));
    CK(hipEventRecord(e0, 0));
    if (so_ssim_l1_fwd(B, H, W, CH, d1, d2, 1, sums, dm, nullptr)) { printf("fwd: %s\n", so_last_error()); return 1; }
    CK(hipEventRecord(e1, 0));
    if (so_ssim_l1_bwd(B, H, W, CH, d1, d2, dm, 0.8f / n, -0.2f / ((H - 10.f) * (W - 10.f) * CH), nullptr, dv, sums, loss, 1, 0.2f, nullptr)) { printf("bwd: %s\n", so_last_error()); return 1; }
    CK(hipEventRecord(e2, 0));
    CK(hipEventSynchronize(e2));
    float f, b;
    CK(hipEventElapsedTime(&f, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
    if (it >= 5) { tf += f; tb += b; }
  }
  std::vector<float> v(n);
  float hs[16], hl[3];
  CK(hipMemcpy(v.data(), dv, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hs, sums, 64, hipMemcpyDeviceToHost));
  if (hs[9] > 0) printf("stamps per wave (s_memtime ticks): barrier %.0f  stage %.0f  horizontal %.0f  vertical+epilogue %.0f | wave total %.0f  (waves %.0f)\n",
                        hs[4] / hs[9], hs[5] / hs[9], hs[6] / hs[9], hs[7] / hs[9], hs[8] / hs[9], hs[9]); CK(hipMemcpy(hl, loss, 12, hipMemcpyDeviceToHost));
  double cs = 0, ca = 0;
  for (size_t i = 0; i < n; ++i) { cs += v[i] * (double)((i % 977) + 1); ca += v[i] < 0 ? -v[i] : v[i]; }
  printf("%dx%d fwd %.1f us  bwd %.1f us  | sums %.3f %.3f loss %.7f %.7f %.7f | vsum %.9e vabs %.9e\n", W, H,
         tf / iters * 1e3, tb / iters * 1e3, hs[0], hs[1], hl[0], hl[1], hl[2], cs, ca);
  for (int rows : {0, 45, 52, 54, 57, 60, 68, 72, 90, 108}) {   // the single kernel on the same inputs, rows per workgroup swept (0 = the launcher's choice)
    float *dv2, *sums2;
    CK(hipMalloc(&dv2, n * 4)); CK(hipMalloc(&sums2, 64));
    float t = 0;
    for (int it = 0; it < iters + 5; ++it) {
      CK(hipMemsetAsync(sums2, 0, 64, 0));
      CK(hipEventRecord(e0, 0));
      if (so_ssim_l1_fused(B, H, W, CH, d1, d2, 1, 0.8f / n, -0.2f / ((H - 10.f) * (W - 10.f) * CH), nullptr, sums2, dv2, getenv("SSIM_BENCH_NO_LOSS_OUT") ? nullptr : sums2 + 2, (int32_t *)(sums2 + 5), 0.2f, rows, nullptr)) { printf("fused: %s\n", so_last_error()); return 1; }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float f;
      CK(hipEventElapsedTime(&f, e0, e1));
      if (it >= 5) t += f;
    }
    std::vector<float> v2(n);
    float hs2[16];
    CK(hipMemcpy(v2.data(), dv2, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hs2, sums2, 64, hipMemcpyDeviceToHost));
    double md = 0, mv = 0;
    size_t nd = 0;
    for (size_t i = 0; i < n; ++i) {
      const double dd = v2[i] > v[i] ? v2[i] - v[i] : v[i] - v2[i];
      if (dd > md) md = dd;
      if (dd != 0) ++nd;
      const double av = v[i] < 0 ? -v[i] : v[i];
      if (av > mv) mv = av;
    }
    printf("%dx%d rows %3d fused %.1f us (two kernels %.1f us) | sums %.3f %.3f | gradient: max |diff| %.3e of max %.3e, %zu of %zu values differ\n",
           W, H, rows, t / iters * 1e3, (tf + tb) / iters * 1e3, hs2[0], hs2[1], md, mv, nd, n);
    CK(hipFree(dv2)); CK(hipFree(sums2));
  }
  return 0;
}
