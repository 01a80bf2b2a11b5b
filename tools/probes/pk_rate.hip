// Issue rate and dependent-chain latency of packed float32 arithmetic on gfx950, next to the scalar forms:
// waves of 256-thread workgroups (one wave per SIMD each), the shader clock around unrolled loops of v_fma_f32 /
// v_pk_fma_f32 (independent accumulators or one dependent chain; plain operands, a broadcast select on one source, three
// distinct register-pair sources).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, long long *cyc, float s) {
  v2f a[8], b = {s, s * 1.5f}, c = {s * 0.25f, s * 0.75f};
  float f[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = v2f{(float)threadIdx.x + i, (float)i}; }
#pragma unroll
  for (int i = 0; i < 16; ++i) f[i] = (float)threadIdx.x * 0.5f + i;
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < 256; ++it) {
    if (MODE == 0) {            // 16 independent scalar FMAs
#pragma unroll
      for (int i = 0; i < 16; ++i) f[i] = __builtin_fmaf(f[i], b.x, c.x);
    } else if (MODE == 1) {     // 8 independent packed FMAs (= 16 float FMAs)
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], b, c);
    } else if (MODE == 2) {     // 16 dependent scalar FMAs
#pragma unroll
      for (int i = 0; i < 16; ++i) f[0] = __builtin_fmaf(f[0], b.x, c.x);
    } else if (MODE == 3) {     // 8 dependent packed FMAs
#pragma unroll
      for (int i = 0; i < 8; ++i) a[0] = __builtin_elementwise_fma(a[0], b, c);
    } else if (MODE == 4) {     // 8 independent packed FMAs, one source a broadcast of a scalar
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], v2f{b.y, b.y}, c);
    } else if (MODE == 5) {     // 8 independent packed multiplies
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = a[i] * b;
    } else if (MODE == 6) {     // 8 independent packed FMAs with three distinct register-pair sources
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[(i + 1) & 7], a[(i + 3) & 7], a[i]);
    } else {                    // 16 independent scalar FMAs with three distinct sources
#pragma unroll
      for (int i = 0; i < 16; ++i) f[i] = __builtin_fmaf(f[(i + 1) & 15], f[(i + 5) & 15], f[i]);
    }
  }
  const long long t1 = clock64();
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> static int run(const char *name, int waves_per_simd, int n_instr) {
  float *out; long long *cyc;
  const int blocks = 256 * waves_per_simd;   // 256-thread blocks: one wave per SIMD each
  CK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CK(hipMalloc(&cyc, blocks * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long h[4096];
  CK(hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost));
  double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[i];
  printf("%-58s waves/SIMD %d: %8.2f clock64 ticks per trip of %2d instructions; kernel %7.1f us\n", name, waves_per_simd, s / blocks / 256.0, n_instr, ms * 1e3);
  CK(hipFree(out)); CK(hipFree(cyc));
  return 0;
}

int main() {
  for (int w = 1; w <= 4; w *= 2) {
    if (run<0>("16 independent v_fma_f32", w, 16)) return 1;
    if (run<1>("8 independent v_pk_fma_f32", w, 8)) return 1;
    if (run<2>("16 dependent v_fma_f32", w, 16)) return 1;
    if (run<3>("8 dependent v_pk_fma_f32", w, 8)) return 1;
    if (run<4>("8 independent v_pk_fma_f32, one source broadcast", w, 8)) return 1;
    if (run<5>("8 independent v_pk_mul_f32", w, 8)) return 1;
    if (run<6>("8 independent v_pk_fma_f32, three distinct sources", w, 8)) return 1;
    if (run<7>("16 independent v_fma_f32, three distinct sources", w, 16)) return 1;
  }
  return 0;
}
