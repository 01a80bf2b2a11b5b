// Issue-rate calibration for gfx950: ns per wave-instruction per SIMD for a few instruction kinds,
// at 1 / 2 / 4 / 8 waves per SIMD.  Every kernel runs the same number of inline-asm instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int kIters = 4096, kUnroll = 8;   // instructions per wave = kIters * kUnroll

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

__global__ void k_fma(float *out, float x, float y) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  for (int it = 0; it < kIters; ++it) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
    REP8(X)
#undef X
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma_dep(float *out, float x, float y) {
  float a = threadIdx.x;
  for (int it = 0; it < kIters; ++it) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
    REP8(X)
#undef X
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k_pkfma(float *out, float x, float y) {
  v2f a[8], xx = {x, y}, yy = {y, x};
  for (int i = 0; i < 8; ++i) a[i] = v2f{(float)threadIdx.x, (float)i};
  for (int it = 0; it < kIters; ++it) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(xx), "v"(yy));
    REP8(X)
#undef X
  }
  v2f s = {0, 0}; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ void k_pkfma_dep(float *out, float x, float y) {
  v2f a = {(float)threadIdx.x, 1.f}, xx = {x, y}, yy = {y, x};
  for (int it = 0; it < kIters; ++it) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(xx), "v"(yy));
    REP8(X)
#undef X
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a.x + a.y;
}
__global__ void k_pkfma_sgpr(float *out, float x, float y) {   // weight from an SGPR pair with op_sel broadcast
  v2f a[8], xx = {x, y};
  for (int i = 0; i < 8; ++i) a[i] = v2f{(float)threadIdx.x, (float)i};
  for (int it = 0; it < kIters; ++it) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "s"(xx), "v"(a[(i + 1) & 7]));
    REP8(X)
#undef X
  }
  v2f s = {0, 0}; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ void k_mad64(float *out, float x, float y) {
  unsigned long long a[8]; unsigned b = threadIdx.x + 3, c = threadIdx.x * 7 + 1;
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  for (int it = 0; it < kIters; ++it) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
    REP8(X)
#undef X
  }
  unsigned long long s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}
__global__ void k_rcp(float *out, float x, float y) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i + 1.5f;
  for (int it = 0; it < kIters; ++it) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
    REP8(X)
#undef X
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_exp(float *out, float x, float y) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < kIters; ++it) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
    REP8(X)
#undef X
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma_salu(float *out, float x, float y) {   // 4 VALU + 4 SALU interleaved
  float a[4]; int s0 = blockIdx.x, s1 = 3;
  for (int i = 0; i < 4; ++i) a[i] = threadIdx.x + i;
  for (int it = 0; it < kIters; ++it) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i & 3]) : "v"(x), "v"(y)); asm volatile("s_add_i32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
    X(0) X(1) X(2) X(3)
#undef X
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a[0] + a[1] + a[2] + a[3] + s0;
}
__global__ void k_dpp(float *out, float x, float y) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  for (int it = 0; it < kIters; ++it) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
    REP8(X)
#undef X
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_ldsb64(float *out, float x, float y) {
  __shared__ v2f buf[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = v2f{(float)i, x};
  __syncthreads();
  v2f a[8];
  for (int i = 0; i < 8; ++i) a[i] = v2f{0, 0};
  unsigned addr = threadIdx.x * 8;
  for (int it = 0; it < kIters; ++it) {
#define X(i) asm volatile("ds_read_b64 %0, %1 offset:" #i "*24" : "=v"(a[i]) : "v"(addr));
    REP8(X)
#undef X
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  v2f s = {0, 0}; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ void k_ldsb32(float *out, float x, float y) {
  __shared__ float buf[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) buf[i] = i;
  __syncthreads();
  float a[8];
  unsigned addr = threadIdx.x * 4;
  for (int it = 0; it < kIters; ++it) {
#define X(i) asm volatile("ds_read_b32 %0, %1 offset:" #i "*12" : "=v"(a[i]) : "v"(addr));
    REP8(X)
#undef X
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v3f __attribute__((ext_vector_type(3)));
__global__ void k_ldsb128(float *out, float x, float y) {
  __shared__ v4f buf[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = v4f{(float)i, x, y, 1.f};
  __syncthreads();
  v4f a[8];
  unsigned addr = threadIdx.x * 16;
  for (int it = 0; it < kIters; ++it) {
#define X(i) asm volatile("ds_read_b128 %0, %1 offset:" #i "*48" : "=v"(a[i]) : "v"(addr));
    REP8(X)
#undef X
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  v4f s = {0, 0, 0, 0}; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
}
__global__ void k_ldsb96(float *out, float x, float y) {
  __shared__ float buf[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = i;
  __syncthreads();
  v3f a[8];
  unsigned addr = threadIdx.x * 12;
  for (int it = 0; it < kIters; ++it) {
#define X(i) asm volatile("ds_read_b96 %0, %1 offset:" #i "*36" : "=v"(a[i]) : "v"(addr));
    REP8(X)
#undef X
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  v3f s = {0, 0, 0}; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z;
}
__global__ void k_lds2b64(float *out, float x, float y) {
  __shared__ v2f buf[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) buf[i] = v2f{(float)i, x};
  __syncthreads();
  v4f a[8];
  unsigned addr = threadIdx.x * 8;
  for (int it = 0; it < kIters; ++it) {
#define X(i) asm volatile("ds_read2_b64 %0, %1 offset0:" #i "*6 offset1:" #i "*6+3" : "=v"(a[i]) : "v"(addr));
    REP8(X)
#undef X
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  v4f s = {0, 0, 0, 0}; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
}
__global__ void k_fma_nop(float *out, float x, float y) {   // does an s_nop cost an issue slot?
  float a[4];
  for (int i = 0; i < 4; ++i) a[i] = threadIdx.x + i;
  for (int it = 0; it < kIters; ++it) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i & 3]) : "v"(x), "v"(y)); asm volatile("v_mov_b32 %0, %0" : "+v"(a[(i + 2) & 3]));
    X(0) X(1) X(2) X(3)
#undef X
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a[0] + a[1] + a[2] + a[3];
}

typedef void (*kern_t)(float *, float, float);
int main() {
  float *out;
  CK(hipMalloc(&out, 1024 * 16 * 64 * 4));
  struct { const char *name; kern_t k; } ks[] = {
      {"v_fma_f32 (8 indep)", k_fma}, {"v_fma_f32 (dependent)", k_fma_dep}, {"v_pk_fma_f32 (8 indep)", k_pkfma},
      {"v_pk_fma_f32 (dependent)", k_pkfma_dep}, {"v_pk_fma_f32 sgpr+op_sel", k_pkfma_sgpr},
      {"v_mad_u64_u32", k_mad64}, {"v_rcp_f32", k_rcp}, {"v_exp_f32", k_exp},
      {"v_fma + s_add alternating", k_fma_salu}, {"v_add_f32_dpp quad_perm", k_dpp},
      {"ds_read_b64", k_ldsb64}, {"ds_read_b32", k_ldsb32}, {"ds_read_b96", k_ldsb96}, {"ds_read_b128", k_ldsb128},
      {"ds_read2_b64", k_lds2b64}, {"v_fma + v_mov alternating", k_fma_nop}};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-28s %10s %10s %10s %10s   (ns per wave-instruction per SIMD; 1024 SIMDs)\n", "kernel", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");
  for (auto &k : ks) {
    printf("%-28s", k.name);
    for (int wps : {1, 2, 4, 8}) {
      const int blocks = 1024 * wps;   // one wave per block; the dispatcher spreads them over all SIMDs
      hipLaunchKernelGGL(k.k, dim3(blocks), dim3(64), 0, 0, out, 1.0001f, 0.5f);
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k.k, dim3(blocks), dim3(64), 0, 0, out, 1.0001f, 0.5f);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf(" %10.3f", ms * 1e6 / ((double)kIters * kUnroll * wps));
    }
    printf("\n");
  }
  return 0;
}
