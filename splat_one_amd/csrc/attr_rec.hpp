// attr_rec.hpp -- where the preprocess kernels read the per-Gaussian attributes from.
//
// AttrSoA : the six float32 tensors the reference keeps in its ParameterDict
//           (/root/reference/utils/gsplat_utils/gsplat_trainer.py:246-257), read in place.
// AttrRec : the float16 attribute record of BASELINE.json configs[4] ("fp16 attributes"): the 56 of 59
//           parameter floats that tolerate it -- quaternion, log-scale and all SH coefficients -- stored as
//           halves in ONE 16-byte-aligned row per Gaussian,
//               +0  f16 quat[4]   +8  f16 log_scale[3]   +14 f16 0   +16 f16 sh[K][3] (sh0 then shN), zero padded
//               row stride = 16 + roundup16(6 K) bytes  (K = 16: 112 B instead of 224 B in six arrays)
//           so a lane fetches its Gaussian with 7 x 16-byte loads of consecutive addresses.  Positions and
//           opacity logits stay float32 in their own arrays (a half has 2e-3 world units of resolution at |x| = 3:
//           several pixels at 1080p).  The float32 masters stay authoritative: the optimiser writes the halves
//           next to its float32 update (adam.hip), so_attr_pack_f16 rebuilds them after densification.
// Arithmetic is float32 in both cases; a half is widened on load.
#pragma once
#include "so_common.hpp"

namespace so {

__host__ __device__ inline int attr_rec_stride_bytes(int K) { return 16 + ((6 * K + 15) & ~15); }

__device__ __forceinline__ float half_lo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)); }
__device__ __forceinline__ float half_hi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }
__device__ __forceinline__ unsigned short f2h_bits(float x) { return __builtin_bit_cast(unsigned short, (_Float16)x); }
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) { return (uint32_t)f2h_bits(lo) | ((uint32_t)f2h_bits(hi) << 16); }

struct AttrSoA {
  const float *log_scales, *quats, *sh0, *shN;
  int K;
  static constexpr bool kActivated = false;   // log-scales / opacity logits: the kernels apply exp / sigmoid
  static constexpr bool kHalfRows = false;
  template <int DEG> struct Coefs {
    const float *c0, *cN;
    __device__ __forceinline__ void get(int k, float c[3]) const {
#ifdef PP_NO_SHN_LOAD
      if (k > 0) { c[0] = 0.1f * k; c[1] = 0.2f; c[2] = 0.3f; return; }
#endif
      const float *cf = (k == 0) ? c0 : cN + 3 * (k - 1);
      c[0] = cf[0]; c[1] = cf[1]; c[2] = cf[2];
    }
  };
  __device__ __forceinline__ void base(int64_t n, float q[4], float ls[3]) const {
    const float4 qq = *reinterpret_cast<const float4 *>(quats + 4 * n);
    q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
    ls[0] = log_scales[3 * n]; ls[1] = log_scales[3 * n + 1]; ls[2] = log_scales[3 * n + 2];
  }
  template <int DEG> __device__ __forceinline__ Coefs<DEG> coefs(int64_t n) const {
    return Coefs<DEG>{sh0 + 3 * n, shN + n * (int64_t)(K - 1) * 3};
  }
};

// AttrAct : what `gsplat.rendering.rasterization` is HANDED at gsplat_trainer.py:477-494 -- post-activation tensors:
//           scales = exp(log_scales) [N,3], opacities = sigmoid(logits) [N] (passed in the kernels' opacity-logit
//           argument), quats [N,4], and ONE coefficient tensor colors = cat(sh0, shN) [N,K,3].  The kernels then skip exp /
//           sigmoid and return the gradients of the activated values; regularisers do not apply (the caller's autograd owns
//           the activations).
struct AttrAct {
  const float *scales, *quats, *coeffs;
  int K;
  static constexpr bool kActivated = true;
  static constexpr bool kHalfRows = false;
  template <int DEG> struct Coefs {
    const float *c;
    __device__ __forceinline__ void get(int k, float o[3]) const { o[0] = c[3 * k]; o[1] = c[3 * k + 1]; o[2] = c[3 * k + 2]; }
  };
  __device__ __forceinline__ void base(int64_t n, float q[4], float ls[3]) const {
    const float4 qq = *reinterpret_cast<const float4 *>(quats + 4 * n);
    q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
    ls[0] = scales[3 * n]; ls[1] = scales[3 * n + 1]; ls[2] = scales[3 * n + 2];
  }
  template <int DEG> __device__ __forceinline__ Coefs<DEG> coefs(int64_t n) const { return Coefs<DEG>{coeffs + n * (int64_t)K * 3}; }
};

struct AttrRec {
  const uint4 *rec;
  int stride16;   // row stride in 16-byte units
  static constexpr bool kActivated = false;
  static constexpr bool kHalfRows = true;     // a fused optimiser step re-packs the rows it read (AdamFuse::half_rows)
  template <int DEG> struct Coefs {
    static constexpr int NH = 3 * (DEG + 1) * (DEG + 1);   // halves used
    static constexpr int NQ = (2 * NH + 15) / 16;           // 16-byte loads
    uint32_t w[4 * NQ];
    __device__ __forceinline__ void get(int k, float c[3]) const {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int i = 3 * k + j;
        c[j] = (i & 1) ? half_hi(w[i >> 1]) : half_lo(w[i >> 1]);
      }
    }
  };
  __device__ __forceinline__ void base(int64_t n, float q[4], float ls[3]) const {
    const uint4 a = rec[n * stride16];
    q[0] = half_lo(a.x); q[1] = half_hi(a.x); q[2] = half_lo(a.y); q[3] = half_hi(a.y);
    ls[0] = half_lo(a.z); ls[1] = half_hi(a.z); ls[2] = half_lo(a.w);
  }
  template <int DEG> __device__ __forceinline__ Coefs<DEG> coefs(int64_t n) const {
    Coefs<DEG> c;
    const uint4 *row = rec + n * stride16 + 1;
#pragma unroll
    for (int t = 0; t < Coefs<DEG>::NQ; ++t) {
      const uint4 v = row[t];
      c.w[4 * t] = v.x; c.w[4 * t + 1] = v.y; c.w[4 * t + 2] = v.z; c.w[4 * t + 3] = v.w;
    }
    return c;
  }
};

}  // namespace so
