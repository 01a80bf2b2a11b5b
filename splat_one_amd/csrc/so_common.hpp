// so_common.hpp -- shared host/device helpers for the gfx950 kernels of libsplat_one_amd.so
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/splat_one_amd.h"

namespace so {

// thread-local error message (defined in common.hip)
void set_error(const char *fmt, ...);

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return SO_ERR_LAUNCH;
  }
  return SO_OK;
}

#define SO_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      ::so::set_error(__VA_ARGS__);      \
      return SO_ERR_INVALID_ARG;         \
    }                                    \
  } while (0)

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;  // CDNA wavefront

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------
// wave64 reductions with DPP (no LDS traffic).  All 64 lanes must be active (EXEC all ones).
// Result is valid in lane 63 after wave_reduce_sum_to_last(); wave_reduce_sum() broadcasts it.
// ---------------------------------------------------------------------------------------------
#if defined(__HIPCC__)
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND_CTRL = false>
__device__ __forceinline__ float dpp_mov(float old, float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                               __builtin_bit_cast(int, v), CTRL, ROW_MASK,
                                                               BANK_MASK, BOUND_CTRL));
#else
  (void)old;
  return v;
#endif
}

// Sum over the 64 lanes; the total lands in lane 63.
__device__ __forceinline__ float wave_reduce_sum_to_last(float v) {
  // row_shr:1,2,3 within rows of 16 (classic GCN reduction), then row_shr:4 / 8, then row_bcast
  v += dpp_mov<0x111, 0xf, 0xf, true>(0.f, v);  // row_shr:1
  v += dpp_mov<0x112, 0xf, 0xf, true>(0.f, v);  // row_shr:2
  v += dpp_mov<0x114, 0xf, 0xe, true>(0.f, v);  // row_shr:4  (bank_mask 0xe)
  v += dpp_mov<0x118, 0xf, 0xc, true>(0.f, v);  // row_shr:8  (bank_mask 0xc)
  v += dpp_mov<0x142, 0xa, 0xf, true>(0.f, v);  // row_bcast:15 (row_mask 0xa)
  v += dpp_mov<0x143, 0xc, 0xf, true>(0.f, v);  // row_bcast:31 (row_mask 0xc)
  return v;
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
  v = wave_reduce_sum_to_last(v);
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#else
  return v;
#endif
}

__device__ __forceinline__ int lane_id() {
#if defined(__HIP_DEVICE_COMPILE__)
  return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#else
  return 0;
#endif
}
#endif

}  // namespace so
