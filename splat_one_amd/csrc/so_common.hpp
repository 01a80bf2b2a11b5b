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

// Periodic images (include/splat_one_amd.h, SO_TILE_WRAP_*): the `tile_size` argument of the binning / rasteriser entry
// points carries, above its low byte, which cameras see an image that is periodic in x (the 360-degree panoramas of
// the spherical camera model: a Gaussian whose footprint crosses x = 0 continues at x = W).
__host__ __device__ inline int tile_size_of(int flags) { return flags & 0xFF; }
__host__ __device__ inline bool wrap_for(int flags, int c) {
  return (flags & SO_TILE_WRAP_ALL) != 0 || (c < 16 && ((flags >> (8 + c)) & 1) != 0);
}
// tile column of a VIRTUAL column x in [-tile_w, 2 tile_w): columns left of 0 / right of tile_w - 1 are the other end
__host__ __device__ inline int wrapx(int x, int tile_w) { return x < 0 ? x + tile_w : (x >= tile_w ? x - tile_w : x); }

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

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a fence over ALL address spaces: in front of its
// s_barrier hipcc waits for every outstanding global store / atomic of the wave (s_waitcnt vmcnt(0) -- stores and
// no-return atomics count in vmcnt on gfx9), i.e. a full L2 round trip per barrier wherever a kernel stores or adds to
// global memory between barriers and exchanges data through LDS alone (the fused SSIM step stores its output row right
// before the barrier; the rasteriser backward has its gradient atomics in flight at the batch boundary).
__device__ __forceinline__ void lds_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#endif
}

// The same for data that only travels between the lanes of ONE wave through LDS (no other wave reads it): order the wave's
// own LDS writes before its LDS reads, nothing else.
__device__ __forceinline__ void wave_lds_sync() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
#endif
}

// Maximum of an int over the 64 lanes as a wave-uniform value (scalar register): four rotations inside the rows, then the
// four row results read from lanes 0 / 16 / 32 / 48 -- 8 vector instructions and no LDS hardware, against 36 + six
// ds_bpermute for the __shfl_xor ladder hipcc builds.  EXEC must be all ones.
__device__ __forceinline__ int wave_max_i32(int x) {
#if defined(__HIP_DEVICE_COMPILE__)
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false));    // quad_perm:[1,0,3,2]
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false));    // quad_perm:[2,3,0,1]
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x124, 0xf, 0xf, false));   // row_ror:4
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false));   // row_ror:8
  const int a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16);
  const int c = __builtin_amdgcn_readlane(x, 32), d = __builtin_amdgcn_readlane(x, 48);
  return max(max(a, b), max(c, d));
#else
  return x;
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

// ---------------------------------------------------------------------------------------------
// Transposing butterfly reduction over the 16 lanes of each DPP row: 8 per-lane values are summed
// in 22 VALU instructions (vs 8 x 4 for eight independent row reductions) and the row total of
// slot s ends in the lanes l of that row with slot_of_lane(l) == s.  `rows_combine` then adds the four
// rows (two ds_bpermute shuffles on the ONE remaining value per lane), after which lanes 0..8 of the
// wave hold the nine totals and issue ONE atomic instruction with nine DISTINCT addresses inside one
// 64-byte record.  Measured on MI355X (tools/dbg_stamps.py): letting the four rows add their partial
// sums to the same addresses in one atomic instruction costs ~130 us per launch of the rasteriser
// backward (same-address lanes serialise in the memory-side atomic unit); distinct addresses in
// one line are almost free.  (v_permlane16/32_swap would do the row combine without LDS hardware,
// but hipcc 7.2 miscompiles `r[0] + r[1]` of that builtin.)  EXEC must be all ones.
//   xor1 / xor2 via quad_perm, rotate-4 / rotate-8 within the row via row_ror.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int slot_of_lane(int lane) {  // 4*b0 + 2*b1 + b2
  return ((lane & 1) << 2) | (lane & 2) | ((lane >> 2) & 1);
}

__device__ __forceinline__ float row_reduce8_transposed(const float (&v)[8], int lane) {
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0;
  float a[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float keep = b0 ? v[j + 4] : v[j], send = b0 ? v[j] : v[j + 4];
    a[j] = keep + dpp_mov<0xB1, 0xf, 0xf, true>(0.f, send);  // quad_perm:[1,0,3,2]
  }
  float c[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float keep = b1 ? a[j + 2] : a[j], send = b1 ? a[j] : a[j + 2];
    c[j] = keep + dpp_mov<0x4E, 0xf, 0xf, true>(0.f, send);  // quad_perm:[2,3,0,1]
  }
  const float keep = b2 ? c[1] : c[0], send = b2 ? c[0] : c[1];
  float e = keep + dpp_mov<0x124, 0xf, 0xf, true>(0.f, send);  // row_ror:4
  e += dpp_mov<0x128, 0xf, 0xf, true>(0.f, e);                 // row_ror:8
  return e;
}

// Adds the four DPP rows lane-wise: every lane l ends with v[l%16] + v[l%16+16] + v[l%16+32] + v[l%16+48].
// gfx950's v_permlane16_swap / v_permlane32_swap exchange the odd rows (upper half-wave) of one register
// with the even rows (lower half-wave) of another; with both operands holding v, the two results are
// (even rows duplicated) and (odd rows duplicated), whose sum is the pairwise row sum -- two VALU
// instructions per level and no LDS hardware.  Written as inline asm: the builtin's struct return is
// miscompiled by hipcc 7.2 when both members are added; the s_nop's are the wait states between a VALU
// write and a cross-lane read of the same register (as for DPP), which the compiler does not insert around
// inline asm -- without them the swap reads stale operands.  EXEC must be all ones.
// (v_mfma_f32_16x16x4_f32 with A = 1 does the same sum in one instruction, but its 8 passes plus the 10 wait
// states before the accumulator can be read measured slower: rasteriser backward 126 vs 112 us.)
__device__ __forceinline__ float rows_combine(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  v = a + b;
  a = v; b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
#else
  return v;
#endif
}

// Sum over the 16 lanes of each row, result in every lane of the row (4 VALU instructions).
__device__ __forceinline__ float row_allreduce_sum(float x) {
  x += dpp_mov<0xB1, 0xf, 0xf, true>(0.f, x);
  x += dpp_mov<0x4E, 0xf, 0xf, true>(0.f, x);
  x += dpp_mov<0x124, 0xf, 0xf, true>(0.f, x);
  x += dpp_mov<0x128, 0xf, 0xf, true>(0.f, x);
  return x;
}

// ---------------------------------------------------------------------------------------------
// Round 3: the whole 64-lane reduction of the rasteriser backward's nine sums as ONE transposing network, priced by
// which lane bits a select is free on.  A transposing stage on lane bit k halves the number of registers: the lanes
// with bit k clear keep the sum of value A, the others of value B.  On gfx950 the select costs nothing where the
// hardware has a write mask or a swap:
//   bit 2 (which quad of a row pair)  DPP bank_mask: row_shl:4 into quads 0,2 / row_shr:4 into quads 1,3 -- 2 per pair
//   bits 4, 5 (which row)             v_permlane16_swap / v_permlane32_swap ARE a select-exchange of two registers --
//                                     swap + add = 2 per pair (rows_combine above spends mov + swap + add on ONE)
//   bits 0, 1 (inside a quad)         no mask: 2 v_cndmask + 1 DPP add per pair -- so these are plain all-reduce steps
// Eight values: bit 2 (8 instructions) -> 4 registers, bit 4 (4) -> 2, bit 5 (2) -> 1, then plain sums over bits 0, 1, 3
// (3 DPP adds): 17.  The ninth (opacity) is a plain wave reduction, 4 DPP adds in the row + row_bcast:15 + row_bcast:31
// (6), whose last instruction writes straight into lanes 56..63 of the result: 23 vector instructions for the nine
// totals against 33 for row_reduce8_transposed + row_allreduce_sum + select + rows_combine.  Result register: lane
// 16 r + 4 q holds slot 2 r + q (r = 0..3, q = 0..1), lane 56 holds slot 8 (reduce9_slot_of_lane).  The s_nop's are the
// wait states between a vector write and a cross-lane read of the same register (inline asm: the compiler adds none);
// independent instructions fill most of them.  EXEC must be all ones.
// ---------------------------------------------------------------------------------------------
constexpr unsigned long long kReduce9Lanes = 0x0111001100110011ull;   // lanes 0,4, 16,20, 32,36, 48,52, 56
__device__ __forceinline__ int reduce9_slot_of_lane(int lane) {       // meaningful in the lanes of kReduce9Lanes
  const int row = lane >> 4, quad = (lane >> 2) & 3;
  return quad < 2 ? 2 * row + quad : 8;
}

__device__ __forceinline__ float wave_reduce9_scattered(const float (&v)[8], float e) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r0, r1, r2, r3;
  asm volatile(
      "s_nop 1\n\t"
      // bit 2: quads 0,2 keep the even value of each pair, quads 1,3 the odd one; opacity's in-row sums ride between
      "v_add_f32_dpp %0, %5, %5 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %0, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %7, %7 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %1, %8, %8 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %9, %9 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %2, %10, %10 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %4, %4, %4 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %11, %11 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %3, %12, %12 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      // bit 4: rows 0,2 keep r0 (r2), rows 1,3 keep r1 (r3)
      "s_nop 0\n\t"
      "v_permlane16_swap_b32 %0, %1\n\t"
      "v_permlane16_swap_b32 %2, %3\n\t"
      "s_nop 1\n\t"
      "v_add_f32 %0, %0, %1\n\t"
      "v_add_f32 %2, %2, %3\n\t"
      "v_add_f32_dpp %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      // bit 5: the lower half-wave keeps r0, the upper r2
      "s_nop 0\n\t"
      "v_permlane32_swap_b32 %0, %2\n\t"
      "s_nop 1\n\t"
      "v_add_f32 %0, %0, %2\n\t"
      // plain sums over bits 0, 1, 3
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      // opacity: row 3 += lane 31 (rows 0+1), written into quads 2,3 of row 3 of the result
      "v_add_f32_dpp %0, %4, %4 row_bcast:31 row_mask:0x8 bank_mask:0xc\n\t"
      "s_nop 1"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "+v"(e)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
  return r0;
#else
  return v[0] + e;
#endif
}

// General 4x4 inverse in double (adjugate / determinant) of one camera-to-world matrix.
__device__ __forceinline__ void camera_inverse_one(const float *__restrict__ src, float *__restrict__ dst) {
  double m[16], inv[16];
  for (int i = 0; i < 16; ++i) m[i] = (double)src[i];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  const double rdet = 1.0 / det;
  for (int i = 0; i < 16; ++i) dst[i] = (float)(inv[i] * rdet);
}

// Device-side Adam schedule (one thread per parameter group, a whole workgroup calls it): learning rate
// lr0 * gamma^step, bias corrections for step+1, then the step counter advances.  hyper[g] = (step size, sqrt(bc2)).
struct AdamSched {
  float lr0[8];        // SO_ADAM_MAX_GROUPS
  float lr_gamma[8];
};
__device__ __forceinline__ void adam_schedule_block(const float *lr0, const float *lr_gamma, int n_groups, double beta1,
                                                    double beta2, int32_t *__restrict__ step_ptr, float2 *__restrict__ hyper) {
  const int step = *step_ptr;                   // optimiser steps completed so far
  const double t = (double)(step + 1);
  if ((int)threadIdx.x < n_groups) {
    const double lr = (double)lr0[threadIdx.x] * pow((double)lr_gamma[threadIdx.x], (double)step);
    hyper[threadIdx.x] = make_float2((float)(lr / (1.0 - pow(beta1, t))), (float)sqrt(1.0 - pow(beta2, t)));
  }
  __syncthreads();
  if (threadIdx.x == 0) *step_ptr = step + 1;
}

// ---- Adam arithmetic shared by adam.hip and the fused tail of k_preprocess_bwd (identical rounding in both)
struct AdamHyper {
  float omb1, b2, omb2, eps;  // (1-beta1), beta2, (1-beta2) rounded to f32 once on the host
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const AdamHyper h,
                                         float step_size, float bc2_sqrt) {
  // sqrt and the two divisions as single hardware instructions (v_sqrt_f32 / v_rcp_f32, 1 ulp each) instead of the
  // correctly rounded sequences (~30 instructions per element): the update differs from torch's by < 4e-7 of its own
  // size (tests/test_gpu_ops.py::test_adam_matches_torch allows 1e-6 lr), and inside the fused backward kernel the
  // arithmetic is on the critical path (6.8M -> 3.6M VALU instructions per launch at 100k Gaussians)
  const float eps = h.eps;
  m = m + (g - m) * h.omb1;
  v = v * h.b2 + h.omb2 * g * g;
  const float denom = __builtin_amdgcn_sqrtf(v) * __builtin_amdgcn_rcpf(bc2_sqrt) + eps;
  p = p - step_size * (m * __builtin_amdgcn_rcpf(denom));
}

typedef float floatx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_nt(const float4 *p) {
  floatx4 v = __builtin_nontemporal_load(reinterpret_cast<const floatx4 *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_nt(float4 *p, float4 a) {
  floatx4 v = {a.x, a.y, a.z, a.w};
  __builtin_nontemporal_store(v, reinterpret_cast<floatx4 *>(p));
}

// What k_preprocess_bwd needs to apply the optimiser itself (so_step_desc.fuse_adam): the six parameter tensors of
// the 3DGS model in the order means, log-scales, quats, opacity logits, sh0, shN with their moments, the per-group
// (step size, sqrt(bias correction 2)) evaluated by so_step_inputs, and the rounded betas / eps.
struct AdamFuse {
  float *p[6], *m[6], *v[6];
  const float2 *hyper;
  AdamHyper h;
  // float16 attribute rows (attr_rec.hpp; nullable): the kernel that applies the update also re-packs the rows of the
  // Gaussians it owns from the new float32 values, so the next forward reads what so_attr_pack_f16 would have written
  void *half_rows = nullptr;
  int half_stride16 = 0;   // row stride in 16-byte units
};

__device__ __forceinline__ int lane_id() {
#if defined(__HIP_DEVICE_COMPILE__)
  return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#else
  return 0;
#endif
}
#endif


// ---------------------------------------------------------------------------------------------
// Where the counter of a binned list lives (round 4).  The binning atomics of so_preprocess_fwd are memory-side returning
// atomics; with the counters of neighbouring tiles next to each other a scene whose splats gather in one image region sends
// most of them to the few 64-byte lines that hold those counters (tools/probes/atomic_skew.hip: 246k adds into 340
// neighbouring counters 55 us, into the same counters 4 KB apart 21 us; spread evenly over all counters 12.6 us either way).
// So RUNS of two neighbouring tiles are dealt over the array: run g of G = M / 2 keeps its two counts at run (g * 1031) mod G
// -- a bijection (1031 is prime; 1033 when it divides G), neighbouring runs 4 KB apart -- in every kernel that reads or bumps
// a BINNED count; the key slots themselves stay at t * bin_cap.  Compact lists (offsets = exclusive scan) are not touched.
// Why runs and not single counters: a large splat is binned by one wave whose 64 lanes bump an 8 x 8 block of neighbouring
// tiles, 8 lines per instruction in plain order and 64 when every counter is on its own.  Measured (tools/gpu_skew_r04.sh,
// so_preprocess_fwd, us; plain / single / runs of 2 / 4 / 8): c2 37.5 / 34.2 / 34.1 / 34.4 / 34.8; dense `ref` regime 135 /
// 186 / 152 / 149 / 147; 100k splats in a fifth of the cube (--cloud-scale 0.2) 96 / 57 / 66 / 73 / 89; 2M there 1014 / 332 /
// 452 / 578 / 842; 400k at 0.4: 126 / 76 / 75 / 78 / 85.
// ---------------------------------------------------------------------------------------------
#ifndef SO_BIN_GROUP_LOG2
#define SO_BIN_GROUP_LOG2 1      // runs of 2^k neighbouring counters stay together and the runs are spread; < 0: plain order
#endif
__host__ __device__ __forceinline__ int64_t bin_counter_index(int64_t t, int64_t M) {
#if SO_BIN_GROUP_LOG2 < 0
  return t;
#else
  const int64_t G = M >> SO_BIN_GROUP_LOG2;                 // whole runs; a ragged tail keeps its place
  if (G < 2 || G >= ((int64_t)1 << 21) || t >= (G << SO_BIN_GROUP_LOG2)) return t;   // (g * 1033 stays below 2^32)
  const uint32_t m = (uint32_t)G, k = (m % 1031u) ? 1031u : 1033u;
  if (m % k == 0u) return t;
  const uint32_t g = (uint32_t)(t >> SO_BIN_GROUP_LOG2), x = g * k;
#if defined(__HIP_DEVICE_COMPILE__)
  // x mod m without the 35-instruction integer division (this runs once per (Gaussian, tile) in so_preprocess_fwd): the
  // quotient is below k <= 1033 (g < m), so a float32 estimate is off by one at most, and the remainder is put right exactly
  const uint32_t q = (uint32_t)((float)x * __builtin_amdgcn_rcpf((float)m));
  uint32_t r = x - q * m;
  if ((int32_t)r < 0) r += m;
  else if (r >= m) r -= m;
#else
  const uint32_t r = x % m;
#endif
  return ((int64_t)r << SO_BIN_GROUP_LOG2) | (t & ((1 << SO_BIN_GROUP_LOG2) - 1));
#endif
}

// ---------------------------------------------------------------------------------------------
// Exact tile culling.  gsplat bins a Gaussian into every tile of the square of half-width ceil(3 sqrt(lambda_max))
// around its centre, but the rasteriser drops a (pixel, Gaussian) pair whose
// alpha = opacity * exp(-sigma) is below 1/255 -- i.e. outside the ellipse sigma(d) <= tau = ln(255 * opacity).  A tile
// whose pixel-centre rectangle lies entirely outside that ellipse contributes to no pixel, exactly; leaving it out
// of the tile's list changes no output and removes 15 % (isotropic trained-like) to 60 % (dense, low-opacity,
// anisotropic) of the per-pixel work.  The minimum of the convex quadratic over the rectangle is 0 if the centre is
// inside, else it lies on one of the four edges.  `tau` carries a safety margin (the caller adds it).  The histogram
// pass and the scatter pass evaluate this in different kernels and MUST agree bit for bit -- a pair that is counted but
// not filled leaves a slot of the tile's list unwritten.  HIP's __fmul_rn / __fadd_rn are plain `*` / `+` to the
// compiler, which fused them into FMAs in one kernel and not in the other (round 2: one disagreement in ~2e8 tests, found
// through exact-size lists in ops.isect_tiles); `#pragma clang fp contract(off)` in each body is what pins the rounding
// (hipcc's default -ffp-contract=fast-honor-pragmas honours it through inlining; checked in the ISA of every user).
// d = pixel - mean; conic (a, b, c): sigma = 0.5 (a dx^2 + c dy^2) + b dx dy.
// ---------------------------------------------------------------------------------------------
#if defined(__HIPCC__)
// (plain operators under the pragma: the bodies of HIP's __fmul_rn / __fadd_rn wrappers are compiled under the
// translation unit's default mode and WOULD be contracted)
__device__ __forceinline__ float cull_edge(float fixed, float lo, float hi, float q_fixed, float q_free, float b, float ratio) {
  // minimise over t in [lo, hi]:  0.5 (q_fixed fixed^2 + q_free t^2) + b fixed t ;  ratio = -b / q_free
#pragma clang fp contract(off)
  float t = ratio * fixed;
  t = fminf(fmaxf(t, lo), hi);
  const float f2 = fixed * fixed, t2 = t * t, ft = fixed * t;
  const float qa = q_fixed * f2, qb = q_free * t2;
  const float quad = qa + qb;
  const float h = 0.5f * quad, cross = b * ft;
  return h + cross;
}

__device__ __forceinline__ bool tile_touches(float mx, float my, float a, float b, float c, float tau, int tx, int ty,
                                             float tile_size) {
#pragma clang fp contract(off)
  const float x0 = (float)tx * tile_size, y0 = (float)ty * tile_size;       // exact: small integers times the tile size
  const float x0h = x0 + 0.5f, y0h = y0 + 0.5f;
  const float xlo = x0h - mx, ylo = y0h - my;
  const float xhi = xlo + (tile_size - 1.f), yhi = ylo + (tile_size - 1.f);
  const float ac = a * c, bb = b * b;
  const float det = ac - bb;
  if (!(a > 0.f) || !(c > 0.f) || !(det > 0.f)) return true;   // degenerate conic: never cull
  if (xlo <= 0.f && xhi >= 0.f && ylo <= 0.f && yhi >= 0.f) return tau >= 0.f;   // centre inside: sigma_min = 0
  const float ry = __fdiv_rn(-b, c), rx = __fdiv_rn(-b, a);
  float best = cull_edge(xlo, ylo, yhi, a, c, b, ry);
  best = fminf(best, cull_edge(xhi, ylo, yhi, a, c, b, ry));
  best = fminf(best, cull_edge(ylo, xlo, xhi, c, a, b, rx));
  best = fminf(best, cull_edge(yhi, xlo, xhi, c, a, b, rx));
  return best <= tau;
}

// tau with its margin: 1 % of alpha at the threshold, far above the float32 error of sigma at |d| ~ image size
__device__ __forceinline__ float cull_tau(float opacity) {
#pragma clang fp contract(off)
  const float x = 255.f * opacity;
  const float l = __logf(x);
  return l + 0.01f;
}
#endif

}  // namespace so
