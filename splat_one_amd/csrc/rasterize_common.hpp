// rasterize_common.hpp -- shared pieces of the K9/K10 tile rasteriser kernels (gfx950, wave64).
#pragma once
#include "so_common.hpp"

namespace so {

constexpr float kAlphaMax = 0.999f;
constexpr float kAlphaMin = 1.f / 255.f;
constexpr float kTStop = 1e-4f;

// XCD-aware tile order: workgroups b and b+8 share an XCD (and its L2), so give each XCD a
// contiguous run of tiles -- neighbouring tiles gather mostly the same Gaussians.  Speed only.
__device__ __forceinline__ int64_t xcd_remap(int64_t b, int64_t n) {
  const int64_t q = n >> 3, r = n & 7;
  const int64_t xcd = b & 7, k = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// Pixel owned by a thread.  16x16 tiles: each of the 4 waves owns an 8x8 quadrant (compact
// footprint -> more whole-wave culls than a 16x4 strip).  8x8 tiles: one wave, row-major.
template <int TS> struct PixelMap;
template <> struct PixelMap<16> {
  __device__ static __forceinline__ void get(int tid, int &lx, int &ly, int &wx0, int &wy0) {
    const int w = tid >> 6, l = tid & 63;
    wx0 = (w & 1) * 8;
    wy0 = (w >> 1) * 8;
    lx = wx0 + (l & 7);
    ly = wy0 + (l >> 3);
  }
};
template <> struct PixelMap<8> {
  __device__ static __forceinline__ void get(int tid, int &lx, int &ly, int &wx0, int &wy0) {
    wx0 = 0;
    wy0 = 0;
    lx = tid & 7;
    ly = tid >> 3;
  }
};

// Axis-aligned bound of {p : opac * exp(-sigma(p)) >= 1/255}, padded for float rounding.  A pixel
// centre outside it provably fails the alpha threshold, so skipping it cannot change any result.
// Degenerate conics (det <= 0) and opacities that can never pass are handled conservatively.
__device__ __forceinline__ float4 alpha_bound_box(float x, float y, float opac, float ca, float cb, float cc) {
  const float inf = __builtin_inff();
  const float tau = fmaxf(__logf(opac * 255.f), 0.f) * 1.0001f + 1e-4f;  // sigma <= tau <=> alpha >= 1/255
  if (!(opac * 255.f >= 0.999f)) return make_float4(inf, -inf, inf, -inf);  // can never contribute
  const float det = ca * cc - cb * cb;
  if (!(det > 0.f) || !(ca > 0.f) || !(cc > 0.f)) return make_float4(-inf, inf, -inf, inf);
  const float s = 2.f * tau / det;
  const float hx = sqrtf(s * cc) * 1.0001f + 0.01f;
  const float hy = sqrtf(s * ca) * 1.0001f + 0.01f;
  return make_float4(x - hx, x + hx, y - hy, y + hy);
}

}  // namespace so
