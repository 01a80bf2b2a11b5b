// rasterize_common.hpp -- shared pieces of the K9/K10 tile rasteriser kernels (gfx950, wave64).
#pragma once
#include "so_common.hpp"

namespace so {

// Loss scalars of the fused training step, finalised by the FIRST thread of the backward rasteriser (the kernel that
// follows the loss kernel in the step): the loss kernel's own last-workgroup finalisation costs it 4.4 us of tail latency
// (two memory round trips at the end of every workgroup); here the sums are simply complete.  sums == NULL: nothing to do.
struct LossFinal {
  const float *sums;   // [0] sum |x - y|, [1] sum of the SSIM map
  float *out;          // [0] loss, [1] mean |x - y|, [2] 1 - mean SSIM
  float w_l1, w_ssim, c_const, a_l1, b_ss;   // loss = w_l1 sums[0] + w_ssim sums[1] + c_const;  means = sums * a_l1 / b_ss
  // skip (nullable; so_rasterization_bwd): the forward's binning pass cut a list (its overflow counter) -> this backward
  // leaves the gradient records at the zeros the forward wrote: a cut image teaches nothing (raster_op.py)
  const int32_t *skip = nullptr;
  // which mapping the packed RGB backward uses (rasterize_bwd.hip): 0 one wave per 8x8 quadrant, 1 one wave per 16x16 tile
  // (rasterize_bwd_tile.hip: fewer instructions, longer per-tile chains -- faster from ~250 list entries per tile on),
  // -1 the process default (SPLAT_ONE_AMD_BWD_TILE, else 0)
  int tile_waves = -1;
  // workgroup -> tile table (nullable; so_step_desc.tile_order, built by so::tile_order_launch): longest list first, for the
  // one-wave-per-tile backward -- a tile there is ONE wave's serial chain, so the kernel ends when its longest tile does;
  // started last, a long tile runs on alone over an emptying machine (round 4: 2.5 resident waves per SIMD of 5)
  const int32_t *tile_order = nullptr;
  // Backward in list SEGMENTS (so_step_desc.bwd_seg_len > 0; quadrant waves, packed RGB, 16x16 tiles): the forward rasteriser left
  // every pixel's (live transmittance, accumulated colour) at each segment boundary of its tile's list in seg_state
  // [(boundary b) * pixels + pixel] (boundary b = after list entry (b + 1) * seg_len - 1); a workgroup of the backward takes ONE
  // segment of one tile and starts from the state at its far end -- long lists become seg_count independent chains.
  // render_colors: the forward's output (the colour accumulated behind a boundary = what was accumulated in all - up to it).
  const float4 *seg_state = nullptr;
  const float *render_colors = nullptr;
  int seg_len = 0, seg_count = 1;
};

// Round 3: the RGB passes of both rasteriser kernels work on packed fp32 pairs (v_pk_mul / v_pk_fma are the
// only vector instructions that do two lanes-worth per issue slot on gfx950 -- SQ_ACTIVE_INST_VALU prices every other one
// of these kernels at ~4 cycles per wave64), staged records laid out so that (ca, cb) and (cb, cc) are register pairs:
//   s_A = (x, y, ca, cb)   s_B = (cb, cc, opacity, cull threshold)   s_C = (red, green, blue, record offset [bwd])
// and ONE statement of the Gaussian's exponent shared by the forward and the backward, so that both take the same
// alpha >= 1/255 decision bit for bit:  q = (k Q) d,  s = d . q,  exp(-sigma) = 2^(-s)  (k = log2 e / 2, see kConicScale).
// (The round-2 scalar form of the RGB pass, kept behind SO_RASTER_V2=0 through round 3, is gone: it was built by no test.)
typedef float raster_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ raster_v2f conic_times(float ca, float cb0, float cb1, float cc, raster_v2f d) {
  raster_v2f q = raster_v2f{ca, cb0} * raster_v2f{d.x, d.x};
  return __builtin_elementwise_fma(raster_v2f{cb1, cc}, raster_v2f{d.y, d.y}, q);      // (ca dx + cb dy, cb dx + cc dy)
}
// The staged conic and cull threshold of the RGB passes are pre-multiplied by log2(e) / 2 (round 3): d . (k Q) d is the
// exponent of 2 directly -- one multiplication less per pass in both kernels -- and the backward undoes the factor once per
// pass on the nine reduced sums (k on the mean2d slots; the 1/2 of dL/d(ca, cc) rides on the same multiplication).
constexpr float kConicScale = 0.72134752044448170368f;        // log2(e) / 2
constexpr float kConicUnscale = 1.38629436111989061883f;      // 2 ln 2
__device__ __forceinline__ float gauss_vis(float scaled_two_sigma) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_exp2f(-scaled_two_sigma);
#else
  return exp2f(-scaled_two_sigma);
#endif
}

constexpr float kAlphaMax = 0.999f;
constexpr float kAlphaMin = 1.f / 255.f;
constexpr float kTStop = 1e-4f;

// XCD-aware tile order: workgroups b and b+8 share an XCD (and its L2).  SO_TILE_ORDER (compile time, measured in
// tools/gpu_tileorder.sh):
//   0  every XCD a contiguous eighth of the tile range -- neighbouring tiles gather mostly the same Gaussians, but an
//      image whose splats sit in one region loads the XCDs unevenly;
//   1  plain order: consecutive tiles on consecutive XCDs (balanced, no locality);
//   2  runs of 8 consecutive tiles per XCD, the runs dealt round-robin (balanced AND local) -- the default since round 2:
//      c2 (uniform cube) forward 61.0 -> 57.9 us, backward 131.7 -> 125.5 us; splats gathered in the middle of the image
//      (bench.py --cloud-scale 0.4) 88 -> 66 us and 204 -> 150 us; order 1 measures the same as 2 within noise.
// Speed only: any bijection gives the same results.
#ifndef SO_TILE_ORDER
#define SO_TILE_ORDER 2
#endif
#ifdef SO_TILE_PERM_EXPERIMENT
// experiment build (tools/experiments/tile_perm.py): workgroup -> tile through a table the host writes (heaviest tiles first)
static __device__ const int32_t *g_tile_perm = nullptr;
#endif
__device__ __forceinline__ int64_t xcd_remap(int64_t b, int64_t n) {
#ifdef SO_TILE_PERM_EXPERIMENT
  if (g_tile_perm) return g_tile_perm[b];
#endif
#if SO_TILE_ORDER == 1
  return b;
#elif SO_TILE_ORDER == 2
  const int64_t grp = b >> 6;
  if ((grp + 1) * 64 > n) return b;                 // the ragged last group keeps plain order
  return grp * 64 + ((b & 7) << 3) + ((b >> 3) & 7);
#else
  const int64_t q = n >> 3, r = n & 7;
  const int64_t xcd = b & 7, k = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
#endif
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

// mask with bit `bit` (uniform, known to be set) cleared: s_bitset0_b64
__device__ __forceinline__ unsigned long long clear_bit(unsigned long long mask, int bit) {
  asm("s_bitset0_b64 %0, %1" : "+s"(mask) : "s"(bit));
  return mask;
}

// A tile's slice of flatten_ids.  Two layouts:
//   compact (gsplat): [offsets[t], offsets[t+1]) with the total n_isects on the device and/or the host -- the device
//     count bounded by the host value when both are given (the capacity of flatten_ids: a list that overflowed its
//     buffer must not be walked past the end);
//   binned (n_dev == NULL, n_host = -cap < 0): every tile owns cap slots, [t cap, t cap + min(offsets[t], cap)), where
//     `offsets` holds the per-tile COUNTS -- no scan, no scatter pass (so_step_desc.bin_capacity).
__device__ __forceinline__ void tile_list_range(int64_t ct, int64_t M, const int32_t *__restrict__ offsets,
                                                const int32_t *__restrict__ n_dev, int64_t n_host, int64_t &lo, int64_t &hi) {
  if (!n_dev && n_host < 0) {
    const int64_t cap = -n_host, cnt = offsets[bin_counter_index(ct, M)];
    lo = ct * cap;
    hi = lo + (cnt < cap ? cnt : cap);
    return;
  }
  int64_t n_isects = n_dev ? (int64_t)*n_dev : n_host;
  if (n_dev && n_host > 0 && n_isects > n_host) n_isects = n_host;
  lo = offsets[ct];
  hi = (ct == M - 1) ? n_isects : (int64_t)offsets[ct + 1];
  if (hi > n_isects) hi = n_isects;
  if (lo > hi) lo = hi;
}

// Axis-aligned bound of {p : opac * exp(-sigma(p)) >= 1/255}, padded for float rounding.  A pixel
// centre outside it provably fails the alpha threshold, so skipping it cannot change any result.
// Degenerate conics (det <= 0) and opacities that can never pass are handled conservatively.
__device__ __forceinline__ float4 alpha_bound_box(float x, float y, float opac, float ca, float cb, float cc) {
  const float inf = __builtin_inff();
  const float tau = fmaxf(__logf(opac * 255.f), 0.f) * 1.0001f + 1e-4f;  // sigma <= tau <=> alpha >= 1/255
  if (!(opac * 255.f >= 0.999f)) return make_float4(inf, -inf, inf, -inf);  // can never contribute
  const float det = ca * cc - cb * cb;
  if (!(det > 0.f) || !(ca > 0.f) || !(cc > 0.f)) return make_float4(-inf, inf, -inf, inf);
  // 1-ulp reciprocal / square root: the padding below is three orders of magnitude wider, and the correctly rounded forms
  // are ~35 instructions per staged list entry in the kernels that have no record to read the box from
  const float s = 2.f * tau * __builtin_amdgcn_rcpf(det);
  const float hx = __builtin_amdgcn_sqrtf(s * cc) * 1.0001f + 0.01f;
  const float hy = __builtin_amdgcn_sqrtf(s * ca) * 1.0001f + 0.01f;
  return make_float4(x - hx, x + hx, y - hy, y + hy);
}

// Exact test "can alpha reach 1/255 anywhere in the axis-aligned rectangle [x0,x1] x [y0,y1] of pixel
// centres?" -- the minimum of sigma(p) = 1/2 (p-mu)^T Q (p-mu) over the rectangle against
// tau = ln(255 opac), with the same conservative padding as alpha_bound_box.  Evaluated by ONE lane
// per candidate Gaussian in the ballot phase (so its cost is amortised over 64 pixels); culls the
// bounding-box corners the box test lets through.
// Split in two since round 3: what depends on the Gaussian alone (the threshold, and the two answers that need no geometry)
// is computed once per Gaussian -- by the kernel that writes the 64-byte record, or when the list entry is staged -- and
// what depends on the rectangle runs per (wave, candidate).
__device__ __forceinline__ float cull_tau(float opac, float ca, float cb, float cc) {
  const float inf = __builtin_inff();
  if (!(opac * 255.f >= 0.999f)) return -inf;                      // can never contribute: no rectangle is hit
  const float det = ca * cc - cb * cb;
  if (!(det > 0.f) || !(ca > 0.f) || !(cc > 0.f)) return inf;      // degenerate conic: never cull
  const float tau = fmaxf(__logf(opac * 255.f), 0.f) * 1.0001f + 1e-4f;
  return tau * 1.0005f + 1e-3f;
}

__device__ __forceinline__ bool ellipse_hits_rect(float mx, float my, float tau, float ca, float cb, float cc,
                                                  float x0, float x1, float y0, float y1) {
  // translate so the Gaussian centre is the origin; pad the rectangle by 0.01 px
  const float ax0 = x0 - mx - 0.01f, ax1 = x1 - mx + 0.01f, ay0 = y0 - my - 0.01f, ay1 = y1 - my + 0.01f;
  // The minimum over the rectangle of a convex quadratic whose own minimum (the centre, sigma = 0) lies outside it is on an
  // edge the centre can see -- sigma grows along every ray from the centre, so the first point of the rectangle a ray meets
  // beats everything behind it: at most ONE vertical and ONE horizontal edge (round 3; all four were evaluated before).
  // On an edge: 1-D parabola, clamp the unconstrained minimiser.
  const float inf = __builtin_inff();
  const bool in_x = ax0 <= 0.f && ax1 >= 0.f, in_y = ay0 <= 0.f && ay1 >= 0.f;
  float best;
  {  // edge x = xe:  sigma(y) = 1/2 (ca xe^2 + cc y^2) + cb xe y,  y* = -cb xe / cc
    const float xe = ax0 > 0.f ? ax0 : ax1;
    const float rcc = __builtin_amdgcn_rcpf(cc);   // (1-ulp reciprocal: the test is padded by 1e-3, and a division is 12 instructions)
    const float y = fminf(fmaxf(-cb * xe * rcc, ay0), ay1);
    const float sv = 0.5f * (ca * xe * xe + cc * y * y) + cb * xe * y;
    best = in_x ? inf : sv;
  }
  {  // edge y = ye
    const float ye = ay0 > 0.f ? ay0 : ay1;
    const float rca = __builtin_amdgcn_rcpf(ca);
    const float x = fminf(fmaxf(-cb * ye * rca, ax0), ax1);
    const float sh = 0.5f * (ca * x * x + cc * ye * ye) + cb * x * ye;
    best = fminf(best, in_y ? inf : sh);
  }
  if (in_x && in_y) best = 0.f;   // centre inside
  return !(best > tau);   // (+inf: always, -inf: never; a NaN -- non-finite conic -- is never culled, as before the split)
}

}  // namespace so
