// rasterize_bwd_tile.hip -- K10 for RGB from the 64-byte records, ONE WAVE PER 16x16 TILE (round 4).
//
// Replaces gsplat `rasterize_to_pixels` backward (`loss.backward()` at
// /root/reference/utils/gsplat_utils/gsplat_trainer.py:655) like rasterize_bwd.hip; algorithm SURVEY.md B.2.
//
// Why a second mapping.  k_rasterize_bwd gives each of a tile's four 8x8 quadrants its own wave; a pass = one (quadrant,
// Gaussian) pair = ~32 vector instructions of per-pixel arithmetic + 23 of cross-lane reduction + one atomic instruction.
// The round-4 ablations (profiles/r04_experiments.json: `rasterize_bwd_ablation`) price the reduction at 29 % of the kernel
// and the atomics at 3.5 %: the instruction count is what matters, and a Gaussian of the c2 scene meets 2.0 quadrants of its
// tile on average (tools/passsim.py: 506k quadrant passes for 254k (tile, Gaussian) pairs), so half of the reductions add up
// what the next wave's reduction adds up again, into the same nine addresses.  Here a wave owns the whole tile -- lane
// (lx, ly) holds the per-pixel state (T, colour behind, upstream gradient, last contributor) of its pixel in EACH quadrant,
// four sets of registers -- walks the tile's list once, runs the per-pixel block only for the quadrants a Gaussian can
// reach (a wave-uniform mask from the candidate's own exact ellipse-vs-quadrant tests: scalar branches, no lane wasted on a
// quadrant the old kernel would not have visited either), accumulates the nine per-lane sums over those blocks with the
// multiply-adds that form them, and pays ONE reduction and ONE atomic instruction per (tile, Gaussian).  No workgroup
// barrier (one wave), 3 KB of LDS, list entries staged 64 at a time by the lanes that test them.
// Same arithmetic per pixel as k_rasterize_bwd (rasterize_common.hpp: conic_times / gauss_vis, bit for bit the forward's
// alpha >= 1/255 decision); the sums differ from it only in the order of additions.
#include "rasterize_common.hpp"

namespace so {

typedef raster_v2f tile_v2f;

template <bool SMALL>
__global__ void __launch_bounds__(64)
k_rasterize_bwd_tile(int C, int N, int W, int H, int tile_w, int tile_h, const float *__restrict__ rec,
                     const float *__restrict__ backgrounds, const int32_t *__restrict__ offsets,
                     const int32_t *__restrict__ flatten_ids, const int32_t *__restrict__ n_isects_dev,
                     int64_t n_isects_host, const float *__restrict__ render_alphas, const int32_t *__restrict__ last_ids,
                     const float *__restrict__ v_render_colors, const float *__restrict__ v_render_alphas,
                     float *__restrict__ vrec, int wrap_flags, const LossFinal fin) {
  constexpr int TS = 16, STAGE = 64;
  if (fin.sums && blockIdx.x == 0 && threadIdx.x == 0) {   // (see LossFinal: the loss kernel before this one has completed)
    const float l1m = fin.sums[0] * fin.a_l1, ssm = fin.sums[1] * fin.b_ss;
    fin.out[0] = fin.w_l1 / fin.a_l1 * l1m + fin.w_ssim / fin.b_ss * ssm + fin.c_const;
    fin.out[1] = l1m;
    fin.out[2] = 1.f - ssm;
  }
  if (fin.skip && *fin.skip != 0) return;   // uniform over the grid
  // staged per Gaussian, as in k_rasterize_bwd: A = (x, y, ca, cb), B = (cb, cc, opacity, .), C = (r, g, b, record offset)
  __shared__ float4 s_A[STAGE];
  __shared__ float4 s_B[STAGE];
  __shared__ float4 s_C[STAGE];

  const int n_tiles = tile_w * tile_h;
  const int M = C * n_tiles;
  const int lane = threadIdx.x;
  const int ct = fin.tile_order ? fin.tile_order[blockIdx.x] : (int)xcd_remap(blockIdx.x, M);
  const int c = ct / n_tiles;
  const int t = ct - c * n_tiles;
  const int ty = t / tile_w, tx = t - ty * tile_w;
  const bool wrap = wrap_for(wrap_flags, c);
  const float wrap_w = (float)W, wrap_cx = (float)(tx * TS) + 0.5f * (float)TS;
  const int lx = lane & 7, ly = lane >> 3;

  int64_t lo, hi;
  tile_list_range(ct, M, offsets, n_isects_dev, n_isects_host, lo, hi);
  if (hi <= lo) return;

  // per-pixel state of this lane's pixel in quadrant q = 2 qy + qx: pixel (tx 16 + 8 qx + lx, ty 16 + 8 qy + ly)
  float T[4], behind[4], vc0[4], vc1[4], vc2[4];
  int32_t rel[4];            // candidate tt of the current batch contributes to the pixel iff tt >= rel (batch_end - last contributor)
  const tile_v2f pxy0 = {(float)(tx * TS + lx) + 0.5f, (float)(ty * TS + ly) + 0.5f};   // quadrant q: + (8 (q & 1), 8 (q >> 1))
  float bg0 = 0.f, bg1 = 0.f, bg2 = 0.f;
  if (backgrounds) { bg0 = backgrounds[c * 3]; bg1 = backgrounds[c * 3 + 1]; bg2 = backgrounds[c * 3 + 2]; }
  int32_t lane_last = (int32_t)lo - 1;
  // (every load unconditional, at an address clamped into the image: the 24 requests of a lane leave back to back and are
  // waited for once -- as `inside ? load : 0` each sat in its own exec-masked block, one round trip after the other)
  float alpha_in[4], va_in[4];
  int32_t last_in[4];
  bool inside[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int j = tx * TS + 8 * (q & 1) + lx, i = ty * TS + 8 * (q >> 1) + ly;
    inside[q] = (i < H) && (j < W);
    const int64_t pix = ((int64_t)c * H + min(i, H - 1)) * W + min(j, W - 1);
    alpha_in[q] = render_alphas[pix];
    vc0[q] = v_render_colors[pix * 3];
    vc1[q] = v_render_colors[pix * 3 + 1];
    vc2[q] = v_render_colors[pix * 3 + 2];
    va_in[q] = v_render_alphas[pix];
    last_in[q] = last_ids[pix];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float T_final = inside[q] ? 1.f - alpha_in[q] : 1.f;
    T[q] = T_final;
    vc0[q] = inside[q] ? vc0[q] : 0.f;
    vc1[q] = inside[q] ? vc1[q] : 0.f;
    vc2[q] = inside[q] ? vc2[q] : 0.f;
    const float v_a = inside[q] ? va_in[q] : 0.f;
    const float bg_dot = bg0 * vc0[q] + bg1 * vc1[q] + bg2 * vc2[q];
    behind[q] = T_final * (v_a - bg_dot);
    // last contributor of this pixel; pixels that nothing reached (or outside the image) keep lo - 1: no Gaussian valid
    rel[q] = (inside[q] && T_final < 1.f) ? last_in[q] : (int32_t)lo - 1;       // (the last contributor itself, for now)
    lane_last = max(lane_last, rel[q]);
  }
  const int32_t wave_last = wave_max_i32(lane_last);   // wave-uniform: the tile's last contributor
  if (wave_last < lo) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) rel[q] = wave_last - rel[q];

  // the nine-sum network's lanes (so_common.hpp::wave_reduce9_scattered)
  const bool atom_lane = ((kReduce9Lanes >> lane) & 1ull) != 0;
  const int slot9 = reduce9_slot_of_lane(lane);
  const float unscale9 = slot9 < 2 ? kConicUnscale : ((slot9 == 2 || slot9 == 4) ? 0.5f : 1.f);
  // quadrant rectangles of pixel centres
  const float tx0 = (float)(tx * TS) + 0.5f, ty0 = (float)(ty * TS) + 0.5f;

  for (int64_t batch_end = wave_last; batch_end >= lo; batch_end -= STAGE) {
    wave_lds_sync();   // (the previous batch's broadcast reads are done: one wave, LDS operations in order)
    const int64_t idx = batch_end - lane;
    bool h0 = false, h1 = false, h2 = false, h3 = false;
    if (idx >= lo) {
      const int32_t g = flatten_ids[idx];
      const float4 *r4 = reinterpret_cast<const float4 *>(rec) + 4 * (int64_t)g;
      float4 q0 = r4[0];
      const float4 q1 = r4[1];   // x,y,ca,cb | cc,opac,r,g
      const float4 q2 = r4[2];   // blue, depth, radius, cull threshold
      float4 bx = r4[3];         // the cull box of this Gaussian, computed once in the kernel that wrote the record
      if (wrap) {
        const float shift = wrap_w * rintf((q0.x - wrap_cx) / wrap_w);
        q0.x -= shift; bx.x -= shift; bx.y -= shift;
      }
      q0.z *= kConicScale; q0.w *= kConicScale;                   // conic and threshold in units of the exponent of 2 (as the forward)
      const float cc_s = q1.x * kConicScale, tau_s = q2.w * kConicScale;
      s_A[lane] = q0;
      s_B[lane] = make_float4(q0.w, cc_s, q1.y, tau_s);
      s_C[lane] = make_float4(q1.z, q1.w, q2.x, __uint_as_float(SMALL ? (unsigned)g * 64u : (unsigned)g));
      // which quadrants can this Gaussian reach?  box first, then the exact ellipse-vs-rectangle test (this lane's registers)
      const bool bxl = !(bx.y < tx0 || bx.x > tx0 + 7.f), bxr = !(bx.y < tx0 + 8.f || bx.x > tx0 + 15.f);
      const bool byt = !(bx.w < ty0 || bx.z > ty0 + 7.f), byb = !(bx.w < ty0 + 8.f || bx.z > ty0 + 15.f);
      h0 = bxl && byt && ellipse_hits_rect(q0.x, q0.y, tau_s, q0.z, q0.w, cc_s, tx0, tx0 + 7.f, ty0, ty0 + 7.f);
      h1 = bxr && byt && ellipse_hits_rect(q0.x, q0.y, tau_s, q0.z, q0.w, cc_s, tx0 + 8.f, tx0 + 15.f, ty0, ty0 + 7.f);
      h2 = bxl && byb && ellipse_hits_rect(q0.x, q0.y, tau_s, q0.z, q0.w, cc_s, tx0, tx0 + 7.f, ty0 + 8.f, ty0 + 15.f);
      h3 = bxr && byb && ellipse_hits_rect(q0.x, q0.y, tau_s, q0.z, q0.w, cc_s, tx0 + 8.f, tx0 + 15.f, ty0 + 8.f, ty0 + 15.f);
    }
    const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
    wave_lds_sync();
    unsigned long long uni = m0 | m1 | m2 | m3;
#pragma unroll 1
    while (uni) {
      const int tt = __ffsll((long long)uni) - 1;
      uni = clear_bit(uni, tt);
      const float4 a = s_A[tt];            // x, y, ca, cb
      const float4 b4 = s_B[tt];           // cb, cc, opacity, (cull threshold)
      const float4 c4 = s_C[tt];           // red, green, blue, record offset
      const tile_v2f d0 = tile_v2f{a.x, a.y} - pxy0;
      tile_v2f acc_xy = {0.f, 0.f}, acc_cxz = {0.f, 0.f}, acc_01 = {0.f, 0.f};
      float acc_cy = 0.f, acc_2 = 0.f, acc_op = 0.f;
      bool any = false;
#define SO_TILE_BLOCK(Q, MASK)                                                                                     \
      if ((MASK >> tt) & 1ull) {                                                                                   \
        const tile_v2f d = d0 - tile_v2f{8.f * (Q & 1), 8.f * (Q >> 1)};                                           \
        const tile_v2f qv = conic_times(a.z, a.w, b4.x, b4.y, d);                                                  \
        const float s2 = fmaf(qv.y, d.y, qv.x * d.x);                                                              \
        const float vis = gauss_vis(s2);                                                                           \
        const float ov = b4.z * vis;                                                                               \
        const float alpha = fminf(kAlphaMax, ov);                                                                  \
        const bool valid = (tt >= rel[Q]) && !(s2 < 0.f || alpha < kAlphaMin);                                     \
        if (__ballot(valid) != 0ull) {                                                                             \
          any = true;                                                                                              \
          const float alpha_v = valid ? alpha : 0.f;                                                               \
          const float ra = __builtin_amdgcn_rcpf(1.f - alpha_v);                                                   \
          T[Q] *= ra;                                                                                              \
          const float fac = alpha_v * T[Q];                                                                        \
          const float cv = fmaf(c4.z, vc2[Q], fmaf(c4.y, vc1[Q], c4.x * vc0[Q]));                                  \
          float v_alpha = fmaf(T[Q], cv, ra * behind[Q]);                                                          \
          behind[Q] = fmaf(-fac, cv, behind[Q]);                                                                   \
          v_alpha = (valid && ov <= kAlphaMax) ? v_alpha : 0.f;                                                    \
          const float v_sigma = -ov * v_alpha;                                                                     \
          const tile_v2f vs2 = {v_sigma, v_sigma};                                                                 \
          const tile_v2f tq = vs2 * d;                                                                             \
          acc_xy = __builtin_elementwise_fma(vs2, qv, acc_xy);                                                     \
          acc_cxz = __builtin_elementwise_fma(tq, d, acc_cxz);                                                     \
          acc_cy = fmaf(tq.x, d.y, acc_cy);                                                                        \
          acc_01 = __builtin_elementwise_fma(tile_v2f{fac, fac}, tile_v2f{vc0[Q], vc1[Q]}, acc_01);                \
          acc_2 = fmaf(fac, vc2[Q], acc_2);                                                                        \
          acc_op = fmaf(vis, v_alpha, acc_op);                                                                     \
        }                                                                                                          \
      }
      SO_TILE_BLOCK(0, m0)
      SO_TILE_BLOCK(1, m1)
      SO_TILE_BLOCK(2, m2)
      SO_TILE_BLOCK(3, m3)
#undef SO_TILE_BLOCK
      if (!any) continue;
      const float v8[8] = {acc_xy.x, acc_xy.y, acc_cxz.x, acc_cy, acc_cxz.y, acc_01.x, acc_01.y, acc_2};
      const float val = wave_reduce9_scattered(v8, acc_op) * unscale9;
      if (atom_lane) {
        if constexpr (SMALL) {
          const unsigned off = __float_as_uint(c4.w) | ((unsigned)slot9 * 4u);
          atomicAdd(reinterpret_cast<float *>(reinterpret_cast<char *>(vrec) + off), val);
        } else {
          atomicAdd(vrec + (int64_t)__float_as_uint(c4.w) * 16 + slot9, val);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) rel[q] -= STAGE;      // the next batch ends STAGE entries earlier
  }
}

// Workgroup -> tile table, longest list first (so_step_desc.tile_order): ONE workgroup, a counting sort of the C x tiles list
// lengths over 256 length classes (class = length scaled by the longest list; the order inside a class is the arrival order of
// LDS atomics -- any bijection gives the same images and gradients, rasterize_common.hpp::xcd_remap).  Consecutive workgroups
// are dispatched round-robin over the XCDs, so every XCD receives its share of the long tiles first.
constexpr int kOrderThreads = 1024, kOrderClasses = 256, kOrderCache = 15360;   // list lengths kept in LDS (60 KB): one global read pass
__global__ void __launch_bounds__(kOrderThreads)
k_tile_order(int M, const int32_t *__restrict__ offsets, const int32_t *__restrict__ n_isects_dev, int64_t n_isects_host,
             int32_t *__restrict__ order) {
  __shared__ int s_cls[kOrderClasses];
  __shared__ int s_wmax[kOrderThreads / 64];
  __shared__ int s_len[kOrderCache];
  const int tid = threadIdx.x;
  auto length = [&](int t) {
    int64_t lo, hi;
    tile_list_range(t, M, offsets, n_isects_dev, n_isects_host, lo, hi);
    return (int)(hi - lo);
  };
  auto cached = [&](int t) { return t < kOrderCache ? s_len[t] : length(t); };
  int mx = 0;
  for (int t = tid; t < M; t += kOrderThreads) {
    const int len = length(t);
    if (t < kOrderCache) s_len[t] = len;
    mx = max(mx, len);
  }
  mx = wave_max_i32(mx);
  if ((tid & 63) == 0) s_wmax[tid >> 6] = mx;
  if (tid < kOrderClasses) s_cls[tid] = 0;
  __syncthreads();
  mx = 0;
#pragma unroll
  for (int w = 0; w < kOrderThreads / 64; ++w) mx = max(mx, s_wmax[w]);
  const float scale = (float)kOrderClasses / (float)(mx + 1);
  auto cls_of = [&](int len) {                       // class 0 = the longest lists
    const int k = (int)((float)len * scale);
    return kOrderClasses - 1 - (k < kOrderClasses ? k : kOrderClasses - 1);
  };
  // (empty tiles -- most of a skewed view's -- are counted and placed once per WAVE: thousands of LDS atomics on one address
  // cost the first version 21 us on a cloud gathered in a ninth of the image)
  const int n_trips = (M + kOrderThreads - 1) / kOrderThreads;
  for (int trip = 0; trip < n_trips; ++trip) {
    const int t = trip * kOrderThreads + tid;
    const int len = t < M ? cached(t) : -1;
    const unsigned long long zmask = __ballot(len == 0);
    if (len > 0) atomicAdd(&s_cls[cls_of(len)], 1);
    else if (len == 0 && (zmask & ((1ull << (tid & 63)) - 1ull)) == 0ull) atomicAdd(&s_cls[kOrderClasses - 1], __popcll(zmask));
  }
  __syncthreads();
  if (tid < 64) {                                    // exclusive scan of the 256 class counts by one wave: 4 per lane
    int v[4], run = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = s_cls[4 * tid + j]; run += v[j]; }
    int inc = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d, 64);
      if (tid >= d) inc += o;
    }
    int base = inc - run;
#pragma unroll
    for (int j = 0; j < 4; ++j) { s_cls[4 * tid + j] = base; base += v[j]; }
  }
  __syncthreads();
  for (int trip = 0; trip < n_trips; ++trip) {
    const int t = trip * kOrderThreads + tid;
    const int len = t < M ? cached(t) : -1;
    const unsigned long long zmask = __ballot(len == 0);
    const unsigned long long below = zmask & ((1ull << (tid & 63)) - 1ull);
    int zbase = 0;
    if (len == 0 && below == 0ull) zbase = atomicAdd(&s_cls[kOrderClasses - 1], __popcll(zmask));
    if (zmask) zbase = __shfl(zbase, __ffsll((long long)zmask) - 1, 64);      // the leader's base, to every lane of the wave
    if (len > 0) order[atomicAdd(&s_cls[cls_of(len)], 1)] = t;
    else if (len == 0) order[zbase + __popcll(below)] = t;
  }
}

int tile_order_launch(int C, int tile_w, int tile_h, const int32_t *offsets, const int32_t *n_isects_dev, int64_t n_isects_host,
                      int32_t *order, hipStream_t st) {
  const int64_t M = (int64_t)C * tile_w * tile_h;
  SO_REQUIRE(order && offsets && M > 0 && M < ((int64_t)1 << 31), "so_train_step_fwd_bwd: tile_order needs the tile lists and C*tiles < 2^31");
  hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(kOrderThreads), 0, st, (int)M, offsets, n_isects_dev, n_isects_host, order);
  return check_launch("so_train_step_fwd_bwd (tile order)");
}

// internal (rasterize_bwd.hip::rasterize_bwd_packed_launch): 16x16 tiles, no absgrad
int rasterize_bwd_tile_launch(int C, int N, int width, int height, int tile_w, int tile_h, const float *rec,
                              const float *backgrounds, const int32_t *isect_offsets, const int32_t *flatten_ids,
                              const int32_t *n_isects_dev, int64_t n_isects_host, const float *render_alphas,
                              const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
                              float *vrec, int wrap_flags, const LossFinal &fin, hipStream_t st) {
  const dim3 grid((unsigned)((int64_t)C * tile_w * tile_h));
  const bool small = (int64_t)C * N < ((int64_t)1 << 26);
  if (small)
    hipLaunchKernelGGL((k_rasterize_bwd_tile<true>), grid, dim3(64), 0, st, C, N, width, height, tile_w, tile_h, rec, backgrounds,
                       isect_offsets, flatten_ids, n_isects_dev, n_isects_host, render_alphas, last_ids, v_render_colors,
                       v_render_alphas, vrec, wrap_flags, fin);
  else
    hipLaunchKernelGGL((k_rasterize_bwd_tile<false>), grid, dim3(64), 0, st, C, N, width, height, tile_w, tile_h, rec, backgrounds,
                       isect_offsets, flatten_ids, n_isects_dev, n_isects_host, render_alphas, last_ids, v_render_colors,
                       v_render_alphas, vrec, wrap_flags, fin);
  return check_launch("so_rasterize_bwd_packed (tile waves)");
}

}  // namespace so
