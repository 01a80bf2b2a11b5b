// rasterize_wave.hip -- K9/K10 in "one wave per 8x8 pixel quadrant" form (packed records, D = 3).
//
// Same algorithm and results as rasterize_fwd.hip / rasterize_bwd.hip,
// restructured around what the profile of those kernels showed on MI355X: with short tile lists
// they are latency-bound (VALU busy < 25 %), stalled on LDS round trips, workgroup barriers and the
// slowest of a tile's four waves.  Here every 8x8 quadrant of a tile is an independent 64-lane
// workgroup:
//   * staging is lane-parallel over Gaussians: lane l loads flatten id + 64-byte record of list entry
//     l of the current 64-entry batch into REGISTERS and runs the exact ellipse-vs-quadrant test;
//     `__ballot` of the test is the work list;
//   * walking the list broadcasts the chosen lane's record with v_readlane (SGPR operands, no LDS
//     round trip, no barrier anywhere in the kernel);
//   * the next batch's ids and records are prefetched while the current one is walked;
//   * the backward sums per-pixel gradients with the per-row transposing butterfly and issues one
//     atomic instruction into the 64-byte gradient record.
// Each quadrant re-stages its tile's list (4x the gathers, served by L2); in exchange there is no
// LDS allocation, no __syncthreads, and the scheduler sees 4x more, 4x smaller workgroups.
#include "rasterize_common.hpp"

namespace so {

struct StagedBatch {   // one list entry per lane
  float4 q0, q1;       // x, y, ca, cb | cc, opac, r, g
  float b;             // blue
  int32_t g;           // flatten id
};

__device__ __forceinline__ float bcast(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

__device__ __forceinline__ void stage_records(StagedBatch &s, const float4 *__restrict__ rec, bool live) {
  if (live) {
    const float4 *r4 = rec + 4 * (int64_t)s.g;
    s.q0 = r4[0];
    s.q1 = r4[1];
    s.b = reinterpret_cast<const float *>(r4 + 2)[0];
  } else {
    s.q0 = make_float4(0.f, 0.f, 1.f, 0.f);
    s.q1 = make_float4(1.f, 0.f, 0.f, 0.f);   // opacity 0: never hits
    s.b = 0.f;
  }
}

__global__ void __launch_bounds__(64)
k_rasterize_fwd_wave(int C, int W, int H, int tile_w, int tile_h, const float4 *__restrict__ rec,
                     const float *__restrict__ backgrounds, const int32_t *__restrict__ offsets,
                     const int32_t *__restrict__ flatten_ids, const int32_t *__restrict__ n_isects_dev,
                     int64_t n_isects_host, float *__restrict__ render_colors, float *__restrict__ render_alphas,
                     int32_t *__restrict__ last_ids) {
  const int n_tiles = tile_w * tile_h;
  const int64_t M = (int64_t)C * n_tiles;
  const int64_t v = xcd_remap(blockIdx.x, 4 * M);
  const int64_t ct = v >> 2;
  const int quad = (int)(v & 3);
  const int c = (int)(ct / n_tiles);
  const int t = (int)(ct - (int64_t)c * n_tiles);
  const int ty = t / tile_w, tx = t - ty * tile_w;
  const int lane = threadIdx.x;
  const int wx0 = (quad & 1) * 8, wy0 = (quad >> 1) * 8;
  const int j = tx * 16 + wx0 + (lane & 7), i = ty * 16 + wy0 + (lane >> 3);
  const bool inside = (i < H) && (j < W);
  const float px = (float)j + 0.5f, py = (float)i + 0.5f;
  const int64_t pix = ((int64_t)c * H + i) * W + j;
  const float qx0 = (float)(tx * 16 + wx0) + 0.5f, qx1 = qx0 + 7.f;
  const float qy0 = (float)(ty * 16 + wy0) + 0.5f, qy1 = qy0 + 7.f;
  if (__ballot(inside) == 0ull) return;   // quadrant entirely outside the image

  // device count, bounded by the host value when both are given (the capacity of flatten_ids: a list that
  // overflowed its buffer must not be walked past the end)
  int64_t n_isects = n_isects_dev ? (int64_t)*n_isects_dev : n_isects_host;
  if (n_isects_dev && n_isects_host > 0 && n_isects > n_isects_host) n_isects = n_isects_host;
  int64_t lo = offsets[ct];
  int64_t hi = (ct == M - 1) ? n_isects : (int64_t)offsets[ct + 1];
  if (hi > n_isects) hi = n_isects;
  if (lo > hi) lo = hi;

  float T = 1.f, ar = 0.f, ag = 0.f, ab = 0.f;
  int32_t cur_idx = 0;
  bool done = !inside;

  StagedBatch cur, nxt;
  int32_t g_next2 = 0;   // ids two batches ahead
  // prologue: ids of batches 0,1; records of batch 0
  cur.g = (lo + lane < hi) ? flatten_ids[lo + lane] : 0;
  nxt.g = (lo + 64 + lane < hi) ? flatten_ids[lo + 64 + lane] : 0;
  stage_records(cur, rec, lo + lane < hi);

  for (int64_t bs = lo; bs < hi; bs += 64) {
    if (__ballot(!done) == 0ull) break;
    // land `cur` before the prefetch is issued (see the backward kernel)
    asm volatile("" ::"v"(cur.q0.x), "v"(cur.q0.y), "v"(cur.q0.z), "v"(cur.q0.w), "v"(cur.q1.x), "v"(cur.q1.y),
                 "v"(cur.q1.z), "v"(cur.q1.w), "v"(cur.b), "v"(cur.g), "v"(nxt.g));
    // prefetch: records of the next batch, ids of the one after
    stage_records(nxt, rec, bs + 64 + lane < hi);
    g_next2 = (bs + 128 + lane < hi) ? flatten_ids[bs + 128 + lane] : 0;
    const bool live = bs + lane < hi;
    const bool hit = live && ellipse_hits_rect(cur.q0.x, cur.q0.y, cur.q1.y, cur.q0.z, cur.q0.w, cur.q1.x, qx0, qx1, qy0, qy1);
    unsigned long long mask = __ballot(hit);
    while (mask) {
      if (__ballot(!done) == 0ull) break;
      const int bit = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const float gx = bcast(cur.q0.x, bit), gy = bcast(cur.q0.y, bit);
      const float ca = bcast(cur.q0.z, bit), cb = bcast(cur.q0.w, bit), cc = bcast(cur.q1.x, bit);
      const float op = bcast(cur.q1.y, bit);
      if (!done) {
        const float dx = gx - px, dy = gy - py;
        const float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
        const float alpha = fminf(kAlphaMax, op * __expf(-sigma));
        if (!(sigma < 0.f || alpha < kAlphaMin)) {
          const float next_T = T * (1.f - alpha);
          if (next_T <= kTStop) {
            done = true;
          } else {
            const float vis = alpha * T;
            ar += bcast(cur.q1.z, bit) * vis;
            ag += bcast(cur.q1.w, bit) * vis;
            ab += bcast(cur.b, bit) * vis;
            cur_idx = (int32_t)(bs + bit);
            T = next_T;
          }
        }
      }
    }
    cur = nxt;
    nxt.g = g_next2;
  }
  if (inside) {
    render_alphas[pix] = 1.f - T;
    float br = 0.f, bg = 0.f, bb = 0.f;
    if (backgrounds) { br = backgrounds[c * 3]; bg = backgrounds[c * 3 + 1]; bb = backgrounds[c * 3 + 2]; }
    render_colors[pix * 3] = ar + T * br;
    render_colors[pix * 3 + 1] = ag + T * bg;
    render_colors[pix * 3 + 2] = ab + T * bb;
    last_ids[pix] = cur_idx;
  }
}

// DBG: diagnostic build only (so_debug_rasterize_bwd_wave_stamps): per-wave s_memtime stamps and pass
// counts go to a buffer of their own; the product entry points instantiate DBG = false.
template <bool ABS, bool DBG>
__global__ void __launch_bounds__(64)
k_rasterize_bwd_wave(int C, int W, int H, int tile_w, int tile_h, const float4 *__restrict__ rec,
                     const float *__restrict__ backgrounds, const int32_t *__restrict__ offsets,
                     const int32_t *__restrict__ flatten_ids, const int32_t *__restrict__ n_isects_dev,
                     int64_t n_isects_host, const float *__restrict__ render_alphas,
                     const int32_t *__restrict__ last_ids, const float *__restrict__ v_render_colors,
                     const float *__restrict__ v_render_alphas, float *__restrict__ vrec,
                     unsigned long long *__restrict__ dbg, int dbg_variant) {
  unsigned long long t_start = 0, t_loop = 0;
  int n_pass = 0, n_valid = 0, n_batches = 0;
  if (DBG) t_start = __builtin_amdgcn_s_memtime();
  const int n_tiles = tile_w * tile_h;
  const int64_t M = (int64_t)C * n_tiles;
  const int64_t v = xcd_remap(blockIdx.x, 4 * M);
  const int64_t ct = v >> 2;
  const int quad = (int)(v & 3);
  const int c = (int)(ct / n_tiles);
  const int t = (int)(ct - (int64_t)c * n_tiles);
  const int ty = t / tile_w, tx = t - ty * tile_w;
  const int lane = threadIdx.x;
  const int wx0 = (quad & 1) * 8, wy0 = (quad >> 1) * 8;
  const int j = tx * 16 + wx0 + (lane & 7), i = ty * 16 + wy0 + (lane >> 3);
  const bool inside = (i < H) && (j < W);
  const float px = (float)j + 0.5f, py = (float)i + 0.5f;
  const int64_t pix = ((int64_t)c * H + i) * W + j;
  const float qx0 = (float)(tx * 16 + wx0) + 0.5f, qx1 = qx0 + 7.f;
  const float qy0 = (float)(ty * 16 + wy0) + 0.5f, qy1 = qy0 + 7.f;

  // device count, bounded by the host value when both are given (the capacity of flatten_ids: a list that
  // overflowed its buffer must not be walked past the end)
  int64_t n_isects = n_isects_dev ? (int64_t)*n_isects_dev : n_isects_host;
  if (n_isects_dev && n_isects_host > 0 && n_isects > n_isects_host) n_isects = n_isects_host;
  int64_t lo = offsets[ct];
  int64_t hi = (ct == M - 1) ? n_isects : (int64_t)offsets[ct + 1];
  if (hi > n_isects) hi = n_isects;
  if (hi <= lo) return;

  const float T_final = inside ? 1.f - render_alphas[pix] : 1.f;
  float T = T_final;
  float buf_r = 0.f, buf_g = 0.f, buf_b = 0.f;
  float vr = 0.f, vg = 0.f, vb = 0.f, v_a = 0.f;
  if (inside) {
    vr = v_render_colors[pix * 3]; vg = v_render_colors[pix * 3 + 1]; vb = v_render_colors[pix * 3 + 2];
    v_a = v_render_alphas[pix];
  }
  float bg_dot = 0.f;
  if (backgrounds) bg_dot = backgrounds[c * 3] * vr + backgrounds[c * 3 + 1] * vg + backgrounds[c * 3 + 2] * vb;
  int32_t bin_final = (int32_t)lo - 1;
  if (inside && T_final < 1.f) bin_final = last_ids[pix];
  int32_t wave_last = bin_final;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) wave_last = max(wave_last, __shfl_xor(wave_last, d, 64));
  if (wave_last < lo) return;   // uniform: nothing reached this quadrant

  // output slot of this lane (see rasterize_bwd.hip): (l&15) < 8 butterfly slots, 8 opacity, 9/10 abs
  const int l15 = lane & 15;
  const int slot = l15 < 8 ? slot_of_lane(lane) : l15;
  float *out_base = vrec + slot;
  const bool out_lane = l15 <= (ABS ? 10 : 8);

  StagedBatch cur, nxt;
  int32_t g_next2 = 0;
  const int64_t first_end = wave_last;   // batches walk back to front from the wave's last contributor
  cur.g = (first_end - lane >= lo) ? flatten_ids[first_end - lane] : 0;
  nxt.g = (first_end - 64 - lane >= lo) ? flatten_ids[first_end - 64 - lane] : 0;
  stage_records(cur, rec, first_end - lane >= lo);
  if (DBG) {
    // force the prologue loads to land so that the stamp separates prologue from the walk
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t_loop = __builtin_amdgcn_s_memtime();
  }

  for (int64_t batch_end = first_end; batch_end >= lo; batch_end -= 64) {
    if (DBG) ++n_batches;
    // Every register of `cur` is touched here, BEFORE the prefetch is issued: the compiler then waits
    // for these loads once per batch at this point.  Left to first use inside the pass loop (which
    // issues atomics, so the VM counter is untrackable) it emits `s_waitcnt vmcnt(0)` per pass and
    // every pass stalls on the previous pass's atomic (measured: 2.9k of 5.2k cycles per pass).
    asm volatile("" ::"v"(cur.q0.x), "v"(cur.q0.y), "v"(cur.q0.z), "v"(cur.q0.w), "v"(cur.q1.x), "v"(cur.q1.y),
                 "v"(cur.q1.z), "v"(cur.q1.w), "v"(cur.b), "v"(cur.g), "v"(nxt.g));
    stage_records(nxt, rec, batch_end - 64 - lane >= lo);
    g_next2 = (batch_end - 128 - lane >= lo) ? flatten_ids[batch_end - 128 - lane] : 0;
    const bool live = batch_end - lane >= lo;
    const bool hit = live && ellipse_hits_rect(cur.q0.x, cur.q0.y, cur.q1.y, cur.q0.z, cur.q0.w, cur.q1.x, qx0, qx1, qy0, qy1);
    unsigned long long mask = __ballot(hit);
    while (mask) {
      const int bit = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const float gx = bcast(cur.q0.x, bit), gy = bcast(cur.q0.y, bit);
      const float ca = bcast(cur.q0.z, bit), cb = bcast(cur.q0.w, bit), cc = bcast(cur.q1.x, bit);
      const float opac = bcast(cur.q1.y, bit);
      const float dx = gx - px, dy = gy - py;
      const float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
      const float vis = __expf(-sigma);
      const float alpha = fminf(kAlphaMax, opac * vis);
      const bool valid = inside && (batch_end - bit <= bin_final) && !(sigma < 0.f || alpha < kAlphaMin);
      if (DBG) ++n_pass;
      if (__ballot(valid) == 0ull) continue;
      if (DBG) ++n_valid;
      float g_r = 0.f, g_g = 0.f, g_b = 0.f;
      float g_cx = 0.f, g_cy = 0.f, g_cz = 0.f, g_x = 0.f, g_y = 0.f, g_ax = 0.f, g_ay = 0.f, g_op = 0.f;
      if (valid) {
        const float cr = bcast(cur.q1.z, bit), cg = bcast(cur.q1.w, bit), cbl = bcast(cur.b, bit);
        const float ra = __builtin_amdgcn_rcpf(1.f - alpha);
        T *= ra;
        const float fac = alpha * T;
        g_r = fac * vr; g_g = fac * vg; g_b = fac * vb;
        float v_alpha = (cr * T - buf_r * ra) * vr + (cg * T - buf_g * ra) * vg + (cbl * T - buf_b * ra) * vb;
        buf_r += cr * fac; buf_g += cg * fac; buf_b += cbl * fac;
        v_alpha += T_final * ra * v_a;
        v_alpha -= T_final * ra * bg_dot;
        if (opac * vis <= kAlphaMax) {
          const float v_sigma = -opac * vis * v_alpha;
          g_cx = 0.5f * v_sigma * dx * dx;
          g_cy = v_sigma * dx * dy;
          g_cz = 0.5f * v_sigma * dy * dy;
          g_x = v_sigma * (ca * dx + cb * dy);
          g_y = v_sigma * (cb * dx + cc * dy);
          if (ABS) { g_ax = fabsf(g_x); g_ay = fabsf(g_y); }
          g_op = vis * v_alpha;
        }
      }
      const float v8[8] = {g_x, g_y, g_cx, g_cy, g_cz, g_r, g_g, g_b};
      float val;
      float r_op;
      if (DBG && (dbg_variant & 2)) {   // ablation: no cross-lane reduction
        val = g_x + g_y + g_cx + g_cy + g_cz + g_r + g_g + g_b;
        r_op = g_op;
      } else {
        val = row_reduce8_transposed(v8, lane);
        r_op = row_allreduce_sum(g_op);
      }
      if (l15 == 8) val = r_op;
      if (ABS) {
        const float r_ax = row_allreduce_sum(g_ax), r_ay = row_allreduce_sum(g_ay);
        if (l15 == 9) val = r_ax;
        if (l15 == 10) val = r_ay;
      }
      const int32_t gid = __builtin_amdgcn_readlane(cur.g, bit);
      if (DBG && (dbg_variant & 4)) {   // ablation: the four rows add to the same addresses (slow!)
        if (out_lane && val != 0.f) atomicAdd(out_base + (int64_t)gid * 16, val);
      } else if (DBG && (dbg_variant & 1)) {   // ablation: no atomic (value kept alive)
        asm volatile("" ::"v"(val), "s"(gid));
      } else {
        val = rows_combine(val);     // nine lanes, nine distinct addresses, one 64-byte record
        if (lane <= (ABS ? 10 : 8) && val != 0.f) atomicAdd(out_base + (int64_t)gid * 16, val);
      }
    }
    cur = nxt;
    nxt.g = g_next2;
  }
  if (DBG && lane == 0) {
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    unsigned long long *o = dbg + 6 * (int64_t)blockIdx.x;
    o[0] = t_start; o[1] = t_loop; o[2] = t_end; o[3] = (unsigned long long)n_pass; o[4] = (unsigned long long)n_valid;
    o[5] = (unsigned long long)n_batches;
  }
}

}  // namespace so

extern "C" int so_rasterize_fwd_wave(int C, int N, int width, int height, const float *rec, const float *backgrounds,
                                     const int32_t *isect_offsets, const int32_t *flatten_ids,
                                     const int32_t *n_isects_dev, int64_t n_isects_host, float *render_colors,
                                     float *render_alphas, int32_t *last_ids, void *stream) {
  SO_REQUIRE(C >= 0 && N >= 0 && width > 0 && height > 0, "so_rasterize_fwd_wave: bad sizes");
  if (C == 0) return SO_OK;
  SO_REQUIRE(isect_offsets && render_colors && render_alphas && last_ids && (N == 0 || rec), "so_rasterize_fwd_wave: null pointer");
  SO_REQUIRE((((uintptr_t)rec) & 63) == 0, "so_rasterize_fwd_wave: rec must be 64-byte aligned");
  const int tile_w = (width + 15) / 16, tile_h = (height + 15) / 16;
  const int64_t blocks = 4 * (int64_t)C * tile_w * tile_h;
  SO_REQUIRE(blocks < ((int64_t)1 << 31), "so_rasterize_fwd_wave: grid too large");
  hipLaunchKernelGGL(so::k_rasterize_fwd_wave, dim3((unsigned)blocks), dim3(64), 0, so::as_stream(stream), C, width, height,
                     tile_w, tile_h, reinterpret_cast<const float4 *>(rec), backgrounds, isect_offsets, flatten_ids,
                     n_isects_dev, n_isects_host, render_colors, render_alphas, last_ids);
  return so::check_launch("so_rasterize_fwd_wave");
}

extern "C" int so_rasterize_bwd_wave(int C, int N, int width, int height, const float *rec, const float *backgrounds,
                                     const int32_t *isect_offsets, const int32_t *flatten_ids,
                                     const int32_t *n_isects_dev, int64_t n_isects_host, const float *render_alphas,
                                     const int32_t *last_ids, const float *v_render_colors,
                                     const float *v_render_alphas, float *vrec, int absgrad, void *stream) {
  SO_REQUIRE(C >= 0 && N >= 0 && width > 0 && height > 0, "so_rasterize_bwd_wave: bad sizes");
  if (C == 0 || N == 0) return SO_OK;
  SO_REQUIRE(rec && isect_offsets && render_alphas && last_ids && v_render_colors && v_render_alphas && vrec,
             "so_rasterize_bwd_wave: null pointer");
  SO_REQUIRE(((((uintptr_t)rec) | ((uintptr_t)vrec)) & 63) == 0, "so_rasterize_bwd_wave: records must be 64-byte aligned");
  const int tile_w = (width + 15) / 16, tile_h = (height + 15) / 16;
  const int64_t blocks = 4 * (int64_t)C * tile_w * tile_h;
  SO_REQUIRE(blocks < ((int64_t)1 << 31), "so_rasterize_bwd_wave: grid too large");
  hipStream_t st = so::as_stream(stream);
  if (absgrad)
    hipLaunchKernelGGL((so::k_rasterize_bwd_wave<true, false>), dim3((unsigned)blocks), dim3(64), 0, st, C, width, height, tile_w,
                       tile_h, reinterpret_cast<const float4 *>(rec), backgrounds, isect_offsets, flatten_ids,
                       n_isects_dev, n_isects_host, render_alphas, last_ids, v_render_colors, v_render_alphas, vrec, nullptr, 0);
  else
    hipLaunchKernelGGL((so::k_rasterize_bwd_wave<false, false>), dim3((unsigned)blocks), dim3(64), 0, st, C, width, height, tile_w,
                       tile_h, reinterpret_cast<const float4 *>(rec), backgrounds, isect_offsets, flatten_ids,
                       n_isects_dev, n_isects_host, render_alphas, last_ids, v_render_colors, v_render_alphas, vrec, nullptr, 0);
  return so::check_launch("so_rasterize_bwd_wave");
}

/* Diagnostic build of the same kernel: stamps[6*blocks] (u64) = {t_start, t_loop, t_end (s_memtime
 * ticks), passes, passes with a valid pixel, batches} per wave.  Never used by the product path. */
extern "C" int so_debug_rasterize_bwd_wave_stamps(int C, int N, int width, int height, const float *rec,
                                                  const int32_t *isect_offsets, const int32_t *flatten_ids,
                                                  const int32_t *n_isects_dev, const float *render_alphas,
                                                  const int32_t *last_ids, const float *v_render_colors,
                                                  const float *v_render_alphas, float *vrec,
                                                  unsigned long long *stamps, int variant, void *stream) {
  SO_REQUIRE(C > 0 && N > 0 && rec && stamps, "so_debug_rasterize_bwd_wave_stamps: bad arguments");
  const int tile_w = (width + 15) / 16, tile_h = (height + 15) / 16;
  const int64_t blocks = 4 * (int64_t)C * tile_w * tile_h;
  hipLaunchKernelGGL((so::k_rasterize_bwd_wave<false, true>), dim3((unsigned)blocks), dim3(64), 0, so::as_stream(stream), C,
                     width, height, tile_w, tile_h, reinterpret_cast<const float4 *>(rec), nullptr, isect_offsets,
                     flatten_ids, n_isects_dev, 0, render_alphas, last_ids, v_render_colors, v_render_alphas, vrec, stamps, variant);
  return so::check_launch("so_debug_rasterize_bwd_wave_stamps");
}
