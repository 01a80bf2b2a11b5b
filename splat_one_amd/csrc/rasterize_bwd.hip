// rasterize_bwd.hip -- K10: backward of the tile rasteriser, one workgroup per tile (gfx950).
//
// Replaces gsplat `rasterize_to_pixels` backward (`loss.backward()` at
// /root/reference/utils/gsplat_utils/gsplat_trainer.py:655).  Algorithm: SURVEY.md B.2.
//
// Each pixel walks its tile list back to front from `last_ids`, rebuilding T; per-pixel gradient
// contributions are summed over the wave with DPP row reductions (no LDS round trip) and one lane
// per wave issues the float atomics.  As in the forward, a wave first ballots which staged
// Gaussians can touch its 8x8 quadrant at all and only walks those; whole batches behind the
// tile's last contributor are never loaded.
#include "rasterize_common.hpp"

// The D == 3 pass works on packed fp32 pairs with the conic applied ONCE per pass (rasterize_common.hpp, round 3):
// q = Q d serves sigma = 1/2 d.q AND the position gradient v_sigma q.

namespace so {

typedef raster_v2f bwd_v2f;

// PACKED: inputs from the 64-byte records rec[g] (passed through `colors`), gradients into the
// 64-byte records vrec[g] = {v_x,v_y,v_ca,v_cb,v_cc,v_r,v_g,v_b | v_opac,abs_x,abs_y,..} (passed
// through `v_colors`): one atomic instruction = one memory-side request per (quadrant, Gaussian).
template <int D, int TS, bool ABS, bool PACKED, bool SMALL = false>
__global__ void __launch_bounds__(TS *TS)
k_rasterize_bwd(int C, int N, int W, int H, int tile_w, int tile_h, const float2 *__restrict__ means2d,
                const float *__restrict__ conics, const float *__restrict__ colors,
                const float *__restrict__ opacities, const float *__restrict__ backgrounds,
                const uint8_t *__restrict__ tile_masks, const int32_t *__restrict__ offsets,
                const int32_t *__restrict__ flatten_ids, const int32_t *__restrict__ n_isects_dev,
                int64_t n_isects_host, const float *__restrict__ render_alphas,
                const int32_t *__restrict__ last_ids, const float *__restrict__ v_render_colors,
                const float *__restrict__ v_render_alphas, float *__restrict__ v_means2d,
                float *__restrict__ v_means2d_abs, float *__restrict__ v_conics, float *__restrict__ v_colors,
                float *__restrict__ v_opacities, int wrap_flags, const LossFinal fin) {
  constexpr int BLOCK = TS * TS;
  // list entries staged per batch.  SO_BWD_STAGE < BLOCK: less LDS per workgroup (17 KB at 256 entries), so that more
  // workgroups fit a CU and a new one can start as soon as four wave slots are free (the four waves of a tile finish at
  // different times); lists longer than the stage take more batches.
#ifndef SO_BWD_STAGE
#define SO_BWD_STAGE 256
#endif
  constexpr int STAGE = (SO_BWD_STAGE < BLOCK) ? SO_BWD_STAGE : BLOCK;
  if (fin.sums && blockIdx.x == 0 && threadIdx.x == 0) {   // (see LossFinal: the loss kernel before this one has completed)
    const float l1m = fin.sums[0] * fin.a_l1, ssm = fin.sums[1] * fin.b_ss;
    fin.out[0] = fin.w_l1 / fin.a_l1 * l1m + fin.w_ssim / fin.b_ss * ssm + fin.c_const;
    fin.out[1] = l1m;
    fin.out[2] = 1.f - ssm;
  }
  if (fin.skip && *fin.skip != 0) return;   // uniform over the grid
  constexpr int NWAVES = (BLOCK + 63) / 64;
  // staged per Gaussian (same records as the forward): A = (x, y, conic a, conic b),
  // B = (conic c, opacity [, r, g when D == 3]), remaining colour channels in s_col
  constexpr int DC = (D == 3) ? 1 : D;
  __shared__ float4 s_A[STAGE];
  __shared__ float4 s_B[STAGE];
  __shared__ float4 s_box[STAGE];
  __shared__ float s_col[(D == 3) ? 1 : STAGE * DC];
  __shared__ float4 s_C[(D == 3) ? STAGE : 1];   // RGB: blue, 16-byte strided like s_A / s_B (one address register per pass)
  __shared__ int32_t s_id[STAGE];
  __shared__ int32_t s_wave_last[NWAVES];

  // the host checked C * tile_w * tile_h < 2^31 (32-bit index arithmetic)
  const int n_tiles = tile_w * tile_h;
  const int M = C * n_tiles;
  // list segments (LossFinal::seg_len): workgroup b of the grid of seg_count x M takes segment b / M of the tile that workgroup
  // b % M of the plain grid would take -- the first segments of all tiles first, in the tile order
  const int seg = fin.seg_len > 0 ? (int)(blockIdx.x / (unsigned)M) : 0;
  const unsigned bt = fin.seg_len > 0 ? blockIdx.x - (unsigned)seg * (unsigned)M : blockIdx.x;
  const int ct = fin.tile_order ? fin.tile_order[bt] : (int)xcd_remap(bt, M);   // (LossFinal::tile_order: longest list first)
  if (tile_masks && !tile_masks[ct]) return;
  const int c = ct / n_tiles;
  const int t = ct - c * n_tiles;
  const int ty = t / tile_w, tx = t - ty * tile_w;
  // periodic image (SO_TILE_WRAP_*): every staged Gaussian is shifted by the multiple of W that brings it closest to
  // this tile, so a footprint that crosses the +-pi seam of a panorama continues on the other side
  const bool wrap = wrap_for(wrap_flags, c);
  const float wrap_w = (float)W, wrap_cx = (float)(tx * TS) + 0.5f * (float)TS;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int lx, ly, wx0, wy0;
  PixelMap<TS>::get(tid, lx, ly, wx0, wy0);
  const int j = tx * TS + lx, i = ty * TS + ly;
  const bool inside = (i < H) && (j < W);
  const float px = (float)j + 0.5f, py = (float)i + 0.5f;
  const int64_t pix = ((int64_t)c * H + i) * W + j;
  const float qx0 = (float)(tx * TS + wx0) + 0.5f, qx1 = qx0 + 7.f;
  const float qy0 = (float)(ty * TS + wy0) + 0.5f, qy1 = qy0 + 7.f;

  int64_t lo, hi;
  tile_list_range(ct, M, offsets, n_isects_dev, n_isects_host, lo, hi);
  if (hi <= lo) return;  // uniform over the block
  // this workgroup's part of the list: [seg_lo, seg_hi] (the whole list without segments)
  const int64_t seg_lo = fin.seg_len > 0 ? lo + (int64_t)seg * fin.seg_len : lo;
  const int64_t seg_hi = (fin.seg_len > 0 && seg < fin.seg_count - 1 && seg_lo + fin.seg_len < hi) ? seg_lo + fin.seg_len - 1 : hi - 1;   // (the last segment: all that is left)
  if (seg_lo >= hi) return;   // uniform

  const float T_final = inside ? 1.f - render_alphas[pix] : 1.f;
  float T = T_final;
  // The reference keeps a per-channel "colour behind this Gaussian" buffer and needs only its dot product
  // with the pixel's upstream colour gradient: carry that scalar (buf_dot = sum_k buffer[k] * v_c[k]).
  float v_c[D];
  float buf_dot = 0.f;
  float bg_dot = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    v_c[k] = inside ? v_render_colors[pix * D + k] : 0.f;
    if (backgrounds) bg_dot += backgrounds[c * D + k] * v_c[k];
  }
  const float v_a = inside ? v_render_alphas[pix] : 0.f;
  const float tf_bg = T_final * (v_a - bg_dot);
  // last contributor of this pixel; pixels that nothing reached keep lo-1 (no Gaussian valid)
  int32_t bin_final = (int32_t)lo - 1;
  if (inside && T_final < 1.f) bin_final = last_ids[pix];
  // a pixel whose walk goes on behind this segment starts from the state the forward left at the segment's far end; one
  // whose last contributor lies before the segment has nothing to do here
  const bool from_boundary = fin.seg_len > 0 && (int64_t)bin_final > seg_hi;
  if (fin.seg_len > 0) bin_final = (int64_t)bin_final < seg_lo ? (int32_t)lo - 1 : (from_boundary ? (int32_t)seg_hi : bin_final);
  // wave / block maxima
  const int32_t wave_last = wave_max_i32(bin_final);   // wave-uniform (scalar)
  if (lane == 0) s_wave_last[wid] = wave_last;
  __syncthreads();
  int32_t block_last = s_wave_last[0];
#pragma unroll
  for (int w = 1; w < NWAVES; ++w) block_last = max(block_last, s_wave_last[w]);
  if (block_last < lo) return;  // uniform

  // D == 3: per-row transposing butterfly; lane l of each DPP row owns one output slot
  //   (l&15) < 8 : slot_of_lane(l) in {v_x,v_y,v_ca,v_cb,v_cc,v_r,v_g,v_b};  8: v_opac;  9,10: abs x,y
  const int l15 = lane & 15;
  const int slot = l15 < 8 ? slot_of_lane(lane) : l15;
  float *out_base = nullptr;
  int out_stride = 0;
  if (D == 3) {
    if (PACKED) { out_base = v_colors + slot; out_stride = 16; }
    else if (slot < 2) { out_base = v_means2d + slot; out_stride = 2; }
    else if (slot < 5) { out_base = v_conics + (slot - 2); out_stride = 3; }
    else if (slot < 8) { out_base = v_colors + (slot - 5); out_stride = 3; }
    else if (slot == 8) { out_base = v_opacities; out_stride = 1; }
    else if (ABS && slot < 11) { out_base = v_means2d_abs + (slot - 9); out_stride = 2; }
  }

  // the nine-sum network's lanes (so_common.hpp::wave_reduce9_scattered): RGB without absgrad, and -- round 4 -- the first
  // three channels + geometry of every D >= 4 call (RGB+ED, what the depth-supervision loss renders: gsplat_trainer.py:595)
  const bool atom_lane = ((kReduce9Lanes >> lane) & 1ull) != 0;
  const int slot9 = reduce9_slot_of_lane(lane);
  float *out_base9 = nullptr;
  int out_stride9 = 0;
  if (D >= 3 && !PACKED) {
    if (slot9 < 2) { out_base9 = v_means2d + slot9; out_stride9 = 2; }
    else if (slot9 < 5) { out_base9 = v_conics + (slot9 - 2); out_stride9 = 3; }
    else if (slot9 < 8) { out_base9 = v_colors + (slot9 - 5); out_stride9 = D; }
    else { out_base9 = v_opacities; out_stride9 = 1; }
  } else if (D == 3) {
    out_base9 = v_colors + slot9; out_stride9 = 16;
  }
  // what undoes the units of the staged conic on the reduced sums: 2 ln 2 on the mean2d slots (and their absolute
  // values), the 1/2 of dL/d(ca, cc) -- by slot, for the lanes of either reduction
  const float unscale9 = slot9 < 2 ? kConicUnscale : ((slot9 == 2 || slot9 == 4) ? 0.5f : 1.f);
  const float unscale = (slot < 2 || slot == 9 || slot == 10) ? kConicUnscale : ((slot == 2 || slot == 4) ? 0.5f : 1.f);
  const bwd_v2f pxy = {px, py};
  float behind = tf_bg;   // tf_bg - buf_dot of the scalar form below
  if constexpr (D == 3) {
    if (from_boundary) {
      // (T, colour accumulated up to the boundary) of this pixel; the colour behind it = all that was accumulated - that
      const float4 st = fin.seg_state[(int64_t)seg * ((int64_t)C * H * W) + pix];
      T = st.x;
      float all_dot = 0.f, upto_dot = st.y * v_c[0] + st.z * v_c[1] + st.w * v_c[2];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float acc_k = fin.render_colors[pix * 3 + k] - (backgrounds ? T_final * backgrounds[c * 3 + k] : 0.f);
        all_dot = fmaf(acc_k, v_c[k], all_dot);
      }
      behind = tf_bg - (all_dot - upto_dot);
    }
  }

  for (int64_t batch_end = block_last; batch_end >= seg_lo; batch_end -= STAGE) {
    lds_barrier();
    const int64_t idx = batch_end - tid;
    if (tid < STAGE && idx >= seg_lo) {
      const int32_t g = flatten_ids[idx];
      s_id[tid] = g;
      if (PACKED) {
        const float4 *r4 = reinterpret_cast<const float4 *>(colors) + 4 * (int64_t)g;
        float4 q0 = r4[0];
        const float4 q1 = r4[1];   // x,y,ca,cb | cc,opac,r,g
        float4 bx = r4[3];         // the cull box of this Gaussian, computed once in the kernel that wrote the record
        if (wrap) {
          const float shift = wrap_w * rintf((q0.x - wrap_cx) / wrap_w);
          q0.x -= shift; bx.x -= shift; bx.y -= shift;
        }
        s_box[tid] = bx;
        // (x, y, ca, cb) | (cb, cc, opacity) | (red, green, blue): (ca, cb) and (cb, cc) are aligned register pairs after the loads
        // s_C.w: the byte offset of this Gaussian's gradient record, so that the atomic's address needs no further LDS read
        const float4 q2 = r4[2];                                    // blue, depth, radius, cull threshold
        q0.z *= kConicScale; q0.w *= kConicScale;                   // conic and threshold in units of the exponent of 2 (as the forward)
        s_A[tid] = q0;
        s_B[tid] = make_float4(q0.w, q1.x * kConicScale, q1.y, q2.w * kConicScale);
        s_C[tid] = make_float4(q1.z, q1.w, q2.x, __uint_as_float((unsigned)g * 64u));
      } else {
        float2 xy = means2d[g];
        if (wrap) xy.x -= wrap_w * rintf((xy.x - wrap_cx) / wrap_w);
        const float op = opacities[g];
        const float ca = conics[3 * (int64_t)g], cb = conics[3 * (int64_t)g + 1], cc = conics[3 * (int64_t)g + 2];
        s_A[tid] = (D == 3) ? make_float4(xy.x, xy.y, ca * kConicScale, cb * kConicScale) : make_float4(xy.x, xy.y, ca, cb);
        s_box[tid] = alpha_bound_box(xy.x, xy.y, op, ca, cb, cc);
        if (D == 3) {
          s_B[tid] = make_float4(cb * kConicScale, cc * kConicScale, op, cull_tau(op, ca, cb, cc) * kConicScale);
          s_C[tid] = make_float4(colors[(int64_t)g * D], colors[(int64_t)g * D + 1], colors[(int64_t)g * D + 2], 0.f);
        } else {
          s_B[tid] = make_float4(cc, op, cull_tau(op, ca, cb, cc), 0.f);
#pragma unroll
          for (int k = 0; k < D; ++k) s_col[tid * DC + k] = colors[(int64_t)g * D + k];
        }
      }
    }
    lds_barrier();
    const int batch_size = (int)((batch_end + 1 - seg_lo) < STAGE ? (batch_end + 1 - seg_lo) : STAGE);
    const int32_t rel_final = (int32_t)(batch_end - bin_final);   // candidate tt contributes to this pixel iff tt >= rel_final
    const int32_t rel_wave = (int32_t)(batch_end - wave_last);
#pragma unroll 1
    for (int chunk0 = 0; chunk0 < batch_size; chunk0 += 64) {
      const int cand = chunk0 + lane;
      bool hit = false;
      if (cand < batch_size && cand >= rel_wave) {
        const float4 bx = s_box[cand];
        hit = !(bx.y < qx0 || bx.x > qx1 || bx.w < qy0 || bx.z > qy1);
        if (hit) {   // bounding boxes overlap: settle it with the exact ellipse-rectangle test
          const float4 a = s_A[cand];
          const float4 bq = s_B[cand];
          // (x, y, cull threshold of this Gaussian, conic) against this wave's quadrant
          if (D == 3) hit = ellipse_hits_rect(a.x, a.y, bq.w, a.z, a.w, bq.y, qx0, qx1, qy0, qy1);
          else hit = ellipse_hits_rect(a.x, a.y, bq.z, a.z, a.w, bq.x, qx0, qx1, qy0, qy1);
        }
      }
      unsigned long long mask = __ballot(hit);
      while (mask) {
        const int bit = __ffsll((long long)mask) - 1;
        mask = clear_bit(mask, bit);
        const int tt = chunk0 + bit;
        if constexpr (D == 3) {
          const float4 a = s_A[tt];            // x, y, ca, cb
          const float4 b4 = s_B[tt];           // cb, cc, opacity, (cull threshold)
          const float4 c4 = s_C[tt];           // red, green, blue, record offset (issued with the other two: one address register, one wait)
          const bwd_v2f d = bwd_v2f{a.x, a.y} - pxy;
          // q = Q d = (ca dx + cb dy, cb dx + cc dy):  sigma = 1/2 d.q,  d sigma / d mean = q
          const bwd_v2f q = conic_times(a.z, a.w, b4.x, b4.y, d);
          const float s2 = fmaf(q.y, d.y, q.x * d.x);                       // sigma log2(e): the staged conic is pre-scaled
          const float vis = gauss_vis(s2);                                  // exp(-sigma), bit for bit the forward's
          const float ov = b4.z * vis;
          const float alpha = fminf(kAlphaMax, ov);
          const bool valid = (tt >= rel_final) && !(s2 < 0.f || alpha < kAlphaMin);
          // keeps the colour / offset reads where they are written (hipcc sinks them below the branch, which costs two address moves)
          asm volatile("" ::"v"(c4.x), "v"(c4.y), "v"(c4.z), "v"(c4.w));
          if (__ballot(valid) == 0ull) continue;
          const float alpha_v = valid ? alpha : 0.f;
          const float ra = __builtin_amdgcn_rcpf(1.f - alpha_v);
          T *= ra;
          const float fac = alpha_v * T;
          const float cv = fmaf(c4.z, v_c[2], fmaf(c4.y, v_c[1], c4.x * v_c[0]));
          const bwd_v2f g01 = bwd_v2f{fac, fac} * bwd_v2f{v_c[0], v_c[1]};
          const float g2 = fac * v_c[2];
          // behind = T_final (v_a - bg.v_c) - sum of the colours behind this Gaussian . v_c, carried as one scalar
          float v_alpha = fmaf(T, cv, ra * behind);
          behind = fmaf(-fac, cv, behind);
          v_alpha = (valid && ov <= kAlphaMax) ? v_alpha : 0.f;             // the clamp at 0.999 has zero slope
          const float v_sigma = -ov * v_alpha;
          const float g_op = vis * v_alpha;
          const bwd_v2f vs2 = {v_sigma, v_sigma};
          // per pixel in the staged units, put right once per pass on the reduced sums (unscale9 / unscale below)
          const bwd_v2f gxy = vs2 * q;                                      // k d L / d mean2d
          const bwd_v2f t = vs2 * d;
          const bwd_v2f gcxz = t * d;                                       // 2 d L / d (ca, cc)
          const float g_cy = t.x * d.y;                                     // d L / d cb
          const float v8[8] = {gxy.x, gxy.y, gcxz.x, g_cy, gcxz.y, g01.x, g01.y, g2};
          if constexpr (!ABS) {
            // nine sums over the wave as one network (so_common.hpp): nine lanes, ONE atomic instruction, nine addresses
#if defined(SO_ABL_NORED)    // ablation (tools/gpu_bwd_ablate.sh): no cross-lane reduction -- WRONG gradients, timing only
            const float val = (v8[0] + v8[1] + v8[2] + v8[3] + v8[4] + v8[5] + v8[6] + v8[7] + g_op) * unscale9;
#else
            const float val = wave_reduce9_scattered(v8, g_op) * unscale9;
#endif
#if defined(SO_ABL_NOATOM)   // ablation: the reduced value is kept alive but never added -- ZERO gradients, timing only
            asm volatile("" ::"v"(val));
            if (false) {
#else
            if (atom_lane) {
#endif
              if constexpr (PACKED && SMALL) {
                const unsigned off = __float_as_uint(c4.w) | ((unsigned)slot9 * 4u);
                atomicAdd(reinterpret_cast<float *>(reinterpret_cast<char *>(v_colors) + off), val);
              } else {
                atomicAdd(out_base9 + (int64_t)s_id[tt] * out_stride9, val);
              }
            }
            continue;
          }
          float val = row_reduce8_transposed(v8, lane);
          const float r_op = row_allreduce_sum(g_op);
          if (l15 == 8) val = r_op;
          if (ABS) {
            const float r_ax = row_allreduce_sum(fabsf(gxy.x)), r_ay = row_allreduce_sum(fabsf(gxy.y));
            if (l15 == 9) val = r_ax;
            if (l15 == 10) val = r_ay;
          }
          val = rows_combine(val) * unscale;
          if (lane <= (ABS ? 10 : 8) && val != 0.f) {
            if constexpr (PACKED && SMALL) {
              // one 64-byte record per Gaussian: 32-bit byte offset from the uniform base (C N 64 B < 4 GB, checked by the launcher)
              const unsigned off = (unsigned)s_id[tt] * 64u + (unsigned)slot * 4u;
              atomicAdd(reinterpret_cast<float *>(reinterpret_cast<char *>(v_colors) + off), val);
            } else {
              atomicAdd(out_base + (int64_t)s_id[tt] * out_stride, val);
            }
          }
          continue;
        } else {
        // ---- D != 3 (depth channel, N-D features): the scalar pass
        const float4 a = s_A[tt];
        const float4 bq = s_B[tt];
        const float opac = bq.y;
        const float dx = a.x - px, dy = a.y - py;
        const float sigma = 0.5f * (a.z * dx * dx + bq.x * dy * dy) + a.w * dx * dy;
        const float vis = __expf(-sigma);
        const float ov = opac * vis;
        const float alpha = fminf(kAlphaMax, ov);
        // pixels outside the image carry bin_final = lo - 1, i.e. rel_final > every tt
        const bool valid = (tt >= rel_final) && !(sigma < 0.f || alpha < kAlphaMin);
        if (__ballot(valid) == 0ull) continue;
        // Branch-free from here: a lane that does not take part uses alpha 0, which leaves T and the
        // colour buffer untouched and makes its gradient terms vanish.
        const float alpha_v = valid ? alpha : 0.f;
        const float ra = __builtin_amdgcn_rcpf(1.f - alpha_v);   // 1 ulp; alpha <= 0.999
        T *= ra;
        const float fac = alpha_v * T;
        float g_col[D];
        float cv = 0.f;                          // sum_k colour[k] * v_c[k]
#pragma unroll
        for (int k = 0; k < D; ++k) {
          cv = fmaf(s_col[tt * DC + k], v_c[k], cv);
          g_col[k] = fac * v_c[k];
        }
        // v_alpha = sum_k (c_k T - buffer_k ra) v_c[k] + T_final ra (v_a - bg . v_c)
        const float v_alpha = fmaf(T, cv, ra * (tf_bg - buf_dot));
        buf_dot = fmaf(fac, cv, buf_dot);
        const bool grad_on = valid && (ov <= kAlphaMax);   // the clamp at 0.999 has zero slope
        const float v_sigma = grad_on ? -ov * v_alpha : 0.f;
        const float g_op = grad_on ? vis * v_alpha : 0.f;
        const float t1 = v_sigma * dx, t2 = v_sigma * dy;
        const float g_cx = 0.5f * (t1 * dx), g_cy = t1 * dy, g_cz = 0.5f * (t2 * dy);
        const float g_x = fmaf(a.z, t1, a.w * t2), g_y = fmaf(a.w, t1, bq.x * t2);
        float g_ax = 0.f, g_ay = 0.f;
        if (ABS) { g_ax = fabsf(g_x); g_ay = fabsf(g_y); }
        if constexpr (D >= 4 && !ABS) {
          // geometry + the first three channels through the nine-sum network (ONE atomic instruction, nine addresses);
          // the channels beyond (the depth of RGB+ED, N-D features) as plain wave sums into lane 63.  Until round 3 all
          // 6 + D sums went the second way: 10 single-lane atomics per pass for RGB+ED.  (With absgrad the old form stays:
          // |gradient| sums and gradient sums are then added in the SAME order, so absgrad >= |grad| holds to the last bit.)
          const float v8[8] = {g_x, g_y, g_cx, g_cy, g_cz, g_col[0], g_col[1], g_col[2]};
          const float val9 = wave_reduce9_scattered(v8, g_op);
#pragma unroll
          for (int k = 3; k < D; ++k) g_col[k] = wave_reduce_sum_to_last(g_col[k]);
          const int64_t g = s_id[tt];
          if (atom_lane) atomicAdd(out_base9 + g * out_stride9, val9);
          if (lane == 63) {
#pragma unroll
            for (int k = 3; k < D; ++k) atomicAdd(v_colors + g * D + k, g_col[k]);
          }
          continue;
        }
        // one or two channels: wave sums land in lane 63
#pragma unroll
        for (int k = 0; k < D; ++k) g_col[k] = wave_reduce_sum_to_last(g_col[k]);
        const float w_cx = wave_reduce_sum_to_last(g_cx), w_cy = wave_reduce_sum_to_last(g_cy);
        const float w_cz = wave_reduce_sum_to_last(g_cz);
        const float w_x = wave_reduce_sum_to_last(g_x), w_y = wave_reduce_sum_to_last(g_y);
        if (ABS) { g_ax = wave_reduce_sum_to_last(g_ax); g_ay = wave_reduce_sum_to_last(g_ay); }
        const float w_op = wave_reduce_sum_to_last(g_op);
        if (lane == 63) {
          const int64_t g = s_id[tt];
#pragma unroll
          for (int k = 0; k < D; ++k) atomicAdd(v_colors + g * D + k, g_col[k]);
          atomicAdd(v_conics + 3 * g, w_cx);
          atomicAdd(v_conics + 3 * g + 1, w_cy);
          atomicAdd(v_conics + 3 * g + 2, w_cz);
          atomicAdd(v_means2d + 2 * g, w_x);
          atomicAdd(v_means2d + 2 * g + 1, w_y);
          if (ABS) {
            atomicAdd(v_means2d_abs + 2 * g, g_ax);
            atomicAdd(v_means2d_abs + 2 * g + 1, g_ay);
          }
          atomicAdd(v_opacities + g, w_op);
        }
        }   // D != 3
      }
    }
  }
}

template <int D>
static int launch_bwd(int TS, bool abs_, dim3 grid, hipStream_t st, int C, int N, int W, int H, int tile_w, int tile_h,
                      const float *means2d, const float *conics, const float *colors, const float *opacities,
                      const float *backgrounds, const uint8_t *tile_masks, const int32_t *offsets,
                      const int32_t *flatten_ids, const int32_t *n_dev, int64_t n_host, const float *ra,
                      const int32_t *last, const float *v_rc, const float *v_ra, float *v_m, float *v_abs, float *v_cn,
                      float *v_col, float *v_op, int wrap_flags) {
  const float2 *m2 = reinterpret_cast<const float2 *>(means2d);
#define SO_GO(TSV, ABSV)                                                                                         \
  hipLaunchKernelGGL((k_rasterize_bwd<D, TSV, ABSV, false>), grid, dim3(TSV * TSV), 0, st, C, N, W, H, tile_w, tile_h, \
                     m2, conics, colors, opacities, backgrounds, tile_masks, offsets, flatten_ids, n_dev,       \
                     n_host, ra, last, v_rc, v_ra, v_m, v_abs, v_cn, v_col, v_op, wrap_flags, LossFinal{})
  if (TS == 16) { if (abs_) SO_GO(16, true); else SO_GO(16, false); }
  else          { if (abs_) SO_GO(8, true);  else SO_GO(8, false); }
#undef SO_GO
  return check_launch("so_rasterize_bwd");
}

int rasterize_bwd_packed_launch(int C, int N, int width, int height, int tile_size, const float *rec, const float *backgrounds,
                                const int32_t *isect_offsets, const int32_t *flatten_ids, const int32_t *n_isects_dev,
                                int64_t n_isects_host, const float *render_alphas, const int32_t *last_ids,
                                const float *v_render_colors, const float *v_render_alphas, float *vrec, int absgrad,
                                const LossFinal &fin, void *stream);
int rasterize_bwd_tile_launch(int C, int N, int width, int height, int tile_w, int tile_h, const float *rec,
                              const float *backgrounds, const int32_t *isect_offsets, const int32_t *flatten_ids,
                              const int32_t *n_isects_dev, int64_t n_isects_host, const float *render_alphas,
                              const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
                              float *vrec, int wrap_flags, const LossFinal &fin, hipStream_t st);
}  // namespace so

extern "C" int so_rasterize_bwd(int C, int N, int D, int width, int height, int tile_size, const float *means2d,
                                const float *conics, const float *colors, const float *opacities,
                                const float *backgrounds, const uint8_t *tile_masks,
                                const int32_t *isect_offsets, const int32_t *flatten_ids,
                                const int32_t *n_isects_dev, int64_t n_isects_host, const float *render_alphas,
                                const int32_t *last_ids, const float *v_render_colors,
                                const float *v_render_alphas, float *v_means2d, float *v_means2d_abs,
                                float *v_conics, float *v_colors, float *v_opacities, void *stream) {
  SO_REQUIRE(C >= 0 && N >= 0 && width > 0 && height > 0, "so_rasterize_bwd: bad sizes");
  const int wrap_flags = tile_size & ~0xFF;
  tile_size = so::tile_size_of(tile_size);
  SO_REQUIRE(tile_size == 16 || tile_size == 8, "so_rasterize_bwd: tile_size %d not in {8,16}", tile_size);
  SO_REQUIRE(!wrap_flags || width % tile_size == 0, "so_rasterize_bwd: SO_TILE_WRAP_* needs width %% tile_size == 0");
  if (C == 0 || N == 0) return SO_OK;
  SO_REQUIRE(means2d && conics && colors && opacities && isect_offsets && render_alphas && last_ids &&
                 v_render_colors && v_render_alphas && v_means2d && v_conics && v_colors && v_opacities,
             "so_rasterize_bwd: null pointer");
  SO_REQUIRE(n_isects_dev || n_isects_host == 0 || flatten_ids, "so_rasterize_bwd: null flatten_ids");
  const int tile_w = (width + tile_size - 1) / tile_size, tile_h = (height + tile_size - 1) / tile_size;
  SO_REQUIRE((int64_t)C * tile_w * tile_h < (int64_t)INT32_MAX, "so_rasterize_bwd: C*tiles = %lld does not fit 31 bits",
             (long long)C * tile_w * tile_h);
  const dim3 grid((unsigned)((int64_t)C * tile_w * tile_h));
  hipStream_t st = so::as_stream(stream);
#define SO_CASE(DD)                                                                                             \
  case DD:                                                                                                      \
    return so::launch_bwd<DD>(tile_size, v_means2d_abs != nullptr, grid, st, C, N, width, height, tile_w,       \
                              tile_h, means2d, conics, colors, opacities, backgrounds, tile_masks,              \
                              isect_offsets, flatten_ids, n_isects_dev, n_isects_host, render_alphas, last_ids, \
                              v_render_colors, v_render_alphas, v_means2d, v_means2d_abs, v_conics, v_colors,   \
                              v_opacities, wrap_flags);
  switch (D) {
    SO_CASE(1) SO_CASE(2) SO_CASE(3) SO_CASE(4) SO_CASE(5) SO_CASE(8) SO_CASE(9) SO_CASE(16) SO_CASE(17) SO_CASE(32) SO_CASE(33)
    default:
      so::set_error("so_rasterize_bwd: unsupported channel count D=%d", D);
      return SO_ERR_UNSUPPORTED;
  }
#undef SO_CASE
}

extern "C" int so_rasterize_bwd_packed(int C, int N, int width, int height, int tile_size, const float *rec,
                                       const float *backgrounds, const int32_t *isect_offsets,
                                       const int32_t *flatten_ids, const int32_t *n_isects_dev,
                                       int64_t n_isects_host, const float *render_alphas, const int32_t *last_ids,
                                       const float *v_render_colors, const float *v_render_alphas, float *vrec,
                                       int absgrad, void *stream) {
  return so::rasterize_bwd_packed_launch(C, N, width, height, tile_size, rec, backgrounds, isect_offsets, flatten_ids, n_isects_dev,
                                         n_isects_host, render_alphas, last_ids, v_render_colors, v_render_alphas, vrec, absgrad,
                                         so::LossFinal{}, stream);
}

// internal (step.hip): the same launch, with the fused step's loss scalars finalised by its first thread
int so::rasterize_bwd_packed_launch(int C, int N, int width, int height, int tile_size, const float *rec, const float *backgrounds,
                                    const int32_t *isect_offsets, const int32_t *flatten_ids, const int32_t *n_isects_dev,
                                    int64_t n_isects_host, const float *render_alphas, const int32_t *last_ids,
                                    const float *v_render_colors, const float *v_render_alphas, float *vrec, int absgrad,
                                    const so::LossFinal &fin, void *stream) {
  SO_REQUIRE(C >= 0 && N >= 0 && width > 0 && height > 0, "so_rasterize_bwd_packed: bad sizes");
  const int wrap_flags = tile_size & ~0xFF;
  tile_size = so::tile_size_of(tile_size);
  SO_REQUIRE(tile_size == 16 || tile_size == 8, "so_rasterize_bwd_packed: tile_size %d not in {8,16}", tile_size);
  SO_REQUIRE(!wrap_flags || width % tile_size == 0, "so_rasterize_bwd_packed: SO_TILE_WRAP_* needs width %% tile_size == 0");
  // the fused step's loss scalars are written by this kernel's first thread: a skipped launch would leave the previous
  // iteration's values behind (so_train_step_fwd_bwd rejects empty problems before it gets here; keep it that way)
  SO_REQUIRE(!fin.sums || (C > 0 && N > 0), "so_rasterize_bwd_packed: the loss scalars need a launch (C = %d, N = %d)", C, N);
  if (C == 0 || N == 0) return SO_OK;
  SO_REQUIRE(rec && isect_offsets && render_alphas && last_ids && v_render_colors && v_render_alphas && vrec,
             "so_rasterize_bwd_packed: null pointer");
  SO_REQUIRE(((((uintptr_t)rec) | ((uintptr_t)vrec)) & 63) == 0, "so_rasterize_bwd_packed: records must be 64-byte aligned");
  const int tile_w = (width + tile_size - 1) / tile_size, tile_h = (height + tile_size - 1) / tile_size;
  SO_REQUIRE((int64_t)C * tile_w * tile_h < (int64_t)INT32_MAX, "so_rasterize_bwd_packed: C*tiles = %lld does not fit 31 bits",
             (long long)C * tile_w * tile_h);
  hipStream_t st = so::as_stream(stream);
  // list segments (LossFinal::seg_len): quadrant waves on 16x16 tiles only; seg_count workgroups per tile
  const bool tile_waves_asked = fin.tile_waves < 0 ? false : fin.tile_waves != 0;
  SO_REQUIRE(fin.seg_len == 0 || (fin.seg_len % 256 == 0 && fin.seg_count >= 1 && fin.seg_count <= 64 && fin.seg_state && fin.render_colors &&
                                  tile_size == 16 && !tile_waves_asked && (int64_t)C * tile_w * tile_h * fin.seg_count < (int64_t)INT32_MAX),
             "so_rasterize_bwd_packed: list segments need 16x16 tiles, quadrant waves, the forward's segment states and colours");
  const dim3 grid((unsigned)((int64_t)C * tile_w * tile_h * (fin.seg_len > 0 ? fin.seg_count : 1)));
  // 16x16 tiles without absgrad, on request: one wave per tile, one reduction and one atomic per (tile, Gaussian) --
  // rasterize_bwd_tile.hip.  Measured (profiles/r04_experiments.json: rasterize_bwd_tile_waves): 22 % fewer vector
  // instructions, but a tile is then one wave's serial chain -- 957 -> 803 us on the dense regime's 520-entry lists, 73 ->
  // 76 us on c2's 31-entry lists, 177 -> 204 us at 106 entries.  The fused engine asks for it when its probe finds >= 256
  // entries per tile (so_step_desc.raster_impl = 1); SPLAT_ONE_AMD_BWD_TILE=1 / 0 sets the default of every other caller.
  static const bool env_tile = [] { const char *e = getenv("SPLAT_ONE_AMD_BWD_TILE"); return e && e[0] == '1'; }();
  const bool tile_waves = fin.tile_waves < 0 ? env_tile : fin.tile_waves != 0;
  if (tile_size == 16 && !absgrad && tile_waves && fin.seg_len == 0)
    return so::rasterize_bwd_tile_launch(C, N, width, height, tile_w, tile_h, rec, backgrounds, isect_offsets, flatten_ids, n_isects_dev,
                                         n_isects_host, render_alphas, last_ids, v_render_colors, v_render_alphas, vrec, wrap_flags, fin, st);
  // SMALL: every gradient record lies within 4 GB of `vrec` (C N 64 B): the atomic's address is a 32-bit offset from the base
  const bool small = (int64_t)C * N < ((int64_t)1 << 26);
#define SO_GO_(TSV, ABSV, SM)                                                                                     \
  hipLaunchKernelGGL((so::k_rasterize_bwd<3, TSV, ABSV, true, SM>), grid, dim3(TSV * TSV), 0, st, C, N, width, height, \
                     tile_w, tile_h, nullptr, nullptr, rec, nullptr, backgrounds, nullptr, isect_offsets,         \
                     flatten_ids, n_isects_dev, n_isects_host, render_alphas, last_ids, v_render_colors,          \
                     v_render_alphas, nullptr, nullptr, nullptr, vrec, nullptr, wrap_flags, fin)
#define SO_GO(TSV, ABSV) do { if (small) SO_GO_(TSV, ABSV, true); else SO_GO_(TSV, ABSV, false); } while (0)
  if (tile_size == 16) { if (absgrad) SO_GO(16, true); else SO_GO(16, false); }
  else                 { if (absgrad) SO_GO(8, true);  else SO_GO(8, false); }
#undef SO_GO
#undef SO_GO_
  return so::check_launch("so_rasterize_bwd_packed");
}

#ifdef SO_TILE_PERM_EXPERIMENT
extern "C" int so_debug_tile_perm_bwd(const int32_t *perm) {
  return hipMemcpyToSymbol(HIP_SYMBOL(so::g_tile_perm), &perm, sizeof(perm)) == hipSuccess ? 0 : 1;
}
#endif
