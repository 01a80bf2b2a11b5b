// rasterize_rows.hip -- K9 / K10 with ROW-INDEPENDENT passes: the tile rasteriser of the fused engine (gfx950).
//
// Replaces gsplat `rasterize_to_pixels` forward / backward (reached inside `rasterization`,
// /root/reference/utils/gsplat_utils/gsplat_trainer.py:477, and from `loss.backward()`, :655) on the packed
// 64-byte records of so_preprocess_fwd.  Algorithm: SURVEY.md B.1 step 7 / B.2.  Same results as the quadrant kernels
// of rasterize_fwd.hip / rasterize_bwd.hip (bit-identical forward, gradients equal up to atomic order).
//
// Why: in those kernels a wave owns an 8x8 pixel quadrant and one pass evaluates ONE Gaussian on all 64 pixels; on
// the c2 scene only 39 % of the lanes of a pass carry a pixel that reaches alpha >= 1/255 (round-1 SQ counters: VALU
// busy ~1.0, i.e. the kernels are bound by issued instructions, most of them on pixels that contribute nothing).
// Here the quadrant is cut into four 4x4 blocks, one per 16-lane DPP row, and every row walks ITS OWN list of the
// Gaussians that can touch its block: in one pass the four rows evaluate four different Gaussians.  A pass therefore
// costs the same instructions but there are max_r(len_r) of them instead of |union_r list_r| -- measured on c2
// (tools/passsim.py): 0.725x the passes, lane utilisation 39 % -> 54 %.  The backward's per-Gaussian sums need only
// the transposing butterfly over the 16 lanes of a row (no cross-row combine), and each row issues its 9 atomics into
// the 64-byte gradient record of its own Gaussian in the same instruction (36 distinct addresses in 4 lines).
//
// Lists: per staged batch of 256 Gaussians every wave tests each candidate against its quadrant's bounding box and
// then exactly (ellipse of alpha >= 1/255 against the pixel-centre rectangle) against each of its four blocks; the
// hits are compacted in list order with wave ballots + popcounts into LDS byte lists, one per (wave, row).
#include "rasterize_common.hpp"

namespace so {

constexpr int kRB = 256;   // threads per tile = Gaussians staged per batch

struct RowLists {
  uint8_t idx[4][4][kRB];   // [wave][row][position] -> index into the staged batch
};

// bit r set: the Gaussian can reach alpha >= 1/255 at a pixel centre of block r of the quadrant whose first pixel centre
// is (qx0, qy0).  Blocks: r & 1 selects the x half, r >> 1 the y half.
__device__ __forceinline__ unsigned blocks_hit(const float4 a, const float4 bq, float qx0, float qy0) {
  unsigned m = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float x0 = qx0 + (float)((r & 1) * 4), y0 = qy0 + (float)((r >> 1) * 4);
    if (ellipse_hits_rect(a.x, a.y, bq.y, a.z, a.w, bq.x, x0, x0 + 3.f, y0, y0 + 3.f)) m |= 1u << r;
  }
  return m;
}

// Appends the candidates of one 64-wide sub-chunk to the four row lists of this wave (list order preserved).
__device__ __forceinline__ void append_hits(uint8_t (*lists)[kRB], unsigned hm, int cand, int (&cnt)[4], unsigned long long lt) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool h = (hm >> r) & 1u;
    const unsigned long long m = __ballot(h);
    if (h) lists[r][cnt[r] + __popcll(m & lt)] = (uint8_t)cand;
    cnt[r] += __popcll(m);
  }
}

__device__ __forceinline__ int pick4(const int (&v)[4], int row) {
  return row == 0 ? v[0] : (row == 1 ? v[1] : (row == 2 ? v[2] : v[3]));
}

// pixel of a lane: wave w owns the quadrant (w & 1, w >> 1); DPP row r the 4x4 block (r & 1, r >> 1) of it
__device__ __forceinline__ void rows_pixel(int tid, int &lx, int &ly, int &wx0, int &wy0, int &row) {
  const int w = tid >> 6, l = tid & 63;
  row = l >> 4;
  const int l15 = l & 15;
  wx0 = (w & 1) * 8;
  wy0 = (w >> 1) * 8;
  lx = wx0 + (row & 1) * 4 + (l15 & 3);
  ly = wy0 + (row >> 1) * 4 + (l15 >> 2);
}

// ------------------------------------------------------------------------------------------------ forward
__global__ void __launch_bounds__(kRB)
k_rasterize_fwd_rows(int C, int W, int H, int tile_w, int tile_h, const float *__restrict__ rec,
                     const float *__restrict__ backgrounds, const int32_t *__restrict__ offsets,
                     const int32_t *__restrict__ flatten_ids, const int32_t *__restrict__ n_isects_dev, int64_t n_isects_host,
                     float *__restrict__ render_colors, float *__restrict__ render_alphas, int32_t *__restrict__ last_ids) {
  __shared__ float4 s_A[kRB];     // x, y, conic a, conic b
  __shared__ float4 s_B[kRB];     // conic c, opacity, r, g
  __shared__ float4 s_box[kRB];   // bounding box of the alpha >= 1/255 region
  __shared__ float s_blue[kRB];
  __shared__ RowLists s_l;

  const int n_tiles = tile_w * tile_h;
  const int M = C * n_tiles;
  const int ct = (int)xcd_remap(blockIdx.x, M);
  const int c = ct / n_tiles;
  const int t = ct - c * n_tiles;
  const int ty = t / tile_w, tx = t - ty * tile_w;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int lx, ly, wx0, wy0, row;
  rows_pixel(tid, lx, ly, wx0, wy0, row);
  const int j = tx * 16 + lx, i = ty * 16 + ly;
  const bool inside = (i < H) && (j < W);
  const float px = (float)j + 0.5f, py = (float)i + 0.5f;
  const int64_t pix = ((int64_t)c * H + i) * W + j;
  const float qx0 = (float)(tx * 16 + wx0) + 0.5f, qx1 = qx0 + 7.f;
  const float qy0 = (float)(ty * 16 + wy0) + 0.5f, qy1 = qy0 + 7.f;
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;

  int64_t lo, hi;
  tile_list_range(ct, M, offsets, n_isects_dev, n_isects_host, lo, hi);

  // a finished pixel carries T == 0 (every later contribution vanishes arithmetically) and its final transmittance in T_out
  float T = inside ? 1.f : 0.f, T_out = 0.f;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
  int32_t cur_idx = 0;
  uint8_t(*lists)[kRB] = s_l.idx[wv];
  const uint8_t *my_list = lists[row];

  for (int64_t batch_start = lo; batch_start < hi; batch_start += kRB) {
    if (__syncthreads_and(!(T > 0.f))) break;
    const int64_t idx = batch_start + tid;
    if (idx < hi) {
      const int32_t g = flatten_ids[idx];
      const float4 *r4 = reinterpret_cast<const float4 *>(rec) + 4 * (int64_t)g;
      const float4 q0 = r4[0], q1 = r4[1];
      s_A[tid] = q0;
      s_B[tid] = q1;
      s_box[tid] = alpha_bound_box(q0.x, q0.y, q1.y, q0.z, q0.w, q1.x);
      s_blue[tid] = reinterpret_cast<const float *>(r4 + 2)[0];
    }
    __syncthreads();
    const int batch_size = (int)((hi - batch_start) < kRB ? (hi - batch_start) : kRB);
    const int32_t batch_base = (int32_t)batch_start;
    // ---- the four row lists of this wave for the batch
    int cnt[4] = {0, 0, 0, 0};
#pragma unroll 1
    for (int chunk0 = 0; chunk0 < batch_size; chunk0 += 64) {
      const int cand = chunk0 + lane;
      unsigned hm = 0;
      if (cand < batch_size) {
        const float4 bx = s_box[cand];
        if (!(bx.y < qx0 || bx.x > qx1 || bx.w < qy0 || bx.z > qy1)) hm = blocks_hit(s_A[cand], s_B[cand], qx0, qy0);
      }
      append_hits(lists, hm, cand, cnt, lt);
    }
    __builtin_amdgcn_wave_barrier();   // the lists are written and read by this wave only (LDS operations of a wave stay in order)
    const int my_cnt = pick4(cnt, row);
    const int max_cnt = max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3]));
#pragma unroll 1
    for (int it = 0; it < max_cnt; ++it) {
      if (__ballot(T > 0.f) == 0ull) break;
      const bool active = it < my_cnt;
      const int tt = active ? (int)my_list[it] : 0;
      const float4 a = s_A[tt];
      const float4 bq = s_B[tt];
      const float blue = s_blue[tt];
      const float dx = a.x - px, dy = a.y - py;
      const float sigma = 0.5f * (a.z * dx * dx + bq.x * dy * dy) + a.w * dx * dy;
      float alpha = fminf(kAlphaMax, bq.y * __expf(-sigma));
      alpha = (sigma < 0.f) ? 0.f : alpha;
      alpha = (alpha < kAlphaMin) ? 0.f : alpha;
      alpha = active ? alpha : 0.f;
      const float next_T = T * (1.f - alpha);
      const bool stop = next_T <= kTStop;                 // also true for pixels already finished (T == 0)
      T_out += stop ? T : 0.f;
      const float vis = stop ? 0.f : alpha * T;
      T = stop ? 0.f : next_T;
      acc0 = fmaf(bq.z, vis, acc0);
      acc1 = fmaf(bq.w, vis, acc1);
      acc2 = fmaf(blue, vis, acc2);
      cur_idx = (vis > 0.f) ? batch_base + tt : cur_idx;
    }
  }
  if (inside) {
    T += T_out;
    render_alphas[pix] = 1.f - T;
    float *o = render_colors + pix * 3;
    if (backgrounds) {
      o[0] = acc0 + T * backgrounds[c * 3]; o[1] = acc1 + T * backgrounds[c * 3 + 1]; o[2] = acc2 + T * backgrounds[c * 3 + 2];
    } else {
      o[0] = acc0; o[1] = acc1; o[2] = acc2;
    }
    last_ids[pix] = cur_idx;
  }
}

// ------------------------------------------------------------------------------------------------ backward
template <bool ABS>
__global__ void __launch_bounds__(kRB)
k_rasterize_bwd_rows(int C, int W, int H, int tile_w, int tile_h, const float *__restrict__ rec,
                     const float *__restrict__ backgrounds, const int32_t *__restrict__ offsets,
                     const int32_t *__restrict__ flatten_ids, const int32_t *__restrict__ n_isects_dev, int64_t n_isects_host,
                     const float *__restrict__ render_alphas, const int32_t *__restrict__ last_ids,
                     const float *__restrict__ v_render_colors, const float *__restrict__ v_render_alphas,
                     float *__restrict__ vrec) {
  __shared__ float4 s_A[kRB];
  __shared__ float4 s_B[kRB];
  __shared__ float4 s_box[kRB];
  __shared__ float s_blue[kRB];
  __shared__ int32_t s_id[kRB];
  __shared__ int32_t s_wave_last[4];
  __shared__ RowLists s_l;

  const int n_tiles = tile_w * tile_h;
  const int M = C * n_tiles;
  const int ct = (int)xcd_remap(blockIdx.x, M);
  const int c = ct / n_tiles;
  const int t = ct - c * n_tiles;
  const int ty = t / tile_w, tx = t - ty * tile_w;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int lx, ly, wx0, wy0, row;
  rows_pixel(tid, lx, ly, wx0, wy0, row);
  const int j = tx * 16 + lx, i = ty * 16 + ly;
  const bool inside = (i < H) && (j < W);
  const float px = (float)j + 0.5f, py = (float)i + 0.5f;
  const int64_t pix = ((int64_t)c * H + i) * W + j;
  const float qx0 = (float)(tx * 16 + wx0) + 0.5f, qx1 = qx0 + 7.f;
  const float qy0 = (float)(ty * 16 + wy0) + 0.5f, qy1 = qy0 + 7.f;
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;

  int64_t lo, hi;
  tile_list_range(ct, M, offsets, n_isects_dev, n_isects_host, lo, hi);
  if (hi <= lo) return;  // uniform over the block

  const float T_final = inside ? 1.f - render_alphas[pix] : 1.f;
  float T = T_final;
  // only the dot product of the "colour behind this Gaussian" buffer with the pixel's upstream colour gradient is needed
  float v_c0 = 0.f, v_c1 = 0.f, v_c2 = 0.f, buf_dot = 0.f, bg_dot = 0.f;
  if (inside) {
    const float *vp = v_render_colors + pix * 3;
    v_c0 = vp[0]; v_c1 = vp[1]; v_c2 = vp[2];
  }
  if (backgrounds) bg_dot = backgrounds[c * 3] * v_c0 + backgrounds[c * 3 + 1] * v_c1 + backgrounds[c * 3 + 2] * v_c2;
  const float v_a = inside ? v_render_alphas[pix] : 0.f;
  const float tf_bg = T_final * (v_a - bg_dot);
  // last contributor of this pixel; pixels that nothing reached keep lo-1 (no Gaussian valid)
  int32_t bin_final = (int32_t)lo - 1;
  if (inside && T_final < 1.f) bin_final = last_ids[pix];
  // row / wave / tile maxima (shuffles with xor < 16 stay inside the DPP row)
  int32_t row_last = bin_final;
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1) row_last = max(row_last, __shfl_xor(row_last, d, 64));
  int32_t rl[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) rl[r] = __builtin_amdgcn_readlane(row_last, 16 * r);
  const int32_t wave_last = max(max(rl[0], rl[1]), max(rl[2], rl[3]));
  if (lane == 0) s_wave_last[wv] = wave_last;
  __syncthreads();
  const int32_t block_last = max(max(s_wave_last[0], s_wave_last[1]), max(s_wave_last[2], s_wave_last[3]));
  if (block_last < lo) return;  // uniform

  // per-row transposing butterfly: lane l of each row owns one output slot
  //   (l & 15) < 8: slot_of_lane(l) in {v_x, v_y, v_ca, v_cb, v_cc, v_r, v_g, v_b};  8: v_opac;  9, 10: abs x, y
  const int l15 = lane & 15;
  const int slot = l15 < 8 ? slot_of_lane(lane) : l15;
  float *out_base = vrec + slot;
  uint8_t(*lists)[kRB] = s_l.idx[wv];
  const uint8_t *my_list = lists[row];

  for (int64_t batch_end = block_last; batch_end >= lo; batch_end -= kRB) {
    __syncthreads();
    const int64_t idx = batch_end - tid;
    if (idx >= lo) {
      const int32_t g = flatten_ids[idx];
      s_id[tid] = g;
      const float4 *r4 = reinterpret_cast<const float4 *>(rec) + 4 * (int64_t)g;
      const float4 q0 = r4[0], q1 = r4[1];   // x,y,ca,cb | cc,opac,r,g
      s_A[tid] = q0;
      s_B[tid] = q1;
      s_box[tid] = alpha_bound_box(q0.x, q0.y, q1.y, q0.z, q0.w, q1.x);
      s_blue[tid] = reinterpret_cast<const float *>(r4 + 2)[0];
    }
    __syncthreads();
    const int batch_size = (int)((batch_end + 1 - lo) < kRB ? (batch_end + 1 - lo) : kRB);
    const int32_t rel_final = (int32_t)(batch_end - bin_final);   // candidate tt contributes to this pixel iff tt >= rel_final
    int32_t rel_row[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rel_row[r] = (int32_t)(batch_end - rl[r]);
    const int32_t rel_wave = (int32_t)(batch_end - wave_last);
    int cnt[4] = {0, 0, 0, 0};
#pragma unroll 1
    for (int chunk0 = 0; chunk0 < batch_size; chunk0 += 64) {
      const int cand = chunk0 + lane;
      unsigned hm = 0;
      if (cand < batch_size && cand >= rel_wave) {
        const float4 bx = s_box[cand];
        if (!(bx.y < qx0 || bx.x > qx1 || bx.w < qy0 || bx.z > qy1)) {
          hm = blocks_hit(s_A[cand], s_B[cand], qx0, qy0);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (cand < rel_row[r]) hm &= ~(1u << r);     // behind the last contributor of every pixel of that block
        }
      }
      append_hits(lists, hm, cand, cnt, lt);
    }
    __builtin_amdgcn_wave_barrier();
    const int my_cnt = pick4(cnt, row);
    const int max_cnt = max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3]));
#pragma unroll 1
    for (int it = 0; it < max_cnt; ++it) {
      const bool active = it < my_cnt;
      const int tt = active ? (int)my_list[it] : 0;
      const float4 a = s_A[tt];
      const float4 bq = s_B[tt];
      const float opac = bq.y;
      const float dx = a.x - px, dy = a.y - py;
      const float sigma = 0.5f * (a.z * dx * dx + bq.x * dy * dy) + a.w * dx * dy;
      const float vis = __expf(-sigma);
      const float ov = opac * vis;
      const float alpha = fminf(kAlphaMax, ov);
      // pixels outside the image carry bin_final = lo - 1, i.e. rel_final > every tt
      const bool valid = active && (tt >= rel_final) && !(sigma < 0.f || alpha < kAlphaMin);
      if (__ballot(valid) == 0ull) continue;
      // Branch-free from here: a lane that does not take part uses alpha 0, which leaves T and the colour buffer
      // untouched and makes its gradient terms vanish.
      const float alpha_v = valid ? alpha : 0.f;
      const float ra = __builtin_amdgcn_rcpf(1.f - alpha_v);   // 1 ulp; alpha <= 0.999
      T *= ra;
      const float fac = alpha_v * T;
      const float cv = fmaf(s_blue[tt], v_c2, fmaf(bq.w, v_c1, bq.z * v_c0));   // sum_k colour[k] * v_c[k]
      // v_alpha = sum_k (c_k T - buffer_k ra) v_c[k] + T_final ra (v_a - bg . v_c)
      const float v_alpha = fmaf(T, cv, ra * (tf_bg - buf_dot));
      buf_dot = fmaf(fac, cv, buf_dot);
      const bool grad_on = valid && (ov <= kAlphaMax);   // the clamp at 0.999 has zero slope
      const float v_sigma = grad_on ? -ov * v_alpha : 0.f;
      const float g_op = grad_on ? vis * v_alpha : 0.f;
      const float t1 = v_sigma * dx, t2 = v_sigma * dy;
      const float v8[8] = {fmaf(a.z, t1, a.w * t2), fmaf(a.w, t1, bq.x * t2), 0.5f * (t1 * dx), t1 * dy, 0.5f * (t2 * dy),
                           fac * v_c0, fac * v_c1, fac * v_c2};
      float val = row_reduce8_transposed(v8, lane);
      const float r_op = row_allreduce_sum(g_op);
      if (l15 == 8) val = r_op;
      if (ABS) {
        const float r_ax = row_allreduce_sum(fabsf(v8[0])), r_ay = row_allreduce_sum(fabsf(v8[1]));
        if (l15 == 9) val = r_ax;
        if (l15 == 10) val = r_ay;
      }
      // every row adds the totals of ITS Gaussian: 9 (11) distinct addresses inside one 64-byte record per row
      if (l15 <= (ABS ? 10 : 8) && val != 0.f) atomicAdd(out_base + (int64_t)s_id[tt] * 16, val);
    }
  }
}

}  // namespace so

extern "C" int so_rasterize_fwd_rows(int C, int N, int width, int height, const float *rec, const float *backgrounds,
                                     const int32_t *isect_offsets, const int32_t *flatten_ids, const int32_t *n_isects_dev,
                                     int64_t n_isects_host, float *render_colors, float *render_alphas, int32_t *last_ids,
                                     void *stream) {
  SO_REQUIRE(C >= 0 && N >= 0 && width > 0 && height > 0, "so_rasterize_fwd_rows: bad sizes");
  if (C == 0) return SO_OK;
  SO_REQUIRE(isect_offsets && render_colors && render_alphas && last_ids && (N == 0 || rec), "so_rasterize_fwd_rows: null pointer");
  SO_REQUIRE((((uintptr_t)rec) & 63) == 0, "so_rasterize_fwd_rows: rec must be 64-byte aligned");
  const int tile_w = (width + 15) / 16, tile_h = (height + 15) / 16;
  SO_REQUIRE((int64_t)C * tile_w * tile_h < (int64_t)INT32_MAX, "so_rasterize_fwd_rows: C*tiles = %lld does not fit 31 bits",
             (long long)C * tile_w * tile_h);
  hipLaunchKernelGGL(so::k_rasterize_fwd_rows, dim3((unsigned)((int64_t)C * tile_w * tile_h)), dim3(so::kRB), 0, so::as_stream(stream),
                     C, width, height, tile_w, tile_h, rec, backgrounds, isect_offsets, flatten_ids, n_isects_dev, n_isects_host,
                     render_colors, render_alphas, last_ids);
  return so::check_launch("so_rasterize_fwd_rows");
}

extern "C" int so_rasterize_bwd_rows(int C, int N, int width, int height, const float *rec, const float *backgrounds,
                                     const int32_t *isect_offsets, const int32_t *flatten_ids, const int32_t *n_isects_dev,
                                     int64_t n_isects_host, const float *render_alphas, const int32_t *last_ids,
                                     const float *v_render_colors, const float *v_render_alphas, float *vrec, int absgrad,
                                     void *stream) {
  SO_REQUIRE(C >= 0 && N >= 0 && width > 0 && height > 0, "so_rasterize_bwd_rows: bad sizes");
  if (C == 0 || N == 0) return SO_OK;
  SO_REQUIRE(rec && isect_offsets && render_alphas && last_ids && v_render_colors && v_render_alphas && vrec,
             "so_rasterize_bwd_rows: null pointer");
  SO_REQUIRE(((((uintptr_t)rec) | ((uintptr_t)vrec)) & 63) == 0, "so_rasterize_bwd_rows: records must be 64-byte aligned");
  const int tile_w = (width + 15) / 16, tile_h = (height + 15) / 16;
  SO_REQUIRE((int64_t)C * tile_w * tile_h < (int64_t)INT32_MAX, "so_rasterize_bwd_rows: C*tiles = %lld does not fit 31 bits",
             (long long)C * tile_w * tile_h);
  const dim3 grid((unsigned)((int64_t)C * tile_w * tile_h));
  hipStream_t st = so::as_stream(stream);
  if (absgrad)
    hipLaunchKernelGGL((so::k_rasterize_bwd_rows<true>), grid, dim3(so::kRB), 0, st, C, width, height, tile_w, tile_h, rec, backgrounds,
                       isect_offsets, flatten_ids, n_isects_dev, n_isects_host, render_alphas, last_ids, v_render_colors,
                       v_render_alphas, vrec);
  else
    hipLaunchKernelGGL((so::k_rasterize_bwd_rows<false>), grid, dim3(so::kRB), 0, st, C, width, height, tile_w, tile_h, rec, backgrounds,
                       isect_offsets, flatten_ids, n_isects_dev, n_isects_host, render_alphas, last_ids, v_render_colors,
                       v_render_alphas, vrec);
  return so::check_launch("so_rasterize_bwd_rows");
}
