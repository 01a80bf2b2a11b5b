// rasterize_fwd.hip -- K9: front-to-back alpha compositing, one workgroup per tile (gfx950).
//
// Replaces gsplat `rasterize_to_pixels` forward (reached inside `rasterization`,
// /root/reference/utils/gsplat_utils/gsplat_trainer.py:477).  Algorithm: SURVEY.md B.1 step 7.
//
// wave64 structure: a 16x16 tile is 4 waves, each owning an 8x8 pixel quadrant.  The tile's
// depth-sorted list is staged through LDS in batches of one Gaussian per thread (coalesced read
// of flatten_ids, gathered 36+4D B per Gaussian).  Each wave then tests 64 staged Gaussians at a
// time against its quadrant (one lane per Gaussian, `__ballot`) and walks only the set bits, so
// Gaussians that cannot reach alpha >= 1/255 anywhere in the quadrant cost one lane-compare
// instead of 64 pixel evaluations.  The cull is exact: it never changes a pixel.
#include "rasterize_common.hpp"
#include "wave_sort.hpp"          // the prologue sort (SORT): wave_sort_list, sort_mid_chunks, rank_sort_global

namespace so {

// PACKED: the four per-Gaussian inputs come from ONE 64-byte record rec[g] = {x,y,ca,cb | cc,opac,r,g |
// b,..} written by so_preprocess_fwd (one cache line per gathered Gaussian instead of four); passed
// through the `colors` pointer, D must be 3.
// SORT (round 5; PACKED, binned lists, so_step_desc.sort_in_rasteriser): `flatten_ids` is NOT sorted yet -- the tile's keys
// {depth bits, id} stand in `sort_keys` as the binning pass left them, and this workgroup sorts its own list before it walks
// it: <= 256 keys by wave 0 in registers (ids to global memory for the backward AND to LDS for the first batch here, so no
// global round trip), <= 2048 by the whole workgroup (register-sorted chunks + rank search in the 16 KB the staging arrays
// occupy later), longer by the scratch-free rank sort.  One launch and one pass over the keys fewer per iteration where the
// lists are short (c2: the sort kernel's 8.4 us + a launch boundary against ~3 us more in here).
template <int D, int TS, bool PACKED, bool SORT = false>
__global__ void __launch_bounds__(TS *TS)
k_rasterize_fwd(int C, int N, int W, int H, int tile_w, int tile_h, const float2 *__restrict__ means2d,
                const float *__restrict__ conics, const float *__restrict__ colors,
                const float *__restrict__ opacities, const float *__restrict__ backgrounds,
                const uint8_t *__restrict__ tile_masks, const int32_t *__restrict__ offsets,
                const int32_t *__restrict__ flatten_ids, const int32_t *__restrict__ n_isects_dev,
                int64_t n_isects_host, float *__restrict__ render_colors, float *__restrict__ render_alphas,
                int32_t *__restrict__ last_ids, int wrap_flags, const int32_t *__restrict__ tile_order,
                const uint64_t *__restrict__ sort_keys = nullptr, float4 *__restrict__ seg_state = nullptr, int seg_batches = 0,
                int seg_boundaries = 0) {
  static_assert(!SORT || (PACKED && D == 3 && TS == 16), "the prologue sort exists for the packed RGB 16x16 kernel");
  constexpr int BLOCK = TS * TS;
  // staged per Gaussian: A = (x, y, conic a, conic b), B = (conic c, opacity [, r, g when D == 3]),
  // remaining colour channels in s_col -- two 16-byte broadcast reads + one 4-byte read per pass for RGB
  constexpr int DC = (D == 3) ? 1 : D;   // channels kept in s_col
  // (one array carved in four: with SORT the medium-list sort parks its 2048 keys in the same 16 KB before the staging starts)
  __shared__ float4 s_raw[((D == 3) ? 4 : 3) * BLOCK + ((D == 3) ? 0 : 1)];
  float4 *const s_A = s_raw, *const s_B = s_raw + BLOCK;
  float4 *const s_box = s_raw + 2 * BLOCK;   // xmin, xmax, ymin, ymax of the alpha>=1/255 region
  __shared__ float s_col[(D == 3) ? 1 : BLOCK * DC];
  // RGB: blue sits in a third 16-byte-strided array, so one address register (tt * 16) serves all three reads of a pass
  float4 *const s_C = s_raw + 3 * BLOCK;
  __shared__ int32_t s_ids[SORT ? BLOCK : 1];

  // the host checked C * tile_w * tile_h < 2^31: 32-bit index arithmetic (a 64-bit division is ~100 instructions)
  const int n_tiles = tile_w * tile_h;
  const int M = C * n_tiles;
  // tile_order (nullable; LossFinal::tile_order): longest list first -- long-list regimes, where a tile's walk is long
  // against the pixel loads the XCD-local order saves
  const int ct = tile_order ? tile_order[blockIdx.x] : (int)xcd_remap(blockIdx.x, M);
  const int c = ct / n_tiles;
  const int t = ct - c * n_tiles;
  const int ty = t / tile_w, tx = t - ty * tile_w;
  // periodic image (SO_TILE_WRAP_*): every staged Gaussian is shifted by the multiple of W that brings it closest to
  // this tile, so a footprint that crosses the +-pi seam of a panorama continues on the other side
  const bool wrap = wrap_for(wrap_flags, c);
  const float wrap_w = (float)W, wrap_cx = (float)(tx * TS) + 0.5f * (float)TS;
  const int tid = threadIdx.x;
  int lx, ly, wx0, wy0;
  PixelMap<TS>::get(tid, lx, ly, wx0, wy0);
  const int j = tx * TS + lx, i = ty * TS + ly;
  const bool inside = (i < H) && (j < W);
  const float px = (float)j + 0.5f, py = (float)i + 0.5f;
  const int64_t pix = ((int64_t)c * H + i) * W + j;
  // pixel-centre extent of this wave's quadrant
  const float qx0 = (float)(tx * TS + wx0) + 0.5f, qx1 = qx0 + 7.f;
  const float qy0 = (float)(ty * TS + wy0) + 0.5f, qy1 = qy0 + 7.f;

  if (tile_masks && !tile_masks[ct]) {
    if (inside) {
#pragma unroll
      for (int k = 0; k < D; ++k) render_colors[pix * D + k] = backgrounds ? backgrounds[c * D + k] : 0.f;
      render_alphas[pix] = 0.f;
      last_ids[pix] = 0;
    }
    return;
  }

  int64_t lo, hi;
  tile_list_range(ct, M, offsets, n_isects_dev, n_isects_host, lo, hi);
  bool ids_in_lds = false;
  if constexpr (SORT) {
    const int L = (int)(hi - lo);
    if (L > 0) {       // (uniform over the workgroup)
      int32_t *ids_out = const_cast<int32_t *>(flatten_ids);
      if (L <= 256) {
        if (tid < 64) {
          if (L <= 64) wave_sort_list<1>(sort_keys, lo, L, tid, ct, n_tiles, 0, ids_out, nullptr, s_ids);
          else if (L <= 128) wave_sort_list<2>(sort_keys, lo, L, tid, ct, n_tiles, 0, ids_out, nullptr, s_ids);
          else wave_sort_list<4>(sort_keys, lo, L, tid, ct, n_tiles, 0, ids_out, nullptr, s_ids);
        }
        ids_in_lds = true;
      } else if (L <= 2048) {
        sort_mid_chunks<BLOCK>(reinterpret_cast<uint64_t *>(s_raw), sort_keys, lo, L, ct, n_tiles, 0, ids_out, nullptr);
      } else {
        rank_sort_global(sort_keys, lo, L, ct, n_tiles, 0, ids_out, nullptr);
      }
      __syncthreads();   // the ids (LDS, or global memory written by this workgroup) before anybody stages them
    }
  }

  // Per-pixel state.  A finished pixel (outside the image, or transmittance exhausted) has T == 0, which
  // makes every later contribution vanish arithmetically -- the pass body below has no branches.  T_out
  // keeps the transmittance the pixel stopped at (exactly one of T, T_out is non-zero at the end).
  // (RGB, round 3: the same idea with one instruction less per pass -- T_live is the live transmittance or 0, T is never zeroed)
  constexpr bool V2 = (D == 3);
  float T = inside ? 1.f : 0.f, T_out = 0.f;
  float T_live = T;
  int cur_slot = -1;
  float acc[D];
#pragma unroll
  for (int k = 0; k < D; ++k) acc[k] = 0.f;
  int32_t cur_idx = 0;
  const int lane = tid & 63;
  const raster_v2f pxy = {px, py};

  // "some pixel of this wave is still live", written by lane 0 of every wave at the end of a batch and read by all after
  // the barrier that also protects the staged records: one ballot + one LDS word instead of __syncthreads_and (a
  // 35-instruction DPP reduction), and nothing at all for the single-batch lists that are the rule.
  __shared__ int s_live[(BLOCK + 63) / 64];
  const int wid = tid >> 6;

  for (int64_t batch_start = lo; batch_start < hi; batch_start += BLOCK) {
    if (batch_start != lo) {   // uniform
      lds_barrier();
      int live = 0;
#pragma unroll
      for (int w = 0; w < (BLOCK + 63) / 64; ++w) live |= s_live[w];
      if (!live) break;        // uniform: every wave read the same words
    }
    const int64_t idx = batch_start + tid;
    if (idx < hi) {
      const int32_t g = (SORT && ids_in_lds) ? s_ids[tid] : flatten_ids[idx];
      if (PACKED) {
        const float4 *r4 = reinterpret_cast<const float4 *>(colors) + 4 * (int64_t)g;
        float4 q0 = r4[0];
        const float4 q1 = r4[1];   // x,y,ca,cb | cc,opac,r,g
        const float4 q2 = r4[2];   // blue, depth, radius, cull threshold
        float4 bx = r4[3];         // the cull box of this Gaussian, computed once in the kernel that wrote the record
        // (all four quarters of the record requested together: left to itself hipcc requests the two words it needs of the
        // third quarter only after the first two have arrived -- one more round trip per staged batch)
        asm volatile("" ::"v"(q2.x), "v"(q2.w));
        if (wrap) {
          const float shift = wrap_w * rintf((q0.x - wrap_cx) / wrap_w);
          q0.x -= shift; bx.x -= shift; bx.y -= shift;
        }
        s_box[tid] = bx;
        q0.z *= kConicScale; q0.w *= kConicScale;                   // conic and threshold in units of the exponent of 2
        s_A[tid] = q0;
        s_B[tid] = make_float4(q0.w, q1.x * kConicScale, q1.y, q2.w * kConicScale);   // cb, cc, opacity, cull threshold
        s_C[tid] = make_float4(q1.z, q1.w, q2.x, 0.f);              // red, green, blue
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
    const int batch_size = (int)((hi - batch_start) < BLOCK ? (hi - batch_start) : BLOCK);
    const int32_t batch_base = (int32_t)batch_start;
#pragma unroll 1
    for (int chunk0 = 0; chunk0 < batch_size; chunk0 += 64) {
      if constexpr (V2) {
        if (__ballot(T_live > 0.f) == 0ull) break;   // (the pass loop tests this itself after every pass, with the compare it has)
      }
      const int cand = chunk0 + lane;
      bool hit = false;
      if (cand < batch_size) {
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
        if constexpr (!V2) {
          if (__ballot(T > 0.f) == 0ull) break;
        }
        const int bit = __ffsll((long long)mask) - 1;
        mask = clear_bit(mask, bit);                // one scalar instruction (mask &= mask - 1 is three)
        const int tt = chunk0 + bit;
        if constexpr (D == 3) {
          const float4 a = s_A[tt];                 // x, y, ca, cb
          const float4 b4 = s_B[tt];                // cb, cc, opacity
          const float4 c4 = s_C[tt];                // red, green, blue (issued with the other two reads, not at its use)
          const raster_v2f d = raster_v2f{a.x, a.y} - pxy;
          const raster_v2f q = conic_times(a.z, a.w, b4.x, b4.y, d);
          const float s2 = fmaf(q.y, d.y, q.x * d.x);                       // sigma log2(e) (pre-scaled conic): the backward recomputes exactly this
          float alpha = fminf(kAlphaMax, b4.z * gauss_vis(s2));
          alpha = (s2 < 0.f || alpha < kAlphaMin) ? 0.f : alpha;            // skipped Gaussian == zero alpha (one select)
          // T_live is the transmittance while the pixel is live and 0 afterwards (every later contribution vanishes
          // arithmetically); T keeps the value the pixel stopped at
          const float next_T = T_live * (1.f - alpha);
          const bool stop = next_T <= kTStop;       // also true for pixels already finished (T_live == 0)
          const float vis = stop ? 0.f : alpha * T_live;
          T = stop ? T : next_T;
          T_live = stop ? 0.f : next_T;
          acc[0] = fmaf(c4.x, vis, acc[0]);
          acc[1] = fmaf(c4.y, vis, acc[1]);
          acc[2] = fmaf(c4.z, vis, acc[2]);
          // the last contributor as the LDS address the pass already holds in a register; turned into a list index per batch
          cur_slot = (vis > 0.f) ? tt * 16 : cur_slot;
          if (__ballot(!stop) == 0ull) break;       // the wave's last live pixel has just stopped
          continue;
        } else {
        // ---- D != 3 (depth channel, N-D features): the scalar pass
        const float4 a = s_A[tt];
        const float4 bq = s_B[tt];
        const float dx = a.x - px, dy = a.y - py;
        const float sigma = 0.5f * (a.z * dx * dx + bq.x * dy * dy) + a.w * dx * dy;
        float alpha = fminf(kAlphaMax, bq.y * __expf(-sigma));
        alpha = (sigma < 0.f) ? 0.f : alpha;
        alpha = (alpha < kAlphaMin) ? 0.f : alpha;          // skipped Gaussian == zero alpha
        const float next_T = T * (1.f - alpha);
        const bool stop = next_T <= kTStop;                 // also true for pixels already finished (T == 0)
        T_out += stop ? T : 0.f;
        const float vis = stop ? 0.f : alpha * T;
        T = stop ? 0.f : next_T;
#pragma unroll
        for (int k = 0; k < D; ++k) acc[k] = fmaf(s_col[tt * DC + k], vis, acc[k]);
        cur_idx = (vis > 0.f) ? batch_base + tt : cur_idx;
        }   // D != 3
      }
    }
    if constexpr (V2) {
      cur_idx = (cur_slot >= 0) ? batch_base + (cur_slot >> 4) : cur_idx;
      cur_slot = -1;
    }
    if (batch_start + BLOCK < hi) {   // uniform; read after the next iteration's barrier, overwritten only after the one below it
      const bool wave_live = __ballot((V2 ? T_live : T) > 0.f) != 0ull;
      if (lane == 0) s_live[wid] = wave_live ? 1 : 0;
      if constexpr (V2 && TS == 16) {
        // a segment boundary of the backward (LossFinal::seg_state): this pixel's live transmittance and accumulated colour after
        // the last entry of the segment (finished pixels: T_live == 0, never read -- their last contributor lies before)
        if (seg_state) {
          const int done = (int)((batch_start - lo) / BLOCK) + 1;          // batches walked so far (uniform)
          if (done % seg_batches == 0 && done / seg_batches <= seg_boundaries && inside)      // (the last segment takes all that is left)
            seg_state[(int64_t)(done / seg_batches - 1) * ((int64_t)C * H * W) + pix] = make_float4(T_live, acc[0], acc[1], acc[2]);
        }
      }
    }
  }
  if (inside) {
    T += T_out;
    render_alphas[pix] = 1.f - T;
#pragma unroll
    for (int k = 0; k < D; ++k) render_colors[pix * D + k] = backgrounds ? acc[k] + T * backgrounds[c * D + k] : acc[k];
    last_ids[pix] = cur_idx;
  }
}

template <int D>
static int launch_fwd(int TS, dim3 grid, hipStream_t st, int C, int N, int W, int H, int tile_w, int tile_h,
                      const float *means2d, const float *conics, const float *colors, const float *opacities,
                      const float *backgrounds, const uint8_t *tile_masks, const int32_t *offsets,
                      const int32_t *flatten_ids, const int32_t *n_dev, int64_t n_host, float *rc, float *ra,
                      int32_t *last, int wrap_flags) {
  const float2 *m2 = reinterpret_cast<const float2 *>(means2d);
  if (TS == 16)
    hipLaunchKernelGGL((k_rasterize_fwd<D, 16, false>), grid, dim3(256), 0, st, C, N, W, H, tile_w, tile_h, m2, conics, colors,
                       opacities, backgrounds, tile_masks, offsets, flatten_ids, n_dev, n_host, rc, ra, last, wrap_flags, (const int32_t *)nullptr,
                       (const uint64_t *)nullptr);
  else
    hipLaunchKernelGGL((k_rasterize_fwd<D, 8, false>), grid, dim3(64), 0, st, C, N, W, H, tile_w, tile_h, m2, conics, colors,
                       opacities, backgrounds, tile_masks, offsets, flatten_ids, n_dev, n_host, rc, ra, last, wrap_flags, (const int32_t *)nullptr,
                       (const uint64_t *)nullptr);
  return check_launch("so_rasterize_fwd");
}

}  // namespace so

extern "C" int so_rasterize_fwd(int C, int N, int D, int width, int height, int tile_size, const float *means2d,
                                const float *conics, const float *colors, const float *opacities,
                                const float *backgrounds, const uint8_t *tile_masks,
                                const int32_t *isect_offsets, const int32_t *flatten_ids,
                                const int32_t *n_isects_dev, int64_t n_isects_host, float *render_colors,
                                float *render_alphas, int32_t *last_ids, void *stream) {
  SO_REQUIRE(C >= 0 && N >= 0 && width > 0 && height > 0, "so_rasterize_fwd: bad sizes");
  const int wrap_flags = tile_size & ~0xFF;
  tile_size = so::tile_size_of(tile_size);
  SO_REQUIRE(tile_size == 16 || tile_size == 8, "so_rasterize_fwd: tile_size %d not in {8,16}", tile_size);
  SO_REQUIRE(!wrap_flags || width % tile_size == 0, "so_rasterize_fwd: SO_TILE_WRAP_* needs width %% tile_size == 0");
  if (C == 0) return SO_OK;
  SO_REQUIRE(isect_offsets && render_colors && render_alphas && last_ids, "so_rasterize_fwd: null pointer");
  SO_REQUIRE(N == 0 || (means2d && conics && colors && opacities), "so_rasterize_fwd: null pointer");
  SO_REQUIRE(n_isects_dev || n_isects_host == 0 || flatten_ids, "so_rasterize_fwd: null flatten_ids");
  const int tile_w = (width + tile_size - 1) / tile_size, tile_h = (height + tile_size - 1) / tile_size;
  SO_REQUIRE((int64_t)C * tile_w * tile_h < (int64_t)INT32_MAX, "so_rasterize_fwd: C*tiles = %lld does not fit 31 bits",
             (long long)C * tile_w * tile_h);
  const dim3 grid((unsigned)((int64_t)C * tile_w * tile_h));
  hipStream_t st = so::as_stream(stream);
#define SO_CASE(DD)                                                                                              \
  case DD:                                                                                                       \
    return so::launch_fwd<DD>(tile_size, grid, st, C, N, width, height, tile_w, tile_h, means2d, conics, colors, \
                              opacities, backgrounds, tile_masks, isect_offsets, flatten_ids, n_isects_dev,      \
                              n_isects_host, render_colors, render_alphas, last_ids, wrap_flags);
  switch (D) {
    SO_CASE(1) SO_CASE(2) SO_CASE(3) SO_CASE(4) SO_CASE(5) SO_CASE(8) SO_CASE(9) SO_CASE(16) SO_CASE(17) SO_CASE(32) SO_CASE(33)
    default:
      so::set_error("so_rasterize_fwd: unsupported channel count D=%d", D);
      return SO_ERR_UNSUPPORTED;
  }
#undef SO_CASE
}

namespace so {
int rasterize_fwd_packed_launch(int C, int N, int width, int height, int tile_size, const float *rec, const float *backgrounds,
                                const int32_t *isect_offsets, const int32_t *flatten_ids, const int32_t *n_isects_dev,
                                int64_t n_isects_host, float *render_colors, float *render_alphas, int32_t *last_ids,
                                const int32_t *tile_order, void *stream, const uint64_t *sort_keys = nullptr, float *seg_state = nullptr,
                                int seg_len = 0, int seg_count = 1);
}
extern "C" int so_rasterize_fwd_packed(int C, int N, int width, int height, int tile_size, const float *rec,
                                       const float *backgrounds, const int32_t *isect_offsets,
                                       const int32_t *flatten_ids, const int32_t *n_isects_dev,
                                       int64_t n_isects_host, float *render_colors, float *render_alphas,
                                       int32_t *last_ids, void *stream) {
  return so::rasterize_fwd_packed_launch(C, N, width, height, tile_size, rec, backgrounds, isect_offsets, flatten_ids, n_isects_dev,
                                         n_isects_host, render_colors, render_alphas, last_ids, nullptr, stream);
}
// internal (step.hip): the same with a workgroup -> tile table (so_step_desc.tile_order; nullable)
int so::rasterize_fwd_packed_launch(int C, int N, int width, int height, int tile_size, const float *rec, const float *backgrounds,
                                    const int32_t *isect_offsets, const int32_t *flatten_ids, const int32_t *n_isects_dev,
                                    int64_t n_isects_host, float *render_colors, float *render_alphas, int32_t *last_ids,
                                    const int32_t *tile_order, void *stream, const uint64_t *sort_keys, float *seg_state, int seg_len,
                                    int seg_count) {
  SO_REQUIRE(C >= 0 && N >= 0 && width > 0 && height > 0, "so_rasterize_fwd_packed: bad sizes");
  SO_REQUIRE(!seg_state || (seg_len > 0 && seg_len % 256 == 0 && so::tile_size_of(tile_size) == 16 && (((uintptr_t)seg_state) & 15) == 0),
             "so_rasterize_fwd_packed: segment states need 16x16 tiles, seg_len a multiple of 256 and a 16-byte aligned buffer");
  float4 *const seg4 = reinterpret_cast<float4 *>(seg_state);
  const int seg_batches = seg_state ? seg_len / 256 : 0;
  // sort_keys (so_step_desc.sort_in_rasteriser): binned lists only (n_isects_dev == NULL, n_isects_host = -slots), 16x16 tiles
  SO_REQUIRE(!sort_keys || (!n_isects_dev && n_isects_host < 0 && so::tile_size_of(tile_size) == 16 && flatten_ids),
             "so_rasterize_fwd_packed: the prologue sort needs binned lists and 16x16 tiles");
  const int wrap_flags = tile_size & ~0xFF;
  tile_size = so::tile_size_of(tile_size);
  SO_REQUIRE(tile_size == 16 || tile_size == 8, "so_rasterize_fwd_packed: tile_size %d not in {8,16}", tile_size);
  SO_REQUIRE(!wrap_flags || width % tile_size == 0, "so_rasterize_fwd_packed: SO_TILE_WRAP_* needs width %% tile_size == 0");
  if (C == 0) return SO_OK;
  SO_REQUIRE(isect_offsets && render_colors && render_alphas && last_ids && (N == 0 || rec), "so_rasterize_fwd_packed: null pointer");
  SO_REQUIRE((((uintptr_t)rec) & 63) == 0, "so_rasterize_fwd_packed: rec must be 64-byte aligned");
  const int tile_w = (width + tile_size - 1) / tile_size, tile_h = (height + tile_size - 1) / tile_size;
  SO_REQUIRE((int64_t)C * tile_w * tile_h < (int64_t)INT32_MAX, "so_rasterize_fwd_packed: C*tiles = %lld does not fit 31 bits",
             (long long)C * tile_w * tile_h);
  const dim3 grid((unsigned)((int64_t)C * tile_w * tile_h));
  hipStream_t st = so::as_stream(stream);
  if (tile_size == 16 && sort_keys)
    hipLaunchKernelGGL((so::k_rasterize_fwd<3, 16, true, true>), grid, dim3(256), 0, st, C, N, width, height, tile_w, tile_h,
                       nullptr, nullptr, rec, nullptr, backgrounds, nullptr, isect_offsets, flatten_ids, n_isects_dev,
                       n_isects_host, render_colors, render_alphas, last_ids, wrap_flags, tile_order, sort_keys, seg4, seg_batches, seg_count - 1);
  else if (tile_size == 16)
    hipLaunchKernelGGL((so::k_rasterize_fwd<3, 16, true>), grid, dim3(256), 0, st, C, N, width, height, tile_w, tile_h,
                       nullptr, nullptr, rec, nullptr, backgrounds, nullptr, isect_offsets, flatten_ids, n_isects_dev,
                       n_isects_host, render_colors, render_alphas, last_ids, wrap_flags, tile_order, (const uint64_t *)nullptr, seg4, seg_batches, seg_count - 1);
  else
    hipLaunchKernelGGL((so::k_rasterize_fwd<3, 8, true>), grid, dim3(64), 0, st, C, N, width, height, tile_w, tile_h,
                       nullptr, nullptr, rec, nullptr, backgrounds, nullptr, isect_offsets, flatten_ids, n_isects_dev,
                       n_isects_host, render_colors, render_alphas, last_ids, wrap_flags, tile_order, (const uint64_t *)nullptr);
  return so::check_launch("so_rasterize_fwd_packed");
}

#ifdef SO_TILE_PERM_EXPERIMENT
extern "C" int so_debug_tile_perm_fwd(const int32_t *perm) {
  return hipMemcpyToSymbol(HIP_SYMBOL(so::g_tile_perm), &perm, sizeof(perm)) == hipSuccess ? 0 : 1;
}
#endif
