// rasterize_gp.hip -- EXPERIMENT: a Gaussian-parallel backward of the tile rasteriser for LONG per-tile lists
// (numerics: gp_backward_model.py).  Not part of the product: built into a variant library by
// tools/build_gp_variant.sh and compared with so_rasterize_bwd_packed on the same inputs by tools/experiments/dbg_gp.py.
//
// One wave per 16x16 tile.  A batch = 64 consecutive entries of the tile's depth-sorted list, one Gaussian per lane, its
// record in registers.  The tile's 256 pixels are walked in a skewed pipeline: at step t lane i handles pixel t - i, takes
// the pixel's front-to-back state (T before this Gaussian, running sum of alpha T (c . v_c)) from lane i-1 with one DPP
// wave shift per value, and hands the updated state on; lane 0 reads the state the previous batch left in LDS, lane 63
// writes it back.  The colour BEHIND a Gaussian, which the product kernel accumulates back to front, is the difference
// total - running (total = (C_final - T_final bg) . v_c, known per pixel from the forward's output).  Gradients stay in the
// lane's registers for the whole batch: no cross-lane reduction, and ONE atomic record per (tile, Gaussian).
// RGB, no backgrounds, no masks, no periodic images, no absgrad: what the dense-regime benchmark needs.
#include "../../splat_one_amd/csrc/rasterize_common.hpp"

namespace so {

__global__ void __launch_bounds__(64)
k_rasterize_bwd_gp(int C, int N, int W, int H, int tile_w, int tile_h, const float *__restrict__ rec,
                   const int32_t *__restrict__ offsets, const int32_t *__restrict__ flatten_ids,
                   const int32_t *__restrict__ n_isects_dev, int64_t n_isects_host, const float *__restrict__ render_colors,
                   const float *__restrict__ render_alphas, const int32_t *__restrict__ last_ids,
                   const float *__restrict__ v_render_colors, const float *__restrict__ v_render_alphas,
                   float *__restrict__ vrec, int min_len) {
  // per pixel, in the order the pipeline walks them; pixels whose last contributor lies before the current batch are
  // compacted away at the start of every batch, so a batch costs (live pixels + 63) steps
  __shared__ float4 s_px[256];     // v_c[0..2], total
  __shared__ float4 s_pc[256];     // T_final * v_alpha_out, last contributor (int bits), pixel centre x, y
  __shared__ float2 s_state[256];  // T, running sum -- entry state of the current batch
  const int n_tiles = tile_w * tile_h;
  const int M = C * n_tiles;
  const int ct = (int)xcd_remap(blockIdx.x, M);
  const int c = ct / n_tiles;
  const int t = ct - c * n_tiles;
  const int ty = t / tile_w, tx = t - ty * tile_w;
  const int lane = threadIdx.x;
  int64_t lo, hi;
  tile_list_range(ct, M, offsets, n_isects_dev, n_isects_host, lo, hi);
  if (hi - lo < (int64_t)min_len || hi <= lo) return;   // short lists stay with the quadrant kernel

  int32_t block_last = (int32_t)lo - 1;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = lane + 64 * k;
    const int j = tx * 16 + (p & 15), i = ty * 16 + (p >> 4);
    const bool inside = (i < H) && (j < W);
    const int64_t pix = ((int64_t)c * H + i) * W + j;
    const float T_final = inside ? 1.f - render_alphas[pix] : 1.f;
    float vc[3] = {0.f, 0.f, 0.f}, total = 0.f;
    if (inside) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        vc[ch] = v_render_colors[pix * 3 + ch];
        total = fmaf(render_colors[pix * 3 + ch], vc[ch], total);
      }
    }
    const float v_a = inside ? v_render_alphas[pix] : 0.f;
    int32_t bin_final = (int32_t)lo - 1;
    if (inside && T_final < 1.f) bin_final = last_ids[pix];
    block_last = max(block_last, bin_final);
    s_px[p] = make_float4(vc[0], vc[1], vc[2], total);
    s_pc[p] = make_float4(T_final * v_a, __int_as_float(bin_final), (float)j + 0.5f, (float)i + 0.5f);
    s_state[p] = make_float2(1.f, 0.f);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) block_last = max(block_last, __shfl_xor(block_last, d, 64));
  __syncthreads();
  if (block_last < lo) return;
  int n_live = 256;

#pragma unroll 1
  for (int64_t base = lo; base <= block_last; base += 64) {
    {   // drop the pixels that are finished before this batch (stable compaction of the three arrays, one wave)
      float4 a4[4], b4[4];
      float2 c2[4];
      int pos[4];
      int run = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int q = lane + 64 * k;
        a4[k] = s_px[q]; b4[k] = s_pc[q]; c2[k] = s_state[q];
        const bool keep = q < n_live && (int64_t)__float_as_int(b4[k].y) >= base;
        const unsigned long long m = __ballot(keep);
        pos[k] = keep ? run + __popcll(m & ((1ull << lane) - 1ull)) : -1;
        run += __popcll(m);
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (pos[k] >= 0) { s_px[pos[k]] = a4[k]; s_pc[pos[k]] = b4[k]; s_state[pos[k]] = c2[k]; }
      n_live = run;
      __syncthreads();
      if (n_live == 0) break;
    }
    const int64_t idx = base + lane;
    const bool has = idx <= block_last;
    int32_t g = 0;
    float4 q0 = make_float4(0.f, 0.f, 1.f, 0.f), q1 = make_float4(1.f, 0.f, 0.f, 0.f);
    float blue = 0.f;
    if (has) {
      g = flatten_ids[idx];
      const float4 *r4 = reinterpret_cast<const float4 *>(rec) + 4 * (int64_t)g;
      q0 = r4[0];
      q1 = r4[1];   // x,y,ca,cb | cc,opac,r,g
      blue = reinterpret_cast<const float *>(r4 + 2)[0];
    }
    const float opac = has ? q1.y : 0.f;
    const int32_t my_idx = (int32_t)idx;
    float a_x = 0.f, a_y = 0.f, a_ca = 0.f, a_cb = 0.f, a_cc = 0.f, a_r = 0.f, a_g = 0.f, a_b = 0.f, a_op = 0.f;
    float T_pass = 1.f, run_pass = 0.f;
#pragma unroll 1
    for (int step = 0; step < n_live + 63; ++step) {
      float T_in = dpp_mov<0x138, 0xf, 0xf, false>(T_pass, T_pass);        // wave_shr:1 -- lane i takes lane i-1's value
      float run_in = dpp_mov<0x138, 0xf, 0xf, false>(run_pass, run_pass);
      const int p = step - lane;
      const bool in_range = p >= 0 && p < n_live;
      const int pc = in_range ? p : 0;
      if (lane == 0) {
        const float2 st = s_state[pc];
        T_in = st.x;
        run_in = st.y;
      }
      const float4 pv = s_px[pc];
      const float4 pq = s_pc[pc];
      const float dx = q0.x - pq.z, dy = q0.y - pq.w;
      const float sigma = 0.5f * (q0.z * dx * dx + q1.x * dy * dy) + q0.w * dx * dy;
      const float vis = __expf(-sigma);
      const float ov = opac * vis;
      const float alpha = fminf(kAlphaMax, ov);
      const bool valid = in_range && has && (my_idx <= __float_as_int(pq.y)) && !(sigma < 0.f || alpha < kAlphaMin);
      const float alpha_v = valid ? alpha : 0.f;
      const float fac = alpha_v * T_in;
      const float cv = fmaf(blue, pv.z, fmaf(q1.w, pv.y, q1.z * pv.x));
      const float run_out = fmaf(fac, cv, run_in);
      const float ra = __builtin_amdgcn_rcpf(1.f - alpha_v);
      // v_alpha = (c . v_c) T - (colour behind . v_c) / (1 - alpha) + T_final v_alpha_out / (1 - alpha)
      const float v_alpha = fmaf(T_in, cv, ra * (pq.x - (pv.w - run_out)));
      const bool grad_on = valid && (ov <= kAlphaMax);
      const float v_sigma = grad_on ? -ov * v_alpha : 0.f;
      a_op += grad_on ? vis * v_alpha : 0.f;
      const float t1 = v_sigma * dx, t2 = v_sigma * dy;
      a_ca = fmaf(0.5f * t1, dx, a_ca);
      a_cb = fmaf(t1, dy, a_cb);
      a_cc = fmaf(0.5f * t2, dy, a_cc);
      a_x += fmaf(q0.z, t1, q0.w * t2);
      a_y += fmaf(q0.w, t1, q1.x * t2);
      a_r = fmaf(fac, pv.x, a_r);
      a_g = fmaf(fac, pv.y, a_g);
      a_b = fmaf(fac, pv.z, a_b);
      T_pass = T_in * (1.f - alpha_v);
      run_pass = run_out;
      if (lane == 63 && in_range) s_state[p] = make_float2(T_pass, run_pass);
    }
    if (has) {
      float *o = vrec + 16 * (int64_t)g;
      if (a_x != 0.f) atomicAdd(o + 0, a_x);
      if (a_y != 0.f) atomicAdd(o + 1, a_y);
      if (a_ca != 0.f) atomicAdd(o + 2, a_ca);
      if (a_cb != 0.f) atomicAdd(o + 3, a_cb);
      if (a_cc != 0.f) atomicAdd(o + 4, a_cc);
      if (a_r != 0.f) atomicAdd(o + 5, a_r);
      if (a_g != 0.f) atomicAdd(o + 6, a_g);
      if (a_b != 0.f) atomicAdd(o + 7, a_b);
      if (a_op != 0.f) atomicAdd(o + 8, a_op);
    }
  }
}

}  // namespace so

/* Same inputs as so_rasterize_bwd_packed plus the forward's colours; tiles whose list is shorter than min_len are left
 * alone (the quadrant kernel's share in a mixed run). */
extern "C" int so_exp_rasterize_bwd_gp(int C, int N, int width, int height, int tile_size, const float *rec,
                                       const int32_t *isect_offsets, const int32_t *flatten_ids, const int32_t *n_isects_dev,
                                       int64_t n_isects_host, const float *render_colors, const float *render_alphas,
                                       const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
                                       float *vrec, int min_len, void *stream) {
  SO_REQUIRE(C > 0 && N > 0 && width > 0 && height > 0 && tile_size == 16, "so_exp_rasterize_bwd_gp: 16x16 tiles only");
  SO_REQUIRE(rec && isect_offsets && flatten_ids && render_colors && render_alphas && last_ids && v_render_colors &&
                 v_render_alphas && vrec, "so_exp_rasterize_bwd_gp: null pointer");
  const int tile_w = (width + 15) / 16, tile_h = (height + 15) / 16;
  const dim3 grid((unsigned)((int64_t)C * tile_w * tile_h));
  hipLaunchKernelGGL(so::k_rasterize_bwd_gp, grid, dim3(64), 0, so::as_stream(stream), C, N, width, height, tile_w, tile_h, rec,
                     isect_offsets, flatten_ids, n_isects_dev, n_isects_host, render_colors, render_alphas, last_ids,
                     v_render_colors, v_render_alphas, vrec, min_len);
  return so::check_launch("so_exp_rasterize_bwd_gp");
}
