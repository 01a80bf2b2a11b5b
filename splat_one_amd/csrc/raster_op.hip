// raster_op.hip -- `gsplat.rendering.rasterization` as ONE C-ABI call each way.
//
// The reference reaches the whole forward through one Python call
//   /root/reference/utils/gsplat_utils/gsplat_trainer.py:477-494   rasterization(means, quats, scales, opacities, colors,
//                                                                  viewmats, Ks, width, height, sh_degree=..., ...)
// and the whole backward through `loss.backward()` (:655).  so_rasterization_fwd / so_rasterization_bwd are those two
// halves for the COMMON SHAPE of that call -- dense layout, SH coefficients shared by the cameras, fixed poses, RGB --
// on the tensors the call is handed (post-activation scales / opacities, one [N,K,3] coefficient tensor; attr_rec.hpp
// AttrAct), or on the raw parameters (so_raster_desc.activated = 0).  Launch sequence = the fused training step's
// (step.hip) without loss and optimiser:
//   forward   k_preprocess_fwd (projection + SH colour + binned keys)  ->  so_isect_sort_bins  ->  so_rasterize_fwd_packed
//             -> k_bins_status (fullest tile / overflow of THIS call to host-mapped memory: the caller sizes its bins one
//                call late without ever synchronising)
//   backward  so_rasterize_bwd_packed (gradient records)  ->  k_preprocess_bwd  ->  screen-space gradients unpacked for
//             the densification strategy (`info["means2d"].grad` / `.absgrad`, gsplat_trainer.py:616-622, 744-752)
// No allocation, no synchronisation, no host read-back; hipGraph-capturable like every other entry point.
#include "rasterize_common.hpp"

namespace so {
int preprocess_fwd_act(int C, int N, int K, int sh_degree, const float *means, const float *scales, const float *quats,
                       const float *opacities_in, const float *coeffs, const float *viewmats, const float *Ks, int width,
                       int height, float eps2d, float near_plane, float far_plane, float radius_clip, int camera_model,
                       int antialiased, int tile_size, int32_t *tile_counts, float *rec, float *vrec, int tile_cull,
                       uint64_t *bin_keys, int64_t bin_cap, int32_t *bin_overflow, void *stream, int32_t *sub_counts = nullptr,
                       int replicas = 1);
int tile_order_launch(int C, int tile_w, int tile_h, const int32_t *offsets, const int32_t *n_isects_dev, int64_t n_isects_host,
                      int32_t *order, hipStream_t st);
int rasterize_fwd_packed_launch(int C, int N, int width, int height, int tile_size, const float *rec, const float *backgrounds,
                                const int32_t *isect_offsets, const int32_t *flatten_ids, const int32_t *n_isects_dev,
                                int64_t n_isects_host, float *render_colors, float *render_alphas, int32_t *last_ids,
                                const int32_t *tile_order, void *stream, const uint64_t *sort_keys = nullptr, float *seg_state = nullptr,
                                int seg_len = 0, int seg_count = 1);
int bins_gather_launch(int64_t M, int R, int32_t *sub_counts, int32_t *tile_counts, uint64_t *bin_keys, int64_t bin_cap,
                       int32_t *eff_fullest, hipStream_t st);
int preprocess_bwd_act(int C, int N, int K, int sh_degree, const float *means, const float *scales, const float *quats,
                       const float *opacities_in, const float *coeffs, const float *viewmats, const float *Ks, int width,
                       int height, float eps2d, int camera_model, int antialiased, float *v_means, float *v_scales,
                       float *v_quats, float *v_opacities, float *v_sh0, float *v_shN, const float *vrec, const float *rec,
                       void *stream);
int preprocess_fwd_n(int C, int N, int K, int sh_degree, const float *means, const float *log_scales, const float *quats,
                     const float *logit_opacities, const float *sh0, const float *shN, const float *viewmats, const float *Ks,
                     int width, int height, float eps2d, float near_plane, float far_plane, float radius_clip, int camera_model,
                     int antialiased, int tile_size, int32_t *radii, float *means2d, float *depths, float *conics,
                     float *opacities, float *colors, int32_t *tiles_per_gauss, int32_t *tile_counts, float *rec, float *vrec,
                     int32_t *tile_slots, int tile_cull, uint64_t *bin_keys, int64_t bin_cap, int32_t *bin_overflow,
                     const int32_t *n_dev, void *stream, int32_t *sub_counts = nullptr, int replicas = 1);
int preprocess_bwd_n(int C, int N, int K, int sh_degree, const float *means, const float *log_scales, const float *quats,
                     const float *logit_opacities, const float *sh0, const float *shN, const float *viewmats, const float *Ks,
                     int width, int height, float eps2d, int camera_model, int antialiased, const int32_t *radii,
                     const float *opacities, const float *colors, float opacity_reg, float scale_reg, float *v_means,
                     float *v_log_scales, float *v_quats, float *v_logit_opacities, float *v_sh0, float *v_shN, float *grad2d,
                     float *count, const float *vrec, int absgrad_stats, const int32_t *skip_flag, float *skip_out,
                     const int32_t *n_dev, const float *rec, void *stream, int64_t row_begin = 0, int64_t row_end = 0);
int rasterize_bwd_packed_launch(int C, int N, int width, int height, int tile_size, const float *rec, const float *backgrounds,
                                const int32_t *isect_offsets, const int32_t *flatten_ids, const int32_t *n_isects_dev,
                                int64_t n_isects_host, const float *render_alphas, const int32_t *last_ids,
                                const float *v_render_colors, const float *v_render_alphas, float *vrec, int absgrad,
                                const LossFinal &fin, void *stream);
int rec_unpack_means2d(int64_t n, const float *vrec, float *v_means2d, float *v_means2d_abs, void *stream);

// Fullest tile and overflow flag of the binning pass that just ran, published to host-mapped memory
// {max tile count, overflow, seq, total intersections (clamped counts)}: one workgroup, a strided max over the M counts.
__global__ void __launch_bounds__(1024)
k_bins_status(const int32_t *__restrict__ tile_counts, int64_t M, int64_t bin_cap, const int32_t *__restrict__ overflow,
              int32_t *__restrict__ status, int32_t seq, int32_t *__restrict__ eff_fullest) {
  __shared__ int s_max[16];
  __shared__ unsigned long long s_sum[16];
  int mx = 0;
  unsigned long long sum = 0;
  for (int64_t i = threadIdx.x; i < M; i += blockDim.x) {
    const int c = tile_counts[i];
    mx = c > mx ? c : mx;
    sum += (unsigned long long)(c < bin_cap ? c : bin_cap);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_down(mx, off);
    mx = o > mx ? o : mx;
    sum += __shfl_down(sum, off);
  }
  if ((threadIdx.x & 63) == 0) { s_max[threadIdx.x >> 6] = mx; s_sum[threadIdx.x >> 6] = sum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { mx = s_max[w] > mx ? s_max[w] : mx; sum += s_sum[w]; }
    if (eff_fullest) {        // replicated counters: R x the fullest SLICE, left by k_bins_gather when a slice overflowed
      const int e = *eff_fullest;
      mx = e > mx ? e : mx;
      *eff_fullest = 0;
    }
    status[0] = mx;
    status[1] = *overflow;
    status[3] = (int32_t)(sum > 0x7fffffffull ? 0x7fffffffull : sum);
    __threadfence_system();
    status[2] = seq;      // written last: the host trusts the other words once it sees the sequence number of its call
  }
}

__global__ void __launch_bounds__(256) k_zero_words(uint32_t *__restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0u;
}

static int wrap_flags_of(int camera_model, int C, int W, int ts) {
  int wrap = 0;
  if (W % ts == 0) {
    if (!(camera_model & SO_CAM_PER_VIEW)) wrap = camera_model == SO_CAM_SPHERICAL ? SO_TILE_WRAP_ALL : 0;
    else
      for (int c = 0; c < C && c < 16; ++c)
        if (((camera_model >> (2 * c)) & 3) == SO_CAM_SPHERICAL) wrap |= SO_TILE_WRAP_CAM(c);
  }
  return wrap;
}

static int check_desc(const so_raster_desc *d, const char *what) {
  SO_REQUIRE(d != nullptr, "%s: null descriptor", what);
  SO_REQUIRE(d->abi_size == (int32_t)sizeof(so_raster_desc), "%s: descriptor size %d != %d (ABI mismatch)", what, d->abi_size,
             (int)sizeof(so_raster_desc));
  SO_REQUIRE(d->C > 0 && d->N >= 0 && d->K >= 1 && d->width > 0 && d->height > 0 && (d->tile_size == 16 || d->tile_size == 8),
             "%s: bad sizes", what);
  SO_REQUIRE(d->bin_capacity > 0, "%s: bin_capacity must be positive", what);
  const int tile_w = (d->width + d->tile_size - 1) / d->tile_size, tile_h = (d->height + d->tile_size - 1) / d->tile_size;
  SO_REQUIRE((int64_t)d->C * tile_w * tile_h * d->bin_capacity < ((int64_t)1 << 31), "%s: C*tiles*bin_capacity does not fit 31 bits", what);
  SO_REQUIRE(d->counters && d->flatten_ids && d->rec, "%s: null workspace pointer", what);
  SO_REQUIRE(d->bin_replicas <= 1 || (d->bin_sub_counts && d->bin_replicas <= 64 && d->bin_capacity % d->bin_replicas == 0),
             "%s: bin_replicas needs bin_sub_counts and bin_capacity %% bin_replicas == 0", what);
  return SO_OK;
}
}  // namespace so

extern "C" int so_rasterization_fwd(const so_raster_desc *d, void *stream) {
  int rc = so::check_desc(d, "so_rasterization_fwd");
  if (rc != SO_OK) return rc;
  const int C = d->C, N = d->N, W = d->width, H = d->height, ts = d->tile_size;
  const int tile_w = (W + ts - 1) / ts, tile_h = (H + ts - 1) / ts;
  const int64_t M = (int64_t)C * tile_w * tile_h;
  SO_REQUIRE(d->key_buf && d->render_colors && d->render_alphas && d->last_ids, "so_rasterization_fwd: null output / scratch pointer");
  hipStream_t st = so::as_stream(stream);
  // counters: tile_counts[M] | long-list scratch of the sort [M + 1] | n_isects (unused) | overflow
  int32_t *tile_counts = d->counters, *cursor = d->counters + M, *overflow = d->counters + 2 * M + 2;
  const int reps = d->bin_replicas > 1 ? d->bin_replicas : 1;
  {
    int64_t g = (2 * M + 3 + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(so::k_zero_words, dim3((unsigned)g), dim3(256), 0, st, reinterpret_cast<uint32_t *>(d->counters), 2 * M + 3);
  }
  if (N > 0) {
    if (d->activated)
      rc = so::preprocess_fwd_act(C, N, d->K, d->sh_degree, d->means, d->scales, d->quats, d->opacities, d->sh0, d->viewmats, d->Ks,
                                  W, H, d->eps2d, d->near_plane, d->far_plane, d->radius_clip, d->camera_model, d->antialiased, ts,
                                  tile_counts, d->rec, d->vrec, d->tile_cull, d->key_buf, d->bin_capacity, overflow, stream,
                                  d->bin_sub_counts, reps);
    else
      rc = so::preprocess_fwd_n(C, N, d->K, d->sh_degree, d->means, d->scales, d->quats, d->opacities, d->sh0, d->shN, d->viewmats,
                                d->Ks, W, H, d->eps2d, d->near_plane, d->far_plane, d->radius_clip, d->camera_model, d->antialiased,
                                ts, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, tile_counts, d->rec, d->vrec,
                                nullptr, d->tile_cull, d->key_buf, d->bin_capacity, overflow, nullptr, stream, d->bin_sub_counts, reps);
    if (rc != SO_OK) return rc;
    if (reps > 1) {      // (replicated bin counters, see so_step_desc.bin_replicas: close the slices of every bin up before the sort)
      rc = so::bins_gather_launch(M, reps, d->bin_sub_counts, tile_counts, d->key_buf, d->bin_capacity, d->bin_sub_counts + (int64_t)reps * M, st);
      if (rc != SO_OK) return rc;
    }
    rc = so_isect_sort_bins(C, tile_w, tile_h, tile_counts, d->bin_capacity, d->key_buf, d->flatten_ids, cursor, stream);
    if (rc != SO_OK) return rc;
  }
  const int wrap = so::wrap_flags_of(d->camera_model, C, W, ts);
  if (d->tile_order && N > 0) {      // longest list first (so_raster_desc.tile_order; the caller asks for it where lists are long)
    rc = so::tile_order_launch(C, tile_w, tile_h, tile_counts, nullptr, -d->bin_capacity, d->tile_order, st);
    if (rc != SO_OK) return rc;
  }
  rc = so::rasterize_fwd_packed_launch(C, N, W, H, ts | wrap, d->rec, d->backgrounds, tile_counts, d->flatten_ids, nullptr,
                                       -d->bin_capacity, d->render_colors, d->render_alphas, d->last_ids,
                                       N > 0 ? d->tile_order : nullptr, stream);
  if (rc != SO_OK) return rc;
  if (d->status_out) {
    hipLaunchKernelGGL(so::k_bins_status, dim3(1), dim3(1024), 0, st, tile_counts, M, d->bin_capacity, overflow, d->status_out, d->seq,
                       reps > 1 ? d->bin_sub_counts + (int64_t)reps * M : (int32_t *)nullptr);
    rc = so::check_launch("so_rasterization_fwd (status)");
  }
  return rc;
}

extern "C" int so_rasterization_bwd(const so_raster_desc *d, void *stream) {
  int rc = so::check_desc(d, "so_rasterization_bwd");
  if (rc != SO_OK) return rc;
  const int C = d->C, N = d->N, W = d->width, H = d->height, ts = d->tile_size;
  if (N == 0) return SO_OK;
  SO_REQUIRE(d->vrec && d->render_alphas && d->last_ids && d->v_render_colors && d->v_render_alphas,
             "so_rasterization_bwd: null pointer (vrec / saved forward outputs / incoming gradients)");
  SO_REQUIRE(d->v_means && d->v_quats && d->v_scales && d->v_opacities && d->v_sh0 && (d->v_shN || d->K == 1),
             "so_rasterization_bwd: null gradient output");
  const int wrap = so::wrap_flags_of(d->camera_model, C, W, ts);
  // counters: tile_counts[M] | sort scratch [M + 1] | n_isects (unused) | overflow of the forward's binning pass
  const int64_t M = (int64_t)C * ((W + ts - 1) / ts) * ((H + ts - 1) / ts);
  so::LossFinal fin{};
  fin.skip = d->counters + 2 * M + 2;
  fin.tile_waves = d->raster_impl;           // (-1: the process default; rasterize_bwd.hip)
  fin.tile_order = d->tile_order;            // (the table the forward of this call built)
  rc = so::rasterize_bwd_packed_launch(C, N, W, H, ts | wrap, d->rec, d->backgrounds, d->counters, d->flatten_ids, nullptr,
                                       -d->bin_capacity, d->render_alphas, d->last_ids, d->v_render_colors, d->v_render_alphas,
                                       d->vrec, d->absgrad, fin, stream);
  if (rc != SO_OK) return rc;
  if (d->activated)
    rc = so::preprocess_bwd_act(C, N, d->K, d->sh_degree, d->means, d->scales, d->quats, d->opacities, d->sh0, d->viewmats, d->Ks, W, H,
                                d->eps2d, d->camera_model, d->antialiased, d->v_means, d->v_scales, d->v_quats, d->v_opacities,
                                d->v_sh0, d->v_shN, d->vrec, d->rec, stream);
  else
    rc = so::preprocess_bwd_n(C, N, d->K, d->sh_degree, d->means, d->scales, d->quats, d->opacities, d->sh0, d->shN, d->viewmats, d->Ks,
                              W, H, d->eps2d, d->camera_model, d->antialiased, nullptr, nullptr, nullptr, 0.f, 0.f, d->v_means,
                              d->v_scales, d->v_quats, d->v_opacities, d->v_sh0, d->v_shN, nullptr, nullptr, d->vrec, 0, nullptr,
                              nullptr, nullptr, d->rec, stream);
  if (rc != SO_OK) return rc;
  if (d->v_means2d) rc = so::rec_unpack_means2d((int64_t)C * N, d->vrec, d->v_means2d, d->absgrad ? d->v_means2d_abs : nullptr, stream);
  return rc;
}
