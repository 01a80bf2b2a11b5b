// step.hip -- the whole training iteration as two C-ABI calls (hipGraph-capturable).
//
// so_train_step_fwd_bwd : one iteration of the hot loop of
//   /root/reference/utils/gsplat_utils/gsplat_trainer.py:586-655  (render -> loss -> backward)
//   as a fixed sequence of launches on one stream, from RAW parameters to gradients of the raw
//   parameters; every buffer is caller-owned and static, nothing is read back to the host.
// so_train_step_optimize : :726-742 (all Adam steps + ExponentialLR on the means) with the step
//   counter, learning-rate schedule and bias corrections evaluated on the device.
// Between the two the caller may all-reduce the gradients (view-sharded data parallelism).
#include "rasterize_common.hpp"   // (LossFinal; includes so_common.hpp)
#include "attr_rec.hpp"           // (attr_rec_stride_bytes)

#include <vector>

namespace so {
// Optional per-stage timing with HIP events on the launch stream (bench.py's roofline leg).
static const char *kStageNames[] = {"so_preprocess_fwd", "so_isect_scan", "so_isect_fill", "so_rasterize_fwd",
                                    "so_ssim_l1_fwd", "so_ssim_l1_bwd", "so_rasterize_bwd", "so_preprocess_bwd",
                                    "so_adam_step_dev", "so_ssim_l1_fused", "so_tile_order", "so_bins_gather"};
constexpr int kNumStages = 12;
// per calling thread: the trainer thread's timers neither see nor are switched by a viewer thread's renders
static thread_local bool g_prof_on = false;
static thread_local std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_events[kNumStages];

struct StageTimer {
  int stage;
  hipStream_t st;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  StageTimer(int s, hipStream_t stream) : stage(s), st(stream) {
    if (g_prof_on) {
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0, st);
    }
  }
  ~StageTimer() {
    if (e0) {
      (void)hipEventRecord(e1, st);
      g_prof_events[stage].emplace_back(e0, e1);
    }
  }
};
}  // namespace so

namespace so {
// plain zero-fill kernel (keeps driver memset nodes out of captured graphs)
__global__ void __launch_bounds__(256) k_zero_u32(uint32_t *__restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0u;
}
int ssim_l1_fwd_launch(int B, int H, int W, int CH, const float *img1, const float *img2, const float *const *img2_slot,
                       int padding_valid, float *sums, float *dmaps, void *stream);
int ssim_l1_bwd_launch(int B, int H, int W, int CH, const float *img1, const float *img2, const float *const *img2_slot,
                       const float *dmaps, float w_l1, float w_ssim, const float *v_loss, float *v_img1, const float *sums,
                       float *loss_out, int padding_valid, float loss_const, void *stream);
int rasterize_bwd_packed_launch(int C, int N, int width, int height, int tile_size, const float *rec, const float *backgrounds,
                                const int32_t *isect_offsets, const int32_t *flatten_ids, const int32_t *n_isects_dev,
                                int64_t n_isects_host, const float *render_alphas, const int32_t *last_ids,
                                const float *v_render_colors, const float *v_render_alphas, float *vrec, int absgrad,
                                const LossFinal &fin, void *stream);
int rasterize_fwd_packed_launch(int C, int N, int width, int height, int tile_size, const float *rec, const float *backgrounds,
                                const int32_t *isect_offsets, const int32_t *flatten_ids, const int32_t *n_isects_dev,
                                int64_t n_isects_host, float *render_colors, float *render_alphas, int32_t *last_ids,
                                const int32_t *tile_order, void *stream, const uint64_t *sort_keys = nullptr, float *seg_state = nullptr,
                                int seg_len = 0, int seg_count = 1);
int tile_order_launch(int C, int tile_w, int tile_h, const int32_t *offsets, const int32_t *n_isects_dev, int64_t n_isects_host,
                      int32_t *order, hipStream_t st);
int ssim_l1_fused_launch(int B, int H, int W, int CH, const float *img1, const float *img2, const float *const *img2_slot,
                         int padding_valid, float w_l1, float w_ssim, const float *v_loss, float *sums, float *v_img1,
                         float *loss_out, int32_t *ticket, float loss_const, int rows, void *stream);

int preprocess_bwd_fused_adam(int C, int N, int K, int sh_degree, const float *means, const float *log_scales, const float *quats,
                              const float *logit_opacities, const float *sh0, const float *shN, const float *viewmats,
                              const float *Ks, int width, int height, float eps2d, int camera_model, int antialiased,
                              const int32_t *radii, const float *opacities, const float *colors, float opacity_reg,
                              float scale_reg, float *grad2d, float *count, const float *vrec, int absgrad_stats,
                              const int32_t *skip_flag, float *skip_out, const AdamFuse &fuse, void *stream,
                              const int32_t *n_dev, const float *rec);
int preprocess_bwd_fused_adam_f16(int C, int N, int K, int sh_degree, const float *means, const float *logit_opacities,
                                  const void *arec, const float *viewmats, const float *Ks, int width, int height, float eps2d,
                                  int camera_model, int antialiased, const int32_t *radii, const float *opacities,
                                  const float *colors, float opacity_reg, float scale_reg, float *grad2d, float *count,
                                  const float *vrec, int absgrad_stats, const int32_t *skip_flag, float *skip_out,
                                  const AdamFuse &fuse, void *stream, const int32_t *n_dev, const float *rec);
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
int preprocess_fwd_n_f16(int C, int N, int K, int sh_degree, const float *means, const float *logit_opacities, const void *arec,
                         const float *viewmats, const float *Ks, int width, int height, float eps2d, float near_plane,
                         float far_plane, float radius_clip, int camera_model, int antialiased, int tile_size, int32_t *radii,
                         float *means2d, float *depths, float *conics, float *opacities, float *colors, int32_t *tiles_per_gauss,
                         int32_t *tile_counts, float *rec, float *vrec, int32_t *tile_slots, int tile_cull, uint64_t *bin_keys,
                         int64_t bin_cap, int32_t *bin_overflow, const int32_t *n_dev, void *stream, int32_t *sub_counts = nullptr,
                         int replicas = 1);
int preprocess_bwd_n_f16(int C, int N, int K, int sh_degree, const float *means, const float *logit_opacities, const void *arec,
                         const float *viewmats, const float *Ks, int width, int height, float eps2d, int camera_model,
                         int antialiased, const int32_t *radii, const float *opacities, const float *colors, float opacity_reg,
                         float scale_reg, float *v_means, float *v_log_scales, float *v_quats, float *v_logit_opacities,
                         float *v_sh0, float *v_shN, float *grad2d, float *count, const float *vrec, int absgrad_stats,
                         const int32_t *skip_flag, float *skip_out, const int32_t *n_dev, void *stream, int64_t row_begin = 0,
                         int64_t row_end = 0);

// Per-iteration inputs in one launch (see so_step_inputs in the header).  Workgroup 0 does the small serial
// pieces (one lane per camera / per Ks entry / per Adam group); every workgroup zeroes its share of the counters.
__global__ void __launch_bounds__(256)
k_step_inputs(int C, const float *__restrict__ c2w, const float *__restrict__ Ks_src, float *__restrict__ w2c,
              float *__restrict__ Ks_dst, const float *pixels, const float **pixels_slot, uint32_t *__restrict__ counters,
              int64_t n_zero, AdamSched sch, int n_groups, double beta1, double beta2, int32_t *__restrict__ step_ptr,
              int32_t *status_out, int64_t status_at, int32_t seq, int64_t n_lists, int32_t *__restrict__ lists_stat,
              const int32_t *__restrict__ order_src, int32_t *__restrict__ order_dst, int64_t n_order) {
  // order_src (nullable): a workgroup -> tile table the caller kept for THIS view (so_step_desc.tile_order_ready) -- copied into the
  // step's table here, in the launch that runs anyway
  if (order_src)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_order; i += (int64_t)gridDim.x * blockDim.x) order_dst[i] = order_src[i];
  // status_out (host-mapped, nullable): what the PREVIOUS iteration left in counters[status_at], [status_at+1]
  // (n_isects, overflow), read by one thread before that pair is zeroed, so no other workgroup races it
  // lists_stat (device, int32[4], nullable; with status_out): the first n_lists counters are the per-tile list lengths of the
  // previous iteration (binned lists) -- their maximum and sum are gathered while they are zeroed, into the pair
  // lists_stat[2 (seq & 1)], and the pair the previous launch completed is published as status_out[3], [4]: the host follows
  // the growth of the lists (bins, choice of the backward rasteriser) without ever reading the device
  uint32_t l_max = 0u, l_sum = 0u;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_zero; i += (int64_t)gridDim.x * blockDim.x) {
    if (lists_stat && status_out && i < n_lists) {
      const uint32_t v = counters[i];
      l_max = v > l_max ? v : l_max;
      l_sum += v;
    }
    if (!status_out || (i != status_at && i != status_at + 1)) counters[i] = 0u;
  }
  if (lists_stat && status_out && n_lists > 0) {
    const int32_t w_max = wave_max_i32((int32_t)l_max);
    uint32_t w_sum = l_sum;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) w_sum += (uint32_t)__shfl_xor((int)w_sum, d, 64);
    if ((threadIdx.x & 63) == 0 && w_sum) {
      atomicMax(lists_stat + 2 * (seq & 1), w_max);
      atomicAdd(lists_stat + 2 * (seq & 1) + 1, (int32_t)w_sum);
    }
  }
  // three independent jobs on three workgroups (when the grid has them), so that the kernel lasts as long as the longest of
  // them and not as long as their sum: the status report waits for its stores to reach host memory, the schedule runs
  // three double-precision pow() per group
  const unsigned b_status = gridDim.x > 1 ? 1u : 0u, b_sched = gridDim.x > 2 ? 2u : 0u;
#ifndef SO_SI_NO_STATUS      // (SO_SI_NO_*: ablation builds of tools/gpu_r05_l.sh -- what of this kernel's 4.7 us is which job)
  if (blockIdx.x == b_status && status_out && threadIdx.x == 0) {
    const int32_t n_prev = (int32_t)counters[status_at], ov_prev = (int32_t)counters[status_at + 1];
    if (status_at < n_zero) counters[status_at] = 0u;
    if (status_at + 1 < n_zero) counters[status_at + 1] = 0u;
    status_out[0] = n_prev;
    status_out[1] = ov_prev;
    if (lists_stat) {   // the pair of the launch before this one: complete, and free again for the launch after this one
      int32_t *done = lists_stat + 2 * ((seq + 1) & 1);
      status_out[3] = done[0];
      status_out[4] = done[1];
      done[0] = 0;
      done[1] = 0;
    }
    __threadfence_system();
    status_out[2] = seq;      // written last: the host trusts [0], [1] once it sees its own sequence number here
  }
#endif
#ifndef SO_SI_NO_CAM
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) camera_inverse_one(c2w + 16 * c, w2c + 16 * c);
    if (Ks_dst)
      for (int i = threadIdx.x; i < 9 * C; i += blockDim.x) Ks_dst[i] = Ks_src[i];
    if (pixels_slot && threadIdx.x == 0) *pixels_slot = pixels;
  }
#endif
#ifndef SO_SI_NO_SCHED
  if (blockIdx.x == b_sched && n_groups > 0)   // (block-uniform: adam_schedule_block has a barrier)
    adam_schedule_block(sch.lr0, sch.lr_gamma, n_groups, beta1, beta2, step_ptr, reinterpret_cast<float2 *>(step_ptr + 2));
#endif
}

static inline void zero_async(void *p, int64_t n_words, hipStream_t st) {
  int64_t g = (n_words + 255) / 256;
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(k_zero_u32, dim3((unsigned)g), dim3(256), 0, st, reinterpret_cast<uint32_t *>(p), n_words);
}
}  // namespace so

extern "C" int so_profile_enable(int enabled) {
  so::g_prof_on = enabled != 0;
  return SO_OK;
}
extern "C" int so_profile_num_stages(void) { return so::kNumStages; }
extern "C" const char *so_profile_stage_name(int i) { return (i >= 0 && i < so::kNumStages) ? so::kStageNames[i] : ""; }
// Synchronises the device; writes the summed milliseconds and call counts per stage; clears the log.
extern "C" int so_profile_read(float *host_ms_sum, int *host_calls) {
  SO_REQUIRE(host_ms_sum && host_calls, "so_profile_read: null pointer");
  if (hipDeviceSynchronize() != hipSuccess) return SO_ERR_LAUNCH;
  for (int s = 0; s < so::kNumStages; ++s) {
    float sum = 0.f;
    for (auto &pr : so::g_prof_events[s]) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, pr.first, pr.second);
      sum += ms;
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
    host_ms_sum[s] = sum;
    host_calls[s] = (int)so::g_prof_events[s].size();
    so::g_prof_events[s].clear();
  }
  return SO_OK;
}
// used by adam.hip's device-scheduled step
extern "C" void so_profile_stage_begin_end(int stage, int begin, void *stream) {
  static thread_local so::StageTimer *cur = nullptr;
  if (begin) { cur = new so::StageTimer(stage, so::as_stream(stream)); }
  else if (cur) { delete cur; cur = nullptr; }
}

// what of the iteration one call runs: all of it; the forward only (eval / viewer); everything but the last stage (HEAD);
// the last stage -- the backward of projection / SH / activations -- on a row range of the Gaussians (TAIL)
enum StepPart { STEP_ALL = 0, STEP_FORWARD = 1, STEP_HEAD = 2, STEP_TAIL = 3 };
static int step_impl(const so_step_desc *d, void *stream, StepPart part, int64_t row_begin = 0, int64_t row_end = 0);

extern "C" int so_train_step_fwd_bwd(const so_step_desc *d, void *stream) { return step_impl(d, stream, STEP_ALL); }

// Data-parallel replicas (gsplat_trainer.py:266-278 scales the hyper-parameters for them): the same
// iteration as so_train_step_fwd_bwd cut in two so that the gradient exchange can start before the backward has finished.
//   so_train_step_head      forward, loss, rasteriser backward (the per-view gradient records are complete)
//   so_train_step_bwd_rows  the per-Gaussian backward for rows [row_begin, row_end) of every gradient tensor (row_begin a
//                           multiple of 64; one lane per Gaussian, no dependence between rows): the caller launches it chunk
//                           by chunk and starts the reduce-scatter of chunk i while chunk i + 1 runs
// head + rows [0, N) == so_train_step_fwd_bwd bit for bit (tests/test_gpu_engine.py).  Not with fuse_adam.
extern "C" int so_train_step_head(const so_step_desc *d, void *stream) { return step_impl(d, stream, STEP_HEAD); }
extern "C" int so_train_step_bwd_rows(const so_step_desc *d, int64_t row_begin, int64_t row_end, void *stream) {
  SO_REQUIRE(row_begin >= 0 && row_begin % 64 == 0 && row_end > row_begin, "so_train_step_bwd_rows: bad row range [%lld, %lld)",
             (long long)row_begin, (long long)row_end);
  return step_impl(d, stream, STEP_TAIL, row_begin, row_end);
}

extern "C" int so_step_inputs(int C, const float *camtoworlds, const float *Ks_src, float *viewmats, float *Ks_dst,
                              const float *pixels, const float **pixels_slot, int32_t *counters, int64_t n_zero,
                              int n_groups, const float *lr0, const float *lr_gamma, double beta1, double beta2,
                              int32_t *step_counter, int32_t *status_out, int64_t status_at, int32_t seq, int64_t n_lists,
                              int32_t *lists_stat, const int32_t *order_src, int32_t *order_dst, int64_t n_order, void *stream) {
  SO_REQUIRE(order_src == nullptr || (order_dst && n_order > 0), "so_step_inputs: order_src needs order_dst and n_order");
  SO_REQUIRE(C >= 0 && n_zero >= 0 && n_groups >= 0 && n_groups <= SO_ADAM_MAX_GROUPS, "so_step_inputs: bad sizes");
  SO_REQUIRE(C == 0 || (camtoworlds && viewmats), "so_step_inputs: null camera pointers");
  SO_REQUIRE((Ks_src == nullptr) == (Ks_dst == nullptr), "so_step_inputs: Ks_src and Ks_dst go together");
  SO_REQUIRE(pixels_slot == nullptr || pixels != nullptr, "so_step_inputs: pixels_slot without pixels");
  SO_REQUIRE(n_zero == 0 || counters, "so_step_inputs: null counters");
  SO_REQUIRE(n_groups == 0 || (lr0 && lr_gamma && step_counter), "so_step_inputs: null schedule pointers");
  SO_REQUIRE(status_out == nullptr || (counters && status_at >= 0), "so_step_inputs: status_out needs counters and status_at");
  SO_REQUIRE(lists_stat == nullptr || (status_out && n_lists >= 0 && n_lists <= status_at), "so_step_inputs: lists_stat needs status_out and n_lists <= status_at");
  so::AdamSched S{};
  for (int i = 0; i < n_groups; ++i) { S.lr0[i] = lr0[i]; S.lr_gamma[i] = lr_gamma[i]; }
  int64_t g = (n_zero + 1023) / 1024;
  if (g < 1) g = 1;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(so::k_step_inputs, dim3((unsigned)g), dim3(256), 0, so::as_stream(stream), C, camtoworlds, Ks_src, viewmats,
                     Ks_dst, pixels, pixels_slot, reinterpret_cast<uint32_t *>(counters), n_zero, S, n_groups, beta1, beta2,
                     step_counter, status_out, status_at, seq, n_lists, lists_stat, order_src, order_dst, n_order);
  return so::check_launch("so_step_inputs");
}

// Forward only (render_colors / render_alphas / last_ids of the current views): the eval / viewer
// path of gsplat_trainer.py:779-940 on the same static buffers, also hipGraph-capturable.
extern "C" int so_render_forward(const so_step_desc *d, void *stream) { return step_impl(d, stream, STEP_FORWARD); }

namespace so {
int bins_gather_launch(int64_t M, int R, int32_t *sub_counts, int32_t *tile_counts, uint64_t *bin_keys, int64_t bin_cap,
                       int32_t *eff_fullest, hipStream_t st);
}
static int step_impl(const so_step_desc *d, void *stream, StepPart part, int64_t row_begin, int64_t row_end) {
  const bool forward_only = part == STEP_FORWARD;
  SO_REQUIRE(d != nullptr, "so_train_step_fwd_bwd: null descriptor");
  SO_REQUIRE(d->abi_size == (int32_t)sizeof(so_step_desc), "so_train_step_fwd_bwd: descriptor size %d != %d (ABI mismatch)",
             d->abi_size, (int)sizeof(so_step_desc));
  const int C = d->C, N = d->N, K = d->K, W = d->width, H = d->height, ts = d->tile_size;
  SO_REQUIRE(C > 0 && N > 0 && W > 0 && H > 0 && (ts == 16 || ts == 8), "so_train_step_fwd_bwd: bad sizes");
  const int tile_w = (W + ts - 1) / ts, tile_h = (H + ts - 1) / ts;
  const int64_t M = (int64_t)C * tile_w * tile_h;
  hipStream_t st = so::as_stream(stream);
  int rc;
  // counters: tile_counts[M] | cursor[M] | long-list length | n_isects | overflow     loss_sums: l1, ssim
  int32_t *tile_counts = d->counters, *cursor = d->counters + M, *n_isects = d->counters + 2 * M + 1,
          *overflow = d->counters + 2 * M + 2;
  SO_REQUIRE(part == STEP_ALL || part == STEP_FORWARD || !d->fuse_adam, "so_train_step_head / _bwd_rows: not with fuse_adam");
  if (part == STEP_TAIL) {
    // (the head of this iteration has run: counters, records and gradient records are in place)
  } else if (d->inputs_staged) {
    // so_step_inputs zeroed the counters (and the loss sums behind them) for this iteration
  } else if (reinterpret_cast<int32_t *>(d->loss_sums) == d->counters + 2 * M + 3) {
    so::zero_async(d->counters, 2 * M + 5, st);   // loss sums packed right behind the counters: one launch
  } else {
    so::zero_async(d->counters, 2 * M + 3, st);
    so::zero_async(d->loss_sums, 2, st);
  }
#define SO_TRY(call) do { rc = (call); if (rc != SO_OK) return rc; } while (0)
#define SO_STAGE(i, call) do { so::StageTimer _t(i, st); SO_TRY(call); } while (0)
  // binned lists: every tile owns bin_capacity slots of key_buf / flatten_ids; the forward kernel's returning atomics
  // place the keys, so the scan and the scatter pass do not exist (tile_counts doubles as the per-tile list length)
  const int64_t bins = d->bin_capacity;
  if (part != STEP_TAIL) {
  SO_REQUIRE(d->raster_impl == 0 || d->raster_impl == 1, "so_train_step_fwd_bwd: raster_impl must be 0 (a wave per 8x8 quadrant) or 1 "
             "(backward: a wave per 16x16 tile)");
  SO_REQUIRE(bins >= 0, "so_train_step_fwd_bwd: bad bin_capacity");
  // periodic views: spherical cameras, when the tile grid lines up across the seam (so_preprocess_fwd derives the same
  // from camera_model; the fill pass and the rasteriser take it as flags in their tile_size argument)
  int wrap_flags = 0;
  if (W % ts == 0) {
    if (!(d->camera_model & SO_CAM_PER_VIEW)) wrap_flags = d->camera_model == SO_CAM_SPHERICAL ? SO_TILE_WRAP_ALL : 0;
    else
      for (int c = 0; c < C && c < 16; ++c)
        if (((d->camera_model >> (2 * c)) & 3) == SO_CAM_SPHERICAL) wrap_flags |= SO_TILE_WRAP_CAM(c);
  }
  SO_REQUIRE(bins == 0 || M * bins < ((int64_t)1 << 31), "so_train_step_fwd_bwd: C*tiles*bin_capacity does not fit 31 bits");
  int32_t *slots = bins ? nullptr : d->tile_slots;
  uint64_t *bin_keys = bins ? d->key_buf : nullptr;
  // replicated bin counters (few tiles, see k_bins_gather): the projection bumps copy (workgroup % R) of a tile's counter
  const int reps = d->bin_replicas > 1 ? d->bin_replicas : 1;
  SO_REQUIRE(reps == 1 || (bins && d->bin_sub_counts && bins % reps == 0 && (d->n_dev || !d->radii)),
             "so_train_step_fwd_bwd: bin_replicas needs binned lists with bin_capacity %% bin_replicas == 0, bin_sub_counts, and the "
             "device-resident count (n_dev) or record-only views");
  SO_REQUIRE(d->radii || (d->rec && !d->attr_rows_f16 && bins), "so_train_step_fwd_bwd: record-only views (radii == NULL) need rec, float32 attributes and binned lists");
  if (d->n_dev && d->attr_rows_f16)
    SO_STAGE(0, so::preprocess_fwd_n_f16(C, N, K, d->sh_degree, d->means, d->logit_opacities, d->attr_rows_f16, d->viewmats, d->Ks,
                                         W, H, d->eps2d, d->near_plane, d->far_plane, d->radius_clip, d->camera_model,
                                         d->antialiased, ts, d->radii, d->means2d, d->depths, d->conics, d->opacities, d->colors,
                                         d->tiles_per_gauss, tile_counts, d->rec, forward_only ? nullptr : d->vrec, slots,
                                         d->tile_cull, bin_keys, bins, overflow, d->n_dev, stream, d->bin_sub_counts, reps));
  else if (d->n_dev || !d->radii)
    SO_STAGE(0, so::preprocess_fwd_n(C, N, K, d->sh_degree, d->means, d->log_scales, d->quats, d->logit_opacities, d->sh0, d->shN,
                                     d->viewmats, d->Ks, W, H, d->eps2d, d->near_plane, d->far_plane, d->radius_clip,
                                     d->camera_model, d->antialiased, ts, d->radii, d->means2d, d->depths, d->conics,
                                     d->opacities, d->colors, d->tiles_per_gauss, tile_counts, d->rec,
                                     forward_only ? nullptr : d->vrec, slots, d->tile_cull, bin_keys, bins, overflow, d->n_dev, stream,
                                     d->bin_sub_counts, reps));
  else if (d->attr_rows_f16)
    SO_STAGE(0, so_preprocess_fwd_f16(C, N, K, d->sh_degree, d->means, d->logit_opacities, d->attr_rows_f16, d->viewmats, d->Ks,
                                      W, H, d->eps2d, d->near_plane, d->far_plane, d->radius_clip, d->camera_model,
                                      d->antialiased, ts, d->radii, d->means2d, d->depths, d->conics, d->opacities, d->colors,
                                      d->tiles_per_gauss, tile_counts, d->rec, forward_only ? nullptr : d->vrec, 0, slots, d->tile_cull, bin_keys, bins, overflow, stream));
  else
  SO_STAGE(0, so_preprocess_fwd(C, N, K, d->sh_degree, d->means, d->log_scales, d->quats, d->logit_opacities, d->sh0, d->shN,
                           d->viewmats, d->Ks, W, H, d->eps2d, d->near_plane, d->far_plane, d->radius_clip,
                           d->camera_model, d->antialiased, ts, d->radii, d->means2d, d->depths, d->conics,
                           d->opacities, d->colors, d->tiles_per_gauss, tile_counts, d->rec, forward_only ? nullptr : d->vrec, 0, slots, d->tile_cull, bin_keys, bins, overflow, stream));
  // sort_in_rasteriser: the forward rasteriser's workgroups sort their own lists (binned lists, 16x16 tiles; short lists)
  const bool fold_sort = bins && d->sort_in_rasteriser && ts == 16;
  if (reps > 1)    // close the R slices of every bin up into one run; tile_counts gets the run lengths
    SO_STAGE(11, so::bins_gather_launch(M, reps, d->bin_sub_counts, tile_counts, d->key_buf, bins, d->bin_sub_counts + (int64_t)reps * M, st));
  if (fold_sort) {
  } else if (bins) {
    SO_STAGE(2, so_isect_sort_bins(C, tile_w, tile_h, tile_counts, bins, d->key_buf, d->flatten_ids, cursor, stream));
  } else {
    SO_STAGE(1, so_isect_scan(C, tile_w, tile_h, tile_counts, slots ? cursor : nullptr, d->isect_offsets, n_isects, stream));
    SO_STAGE(2, so_isect_fill(C, N, d->means2d, d->radii, d->depths, ts | wrap_flags, tile_w, tile_h, d->isect_offsets, n_isects, cursor,
                         d->isect_capacity, d->key_buf, d->flatten_ids, nullptr, overflow, slots, d->tile_cull ? d->rec : nullptr, stream));
  }
  // list layout handed to the rasteriser: compact (offsets, device count, capacity) or binned (counts, NULL, -slots)
  const int32_t *list_off = bins ? tile_counts : d->isect_offsets;
  const int32_t *list_n = bins ? nullptr : n_isects;
  const int64_t list_cap = bins ? -bins : d->isect_capacity;
  if (d->tile_order && !d->tile_order_ready)   // longest list first (both rasterisers); ready: the caller's own table of this view, kept from an earlier visit
    SO_STAGE(10, so::tile_order_launch(C, tile_w, tile_h, list_off, list_n, list_cap, d->tile_order, st));
  // the backward in list segments (so_step_desc.bwd_seg_len): the forward leaves the per-pixel state at the segment boundaries
  const bool segs = d->bwd_seg_len > 0 && d->bwd_seg_count > 1 && !forward_only;
  SO_REQUIRE(!segs || (ts == 16 && d->raster_impl == 0 && d->bwd_seg_state && d->bwd_seg_len % 256 == 0),
             "so_train_step_fwd_bwd: bwd_seg_len needs 16x16 tiles, raster_impl 0, bwd_seg_state and a multiple of 256");
    SO_STAGE(3, so::rasterize_fwd_packed_launch(C, N, W, H, ts | wrap_flags, d->rec, d->backgrounds, list_off, d->flatten_ids, list_n,
                                                list_cap, d->render_colors, d->render_alphas, d->last_ids, d->tile_order, stream,
                                                fold_sort ? d->key_buf : nullptr, segs ? d->bwd_seg_state : nullptr, segs ? d->bwd_seg_len : 0,
                                                segs ? d->bwd_seg_count : 1));
  if (forward_only) return SO_OK;
  // loss = (1-l) * mean|.| + l * (1 - mean SSIM_valid)
  const float n_l1 = (float)C * H * W * 3.f, n_ss = (float)C * 3.f * (float)(H - 10) * (float)(W - 10);
  so::LossFinal fin{};
  if (!d->dmaps) {   // one kernel, the derivative values never leave the CU; the loss scalars are written by the first
                     // thread of the backward rasteriser below (LossFinal), not by this kernel's last workgroup
    SO_STAGE(9, so::ssim_l1_fused_launch(C, H, W, 3, d->render_colors, d->pixels, d->pixels_indirect, 1, (1.f - d->ssim_lambda) / n_l1,
                          -d->ssim_lambda / n_ss, nullptr, d->loss_sums, d->v_render_colors, nullptr, nullptr, d->ssim_lambda, 0,
                          stream));
    fin = so::LossFinal{d->loss_sums, d->loss_sums + 2, (1.f - d->ssim_lambda) / n_l1, -d->ssim_lambda / n_ss, d->ssim_lambda,
                        1.f / n_l1, 1.f / n_ss};
  } else {           // the forward / backward pair through dmaps (kept for comparison: bench.py --loss-kernels 2)
    SO_STAGE(4, so::ssim_l1_fwd_launch(C, H, W, 3, d->render_colors, d->pixels, d->pixels_indirect, 1, d->loss_sums, d->dmaps, stream));
    SO_STAGE(5, so::ssim_l1_bwd_launch(C, H, W, 3, d->render_colors, d->pixels, d->pixels_indirect, d->dmaps, (1.f - d->ssim_lambda) / n_l1,
                          -d->ssim_lambda / n_ss, nullptr, d->v_render_colors, d->loss_sums, d->loss_sums + 2, 1,
                          d->ssim_lambda, stream));
  }
  // gradients of the intermediates accumulate in the 64-byte records vrec[C*N] (zeroed by the
  // forward preprocess kernel): one atomic request per (tile quadrant, Gaussian), or per (tile, Gaussian) with raster_impl 1
  fin.tile_waves = d->raster_impl == 1 ? 1 : 0;
  fin.tile_order = d->tile_order;
  if (segs) {
    fin.seg_state = reinterpret_cast<const float4 *>(d->bwd_seg_state);
    fin.render_colors = d->render_colors;
    fin.seg_len = d->bwd_seg_len;
    fin.seg_count = d->bwd_seg_count;
  }
    SO_STAGE(6, so::rasterize_bwd_packed_launch(C, N, W, H, ts | wrap_flags, d->rec, d->backgrounds, list_off, d->flatten_ids, list_n,
                                                list_cap, d->render_alphas, d->last_ids, d->v_render_colors, d->zero_v_alphas, d->vrec,
                                                d->absgrad, fin, stream));
  }   // part != STEP_TAIL
  if (part == STEP_HEAD) return SO_OK;
  SO_REQUIRE(part != STEP_TAIL || d->n_dev || !d->radii,
             "so_train_step_bwd_rows: needs the device-resident row count (n_dev) or record-only views (radii == NULL)");
  if (d->fuse_adam) {
    // the optimiser runs inside the backward kernel (gradients never reach HBM); its schedule for this step was
    // evaluated by so_step_inputs into the scratch behind the step counter
    const so_adam_fuse *f = d->fuse_adam;
    SO_REQUIRE(f->step_counter, "so_train_step_fwd_bwd: fuse_adam needs the step counter");
    so::AdamFuse F{};
    for (int g = 0; g < 6; ++g) { F.p[g] = f->groups[g].param; F.m[g] = f->groups[g].exp_avg; F.v[g] = f->groups[g].exp_avg_sq; }
    SO_REQUIRE(F.p[0] == d->means && F.p[1] == d->log_scales && F.p[2] == d->quats && F.p[3] == d->logit_opacities &&
                   F.p[4] == d->sh0 && F.p[5] == d->shN, "so_train_step_fwd_bwd: fuse_adam groups must be the descriptor's six parameter tensors, in order");
    F.hyper = reinterpret_cast<const float2 *>(f->step_counter + 2);
    F.h = so::AdamHyper{(float)(1.0 - f->beta1), (float)f->beta2, (float)(1.0 - f->beta2), (float)f->eps};
    if (d->attr_rows_f16) {   // the rows are read by this kernel and re-packed by it from the updated masters
      F.half_rows = const_cast<void *>(d->attr_rows_f16);
      F.half_stride16 = so::attr_rec_stride_bytes(K) / 16;
      SO_STAGE(7, so::preprocess_bwd_fused_adam_f16(C, N, K, d->sh_degree, d->means, d->logit_opacities, d->attr_rows_f16, d->viewmats,
                                                    d->Ks, W, H, d->eps2d, d->camera_model, d->antialiased, d->radii, d->opacities,
                                                    d->colors, d->opacity_reg, d->scale_reg, d->grad2d, d->count, d->vrec, d->absgrad,
                                                    overflow, d->overflow_flag_out, F, stream, d->n_dev, d->rec));
    } else
    SO_STAGE(7, so::preprocess_bwd_fused_adam(C, N, K, d->sh_degree, d->means, d->log_scales, d->quats, d->logit_opacities, d->sh0,
                                              d->shN, d->viewmats, d->Ks, W, H, d->eps2d, d->camera_model, d->antialiased, d->radii,
                                              d->opacities, d->colors, d->opacity_reg, d->scale_reg, d->grad2d, d->count, d->vrec,
                                              d->absgrad, overflow, d->overflow_flag_out, F, stream, d->n_dev, d->rec));
  } else if (d->n_dev && d->attr_rows_f16)
    SO_STAGE(7, so::preprocess_bwd_n_f16(C, N, K, d->sh_degree, d->means, d->logit_opacities, d->attr_rows_f16, d->viewmats, d->Ks,
                                         W, H, d->eps2d, d->camera_model, d->antialiased, d->radii, d->opacities, d->colors,
                                         d->opacity_reg, d->scale_reg, d->v_means, d->v_log_scales, d->v_quats,
                                         d->v_logit_opacities, d->v_sh0, d->v_shN, d->grad2d, d->count, d->vrec, d->absgrad,
                                         overflow, d->overflow_flag_out, d->n_dev, stream, row_begin, row_end));
  else if (d->n_dev || !d->radii)
    SO_STAGE(7, so::preprocess_bwd_n(C, N, K, d->sh_degree, d->means, d->log_scales, d->quats, d->logit_opacities, d->sh0, d->shN,
                                     d->viewmats, d->Ks, W, H, d->eps2d, d->camera_model, d->antialiased, d->radii, d->opacities,
                                     d->colors, d->opacity_reg, d->scale_reg, d->v_means, d->v_log_scales, d->v_quats,
                                     d->v_logit_opacities, d->v_sh0, d->v_shN, d->grad2d, d->count, d->vrec, d->absgrad,
                                     overflow, d->overflow_flag_out, d->n_dev, d->rec, stream, row_begin, row_end));
  else if (d->attr_rows_f16)
    SO_STAGE(7, so_preprocess_bwd_f16(C, N, K, d->sh_degree, d->means, d->logit_opacities, d->attr_rows_f16, d->viewmats, d->Ks,
                                      W, H, d->eps2d, d->camera_model, d->antialiased, d->radii, d->opacities, d->colors,
                                      d->opacity_reg, d->scale_reg, d->v_means, d->v_log_scales, d->v_quats,
                                      d->v_logit_opacities, d->v_sh0, d->v_shN, d->grad2d, d->count, d->vrec, d->absgrad, 0,
                                      overflow, d->overflow_flag_out, stream));
  else
  SO_STAGE(7, so_preprocess_bwd(C, N, K, d->sh_degree, d->means, d->log_scales, d->quats, d->logit_opacities, d->sh0, d->shN,
                           d->viewmats, d->Ks, W, H, d->eps2d, d->camera_model, d->antialiased, d->radii, d->opacities,
                           d->colors, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d->opacity_reg,
                           d->scale_reg, d->v_means, d->v_log_scales, d->v_quats, d->v_logit_opacities, d->v_sh0,
                           d->v_shN, d->grad2d, d->count, d->vrec, d->absgrad, 0, overflow, d->overflow_flag_out, stream));
#undef SO_STAGE
#undef SO_TRY
  return SO_OK;
}
