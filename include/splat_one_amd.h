/*
 * splat_one_amd.h -- C ABI of the MI355X-native 3D-Gaussian-splatting training path.
 *
 * This is the drop-in boundary for the ONE hot path of inuex35/splat_one: everything that
 * `Runner.rasterize_splats` reaches through `gsplat.rendering.rasterization(...)`
 * (/root/reference/utils/gsplat_utils/gsplat_trainer.py:446-497, call at :477-494), its
 * backward (`loss.backward()` at :655), the optimiser step (:726-742) and the densification
 * strategy hooks (:616-622, :744-763).  In the reference those live in the CUDA extension of the
 * un-vendored gsplat fork (/root/reference/.gitmodules:13-16); each entry point below names the
 * gsplat operator it replaces and the reference line that reaches it.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into HBM unless it is named host_*; tensors are dense,
 *     row-major, float32 / int32 / int64 as declared; "nullable" pointers may be NULL.
 *   - inputs are borrowed; outputs are written into caller-owned buffers; no entry point allocates,
 *     frees or synchronises (all are hipGraph-capturable).  Scratch comes from caller workspaces.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - return value: SO_OK or an SO_ERR_* code; so_last_error() gives a thread-local message.
 *   - (C, N): cameras x Gaussians.  Flattened (camera, Gaussian) index g = c*N + n.
 *   - camera_model: SO_CAM_PINHOLE / SO_CAM_ORTHO / SO_CAM_FISHEYE / SO_CAM_SPHERICAL (trainer
 *     Config.camera_model, gsplat_trainer.py:89; "spherical" is fork-only: the model is the one this
 *     library defines, see enum so_camera_model below); any other value -> SO_ERR_UNSUPPORTED.
 *
 * Reference-side bindings (ctypes / torch autograd.Function) are shown in INTEGRATION.md.
 */
#ifndef SPLAT_ONE_AMD_H
#define SPLAT_ONE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SO_ABI_VERSION 2

enum so_status {
  SO_OK = 0,
  SO_ERR_INVALID_ARG = 1, /* bad shape / null pointer / unsupported size           */
  SO_ERR_UNSUPPORTED = 2, /* valid request this build does not implement           */
  SO_ERR_WORKSPACE = 3,   /* caller workspace too small                            */
  SO_ERR_LAUNCH = 4       /* HIP reported an error at launch                       */
};

enum so_camera_model { SO_CAM_PINHOLE = 0, SO_CAM_ORTHO = 1, SO_CAM_FISHEYE = 2, SO_CAM_SPHERICAL = 3 };
/* SO_CAM_SPHERICAL: the 360-degree (equirectangular) cameras of the reference's data sets (utils/datasets/opensfm.py:
 * 176-193, 430-436).  The fork's own kernel for it is absent from the reference tree: the model is defined by this
 * library (csrc/splat_math.hpp: u = W (atan2(x,z)/2pi + 1/2), v = H (atan2(y,|xz|)/pi + 1/2), depth = range, K unused)
 * -- parity unpinned. */
/* The fused entry points (so_preprocess_fwd / _bwd and their _f16 forms, so_step_desc.camera_model) also take one
 * model PER VIEW -- BASELINE.json configs[4] mixes perspective and fisheye cameras (app/camera_models.py:230-237)
 * in one batch:  SO_CAM_PER_VIEW | m_0 | m_1 << 2 | ... | m_{C-1} << 2 (C-1),  2 bits per view, C <= 15. */
#define SO_CAM_PER_VIEW 0x40000000
#define SO_CAM_PER_VIEW_MAX 15
/* Periodic images.  A spherical (equirectangular) view is periodic in x: a Gaussian whose footprint crosses the
 * +-pi seam (x = 0 / x = W) continues on the other side of the image.  The binning and rasteriser entry points learn
 * which cameras are periodic from flags OR-ed into their `tile_size` argument (its low byte stays the tile size):
 *   tile_size | SO_TILE_WRAP_ALL          every camera;      tile_size | SO_TILE_WRAP_CAM(c)   camera c (c < 16).
 * Binning then walks VIRTUAL tile columns (floor((x - r)/ts) .. ceil((x + r)/ts), not clamped to the image, at most one
 * image width) and files column x under x mod tile_width; the rasteriser shifts every staged Gaussian by the multiple
 * of W that brings it closest to the tile.  Requires width % tile_size == 0 (the virtual tile grid must line up across
 * the seam): callers pass no flag otherwise.  so_preprocess_fwd / so_train_step_fwd_bwd derive the flags from
 * camera_model themselves. */
#define SO_TILE_WRAP_ALL (1 << 24)
#define SO_TILE_WRAP_CAM(c) (1 << (8 + (c)))

int so_abi_version(void);
const char *so_last_error(void);
/* number of compute units / XCDs of the current device (launch sizing on the host side) */
int so_device_cu_count(void);
/* viewmats[C,4,4] = inverse(camtoworlds[C,4,4]) in one launch (replaces `torch.linalg.inv(camtoworlds)`,
 * gsplat_trainer.py:483) */
int so_camera_inverse(int C, const float *camtoworlds, float *viewmats, void *stream);
/* test hook for the wave64 reduction primitives: in[n_waves*64,9] -> out[n_waves,10] */
int so_debug_wave_reduce(int n_waves, const float *in, float *out, void *stream);
/* test hook for the rasteriser backward's nine-sum network (round 3): in[n_waves*64,9] -> out[n_waves,9] */
int so_debug_wave_reduce9(int n_waves, const float *in, float *out, void *stream);
/* test hook for the rasterisers' per-quadrant culling: in[n][10] = {mx, my, opacity, conic a, b, c, x0, x1, y0, y1}
 * -> out[n][2] = {box test, exact test} as 0 / 1 */
int so_debug_cull(int64_t n, const float *in, float *out, void *stream);
/* test hook: out[t] = so_bin_counter_index(t, M) as the DEVICE code evaluates it (t = 0 .. M-1) */
int so_debug_bin_counter_index(int64_t M, int64_t *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * K1/K2  3D -> 2D EWA projection.   Replaces gsplat `fully_fused_projection` (legacy
 * `project_gaussians`) forward/backward, reached from gsplat_trainer.py:477.
 *   means[N,3]; covars6[N,6] (xx,xy,xz,yy,yz,zz) nullable; quats[N,4] (w,x,y,z, un-normalised),
 *   scales[N,3] (used when covars6 == NULL); viewmats[C,4,4] world->camera; Ks[C,3,3].
 *   -> radii[C,N] i32 (0 = culled), means2d[C,N,2], depths[C,N], conics[C,N,3],
 *      compensations[C,N] nullable.  Culled entries are written as zeros.
 * ---------------------------------------------------------------------------------------- */
int so_projection_fwd(int C, int N, const float *means, const float *covars6, const float *quats,
                      const float *scales, const float *viewmats, const float *Ks, int width, int height,
                      float eps2d, float near_plane, float far_plane, float radius_clip, int camera_model,
                      int32_t *radii, float *means2d, float *depths, float *conics, float *compensations,
                      void *stream);

/* Backward.  v_means[N,3], v_quats[N,4], v_scales[N,3] (or v_covars6[N,6]) are OVERWRITTEN with
 * the sum over cameras (one thread owns one Gaussian: no atomics, bitwise reproducible).
 * v_viewmats[C,4,4] nullable: if given it must be zero-initialised; it is accumulated atomically. */
int so_projection_bwd(int C, int N, const float *means, const float *covars6, const float *quats,
                      const float *scales, const float *viewmats, const float *Ks, int width, int height,
                      float eps2d, int camera_model, const int32_t *radii, const float *v_means2d,
                      const float *v_depths, const float *v_conics, const float *v_compensations,
                      float *v_means, float *v_covars6, float *v_quats, float *v_scales, float *v_viewmats,
                      void *stream);

/* K3  packed projection.   Replaces gsplat `fully_fused_projection(packed=True)` (`cfg.packed`,
 * gsplat_trainer.py:133, :487): one row per (camera, Gaussian) pair with a positive radius, in ascending flattened
 * index c*N + n (camera-major; the reference builds coalesced sparse gradients from gaussian_ids, :705-717).  No [C,N]
 * array exists at any point: so_projection_packed is called TWICE --
 *   counting pass (camera_ids == NULL): block_counts[B] i32, vis_masks[16 B] u64, block_offsets[B] i64 with
 *     B = so_projection_packed_blocks(C,N), and *total_dev (i64, device) = nnz are written (projection + wave-ballot counts
 *     + one-workgroup scan); vis_masks holds one bit per (camera, Gaussian) pair: the rows;
 *   writing pass (camera_ids != NULL, after the caller sized its nnz-row outputs; same vis_masks / block_offsets): the
 *     projection is recomputed for the values, WHICH pairs are rows is read from vis_masks (decided once), and row
 *     block_offsets[b] + rank is written: camera_ids / gaussian_ids [nnz] i64, radii[nnz] i32, means2d[nnz,2],
 *     depths[nnz], conics[nnz,3], compensations[nnz] (nullable).
 * so_projection_bwd_packed: one lane per packed row; v_means / v_quats / v_scales (or v_covars6) and the nullable
 *   v_viewmats must be ZERO-initialised and are accumulated with float atomics. */
int64_t so_projection_packed_blocks(int C, int N);
int so_projection_packed(int C, int N, const float *means, const float *covars6, const float *quats, const float *scales,
                         const float *viewmats, const float *Ks, int width, int height, float eps2d, float near_plane,
                         float far_plane, float radius_clip, int camera_model, int32_t *block_counts, uint64_t *vis_masks,
                         int64_t *block_offsets, int64_t *total_dev, int64_t *camera_ids, int64_t *gaussian_ids,
                         int32_t *radii, float *means2d, float *depths, float *conics, float *compensations, void *stream);
int so_projection_bwd_packed(int C, int N, int64_t nnz, const float *means, const float *covars6, const float *quats,
                             const float *scales, const float *viewmats, const float *Ks, int width, int height,
                             float eps2d, int camera_model, const int64_t *camera_ids, const int64_t *gaussian_ids,
                             const float *v_means2d, const float *v_depths, const float *v_conics,
                             const float *v_compensations, float *v_means, float *v_covars6, float *v_quats,
                             float *v_scales, float *v_viewmats, void *stream);

/* ------------------------------------------------------------------------------------------
 * K4/K5  spherical harmonics.   Replaces gsplat `spherical_harmonics` (reached with
 * sh_degree=... from gsplat_trainer.py:591).
 *   dirs[C,N,3] (not normalised); coeffs[N,K,3] (coeffs_per_camera=0) or [C,N,K,3] (=1);
 *   masks[C,N] u8 nullable (0 -> output 0, no gradient); degrees_to_use <= 4, (deg+1)^2 <= K.
 *   -> colors[C,N,3]   (the caller adds 0.5 and clamps, as gsplat's `rasterization` does)
 * Backward: v_coeffs (same shape as coeffs) is OVERWRITTEN (sum over cameras when shared);
 * v_dirs[C,N,3] nullable, overwritten.
 * ---------------------------------------------------------------------------------------- */
/* The whole colour stage of `rasterization(sh_degree=...)` in one launch each way (dense layout, coefficients shared by
 * the cameras): colors[c,n] = max(SH(normalise(means[n] - campos[c])) . coeffs[n] + 0.5, 0) where radii[c,n] > 0, 0.5
 * elsewhere (radii nullable: all visible); the backward OVERWRITES v_coeffs[N,K,3] and v_means[N,3] (sums over the
 * cameras) and takes the clamp from the forward's output `colors`. */
int so_sh_view_colors_fwd(int C, int N, int K, int degrees_to_use, const float *means, const float *campos,
                          const float *coeffs, const int32_t *radii, float *colors, void *stream);
int so_sh_view_colors_bwd(int C, int N, int K, int degrees_to_use, const float *means, const float *campos,
                          const float *coeffs, const int32_t *radii, const float *colors, const float *v_colors,
                          float *v_coeffs, float *v_means, void *stream);
int so_sh_fwd(int C, int N, int K, int degrees_to_use, const float *dirs, const float *coeffs,
              int coeffs_per_camera, const uint8_t *masks, float *colors, void *stream);
int so_sh_bwd(int C, int N, int K, int degrees_to_use, const float *dirs, const float *coeffs,
              int coeffs_per_camera, const uint8_t *masks, const float *v_colors, float *v_coeffs,
              float *v_dirs, void *stream);

/* ------------------------------------------------------------------------------------------
 * K6-K8  tile binning + per-tile depth sort.   Replaces gsplat `isect_tiles` (two launches +
 * cub radix sort) and `isect_offset_encode`, reached inside `rasterization` (:477).
 *
 * MI355X design: counting sort by tile (histogram + scan gives `isect_offsets` directly), then
 * one workgroup per tile sorts its (fp32 depth bits, flatten id) keys in LDS.  Result is identical
 * to the reference's global stable radix sort of (camera | tile | depth) keys: ties in depth are
 * ordered by ascending flatten id.
 *
 * Step 1  so_isect_count : tiles_per_gauss[C,N] i32, tile_counts[C*tile_h*tile_w] i32 (must be
 *         zeroed by the caller), then exclusive scan -> isect_offsets[C,tile_h,tile_w] i32 and
 *         n_isects (1 x i32, device).
 * Step 2  so_isect_fill  : scatters keys into key_buf[capacity] (u64: depth_bits<<32 | g),
 *         sorts each tile, writes flatten_ids[capacity] i32 and, if non-NULL,
 *         isect_ids[capacity] i64 (= cam << (32+tile_bits) | tile << 32 | depth_bits).
 *         `capacity` is the size of the caller's buffers; if n_isects > capacity nothing beyond
 *         capacity is written and *overflow (device i32, nullable) is set to 1.
 *         tile_cursor[C*tile_h*tile_w + 1] i32 must be zeroed by the caller (after the scatter the
 *         array is reused as the work list of tiles whose list exceeds 2048 keys; the extra
 *         element is its length).  Lists up to 16384 keys sort in 128 KiB of LDS, longer ones
 *         fall back to the same network in global memory.
 * ---------------------------------------------------------------------------------------- */
int so_isect_count(int C, int N, const float *means2d, const int32_t *radii, int tile_size, int tile_width,
                   int tile_height, int32_t *tiles_per_gauss, int32_t *tile_counts, int32_t *isect_offsets,
                   int32_t *n_isects, const float *cull_rec /* nullable: exact tile culling, see so_isect_fill */,
                   void *stream);
/* the exclusive scan of step 1 alone (when the histogram was filled by so_preprocess_fwd) */
int so_isect_scan(int C, int tile_width, int tile_height, const int32_t *tile_counts,
                  const int32_t *tile_counts_big /* nullable: added element-wise (so_preprocess_fwd tile_slots) */, int32_t *isect_offsets,
                  int32_t *n_isects, void *stream);
int so_isect_fill(int C, int N, const float *means2d, const int32_t *radii, const float *depths,
                  int tile_size, int tile_width, int tile_height, const int32_t *isect_offsets,
                  const int32_t *n_isects, int32_t *tile_cursor, int64_t capacity, uint64_t *key_buf,
                  int32_t *flatten_ids, int64_t *isect_ids, int32_t *overflow,
                  const int32_t *tile_slots /* nullable, see below */,
                  const float *cull_rec /* nullable, see below */, void *stream);
/* Slotted binning (tile_slots, int32[C*N][SO_TILE_SLOTS], nullable everywhere): so_preprocess_fwd's histogram uses
 * RETURNING atomics for rectangles of <= SO_TILE_SLOTS tiles and keeps what they return -- the Gaussian's slot in each
 * tile's list -- in tile_slots (row-major over the rectangle); larger rectangles are counted apart, in the second half
 * of a tile_counts[2*C*tiles] array.  so_isect_scan then takes both halves (tile_counts_big = tile_counts + C*tiles)
 * and so_isect_fill, given the same tile_slots and tile_cursor = that second half, writes a slotted key to
 * offsets[tile] + slot with no atomic at all and lets the large rectangles fill the tail of each list from the back
 * (counting tile_cursor down to zero).  One round of atomics per iteration instead of two. */
#define SO_TILE_SLOTS 12
/* Binned lists -- the third way to build the per-tile lists, and the fused engine's: so_preprocess_fwd(bin_keys,
 * bin_cap, bin_overflow) gives every (camera, tile) bin_cap key slots; the returning atomic on tile_counts[t] is the slot
 * and the key {depth bits, flatten id} is stored at bin_keys[t * bin_cap + slot] at once (slot >= bin_cap raises
 * *bin_overflow instead).  so_isect_sort_bins then sorts each bin's min(tile_counts[t], bin_cap) keys in place and
 * writes flatten_ids[t * bin_cap + i]; long_list is int32[C*tiles + 1] scratch whose last element is zero on entry.
 * The rasteriser's packed entry points take this layout as isect_offsets = tile_counts, n_isects_dev = NULL,
 * n_isects_host = -bin_cap.
 * Tile t of M = C*tiles keeps its COUNT at tile_counts[so_bin_counter_index(t, M)] (a fixed bijection that deals runs of two
 * neighbouring counters 4 KB apart: the binning atomics of a scene gathered in one image region then spread over many
 * 64-byte lines instead of queueing on a few -- so_preprocess_fwd 1014 -> 452 us with 2M splats in a fifth of the scene);
 * keys and ids stay at t * bin_cap.  Every entry point above applies it; a caller that reads per-tile counts itself goes
 * through the function. */
int64_t so_bin_counter_index(int64_t t, int64_t M);
int so_isect_sort_bins(int C, int tile_width, int tile_height, const int32_t *tile_counts, int64_t bin_cap,
                       uint64_t *bin_keys, int32_t *flatten_ids, int32_t *long_list, void *stream);
/* Exact tile culling (tile_cull of so_preprocess_fwd + cull_rec = its 64-byte records for so_isect_fill; both or
 * neither): gsplat bins a Gaussian into every tile of the square of half-width ceil(3 sqrt(lambda_max)) around its
 * centre, while the rasteriser drops every pair with alpha = opacity exp(-sigma) < 1/255.  A tile whose pixel-centre
 * rectangle lies wholly outside the ellipse sigma <= ln(255 opacity) (+ margin) therefore changes no output and is left
 * out of the lists: 15 % fewer intersections on isotropic trained-like scenes, 60 % on dense anisotropic low-opacity
 * ones.  Images, losses and gradients are unchanged (tests/test_gpu_engine.py); the LISTS differ from gsplat's, so the
 * operator-level isect_tiles culls only when asked to (conics + opacities given: so_rec_pack builds the records;
 * rasterization(tile_cull=True), the default of this build, does). */

/* gsplat `isect_tiles(sort=False)`: Gaussian-major, row-major-tile emission order.
 * cum_tiles[C*N] i64 = inclusive prefix sum of tiles_per_gauss (caller-provided). */
int so_isect_emit_unsorted(int C, int N, const float *means2d, const int32_t *radii, const float *depths,
                           const int64_t *cum_tiles, int tile_size, int tile_width, int tile_height,
                           int64_t *isect_ids, int32_t *flatten_ids, void *stream);

/* gsplat `isect_offset_encode`: offsets[c,ty,tx] = first index of that tile's run in sorted ids */
int so_isect_offset_encode(int64_t n_isects, const int64_t *isect_ids, int C, int tile_width,
                           int tile_height, int32_t *isect_offsets, void *stream);

/* ------------------------------------------------------------------------------------------
 * K9/K10  tile rasteriser.   Replaces gsplat `rasterize_to_pixels` forward/backward.
 *   means2d[C,N,2] conics[C,N,3] colors[C,N,D] opacities[C,N]; backgrounds[C,D] nullable;
 *   tile_masks[C,tile_h,tile_w] u8 nullable; isect_offsets[C,tile_h,tile_w]; flatten_ids[>=n_isects].
 *   n_isects is read from the device pointer `n_isects_dev` when it is non-NULL, else the host
 *   value `n_isects_host` is used; with both given (host value > 0) the smaller one counts -- pass the
 *   capacity of `flatten_ids` there so that an overflowed binning pass is never walked past the buffer.  D in {1,2,3,4,5,8,9,16,17,32,33}.  tile_size in {8,16}.
 *   -> render_colors[C,H,W,D], render_alphas[C,H,W], last_ids[C,H,W] i32.
 * Backward: v_means2d[C,N,2], v_conics[C,N,3], v_colors[C,N,D], v_opacities[C,N] and the
 * nullable v_means2d_abs[C,N,2] (`absgrad`) must be zero-initialised; they are accumulated with
 * float atomics (one per tile and Gaussian).
 * ---------------------------------------------------------------------------------------- */
int so_rasterize_fwd(int C, int N, int D, int width, int height, int tile_size, const float *means2d,
                     const float *conics, const float *colors, const float *opacities,
                     const float *backgrounds, const uint8_t *tile_masks, const int32_t *isect_offsets,
                     const int32_t *flatten_ids, const int32_t *n_isects_dev, int64_t n_isects_host,
                     float *render_colors, float *render_alphas, int32_t *last_ids, void *stream);
int so_rasterize_bwd(int C, int N, int D, int width, int height, int tile_size, const float *means2d,
                     const float *conics, const float *colors, const float *opacities,
                     const float *backgrounds, const uint8_t *tile_masks, const int32_t *isect_offsets,
                     const int32_t *flatten_ids, const int32_t *n_isects_dev, int64_t n_isects_host,
                     const float *render_alphas, const int32_t *last_ids, const float *v_render_colors,
                     const float *v_render_alphas, float *v_means2d, float *v_means2d_abs, float *v_conics,
                     float *v_colors, float *v_opacities, void *stream);

/* Packed-record variants (D = 3): rec / vrec as written by so_preprocess_fwd. */
int so_rasterize_fwd_packed(int C, int N, int width, int height, int tile_size, const float *rec,
                            const float *backgrounds, const int32_t *isect_offsets, const int32_t *flatten_ids,
                            const int32_t *n_isects_dev, int64_t n_isects_host, float *render_colors,
                            float *render_alphas, int32_t *last_ids, void *stream);
int so_rasterize_bwd_packed(int C, int N, int width, int height, int tile_size, const float *rec,
                            const float *backgrounds, const int32_t *isect_offsets, const int32_t *flatten_ids,
                            const int32_t *n_isects_dev, int64_t n_isects_host, const float *render_alphas,
                            const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
                            float *vrec, int absgrad, void *stream);


/* ------------------------------------------------------------------------------------------
 * Optimiser.  Replaces the six torch.optim.Adam steps + zero_grad of gsplat_trainer.py:726-731
 * (hyper-parameters per :266-280) with ONE launch over up to SO_ADAM_MAX_GROUPS tensors.
 *   per group g: param/grad/exp_avg/exp_avg_sq [numel[g]] f32; step_size = lr/bias_correction1,
 *   bc2_sqrt = sqrt(1-beta2^t) are computed by the caller (host) from its step counter.
 *   Semantics = torch.optim.Adam(amsgrad=False, weight_decay=0, maximize=False):
 *     m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= step_size * m / (sqrt(v)/bc2_sqrt + eps)
 *   beta1/beta2/eps are doubles so that (1-beta) is rounded to float32 once, exactly as torch does.
 *   If `zero_grad` != 0 the gradient is zeroed in the same pass.
 *   visibility[g] (u8 per ROW of row_len[g] elements, nullable): rows with 0 are skipped entirely
 *   (gsplat `SelectiveAdam`, gsplat_trainer.py:269-270, 719-728).
 * ---------------------------------------------------------------------------------------- */
#define SO_ADAM_MAX_GROUPS 8
typedef struct so_adam_group {
  float *param;
  float *grad;
  float *exp_avg;
  float *exp_avg_sq;
  const uint8_t *visibility; /* nullable */
  int64_t numel;
  int32_t row_len;
  float lr_step_size; /* lr / (1 - beta1^t) */
  float bc2_sqrt;     /* sqrt(1 - beta2^t)  */
} so_adam_group;
int so_adam_step(int n_groups, const so_adam_group *host_groups, double beta1, double beta2, double eps,
                 int zero_grad, void *stream);
/* The same step for a row / flat piece of a data-parallel replica (one view per GPU, gradients reduce-scattered:
 * SURVEY.md section 8e; the reference scales its hyper-parameters for that batch at gsplat_trainer.py:266-278):
 *   grad_scale  multiplies the gradient on the fly (1 / world: the reduce-scatter delivers the SUM over the ranks);
 *   skip_f32    (nullable, device) a float the same reduce-scatter has summed over the ranks -- non-zero: some rank's
 *               binning pass overflowed, the iteration is void on EVERY rank, nothing is written. */
int so_adam_step_scaled(int n_groups, const so_adam_group *host_groups, double beta1, double beta2, double eps,
                        int zero_grad, const float *skip_f32, float grad_scale, void *stream);

/* Device-scheduled variant for hipGraph replay: `step_counter` is a device int32[2 + 4*SO_ADAM_MAX_GROUPS]
 * scratch whose element 0 is the number of optimiser steps done so far (the rest holds the per-group
 * step sizes of the current step).  A one-block kernel evaluates lr = lr0[g] * lr_gamma[g]^step
 * (ExponentialLR, gsplat_trainer.py:512-516 and :741-742), the bias corrections for t = step+1, and
 * increments the counter afterwards.  Launch arguments are therefore constant across iterations. */
int so_adam_step_dev(int n_groups, const so_adam_group *host_groups, const float *host_lr0,
                     const float *host_lr_gamma, double beta1, double beta2, double eps, int32_t *step_counter,
                     int zero_grad, int schedule_done, const int32_t *skip_if_nonzero_i32,
                     const float *skip_if_nonzero_f32, void *stream);
/* The same step that also keeps the float16 attribute rows (so_attr_pack_f16 below) current: for a group g with
 * offset_bytes[g] >= 0 every updated parameter is ALSO stored, rounded to nearest-even, as a half at
 *   arec + (e / row_len[g]) * stride_bytes + offset_bytes[g] + 2 * (e % row_len[g])        (e = element index),
 * so the float32 masters and their float16 copies never diverge and no separate repack launch is needed.
 * `row_len` of such a group must be its true row length (3, 4, 3(K-1) ...) and numel < 2^31.  shadow == NULL (or
 * shadow->arec == NULL) is so_adam_step_dev. */
typedef struct so_attr_shadow {
  void *arec;
  int32_t stride_bytes;
  int32_t offset_bytes[SO_ADAM_MAX_GROUPS]; /* < 0: no float16 copy of this group */
} so_attr_shadow;
int so_adam_step_dev_shadow(int n_groups, const so_adam_group *host_groups, const float *host_lr0,
                            const float *host_lr_gamma, double beta1, double beta2, double eps,
                            int32_t *step_counter, int zero_grad, int schedule_done,
                            const int32_t *skip_if_nonzero_i32, const float *skip_if_nonzero_f32,
                            const so_attr_shadow *shadow, void *stream);

/* ------------------------------------------------------------------------------------------
 * Fused front end / back end on the RAW parameters (what `Runner.rasterize_splats` holds,
 * gsplat_trainer.py:456-474): exp / sigmoid activations, SH concat, camera centre, projection, SH
 * colour (+0.5, clamp) and the first binning pass in ONE forward kernel; projection bwd + SH bwd +
 * activation bwd + opacity/scale regularisers (:650-653) + the densification statistics of
 * `DefaultStrategy._update_state` (:744-752) in ONE backward kernel.
 *   means[N,3] log_scales[N,3] quats[N,4] logit_opacities[N] sh0[N,1,3] shN[N,K-1,3]
 *   -> radii, means2d, depths, conics, opacities[C,N] (sigmoid, x compensation if antialiased),
 *      colors[C,N,3], tiles_per_gauss[C,N], tile_counts[C*tiles] (+=, zeroed by the caller)
 * Backward overwrites v_means, v_log_scales, v_quats, v_logit_opacities, v_sh0, v_shN and adds
 * to grad2d[N] / count[N] (nullable pair).  v_means2d_abs / v_depths nullable.
 * Packed records (64-byte aligned, nullable): the forward also writes rec[C*N][16] =
 *   {x, y, conic a, b, c, opacity, r, g, b, depth, radius bits, cull threshold, cull box xmin, xmax, ymin, ymax}
 *   (one cache line per Gaussian for the rasteriser's gathers; slots 11..15, round 3, are the rasterisers' own culling
 *   data -- the threshold of the exact quadrant test and the box of the alpha >= 1/255 region, computed once per
 *   Gaussian here or by so_rec_pack: records for the packed rasterisers must come from one of these two) and zeroes
 *   vrec[C*N][16]; so_rasterize_bwd_packed accumulates
 *   {v_x, v_y, v_ca, v_cb, v_cc, v_r, v_g, v_b, v_opacity, abs_x, abs_y, 0...} there and the backward
 *   reads them from `vrec` instead of the five separate v_* arrays.
 * cam_stride (0 = N): row stride between cameras of every per-view array and of rec / vrec -- lets a
 *   rank whose shard is shorter than the exchange buffers use them in place.  tile_counts may be NULL
 *   (no histogram: Gaussian-sharded runs bin after the exchange, on the records of all shards).
 * so_rec_unpack: from rec[n][16] writes means2d[n,2], radii[n], depths[n] (the inputs of so_isect_count /
 *   so_isect_fill) and zeroes vrec[n][16] (nullable).
 * so_shard_flag_put / so_shard_flag_get: a Gaussian-sharded step (the reference's `distributed=True` scheme,
 *   gsplat_trainer.py:236-238, 477-494) sends block j of vrec_full[world][cap][16] to rank j.  put writes
 *   (*overflow != 0) as 1.0f / 0.0f into slot 15 of the first record of every block before that exchange (after
 *   so_rasterize_bwd_packed; slots 11..15 carry no gradient); get, on the received vrec_shard[world][cap][16],
 *   SETS *overflow when any sender's flag is set (it never clears it).  Every rank then skips the same iteration
 *   (so_preprocess_bwd / so_adam_step_dev skip_if_nonzero) without a collective of its own.  world <= 64.
 * ---------------------------------------------------------------------------------------- */
int so_preprocess_fwd(int C, int N, int K, int sh_degree, const float *means, const float *log_scales,
                      const float *quats, const float *logit_opacities, const float *sh0, const float *shN,
                      const float *viewmats, const float *Ks, int width, int height, float eps2d,
                      float near_plane, float far_plane, float radius_clip, int camera_model, int antialiased,
                      int tile_size, int32_t *radii, float *means2d, float *depths, float *conics,
                      float *opacities, float *colors, int32_t *tiles_per_gauss, int32_t *tile_counts,
                      float *rec, float *vrec, int64_t cam_stride, int32_t *tile_slots, int tile_cull,
                      uint64_t *bin_keys /* nullable, see so_isect_sort_bins */, int64_t bin_cap, int32_t *bin_overflow,
                      void *stream);
int so_preprocess_bwd(int C, int N, int K, int sh_degree, const float *means, const float *log_scales,
                      const float *quats, const float *logit_opacities, const float *sh0, const float *shN,
                      const float *viewmats, const float *Ks, int width, int height, float eps2d,
                      int camera_model, int antialiased, const int32_t *radii, const float *opacities,
                      const float *colors, const float *v_means2d, const float *v_means2d_abs,
                      const float *v_depths, const float *v_conics, const float *v_colors,
                      const float *v_opacities, float opacity_reg, float scale_reg, float *v_means,
                      float *v_log_scales, float *v_quats, float *v_logit_opacities, float *v_sh0, float *v_shN,
                      float *grad2d, float *count, const float *vrec, int absgrad_stats, int64_t cam_stride, const int32_t *skip_if_nonzero,
                      float *skip_flag_out, void *stream);
int so_rec_unpack(int64_t n, const float *rec, float *means2d, int32_t *radii, float *depths, float *vrec,
                  void *stream);
/* Operator-level `rasterize_to_pixels` with RGB colours (gsplat_trainer.py:477-494 through `rasterization`): the
 * separate arrays means2d[n,2], conics[n,3], colors[n,3], opacities[n] packed into rec[n][16] for
 * so_rasterize_fwd_packed / so_rasterize_bwd_packed (vrec, nullable, is zeroed; colors may be NULL when the records
 * only serve the exact tile culling of so_isect_count / so_isect_fill), and the gradient records
 * vrec[n][16] spread back into v_means2d / v_conics / v_colors / v_opacities (+ v_means2d_abs, nullable). */
int so_rec_pack(int64_t n, const float *means2d, const float *conics, const float *colors, const float *opacities,
                float *rec, float *vrec, void *stream);
int so_rec_unpack_grads(int64_t n, const float *vrec, float *v_means2d, float *v_conics, float *v_colors,
                        float *v_opacities, float *v_means2d_abs, void *stream);
int so_shard_flag_put(int world, int64_t cap, const int32_t *overflow, float *vrec_full, void *stream);
int so_shard_flag_get(int world, int64_t cap, const float *vrec_shard, int32_t *overflow, void *stream);

/* ------------------------------------------------------------------------------------------
 * float16 attribute storage (BASELINE.json configs[4]: "2M Gaussians ... fp16 attributes").  The 56 of the 59
 * parameter floats per Gaussian that tolerate it are kept as halves in ONE 16-byte aligned row per Gaussian,
 *   arec[N][stride]:  +0 f16 quat[4] | +8 f16 log_scale[3] | +14 f16 0 | +16 f16 sh[K][3] (sh0, then shN), zero pad
 *   stride = so_attr_rec_stride(K) = 16 + roundup16(6 K) bytes   (K = 16: 112 B against 224 B in five arrays),
 * read by so_preprocess_fwd_f16 / so_preprocess_bwd_f16 with 16-byte loads; positions and opacity logits stay
 * float32 in their own arrays (a half resolves 2e-3 world units at |x| = 3, several pixels at 1080p).  All
 * arithmetic is float32: the result equals so_preprocess_fwd / _bwd on the half-rounded attribute values.
 * Gradients are float32 and go to the float32 masters, whose optimiser step refreshes the halves
 * (so_adam_step_dev_shadow); so_attr_pack_f16 (re)builds all rows from the masters, e.g. after densification.
 * The *_f16 entry points take the same arguments as their float32 counterparts with `arec` in place of
 * log_scales / quats / sh0 / shN; the backward reads the rasteriser's gradients from `vrec` only.
 * ---------------------------------------------------------------------------------------- */
int64_t so_attr_rec_stride(int K);
int so_attr_pack_f16(int64_t N, int K, const float *log_scales, const float *quats, const float *sh0,
                     const float *shN, void *arec, void *stream);
/* ... of a device-resident model (so_step_desc.n_dev): N is the capacity, rows [0, *n_dev) are packed (n_dev nullable) */
int so_attr_pack_f16_n(int64_t N, int K, const float *log_scales, const float *quats, const float *sh0,
                       const float *shN, void *arec, const int32_t *n_dev, void *stream);
int so_preprocess_fwd_f16(int C, int N, int K, int sh_degree, const float *means, const float *logit_opacities,
                          const void *arec, const float *viewmats, const float *Ks, int width, int height,
                          float eps2d, float near_plane, float far_plane, float radius_clip, int camera_model,
                          int antialiased, int tile_size, int32_t *radii, float *means2d, float *depths,
                          float *conics, float *opacities, float *colors, int32_t *tiles_per_gauss,
                          int32_t *tile_counts, float *rec, float *vrec, int64_t cam_stride, int32_t *tile_slots,
                          int tile_cull, uint64_t *bin_keys, int64_t bin_cap, int32_t *bin_overflow, void *stream);
int so_preprocess_bwd_f16(int C, int N, int K, int sh_degree, const float *means, const float *logit_opacities,
                          const void *arec, const float *viewmats, const float *Ks, int width, int height,
                          float eps2d, int camera_model, int antialiased, const int32_t *radii,
                          const float *opacities, const float *colors, float opacity_reg, float scale_reg,
                          float *v_means, float *v_log_scales, float *v_quats, float *v_logit_opacities,
                          float *v_sh0, float *v_shN, float *grad2d, float *count, const float *vrec,
                          int absgrad_stats, int64_t cam_stride, const int32_t *skip_if_nonzero,
                          float *skip_flag_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * One training iteration (gsplat_trainer.py:586-655: render -> loss -> backward) as ONE call on
 * caller-owned static buffers: 6 launches with binned lists, fused Adam and the single-kernel loss (up to
 * 11 otherwise), no allocation, no host read-back, capturable in a hipGraph.  Gradients of the raw parameters are overwritten; loss_sums[6] = (sum|x-y|,
 * sum SSIM_valid, loss, l1, 1-SSIM, ticket of so_ssim_l1_fused: an int32 that is 0 before the first step).  dmaps
 * NULL: the loss is the single kernel so_ssim_l1_fused; dmaps[3,C,H,W,3] given: the so_ssim_l1_fwd/bwd pair.  counters: int32[2*C*tiles + 3] (histogram | cursor | long-list length |
 * n_isects | overflow);
 * zero_v_alphas: float[C*H*W] of zeros (the photometric loss does
 * not depend on alpha).  `abi_size` must be sizeof(so_step_desc).
 * ---------------------------------------------------------------------------------------- */
typedef struct so_step_desc {
  /* raw parameters */
  const float *means, *log_scales, *quats, *logit_opacities, *sh0, *shN;
  /* cameras and target */
  const float *viewmats, *Ks, *pixels, *backgrounds /* nullable [C,3] */;
  /* per-view intermediates.  radii, means2d, depths, conics, opacities, colors, tiles_per_gauss may ALL be NULL
   * ("record-only views"; needs rec, float32 attributes and bin_capacity > 0): the forward then writes only the
   * 64-byte records -- which hold the same values -- and the backward reads radius / colour / opacity from them. */
  int32_t *radii;
  float *means2d, *depths, *conics, *opacities, *colors;
  int32_t *tiles_per_gauss, *counters, *isect_offsets;
  uint64_t *key_buf;
  int32_t *flatten_ids;
  float *render_colors, *render_alphas;
  int32_t *last_ids;
  float *loss_sums /* [2] sums, [3] loss, l1, ssimloss, [1] int32 ticket */, *dmaps /* nullable */, *v_render_colors;
  const float *zero_v_alphas;
  float *rec, *vrec; /* [C*N][16] packed records, 64-byte aligned */
  /* gradients of the raw parameters */
  float *v_means, *v_log_scales, *v_quats, *v_logit_opacities, *v_sh0, *v_shN;
  /* densification statistics (nullable pair) */
  float *grad2d, *count;
  int64_t isect_capacity;
  int32_t abi_size, C, N, K, width, height, tile_size, sh_degree, camera_model, antialiased, absgrad;
  int32_t raster_impl; /* backward rasteriser mapping: 0 one wave per 8x8 quadrant (default); 1 one wave per 16x16 tile -- one
                        * reduction and one atomic per (tile, Gaussian) instead of per (quadrant, Gaussian): faster from ~250
                        * list entries per tile on (16x16 tiles, no absgrad; otherwise 0 is used) */
  float eps2d, near_plane, far_plane, radius_clip, ssim_lambda, opacity_reg, scale_reg;
  /* Inputs staged by so_step_inputs (both optional, zero = off):
   *   pixels_indirect  device slot holding the address of this iteration's target image [C,H,W,3]; when set it
   *                    is read INSTEAD of `pixels`, so a resident dataset image is used in place, not copied;
   *   inputs_staged    != 0: `counters` (and the loss sums behind them) are already zero, skip that launch. */
  const float *const *pixels_indirect;
  int32_t inputs_staged;
  int32_t tile_cull; /* != 0: exact tile culling in the binning passes (see so_isect_fill); needs rec */
  /* A binning pass that does not fit `isect_capacity` raises counters[2M+2] (overflow).  The iteration is then
   * VOID: lists are walked only up to the capacity (no out-of-bounds access), so_preprocess_bwd leaves the
   * gradients and densification statistics untouched, and so_adam_step_dev skips when given that flag.
   * overflow_flag_out (nullable): 1.0f / 0.0f copy of the flag, e.g. a spare slot behind the flat gradient
   * buffer so that a gradient all-reduce carries "some rank overflowed" to every rank. */
  float *overflow_flag_out;
  /* float16 attribute rows (nullable; see so_attr_pack_f16): when set, quaternions, log-scales and SH coefficients
   * are read from here instead of log_scales / quats / sh0 / shN (which may then be NULL). */
  const void *attr_rows_f16;
  /* int32[C*N][SO_TILE_SLOTS] scratch (nullable): the binning histogram keeps the slot each returning atomic handed
   * out, so the scatter pass places those keys without a second round of atomics (see so_preprocess_fwd). */
  int32_t *tile_slots;
  /* Binned lists (0 = off): every (camera, tile) owns bin_capacity slots of key_buf / flatten_ids (both then hold
   * C*tiles*bin_capacity entries; isect_offsets / isect_capacity / tile_slots are unused).  The forward kernel's
   * returning histogram atomic IS the slot, so the key goes straight to its place: no scan, no scatter pass.  A tile
   * that receives more than bin_capacity Gaussians raises the overflow flag (void iteration, as above); the caller
   * then enlarges the bins -- at 288 GB of HBM, 8160 tiles x 4096 slots x 12 B = 400 MB is not a constraint. */
  int64_t bin_capacity;
  /* Optimiser fused into the backward (nullable): the last kernel of the step applies Adam to the six parameter
   * tensors itself -- the 236 B/Gaussian of gradient never travel to HBM and back and so_adam_step_dev is not called.
   * The v_* gradient outputs are then NOT written.  Single-GPU steps only (a data-parallel step needs the gradients
   * for its all-reduce); the schedule of this step must have been evaluated by so_step_inputs (n_groups = 6).
   * With attr_rows_f16 (K <= 18) the same kernel re-packs the rows of the Gaussians it updates from their new float32
   * values (round to nearest even, as so_attr_pack_f16): the rows stay == half(masters) without a further pass. */
  const struct so_adam_fuse *fuse_adam;
  /* Device-resident Gaussian count (nullable): when set, `N` above is the CAPACITY of every per-Gaussian buffer (the
   * parameters, their moments, the per-view arrays with row stride N, rec / vrec, grad2d / count) and the number of
   * live Gaussians is read from *n_dev by the kernels themselves -- so_refine_default changes it on the device and a
   * captured step follows without re-capture or host read-back.  With attr_rows_f16 the rows follow their masters through
   * so_attr_pack_f16_n. */
  const int32_t *n_dev;
  /* int32[C*tiles] scratch (nullable): when set, the step builds a workgroup -> tile table by descending list length (one
   * small launch behind the per-tile sort) and BOTH rasterisers take their tiles in that order, longest list first.  For
   * long-list regimes (raster_impl == 1: a tile is one wave's serial chain, so the kernel cannot end before its longest
   * tile does -- started last, that tile runs on alone).  Speed only: any tile order gives the same images and gradients. */
  int32_t *tile_order;
  /* != 0 (binned lists, 16x16 tiles): the per-tile sort is NOT launched; the forward rasteriser's workgroup sorts the list of
   * its own tile before it walks it (<= 256 keys: one wave, in registers; <= 2048: the workgroup; longer: a slow scratch-free
   * rank sort) and writes flatten_ids for the backward.  One launch and one pass over the keys fewer where lists are short
   * everywhere; the same lists, images and gradients either way. */
  int32_t sort_in_rasteriser;
  /* Replicated bin counters (binned lists; needs n_dev or record-only views; bin_capacity % bin_replicas == 0).  R = bin_replicas
   * > 1: the returning atomics of ONE counter serialise (~230 ns each on an MI355X), so on images of few tiles with long lists
   * the binning pass runs at 4 G atomics/s instead of ~19.  With R copies of the counters -- bin_sub_counts: int32[R * C * tiles
   * + 1], zero on entry, kept zero by the step; the last word is raised to R x the fullest slice when a slice overflows -- a
   * workgroup of the projection kernel bumps copy (workgroup % R) and fills slice (workgroup % R) of the bin (bin_capacity / R
   * slots each); one small launch then closes the slices up and writes the list lengths where they always are (`counters`).
   * Same lists after the sort, same images and gradients; a slice that overflows voids the iteration like a bin that overflows. */
  int32_t bin_replicas;
  int32_t *bin_sub_counts;
  /* Backward rasteriser in list SEGMENTS (16x16 tiles, raster_impl 0).  bwd_seg_len > 0 (a multiple of 256): the forward
   * rasteriser writes every pixel's (live transmittance, accumulated colour) at the boundaries of its tile's list -- after entries
   * bwd_seg_len, 2 bwd_seg_len, ... -- into bwd_seg_state (float[(bwd_seg_count - 1) * C * H * W * 4], 16-byte aligned), and the
   * backward runs bwd_seg_count workgroups per tile, each from the state at the far end of its segment: a list of thousands of
   * entries is no longer ONE serial chain (few tiles, or one hot image region).  The last segment takes everything beyond
   * (bwd_seg_count - 1) x bwd_seg_len entries (the caller sizes bwd_seg_len by its fullest list).  Same gradients up to the order of
   * float additions. */
  int32_t bwd_seg_len, bwd_seg_count;
  float *bwd_seg_state;
  /* != 0: tile_order already holds a workgroup -> tile table for these views (any permutation of the C x tiles indices is valid:
   * the order only schedules) -- the step does not build one.  The engine keeps the table it built the last time it saw a view and
   * hands it back through so_step_inputs (order_src): list lengths of a view change slowly over training, the table's launch
   * (~13 us at 1080p) is then paid once every few visits. */
  int32_t tile_order_ready;
} so_step_desc;
typedef struct so_adam_fuse {
  so_adam_group groups[6]; /* means, log_scales, quats, logit_opacities, sh0, shN: param / exp_avg / exp_avg_sq (grad unused) */
  double beta1, beta2, eps;
  int32_t *step_counter;   /* the device scratch of so_adam_step_dev */
} so_adam_fuse;
int so_train_step_fwd_bwd(const so_step_desc *desc, void *stream);
/* The same iteration cut in two for data-parallel replicas (one view per GPU, `cli(main, cfg)` gsplat_trainer.py:998;
 * gradient exchange instead of the reference's Gaussian sharding: SURVEY.md section 8e), so that the exchange starts
 * before the backward has finished:
 *   so_train_step_head      forward, loss, rasteriser backward -- the per-view gradient records vrec are complete;
 *   so_train_step_bwd_rows  the per-Gaussian backward (projection / SH / activations, regularisers, densification
 *                           statistics) for rows [row_begin, row_end) of every gradient tensor; row_begin a multiple of
 *                           64.  One lane per Gaussian, no dependence between rows: the caller launches it chunk by chunk
 *                           and starts the reduce-scatter of chunk i while chunk i + 1 runs.
 * head + rows [0, N) == so_train_step_fwd_bwd bit for bit.  Needs desc->n_dev or record-only views; not with fuse_adam. */
int so_train_step_head(const so_step_desc *desc, void *stream);
int so_train_step_bwd_rows(const so_step_desc *desc, int64_t row_begin, int64_t row_end, void *stream);
/* Everything that changes from one iteration to the next, in ONE launch, so that a captured step needs
 * neither copies nor re-capture (the reference does `inv(camtoworlds)`, `.to(device)` and the scheduler
 * arithmetic on the host each iteration, gsplat_trainer.py:586-603, :483, :749-751):
 *   viewmats[c] = inverse(camtoworlds[c]) (general 4x4, in double);  Ks_dst = Ks_src (nullable pair);
 *   *pixels_slot = pixels (nullable pair; see so_step_desc.pixels_indirect);
 *   counters[0 .. n_zero) = 0 (int32 words; nullable);
 *   n_groups > 0: the Adam schedule of so_adam_step_dev for the step in step_counter[0] (hyper-parameters
 *   written behind the counter, counter advanced) -- pass schedule_done = 1 to so_adam_step_dev then;
 *   status_out (nullable, host-mapped int32[3]): before zeroing, counters[status_at] and [status_at+1]
 *   (n_isects and overflow of the PREVIOUS iteration on these buffers) are published as
 *   {n_isects, overflow, seq} -- the host learns of a void iteration one step late, without a device sync;
 *   lists_stat (nullable, device int32[4], zero before the first call; with status_out, which is then int32[5]): the
 *   first n_lists counters are per-tile list lengths (binned lists) -- their maximum and sum are gathered while they are
 *   zeroed and published by the NEXT call as status_out[3], [4] (written before seq): the host follows the growth of the
 *   lists -- larger bins before a tile overflows, the backward rasteriser that suits the length -- without reading the device. */
int so_step_inputs(int C, const float *camtoworlds, const float *Ks_src, float *viewmats, float *Ks_dst,
                   const float *pixels, const float **pixels_slot, int32_t *counters, int64_t n_zero, int n_groups,
                   const float *lr0, const float *lr_gamma, double beta1, double beta2, int32_t *step_counter,
                   int32_t *status_out, int64_t status_at, int32_t seq, int64_t n_lists, int32_t *lists_stat,
                   const int32_t *order_src /* nullable */, int32_t *order_dst, int64_t n_order, void *stream);
/* the forward stages only (preprocess, binning, sort, rasterise) on the same descriptor: the eval /
 * viewer render of gsplat_trainer.py:779-940; pixels, loss and gradient buffers are not touched */
int so_render_forward(const so_step_desc *desc, void *stream);

/* ------------------------------------------------------------------------------------------
 * `gsplat.rendering.rasterization` WHOLE, one call each way.   The reference renders through ONE Python call,
 *   render_colors, render_alphas, info = rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height,
 *       packed=False, absgrad=..., sparse_grad=False, rasterize_mode=..., distributed=False, camera_model=...,
 *       sh_degree=..., near_plane=..., far_plane=..., render_mode="RGB")              (gsplat_trainer.py:477-494)
 * and differentiates it with loss.backward() (:655).  For the common shape of that call -- dense layout, SH coefficients
 * [N,K,3] shared by the cameras, poses without gradient, three colour channels -- so_rasterization_fwd runs projection +
 * SH colour (+0.5, clamp) + tile binning + per-tile depth sort + rasterisation, and so_rasterization_bwd the backward of
 * all of them, on the tensors the call is handed:
 *   activated = 1 (the gsplat call): scales[N,3] = exp(log-scales), opacities[N] = sigmoid(logits), sh0 = colors[N,K,3]
 *     (ONE coefficient tensor, cat(sh0, shN) at :474; shN unused); gradients come back for exactly these tensors, the
 *     coefficient gradient in two pieces v_sh0[N,3] | v_shN[N,3(K-1)] (band 0 | bands 1..K-1);
 *   activated = 0: the raw parameters (log-scales, opacity logits, sh0[N,1,3], shN[N,K-1,3]) -- exp / sigmoid and their
 *     backward run inside, as in so_preprocess_fwd / _bwd.
 * Per-call buffers (the caller allocates; the backward needs counters, flatten_ids, rec, vrec, render_alphas, last_ids
 * of its forward untouched): counters int32[2*C*tiles + 3] (zeroed HERE), flatten_ids int32[C*tiles*bin_capacity],
 * rec / vrec float[C*N][16] (64-byte aligned; vrec NULL = forward only, nothing saved for a backward).
 * Scratch that only lives inside the forward: key_buf u64[C*tiles*bin_capacity].
 * Lists are BINNED (see so_isect_sort_bins): a tile that receives more than bin_capacity Gaussians raises overflow and is
 * rasterised with its first bin_capacity entries.  status_out (nullable, HOST-MAPPED int32[4]) receives
 * {fullest tile's count, overflow, seq, intersections} when the forward's last kernel has run: the caller enlarges its
 * bins from it one call late, without synchronising (and rejects / repeats a call whose overflow flag was set).
 * Outputs: render_colors[C,H,W,3], render_alphas[C,H,W,1], last_ids[C,H,W]; `info` = strided views of rec:
 *   {x, y, conic a b c, opacity, r, g, b, depth, radius bits, 0 ...}.
 * Backward: v_render_colors[C,H,W,3], v_render_alphas[C,H,W,1] (both required) -> v_means[N,3], v_quats[N,4], v_scales[N,3],
 * v_opacities[N], v_sh0, v_shN (overwritten: one lane owns one Gaussian, sums over the cameras, no atomics) and, when
 * v_means2d[C,N,2] is given, the screen-space gradient the densification strategy accumulates (`info["means2d"].grad`,
 * :616-622, :744-752; v_means2d_abs[C,N,2] with absgrad != 0).
 * ---------------------------------------------------------------------------------------- */
typedef struct so_raster_desc {
  int32_t abi_size, C, N, K, width, height, tile_size, sh_degree, camera_model, antialiased, absgrad, tile_cull, activated, seq;
  int32_t raster_impl; /* backward only: 0 one wave per 8x8 quadrant, 1 one wave per 16x16 tile (as so_step_desc.raster_impl:
                        * faster from ~250 list entries per tile on), -1 the process default (SPLAT_ONE_AMD_BWD_TILE) */
  float eps2d, near_plane, far_plane, radius_clip;
  int64_t bin_capacity;
  /* inputs */
  const float *means, *quats, *scales, *opacities, *sh0, *shN, *viewmats, *Ks, *backgrounds /* nullable [C,3] */;
  /* per-call buffers and scratch */
  int32_t *counters;
  uint64_t *key_buf;
  int32_t *flatten_ids;
  float *rec, *vrec;
  int32_t *status_out; /* nullable, host-mapped */
  /* forward outputs (inputs of the backward) */
  float *render_colors, *render_alphas;
  int32_t *last_ids;
  /* backward */
  const float *v_render_colors, *v_render_alphas;
  float *v_means, *v_quats, *v_scales, *v_opacities, *v_sh0, *v_shN, *v_means2d /* nullable */, *v_means2d_abs /* nullable */;
  /* replicated bin counters, as so_step_desc.bin_replicas / bin_sub_counts (int32[R * C * tiles + 1], zero on entry, kept zero):
   * for images of few tiles; status_out[0] then reports max(fullest list, R x fullest slice of a bin that overflowed) */
  int32_t *bin_sub_counts;
  int32_t bin_replicas;
  /* int32[C*tiles] (nullable), as so_step_desc.tile_order: the forward builds the workgroup -> tile table (longest list first) and
   * both rasterisers follow it; the backward must be given the same buffer, untouched.  Worth it from ~60 list entries per tile on. */
  int32_t *tile_order;
} so_raster_desc;
int so_rasterization_fwd(const so_raster_desc *desc, void *stream);
int so_rasterization_bwd(const so_raster_desc *desc, void *stream);

/* Per-stage HIP-event timing of so_train_step_fwd_bwd / so_adam_step_dev on their launch stream
 * (measurement only; the switch and the event log belong to the CALLING THREAD, so another thread's launches are
 * neither timed nor affected; events cannot be recorded inside a hipGraph replay, so profile un-captured launches).
 * so_profile_read (same thread) synchronises the device, returns the summed
 * milliseconds and call counts of the so_profile_num_stages() stages and clears the log. */
int so_profile_enable(int enabled);
int so_profile_num_stages(void);
const char *so_profile_stage_name(int stage);
int so_profile_read(float *host_ms_sum, int *host_calls);
void so_profile_stage_begin_end(int stage, int begin, void *stream);

/* so_adam_step_dev with the ROW count of every group in device memory (so_step_desc.n_dev): numel[g] is then
 * capacity * row_len[g] and only the first *n_rows_dev rows are stepped. */
int so_adam_step_dev_n(int n_groups, const so_adam_group *host_groups, const float *host_lr0,
                       const float *host_lr_gamma, double beta1, double beta2, double eps, int32_t *step_counter,
                       int zero_grad, int schedule_done, const int32_t *skip_if_nonzero_i32,
                       const float *skip_if_nonzero_f32, const int32_t *n_rows_dev, void *stream);

/* ------------------------------------------------------------------------------------------
 * Device-side densification.   Replaces gsplat `DefaultStrategy._grow_gs` / `_prune_gs` / `reset_opa`
 * (duplicate, split, remove on the parameters AND the Adam moments), which the reference drives from
 * gsplat_trainer.py:744-763 every `refine_every` steps (hooks :616-622; defaults SURVEY.md section 8 a11 / B.3).
 * gsplat reads three counts back to the host per refinement to size its torch.cat's; here the refinement is a
 * stream compaction from one capacity-preallocated model set into another, with N in device memory:
 *   avg = grad2d / max(count, 1);  high = avg > grow_grad2d;  small = max_k exp(log_scale_k) <= grow_scale3d
 *   duplicate = high & small (copy appended, moments zero);  split = high & !small (replaced by two samples
 *   mean + R diag(s) z, z ~ N(0, I), scales / 1.6, moments zero; opacity 1 - sqrt(1 - o) when revised_opacity);
 *   prune (evaluated on the grown set) = sigmoid(logit) < prune_opa, or, when prune_big, max scale > prune_scale3d.
 * Output rows, in gsplat's order: [kept originals | kept duplicates | first children | second children], each in
 * source order.  z is a counter-based function of (seed, step, source row, child, component) -- Philox4x32-10 +
 * Box-Muller, csrc/so_rng.hpp -- so no host generator is involved and replicas agree without communication.
 *   src / dst: two DISJOINT model sets of `capacity` rows each (tensor order: means[.,3], log_scales[.,3], quats[.,4],
 *     logit_opacities[.], sh0[.,3], shN[.,3(K-1)]; p = parameter, m = exp_avg, v = exp_avg_sq);
 *   n_src_dev / n_dst_dev: device int32 (distinct): live rows of src (read), of dst (written);
 *   grad2d / count [capacity]: the densification statistics, zeroed on exit;
 *   scratch: int32[so_refine_scratch_words(capacity)];
 *   report_dev int32[8] (zero-initialised once by the caller; device or HOST-MAPPED memory -- a caller that maps it reads
 *     it without synchronising, element 6 telling which refinement it describes): {duplicated, split, pruned, new N,
 *     overflow, old N, refinements so far, rows the refined set needs}.  overflow = the refined set does not fit
 *     `capacity`: NOTHING is refined then -- dst receives the source rows unchanged (new N = old N), grad2d / count keep
 *     their sums -- and the caller enlarges its buffers (element 7 says by how much) and calls again.
 * so_reset_opacity: gsplat `reset_opa` -- logits clamped to max_logit, their moments zeroed; n_dev nullable.
 * No entry point here synchronises or reads back; all are hipGraph-capturable.
 * ---------------------------------------------------------------------------------------- */
typedef struct so_model_set {
  float *p[6], *m[6], *v[6];
} so_model_set;
typedef struct so_refine_params {
  float grow_grad2d;
  float grow_scale3d;  /* already multiplied by scene_scale */
  float prune_opa;
  float prune_scale3d; /* already multiplied by scene_scale */
  int32_t prune_big;   /* step > reset_every */
  int32_t revised_opacity;
  uint64_t seed;
  int32_t step;
  int32_t reserved;
} so_refine_params;
int64_t so_refine_scratch_words(int64_t capacity);
int so_refine_default(int64_t capacity, int K, const so_model_set *src, const int32_t *n_src_dev,
                      const so_model_set *dst, int32_t *n_dst_dev, float *grad2d, float *count,
                      const so_refine_params *prm, int32_t *scratch, int32_t *report_dev, void *stream);
int so_reset_opacity(int64_t capacity, const int32_t *n_dev, float *logit_opacities, float *exp_avg, float *exp_avg_sq,
                     float max_logit, void *stream);
/* gsplat `DefaultStrategy._update_state` (the `step_post_backward` hook, gsplat_trainer.py:744-752) on the dense layout:
 * for every Gaussian n and every camera c with radii[c,n] > 0:
 *   grad2d[n] += |(v_means2d[c,n].x * sx, v_means2d[c,n].y * sy)|,  count[n] += 1,
 *   radii_state[n] = max(radii_state[n], radii[c,n] * inv_max_wh)   (radii_state nullable)
 * with sx = width / 2 * n_cameras, sy = height / 2 * n_cameras, inv_max_wh = 1 / max(width, height).  (The fused engine
 * accumulates the same inside so_preprocess_bwd.)
 * radii_stride: int32 words between consecutive radii (0 or 1: a dense [C,N] array; 16: the radius slot of the 64-byte
 * records, `info["radii"]` of so_rasterization_fwd). */
int so_strategy_update_state(int C, int64_t N, const float *v_means2d, const int32_t *radii, int64_t radii_stride, float sx,
                             float sy, float inv_max_wh, float *grad2d, float *count, float *radii_state, void *stream);

/* ------------------------------------------------------------------------------------------
 * MCMC densification strategy (the reference's `mcmc` preset, gsplat_trainer.py:975-983, :753-761).
 * so_compute_relocation replaces gsplat `compute_relocation` (K13): for a Gaussian split `ratios[i]`
 *   ways, new_opacity = 1-(1-o)^(1/n) and new_scale = s * o / sum_{i<=n} sum_{k<i} C(i-1,k)(-1)^k
 *   new_opacity^(k+1)/sqrt(k+1); binoms[n_max,n_max] f32, ratios clamped to [1,n_max].
 * so_inject_noise replaces `inject_noise_to_position`: means += Sigma (noise * sig(1-opacity) * scaler)
 *   on the RAW parameters (log scales, logit opacities), sig(x) = 1/(1+exp(-100(x-0.995))).
 * ---------------------------------------------------------------------------------------- */
int so_compute_relocation(int64_t N, const float *opacities, const float *scales, const int32_t *ratios,
                          const float *binoms, int n_max, float *new_opacities, float *new_scales, void *stream);
int so_inject_noise(int64_t N, float *means, const float *log_scales, const float *quats,
                    const float *logit_opacities, const float *noise, float scaler, void *stream);
/* MCMCStrategy on the DEVICE-RESIDENT model (so_step_desc.n_dev): gsplat `relocate` + `sample_add` of one refinement
 * (gsplat_trainer.py:753-761 every refine_every steps of the `mcmc` preset, :975-983) in place on ONE capacity-sized model
 * set, with the Gaussian count in device memory -- nothing is read back, nothing re-allocated, a captured step follows.
 * A draw is a function of (seed, step, phase, sample number): Philox4x32-10 -> uniform -> inverse CDF over a float64
 * device prefix sum of the opacities (gsplat: torch.multinomial from a host-seeded generator).
 *   relocate: dead rows (sigmoid(logit) <= min_opacity, ascending) take the place of rows drawn from the ALIVE ones in
 *     proportion to opacity; a source drawn r times gets (opacity, scale) = so_compute_relocation(., r + 1), clamped to
 *     [min_opacity, 1 - eps], written back as logit / log, its moments (exp_avg, exp_avg_sq of all six tensors) zeroed; the
 *     dead row becomes a copy of the updated source.
 *   sample_add: n_add = min(cap_max, int(1.05 N)) - N rows drawn from ALL rows (after the relocation) the same way, the
 *     sources updated the same way (their moments kept), the copies appended with zero moments; *n_dev += n_add
 *     (capacity >= cap_max is the caller's contract; n_add is clamped to the capacity otherwise).
 *   set: the model (tensor order of so_model_set); binoms[n_max,n_max] as so_compute_relocation takes them;
 *   scratch: int32[so_mcmc_scratch_words(capacity)], 8-byte aligned, ZERO before the first call (left tidy by every call);
 *   report_dev int32[8] (device or host-mapped): {relocated, added, 0, new N, 0, old N, refinements so far, 0}.
 * so_inject_noise_dev: `inject_noise_to_position` of EVERY iteration with the normals drawn on the device from (seed,
 *   step_counter[0], row) and the scale lr * noise_lr evaluated there too, lr = lr0 * lr_gamma^step_counter[0] (the means'
 *   ExponentialLR value after this iteration's optimiser step: step_counter is so_adam_step_dev's) -- constant launch
 *   arguments, hipGraph-capturable behind the optimiser.  n_dev nullable (N = capacity); skip_if_nonzero nullable: ONE
 *   4-byte device word tested for any set bit -- an int32 flag, or the float32 void flag a gradient reduce-scatter summed
 *   over the ranks (+0.0f is all-zero bits; a sum of 0 / 1 flags is never -0.0f). */
typedef struct so_mcmc_params {
  float min_opacity;
  int32_t cap_max;
  uint64_t seed;
  int32_t step;
  int32_t reserved; /* 0; bit 0 / bit 1 set: leave out the relocation / the addition (tests look at one phase at a time) */
} so_mcmc_params;
int64_t so_mcmc_scratch_words(int64_t capacity);
int so_mcmc_refine(int64_t capacity, int K, const so_model_set *set, int32_t *n_dev, const float *binoms, int n_max,
                   const so_mcmc_params *prm, int32_t *scratch, int32_t *report_dev, void *stream);
int so_inject_noise_dev(int64_t capacity, const int32_t *n_dev, float *means, const float *log_scales, const float *quats,
                        const float *logit_opacities, uint64_t seed, const int32_t *step_counter, float lr0, float lr_gamma,
                        float noise_lr, const int32_t *skip_if_nonzero, void *stream);

/* ------------------------------------------------------------------------------------------
 * Photometric loss.  Replaces F.l1_loss + the CUDA-only `fused_ssim(..., padding="valid")` of
 * gsplat_trainer.py:624-628 with one forward and one backward kernel on the rasteriser's own
 * channel-last layout.  SSIM: 11x11 Gaussian window (sigma 1.5), C1=0.01^2, C2=0.03^2.
 *   img1 (rendered; receives the gradient), img2 (target): [B,H,W,CH] f32, CH in {1,3,4}.
 *   sums[2] (zeroed by the caller): sums[0] += sum|img1-img2| ; sums[1] += sum of the SSIM map over
 *   all pixels, or over the interior (5-pixel crop) when padding_valid != 0.
 *   dmaps[3,B,H,W,CH]: derivative maps saved for the backward (nullable when no gradient is needed).
 * Backward: v_img1 = v_loss * ( w_l1 * sign(img1-img2) + w_ssim * d(sum SSIM)/d img1 ), with
 *   v_loss a device scalar (nullable = 1); if loss_out[3] (nullable) is given together with the
 *   forward's `sums`, it receives (w_l1*sums[0] + w_ssim*sums[1] + loss_const, mean|x-y|, 1-mean SSIM).
 * ---------------------------------------------------------------------------------------- */
int so_ssim_l1_fwd(int B, int H, int W, int CH, const float *img1, const float *img2, int padding_valid,
                   float *sums, float *dmaps, void *stream);
int so_ssim_l1_bwd(int B, int H, int W, int CH, const float *img1, const float *img2, const float *dmaps,
                   float w_l1, float w_ssim, const float *v_loss, float *v_img1, const float *sums,
                   float *loss_out, int padding_valid, float loss_const, void *stream);
/* Both in ONE launch, for callers that need the gradient right away (the training step): two chained row-streaming
 * stages per workgroup, the derivative values pass through LDS instead of dmaps.  sums[2] (zeroed by the caller) and
 * v_img1 as above.  loss_out[3] (nullable) is written by the workgroup that finishes last; it needs `ticket`: one
 * int32 that is 0 before the first launch (the kernel returns it to 0) and is not shared by launches that can
 * overlap.  rows = output rows per workgroup, 0 = as few as keep the whole grid resident at once. */
int so_ssim_l1_fused(int B, int H, int W, int CH, const float *img1, const float *img2, int padding_valid,
                     float w_l1, float w_ssim, const float *v_loss, float *sums, float *v_img1, float *loss_out,
                     int32_t *ticket, float loss_const, int rows, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SPLAT_ONE_AMD_H */
