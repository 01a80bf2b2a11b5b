/*
 * oracle/c/raster_oracle.c -- plain-C restatement of tile rasterisation (forward + backward).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py; "parity unpinned": the reference's
 * arithmetic for this path lives in the un-vendored gsplat fork, /root/reference/.gitmodules:13-16).
 * Follows the published front-to-back alpha-compositing algorithm that the reference reaches
 * through `rasterization(...)` at /root/reference/utils/gsplat_utils/gsplat_trainer.py:477-494
 * (SURVEY.md Appendix B.1 step 7 for the forward, B.2 for the backward).
 *
 * Why C: the float64 torch oracle (oracle/torch_oracle.py) is the ground truth for gradients
 * (autograd), but it cannot rasterise 1920x1080 with millions of tile intersections in seconds.
 * This file can; tests/test_oracle_c.py pins it against the torch oracle on small cases.
 *
 * Built twice by oracle/Makefile: REAL=double (parity) and REAL=float (timed CPU baseline).
 * One sequential loop per pixel, exactly as the algorithm is written; OpenMP over tiles.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef REAL
#define REAL double
#endif

#define ALPHA_MAX ((REAL)0.999)
#define ALPHA_MIN ((REAL)(1.0 / 255.0))
#define T_STOP ((REAL)1e-4)
#define MAX_D 8

static inline REAL real_exp(REAL x) { return sizeof(REAL) == 4 ? (REAL)expf((float)x) : (REAL)exp((double)x); }

/* periodic images (equirectangular panoramas; build-defined, see torch_oracle.isect_tiles): a Gaussian is evaluated at
 * the copy  x - W * round((x - tile centre) / W)  nearest to the tile being rasterised */
static inline REAL nearest_copy(REAL x, REAL tile_cx, int W, int periodic) {
  return periodic ? x - (REAL)W * (REAL)rint((double)((x - tile_cx) / (REAL)W)) : x;
}

int oracle_real_size(void) { return (int)sizeof(REAL); }

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* means2d[C*N*2] conics[C*N*3] colors[C*N*D] opacities[C*N] backgrounds[C*D]|NULL
 * isect_offsets[C*tile_h*tile_w] flatten_ids[n_isects]
 * -> render_colors[C*H*W*D] render_alphas[C*H*W] last_ids[C*H*W] */
int oracle_rasterize_fwd(int C, int N, int D, int W, int H, int tile_size, int tile_w, int tile_h, int periodic,
                         const REAL *means2d, const REAL *conics, const REAL *colors,
                         const REAL *opacities, const REAL *backgrounds,
                         const int32_t *isect_offsets, const int32_t *flatten_ids, int64_t n_isects,
                         REAL *render_colors, REAL *render_alphas, int32_t *last_ids) {
  if (D > MAX_D) return -1;
  (void)N;
  const int n_tiles = tile_w * tile_h;
  const int64_t total_tiles = (int64_t)C * n_tiles;
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t ct = 0; ct < total_tiles; ++ct) {
    const int c = (int)(ct / n_tiles), t = (int)(ct % n_tiles);
    const int ty = t / tile_w, tx = t % tile_w;
    const REAL tile_cx = (REAL)(tx * tile_size) + (REAL)0.5 * (REAL)tile_size;
    const int64_t lo = isect_offsets[ct];
    const int64_t hi = (ct == total_tiles - 1) ? n_isects : isect_offsets[ct + 1];
    for (int i = ty * tile_size; i < (ty + 1) * tile_size && i < H; ++i) {
      for (int j = tx * tile_size; j < (tx + 1) * tile_size && j < W; ++j) {
        const REAL px = (REAL)j + (REAL)0.5, py = (REAL)i + (REAL)0.5;
        REAL T = 1, pix[MAX_D] = {0};
        int32_t last = 0;
        for (int64_t k = lo; k < hi; ++k) {
          const int32_t g = flatten_ids[k];
          const REAL dx = nearest_copy(means2d[2 * g], tile_cx, W, periodic) - px, dy = means2d[2 * g + 1] - py;
          const REAL a = conics[3 * g], b = conics[3 * g + 1], cc = conics[3 * g + 2];
          const REAL sigma = (REAL)0.5 * (a * dx * dx + cc * dy * dy) + b * dx * dy;
          REAL alpha = opacities[g] * real_exp(-sigma);
          if (alpha > ALPHA_MAX) alpha = ALPHA_MAX;
          if (sigma < 0 || alpha < ALPHA_MIN) continue;
          const REAL next_T = T * (1 - alpha);
          if (next_T <= T_STOP) break;
          const REAL vis = alpha * T;
          for (int d = 0; d < D; ++d) pix[d] += colors[(int64_t)g * D + d] * vis;
          last = (int32_t)k;
          T = next_T;
        }
        const int64_t p = ((int64_t)c * H + i) * W + j;
        render_alphas[p] = 1 - T;
        for (int d = 0; d < D; ++d)
          render_colors[p * D + d] = backgrounds ? pix[d] + T * backgrounds[c * D + d] : pix[d];
        last_ids[p] = last;
      }
    }
  }
  return 0;
}

/* v_render_colors[C*H*W*D] v_render_alphas[C*H*W] -> (accumulated into zero-initialised)
 * v_means2d[C*N*2] v_means2d_abs[C*N*2]|NULL v_conics[C*N*3] v_colors[C*N*D] v_opacities[C*N] */
int oracle_rasterize_bwd(int C, int N, int D, int W, int H, int tile_size, int tile_w, int tile_h, int periodic,
                         const REAL *means2d, const REAL *conics, const REAL *colors,
                         const REAL *opacities, const REAL *backgrounds,
                         const int32_t *isect_offsets, const int32_t *flatten_ids, int64_t n_isects,
                         const REAL *render_alphas, const int32_t *last_ids,
                         const REAL *v_render_colors, const REAL *v_render_alphas,
                         REAL *v_means2d, REAL *v_means2d_abs, REAL *v_conics, REAL *v_colors,
                         REAL *v_opacities) {
  if (D > MAX_D) return -1;
  (void)N;
  const int n_tiles = tile_w * tile_h;
  const int64_t total_tiles = (int64_t)C * n_tiles;
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t ct = 0; ct < total_tiles; ++ct) {
    const int c = (int)(ct / n_tiles), t = (int)(ct % n_tiles);
    const int ty = t / tile_w, tx = t % tile_w;
    const REAL tile_cx = (REAL)(tx * tile_size) + (REAL)0.5 * (REAL)tile_size;
    const int64_t lo = isect_offsets[ct];
    const int64_t hi = (ct == total_tiles - 1) ? n_isects : isect_offsets[ct + 1];
    if (hi <= lo) continue;
    const int64_t L = hi - lo;
    /* per-tile accumulators: [L][2+2+3+D+1] */
    const int S = 2 + 2 + 3 + D + 1;
    REAL *acc = (REAL *)calloc((size_t)L * S, sizeof(REAL));
    for (int i = ty * tile_size; i < (ty + 1) * tile_size && i < H; ++i) {
      for (int j = tx * tile_size; j < (tx + 1) * tile_size && j < W; ++j) {
        const int64_t p = ((int64_t)c * H + i) * W + j;
        const REAL px = (REAL)j + (REAL)0.5, py = (REAL)i + (REAL)0.5;
        const REAL T_final = 1 - render_alphas[p];
        REAL T = T_final, buffer[MAX_D] = {0};
        const REAL *v_c = v_render_colors + p * D;
        const REAL v_a = v_render_alphas[p];
        REAL bg_dot = 0;
        if (backgrounds)
          for (int d = 0; d < D; ++d) bg_dot += backgrounds[c * D + d] * v_c[d];
        for (int64_t k = last_ids[p]; k >= lo; --k) {
          const int32_t g = flatten_ids[k];
          const REAL dx = nearest_copy(means2d[2 * g], tile_cx, W, periodic) - px, dy = means2d[2 * g + 1] - py;
          const REAL a = conics[3 * g], b = conics[3 * g + 1], cc = conics[3 * g + 2];
          const REAL sigma = (REAL)0.5 * (a * dx * dx + cc * dy * dy) + b * dx * dy;
          const REAL vis = real_exp(-sigma);
          const REAL opac = opacities[g];
          REAL alpha = opac * vis;
          if (alpha > ALPHA_MAX) alpha = ALPHA_MAX;
          if (sigma < 0 || alpha < ALPHA_MIN) continue;
          REAL *A = acc + (k - lo) * S;
          const REAL ra = 1 / (1 - alpha);
          T *= ra;
          const REAL fac = alpha * T;
          REAL v_alpha = 0;
          for (int d = 0; d < D; ++d) {
            const REAL cd = colors[(int64_t)g * D + d];
            A[7 + d] += fac * v_c[d];
            v_alpha += (cd * T - buffer[d] * ra) * v_c[d];
            buffer[d] += cd * fac;
          }
          v_alpha += T_final * ra * v_a;
          v_alpha -= T_final * ra * bg_dot;
          if (opac * vis <= ALPHA_MAX) {
            const REAL v_sigma = -opac * vis * v_alpha;
            const REAL gx = v_sigma * (a * dx + b * dy), gy = v_sigma * (b * dx + cc * dy);
            A[0] += gx;
            A[1] += gy;
            A[2] += gx < 0 ? -gx : gx;
            A[3] += gy < 0 ? -gy : gy;
            A[4] += (REAL)0.5 * v_sigma * dx * dx;
            A[5] += v_sigma * dx * dy;
            A[6] += (REAL)0.5 * v_sigma * dy * dy;
            A[7 + D] += vis * v_alpha;
          }
        }
      }
    }
    for (int64_t k = 0; k < L; ++k) {
      const int32_t g = flatten_ids[lo + k];
      const REAL *A = acc + k * S;
#define ATOMIC_ADD(dst, v) do { REAL _v = (v); if (_v != 0) { _Pragma("omp atomic") (dst) += _v; } } while (0)
      ATOMIC_ADD(v_means2d[2 * g], A[0]);
      ATOMIC_ADD(v_means2d[2 * g + 1], A[1]);
      if (v_means2d_abs) {
        ATOMIC_ADD(v_means2d_abs[2 * g], A[2]);
        ATOMIC_ADD(v_means2d_abs[2 * g + 1], A[3]);
      }
      ATOMIC_ADD(v_conics[3 * g], A[4]);
      ATOMIC_ADD(v_conics[3 * g + 1], A[5]);
      ATOMIC_ADD(v_conics[3 * g + 2], A[6]);
      for (int d = 0; d < D; ++d) ATOMIC_ADD(v_colors[(int64_t)g * D + d], A[7 + d]);
      ATOMIC_ADD(v_opacities[g], A[7 + D]);
    }
    free(acc);
  }
  return 0;
}
