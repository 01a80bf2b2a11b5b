// sh.hip -- K4/K5: real spherical harmonics colour evaluation, forward / backward, for gfx950.
//
// Replaces gsplat `spherical_harmonics` (reached with sh_degree=... from
// /root/reference/utils/gsplat_utils/gsplat_trainer.py:591 via `rasterization`).  HBM-bound: the
// coefficient block (12*K B per Gaussian, 192 B at K=16) is the largest per-Gaussian read of the
// path.  Forward: one lane per (camera, Gaussian).  Backward: one lane per Gaussian looping over
// cameras, so v_coeffs for shared coefficients is a plain store (no atomics, reproducible).
#include "so_common.hpp"
#include "splat_math.hpp"

namespace so {

// VIEW: the colour stage of `rasterization` in one launch -- dirs = means[n] - campos[c] (the `dirs` argument then holds
// means[N,3]), visibility from radii[C,N] > 0 (through `masks`, reinterpreted), and the "+ 0.5, clamp at 0" epilogue.
template <int DEG, bool VIEW>
__global__ void __launch_bounds__(256)
k_sh_fwd(int C, int N, int K, const float *__restrict__ dirs, const float *__restrict__ coeffs, int per_camera,
         const uint8_t *__restrict__ masks, float *__restrict__ colors, const float *__restrict__ campos) {
  const int64_t total = (int64_t)C * N;
  const int32_t *radii = reinterpret_cast<const int32_t *>(masks);
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    float r = 0.f, g = 0.f, b = 0.f;
    if (!masks || (VIEW ? radii[idx] > 0 : masks[idx] != 0)) {
      const int64_t n = idx % N;
      const float *cf = coeffs + (per_camera ? idx : n) * (int64_t)K * 3;
      float x, y, z;
      if (VIEW) {
        const int64_t c = idx / N;
        x = dirs[3 * n] - campos[3 * c]; y = dirs[3 * n + 1] - campos[3 * c + 1]; z = dirs[3 * n + 2] - campos[3 * c + 2];
      } else {
        x = dirs[3 * idx]; y = dirs[3 * idx + 1]; z = dirs[3 * idx + 2];
      }
      const float inorm = rsqrtf(x * x + y * y + z * z);
      x *= inorm; y *= inorm; z *= inorm;
      sh_eval<float>(DEG, x, y, z, [&](int k, float yk, float, float, float) {
        r += yk * cf[3 * k];
        g += yk * cf[3 * k + 1];
        b += yk * cf[3 * k + 2];
      });
    }
    if (VIEW) { r = fmaxf(r + 0.5f, 0.f); g = fmaxf(g + 0.5f, 0.f); b = fmaxf(b + 0.5f, 0.f); }
    colors[3 * idx] = r;
    colors[3 * idx + 1] = g;
    colors[3 * idx + 2] = b;
  }
}

template <int DEG, bool HAS_VDIRS, bool VIEW>
__global__ void __launch_bounds__(256)
k_sh_bwd(int C, int N, int K, const float *__restrict__ dirs, const float *__restrict__ coeffs, int per_camera,
         const uint8_t *__restrict__ masks, const float *__restrict__ v_colors, float *__restrict__ v_coeffs,
         float *__restrict__ v_dirs, const float *__restrict__ campos, const float *__restrict__ colors_out) {
  // VIEW (see k_sh_fwd): dirs = means[N,3]; v_dirs = v_means[N,3], the sum over the cameras; the upstream gradient
  // passes the clamp where the forward's output is positive (colors_out)
  constexpr int NB = (DEG + 1) * (DEG + 1);
  const int32_t *radii = reinterpret_cast<const int32_t *>(masks);
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
    float acc[NB][3];
    float vm[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NB; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.f;
    for (int c = 0; c < C; ++c) {
      const int64_t idx = (int64_t)c * N + n;
      const bool on = !masks || (VIEW ? radii[idx] > 0 : masks[idx] != 0);
      float vd[3] = {0.f, 0.f, 0.f};
      if (on) {
        float vr = v_colors[3 * idx], vg = v_colors[3 * idx + 1], vb = v_colors[3 * idx + 2];
        float dx, dy, dz;
        if (VIEW) {
          if (!(colors_out[3 * idx] > 0.f)) vr = 0.f;
          if (!(colors_out[3 * idx + 1] > 0.f)) vg = 0.f;
          if (!(colors_out[3 * idx + 2] > 0.f)) vb = 0.f;
          dx = dirs[3 * n] - campos[3 * c]; dy = dirs[3 * n + 1] - campos[3 * c + 1]; dz = dirs[3 * n + 2] - campos[3 * c + 2];
        } else {
          dx = dirs[3 * idx]; dy = dirs[3 * idx + 1]; dz = dirs[3 * idx + 2];
        }
        const float inorm = rsqrtf(dx * dx + dy * dy + dz * dz);
        const float x = dx * inorm, y = dy * inorm, z = dz * inorm;
        const float *cf = coeffs + (per_camera ? idx : n) * (int64_t)K * 3;
        float vdn[3] = {0.f, 0.f, 0.f};
        sh_eval<float>(DEG, x, y, z, [&](int k, float yk, float ddx, float ddy, float ddz) {
          acc[k][0] += yk * vr;
          acc[k][1] += yk * vg;
          acc[k][2] += yk * vb;
          if (HAS_VDIRS) {
            const float w = cf[3 * k] * vr + cf[3 * k + 1] * vg + cf[3 * k + 2] * vb;
            vdn[0] += ddx * w;
            vdn[1] += ddy * w;
            vdn[2] += ddz * w;
          }
        });
        if (HAS_VDIRS) {  // through d_n = d/|d|
          const float dot = vdn[0] * x + vdn[1] * y + vdn[2] * z;
          vd[0] = (vdn[0] - dot * x) * inorm;
          vd[1] = (vdn[1] - dot * y) * inorm;
          vd[2] = (vdn[2] - dot * z) * inorm;
        }
      }
      if (VIEW) {
        vm[0] += vd[0]; vm[1] += vd[1]; vm[2] += vd[2];
      } else if (HAS_VDIRS) {
        v_dirs[3 * idx] = vd[0];
        v_dirs[3 * idx + 1] = vd[1];
        v_dirs[3 * idx + 2] = vd[2];
      }
      if (per_camera) {
        float *o = v_coeffs + idx * (int64_t)K * 3;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          o[3 * k] = acc[k][0]; o[3 * k + 1] = acc[k][1]; o[3 * k + 2] = acc[k][2];
          acc[k][0] = acc[k][1] = acc[k][2] = 0.f;
        }
        for (int k = 3 * NB; k < 3 * K; ++k) o[k] = 0.f;
      }
    }
    if (!per_camera) {
      float *o = v_coeffs + n * (int64_t)K * 3;
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        o[3 * k] = acc[k][0]; o[3 * k + 1] = acc[k][1]; o[3 * k + 2] = acc[k][2];
      }
      for (int k = 3 * NB; k < 3 * K; ++k) o[k] = 0.f;
    }
    if (VIEW) { v_dirs[3 * n] = vm[0]; v_dirs[3 * n + 1] = vm[1]; v_dirs[3 * n + 2] = vm[2]; }
  }
}

static inline int sh_grid(int64_t total) {
  int64_t g = ceil_div(total, 256);
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace so

extern "C" int so_sh_fwd(int C, int N, int K, int degrees_to_use, const float *dirs, const float *coeffs,
                         int coeffs_per_camera, const uint8_t *masks, float *colors, void *stream) {
  SO_REQUIRE(C >= 0 && N >= 0 && K >= 1, "so_sh_fwd: bad sizes");
  SO_REQUIRE(degrees_to_use >= 0 && degrees_to_use <= 4, "so_sh_fwd: degrees_to_use %d not in [0,4]", degrees_to_use);
  SO_REQUIRE((degrees_to_use + 1) * (degrees_to_use + 1) <= K, "so_sh_fwd: K=%d too small for degree %d", K, degrees_to_use);
  if ((int64_t)C * N == 0) return SO_OK;
  SO_REQUIRE(dirs && coeffs && colors, "so_sh_fwd: null pointer");
  const dim3 grid(so::sh_grid((int64_t)C * N)), block(256);
  hipStream_t st = so::as_stream(stream);
#define SO_LAUNCH(D) hipLaunchKernelGGL((so::k_sh_fwd<D, false>), grid, block, 0, st, C, N, K, dirs, coeffs, coeffs_per_camera, masks, colors, nullptr)
  switch (degrees_to_use) {
    case 0: SO_LAUNCH(0); break;
    case 1: SO_LAUNCH(1); break;
    case 2: SO_LAUNCH(2); break;
    case 3: SO_LAUNCH(3); break;
    default: SO_LAUNCH(4); break;
  }
#undef SO_LAUNCH
  return so::check_launch("so_sh_fwd");
}

extern "C" int so_sh_bwd(int C, int N, int K, int degrees_to_use, const float *dirs, const float *coeffs,
                         int coeffs_per_camera, const uint8_t *masks, const float *v_colors, float *v_coeffs,
                         float *v_dirs, void *stream) {
  SO_REQUIRE(C >= 0 && N >= 0 && K >= 1, "so_sh_bwd: bad sizes");
  SO_REQUIRE(degrees_to_use >= 0 && degrees_to_use <= 4, "so_sh_bwd: degrees_to_use %d not in [0,4]", degrees_to_use);
  SO_REQUIRE((degrees_to_use + 1) * (degrees_to_use + 1) <= K, "so_sh_bwd: K=%d too small for degree %d", K, degrees_to_use);
  if (N == 0) return SO_OK;
  SO_REQUIRE(dirs && coeffs && v_colors && v_coeffs, "so_sh_bwd: null pointer");
  const dim3 grid(so::sh_grid(N)), block(256);
  hipStream_t st = so::as_stream(stream);
#define SO_LAUNCH(D) hipLaunchKernelGGL((v_dirs ? so::k_sh_bwd<D, true, false> : so::k_sh_bwd<D, false, false>), grid, block, 0, st, C, N, K, dirs, coeffs, coeffs_per_camera, masks, v_colors, v_coeffs, v_dirs, nullptr, nullptr)
  switch (degrees_to_use) {
    case 0: SO_LAUNCH(0); break;
    case 1: SO_LAUNCH(1); break;
    case 2: SO_LAUNCH(2); break;
    case 3: SO_LAUNCH(3); break;
    default: SO_LAUNCH(4); break;
  }
#undef SO_LAUNCH
  return so::check_launch("so_sh_bwd");
}

// The colour stage of `rasterization` (gsplat_trainer.py:477-494 with sh_degree) in one launch each way:
//   colors[c,n] = max(SH(normalise(means[n] - campos[c])) . coeffs[n] + 0.5, 0)   (0.5 where radii[c,n] <= 0)
extern "C" int so_sh_view_colors_fwd(int C, int N, int K, int degrees_to_use, const float *means, const float *campos,
                                     const float *coeffs, const int32_t *radii, float *colors, void *stream) {
  SO_REQUIRE(C >= 0 && N >= 0 && K >= 1, "so_sh_view_colors_fwd: bad sizes");
  SO_REQUIRE(degrees_to_use >= 0 && degrees_to_use <= 4 && (degrees_to_use + 1) * (degrees_to_use + 1) <= K,
             "so_sh_view_colors_fwd: degree %d / K %d", degrees_to_use, K);
  if ((int64_t)C * N == 0) return SO_OK;
  SO_REQUIRE(means && campos && coeffs && colors, "so_sh_view_colors_fwd: null pointer");
  const dim3 grid(so::sh_grid((int64_t)C * N)), block(256);
  hipStream_t st = so::as_stream(stream);
  const uint8_t *m = reinterpret_cast<const uint8_t *>(radii);
#define SO_LAUNCH(D) hipLaunchKernelGGL((so::k_sh_fwd<D, true>), grid, block, 0, st, C, N, K, means, coeffs, 0, m, colors, campos)
  switch (degrees_to_use) {
    case 0: SO_LAUNCH(0); break;
    case 1: SO_LAUNCH(1); break;
    case 2: SO_LAUNCH(2); break;
    case 3: SO_LAUNCH(3); break;
    default: SO_LAUNCH(4); break;
  }
#undef SO_LAUNCH
  return so::check_launch("so_sh_view_colors_fwd");
}

// v_coeffs[N,K,3] and v_means[N,3] are OVERWRITTEN (sums over the cameras); `colors` is the forward's output (the clamp).
extern "C" int so_sh_view_colors_bwd(int C, int N, int K, int degrees_to_use, const float *means, const float *campos,
                                     const float *coeffs, const int32_t *radii, const float *colors, const float *v_colors,
                                     float *v_coeffs, float *v_means, void *stream) {
  SO_REQUIRE(C >= 0 && N >= 0 && K >= 1, "so_sh_view_colors_bwd: bad sizes");
  SO_REQUIRE(degrees_to_use >= 0 && degrees_to_use <= 4 && (degrees_to_use + 1) * (degrees_to_use + 1) <= K,
             "so_sh_view_colors_bwd: degree %d / K %d", degrees_to_use, K);
  if (N == 0) return SO_OK;
  SO_REQUIRE(means && campos && coeffs && colors && v_colors && v_coeffs && v_means, "so_sh_view_colors_bwd: null pointer");
  const dim3 grid(so::sh_grid(N)), block(256);
  hipStream_t st = so::as_stream(stream);
  const uint8_t *m = reinterpret_cast<const uint8_t *>(radii);
#define SO_LAUNCH(D) hipLaunchKernelGGL((so::k_sh_bwd<D, true, true>), grid, block, 0, st, C, N, K, means, coeffs, 0, m, v_colors, v_coeffs, v_means, campos, colors)
  switch (degrees_to_use) {
    case 0: SO_LAUNCH(0); break;
    case 1: SO_LAUNCH(1); break;
    case 2: SO_LAUNCH(2); break;
    case 3: SO_LAUNCH(3); break;
    default: SO_LAUNCH(4); break;
  }
#undef SO_LAUNCH
  return so::check_launch("so_sh_view_colors_bwd");
}
