// projection.hip -- K1/K2: 3D->2D EWA projection forward / backward for gfx950.
//
// Replaces gsplat `fully_fused_projection` (reached from
// /root/reference/utils/gsplat_utils/gsplat_trainer.py:477).  HBM-bound: 40 B read + 28 B written
// per (camera, Gaussian) forward.  One lane per (camera, Gaussian) forward; one lane per Gaussian
// (looping over cameras) backward, so the per-Gaussian gradients need no atomics.
#include "so_common.hpp"
#include "splat_math.hpp"

namespace so {

struct CamParams {  // world->camera rotation/translation + intrinsics, per camera
  float Rw[9];
  float tw[3];
  float fx, fy, cx, cy;
};

__device__ __forceinline__ CamParams load_cam(const float *__restrict__ viewmats, const float *__restrict__ Ks, int c) {
  CamParams p;
  const float *V = viewmats + 16 * c;
  p.Rw[0] = V[0]; p.Rw[1] = V[1]; p.Rw[2] = V[2];
  p.Rw[3] = V[4]; p.Rw[4] = V[5]; p.Rw[5] = V[6];
  p.Rw[6] = V[8]; p.Rw[7] = V[9]; p.Rw[8] = V[10];
  p.tw[0] = V[3]; p.tw[1] = V[7]; p.tw[2] = V[11];
  const float *K = Ks + 9 * c;
  p.fx = K[0]; p.fy = K[4]; p.cx = K[2]; p.cy = K[5];
  return p;
}

template <bool HAS_COV>
__global__ void __launch_bounds__(256)
k_projection_fwd(int C, int N, const float *__restrict__ means, const float *__restrict__ covars6,
                 const float *__restrict__ quats, const float *__restrict__ scales,
                 const float *__restrict__ viewmats, const float *__restrict__ Ks, int W, int H, float eps2d,
                 float near_plane, float far_plane, float radius_clip, int model, int32_t *__restrict__ radii,
                 float *__restrict__ means2d, float *__restrict__ depths, float *__restrict__ conics,
                 float *__restrict__ comps) {
  const int64_t total = (int64_t)C * N;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx / N);
    const int n = (int)(idx - (int64_t)c * N);
    const CamParams cam = load_cam(viewmats, Ks, c);
    float mean[3] = {means[3 * n], means[3 * n + 1], means[3 * n + 2]};
    float cov6[6], q[4], s[3];
    if (HAS_COV) {
#pragma unroll
      for (int k = 0; k < 6; ++k) cov6[k] = covars6[6 * (int64_t)n + k];
    } else {
      const float4 qq = *reinterpret_cast<const float4 *>(quats + 4 * (int64_t)n);
      q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
      s[0] = scales[3 * n]; s[1] = scales[3 * n + 1]; s[2] = scales[3 * n + 2];
    }
    ProjOut<float> o;
    project_fwd<float>(mean, HAS_COV ? cov6 : nullptr, q, s, cam.Rw, cam.tw, cam.fx, cam.fy, cam.cx, cam.cy, W, H,
                       eps2d, near_plane, far_plane, radius_clip, model, o);
    radii[idx] = o.radius;
    *reinterpret_cast<float2 *>(means2d + 2 * idx) = make_float2(o.m2d[0], o.m2d[1]);
    depths[idx] = o.depth;
    conics[3 * idx] = o.conic[0];
    conics[3 * idx + 1] = o.conic[1];
    conics[3 * idx + 2] = o.conic[2];
    if (comps) comps[idx] = o.comp;
  }
}

template <bool HAS_COV, bool HAS_VIEW>
__global__ void __launch_bounds__(256)
k_projection_bwd(int C, int N, const float *__restrict__ means, const float *__restrict__ covars6,
                 const float *__restrict__ quats, const float *__restrict__ scales,
                 const float *__restrict__ viewmats, const float *__restrict__ Ks, int W, int H, float eps2d,
                 int model, const int32_t *__restrict__ radii, const float *__restrict__ v_means2d,
                 const float *__restrict__ v_depths, const float *__restrict__ v_conics,
                 const float *__restrict__ v_comps, float *__restrict__ v_means, float *__restrict__ v_covars6,
                 float *__restrict__ v_quats, float *__restrict__ v_scales, float *__restrict__ v_viewmats) {
  // grid-stride over Gaussians; every lane stays in the loop to the same trip count so that the
  // wave reductions for v_viewmats see EXEC all ones.
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n_iter = (N + stride - 1) / stride;
  for (int64_t it = 0; it < n_iter; ++it) {
    const int64_t n = it * stride + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = n < N;
    float mean[3] = {0, 0, 0}, cov6[6] = {1, 0, 0, 1, 0, 1}, q[4] = {1, 0, 0, 0}, s[3] = {1, 1, 1};
    if (live) {
      mean[0] = means[3 * n]; mean[1] = means[3 * n + 1]; mean[2] = means[3 * n + 2];
      if (HAS_COV) {
#pragma unroll
        for (int k = 0; k < 6; ++k) cov6[k] = covars6[6 * n + k];
      } else {
        const float4 qq = *reinterpret_cast<const float4 *>(quats + 4 * n);
        q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
        s[0] = scales[3 * n]; s[1] = scales[3 * n + 1]; s[2] = scales[3 * n + 2];
      }
    }
    float vm[3] = {0, 0, 0}, vc6[6] = {0, 0, 0, 0, 0, 0}, vq[4] = {0, 0, 0, 0}, vs[3] = {0, 0, 0};
    for (int c = 0; c < C; ++c) {
      const int64_t idx = (int64_t)c * N + n;
      const bool vis = live && radii[idx] > 0;
      float vR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, vt[3] = {0, 0, 0};
      if (vis) {
        const CamParams cam = load_cam(viewmats, Ks, c);
        const float2 vm2 = *reinterpret_cast<const float2 *>(v_means2d + 2 * idx);
        const float v_m2d[2] = {vm2.x, vm2.y};
        const float v_con[3] = {v_conics[3 * idx], v_conics[3 * idx + 1], v_conics[3 * idx + 2]};
        project_bwd<float>(mean, HAS_COV ? cov6 : nullptr, q, s, cam.Rw, cam.tw, cam.fx, cam.fy, cam.cx, cam.cy, W,
                           H, eps2d, model, v_m2d, v_depths ? v_depths[idx] : 0.f, v_con,
                           v_comps ? v_comps[idx] : 0.f, vm, vc6, vq, vs, HAS_VIEW ? vR : nullptr, vt);
      }
      if (HAS_VIEW) {
        float *o = v_viewmats + 16 * c;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const float r = wave_reduce_sum(vR[3 * i + j]);
            if (lane_id() == 0 && r != 0.f) atomicAdd(o + 4 * i + j, r);
          }
          const float r = wave_reduce_sum(vt[i]);
          if (lane_id() == 0 && r != 0.f) atomicAdd(o + 4 * i + 3, r);
        }
      }
    }
    if (live) {
      v_means[3 * n] = vm[0]; v_means[3 * n + 1] = vm[1]; v_means[3 * n + 2] = vm[2];
      if (HAS_COV) {
#pragma unroll
        for (int k = 0; k < 6; ++k) v_covars6[6 * n + k] = vc6[k];
      } else {
        *reinterpret_cast<float4 *>(v_quats + 4 * n) = make_float4(vq[0], vq[1], vq[2], vq[3]);
        v_scales[3 * n] = vs[0]; v_scales[3 * n + 1] = vs[1]; v_scales[3 * n + 2] = vs[2];
      }
    }
  }
}

static inline int grid_for(int64_t total, int block) {
  int64_t g = ceil_div(total, block);
  const int64_t cap = 256 * 16;  // 256 CUs x 16 blocks, grid-stride beyond
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace so

extern "C" int so_projection_fwd(int C, int N, const float *means, const float *covars6, const float *quats,
                                 const float *scales, const float *viewmats, const float *Ks, int width,
                                 int height, float eps2d, float near_plane, float far_plane, float radius_clip,
                                 int camera_model, int32_t *radii, float *means2d, float *depths, float *conics,
                                 float *compensations, void *stream) {
  SO_REQUIRE(C >= 0 && N >= 0 && width > 0 && height > 0, "so_projection_fwd: bad sizes C=%d N=%d %dx%d", C, N, width, height);
  if (camera_model < 0 || camera_model > SO_CAM_SPHERICAL) {
    so::set_error("so_projection_fwd: unsupported camera_model %d", camera_model);
    return SO_ERR_UNSUPPORTED;
  }
  if ((int64_t)C * N == 0) return SO_OK;
  SO_REQUIRE(means && viewmats && Ks && radii && means2d && depths && conics, "so_projection_fwd: null pointer");
  SO_REQUIRE(covars6 || (quats && scales), "so_projection_fwd: need covars6 or quats+scales");
  auto kern = covars6 ? so::k_projection_fwd<true> : so::k_projection_fwd<false>;
  hipLaunchKernelGGL(kern, dim3(so::grid_for((int64_t)C * N, 256)), dim3(256), 0,
                     so::as_stream(stream), C, N, means, covars6, quats, scales, viewmats, Ks, width, height, eps2d,
                     near_plane, far_plane, radius_clip, camera_model, radii, means2d, depths, conics, compensations);
  return so::check_launch("so_projection_fwd");
}

extern "C" int so_projection_bwd(int C, int N, const float *means, const float *covars6, const float *quats,
                                 const float *scales, const float *viewmats, const float *Ks, int width,
                                 int height, float eps2d, int camera_model, const int32_t *radii,
                                 const float *v_means2d, const float *v_depths, const float *v_conics,
                                 const float *v_compensations, float *v_means, float *v_covars6, float *v_quats,
                                 float *v_scales, float *v_viewmats, void *stream) {
  SO_REQUIRE(C >= 0 && N >= 0 && width > 0 && height > 0, "so_projection_bwd: bad sizes");
  if (camera_model < 0 || camera_model > SO_CAM_SPHERICAL) {
    so::set_error("so_projection_bwd: unsupported camera_model %d", camera_model);
    return SO_ERR_UNSUPPORTED;
  }
  if (N == 0) return SO_OK;
  SO_REQUIRE(means && viewmats && Ks && radii && v_means2d && v_conics && v_means, "so_projection_bwd: null pointer");
  SO_REQUIRE(covars6 ? (v_covars6 != nullptr) : (quats && scales && v_quats && v_scales),
             "so_projection_bwd: missing covars6/quats/scales buffers");
  auto kern = covars6 ? (v_viewmats ? so::k_projection_bwd<true, true> : so::k_projection_bwd<true, false>)
                      : (v_viewmats ? so::k_projection_bwd<false, true> : so::k_projection_bwd<false, false>);
  hipLaunchKernelGGL(kern, dim3(so::grid_for(N, 256)), dim3(256), 0, so::as_stream(stream), C, N,
                     means, covars6, quats, scales, viewmats, Ks, width, height, eps2d, camera_model, radii,
                     v_means2d, v_depths, v_conics, v_compensations, v_means, v_covars6, v_quats, v_scales,
                     v_viewmats);
  return so::check_launch("so_projection_bwd");
}
