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

// ---------------------------------------------------------------------------------------------
// Packed projection (gsplat `fully_fused_projection(packed=True)`, K3; `cfg.packed` at
// /root/reference/utils/gsplat_utils/gsplat_trainer.py:133, :487): only the (camera, Gaussian) pairs with a positive
// radius get a row, camera-major -- and, as in gsplat, in ASCENDING flattened index, because the reference builds
// coalesced sparse gradients from `gaussian_ids` (:705-717).  Two passes over the raw inputs and NO [C,N] array:
//   pass 0  every workgroup projects its 1024 consecutive pairs and counts the visible ones (wave ballots)
//   scan    exclusive prefix of the workgroup counts (one workgroup) + the total
//   pass 1  projects again and writes each visible pair at offset[workgroup] + its ballot prefix inside the workgroup
// The projection is recomputed rather than parked in a dense scratch: 40 B of inputs per pair against 36 B of outputs.
// ---------------------------------------------------------------------------------------------
constexpr int kPackItems = 4, kPackChunk = 256 * kPackItems;

template <bool HAS_COV>
__device__ __forceinline__ void project_pair(int64_t idx, int N, const float *__restrict__ means,
                                             const float *__restrict__ covars6, const float *__restrict__ quats,
                                             const float *__restrict__ scales, const float *__restrict__ viewmats,
                                             const float *__restrict__ Ks, int W, int H, float eps2d, float near_plane,
                                             float far_plane, float radius_clip, int model, int &c, int &n, ProjOut<float> &o) {
  c = (int)(idx / N);
  n = (int)(idx - (int64_t)c * N);
  const CamParams cam = load_cam(viewmats, Ks, c);
  float mean[3] = {means[3 * (int64_t)n], means[3 * (int64_t)n + 1], means[3 * (int64_t)n + 2]};
  float cov6[6], q[4], s[3];
  if (HAS_COV) {
#pragma unroll
    for (int k = 0; k < 6; ++k) cov6[k] = covars6[6 * (int64_t)n + k];
  } else {
    const float4 qq = *reinterpret_cast<const float4 *>(quats + 4 * (int64_t)n);
    q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
    s[0] = scales[3 * (int64_t)n]; s[1] = scales[3 * (int64_t)n + 1]; s[2] = scales[3 * (int64_t)n + 2];
  }
  project_fwd<float>(mean, HAS_COV ? cov6 : nullptr, q, s, cam.Rw, cam.tw, cam.fx, cam.fy, cam.cx, cam.cy, W, H, eps2d,
                     near_plane, far_plane, radius_clip, model, o);
}

template <bool HAS_COV, bool WRITE>
__global__ void __launch_bounds__(256)
k_projection_packed(int C, int N, const float *__restrict__ means, const float *__restrict__ covars6,
                    const float *__restrict__ quats, const float *__restrict__ scales, const float *__restrict__ viewmats,
                    const float *__restrict__ Ks, int W, int H, float eps2d, float near_plane, float far_plane,
                    float radius_clip, int model, int32_t *__restrict__ block_counts, uint64_t *__restrict__ vis_masks,
                    const int64_t *__restrict__ block_offsets, int64_t *__restrict__ camera_ids,
                    int64_t *__restrict__ gaussian_ids, int32_t *__restrict__ radii, float *__restrict__ means2d,
                    float *__restrict__ depths, float *__restrict__ conics, float *__restrict__ comps) {
  __shared__ int32_t s_cnt[kPackItems * 4];
  const int64_t total = (int64_t)C * N;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  ProjOut<float> o[kPackItems];
  int cc[kPackItems], nn[kPackItems], rank[kPackItems];
  bool vis[kPackItems];
#pragma unroll
  for (int it = 0; it < kPackItems; ++it) {
    const int64_t idx = (int64_t)blockIdx.x * kPackChunk + it * 256 + threadIdx.x;
    o[it].radius = 0;
    cc[it] = nn[it] = 0;
    if (idx < total)
      project_pair<HAS_COV>(idx, N, means, covars6, quats, scales, viewmats, Ks, W, H, eps2d, near_plane, far_plane,
                            radius_clip, model, cc[it], nn[it], o[it]);
    // WHICH pairs are rows is decided once, by the counting pass (it sized the outputs): the two instantiations of this
    // kernel are optimised differently and need not round a borderline radius / depth test alike
    unsigned long long m;
    const int64_t word = ((int64_t)blockIdx.x * kPackItems + it) * 4 + wv;
    if (WRITE) {
      m = vis_masks[word];
    } else {
      m = __ballot(o[it].radius > 0);
      if (lane == 0) vis_masks[word] = m;
    }
    vis[it] = (m >> lane) & 1ull;
    rank[it] = __popcll(m & lt);
    if (lane == 0) s_cnt[it * 4 + wv] = __popcll(m);
  }
  __syncthreads();
  if (!WRITE) {
    if (threadIdx.x == 0) {
      int32_t t = 0;
#pragma unroll
      for (int e = 0; e < kPackItems * 4; ++e) t += s_cnt[e];
      block_counts[blockIdx.x] = t;
    }
    return;
  }
  const int64_t base = block_offsets[blockIdx.x];
#pragma unroll
  for (int it = 0; it < kPackItems; ++it) {
    if (!vis[it]) continue;
    int32_t before = 0;
    for (int e = 0; e < it * 4 + wv; ++e) before += s_cnt[e];
    const int64_t r = base + before + rank[it];
    camera_ids[r] = cc[it];
    gaussian_ids[r] = nn[it];
    radii[r] = o[it].radius > 0 ? o[it].radius : 0;     // (a row this pass would not have kept: radius 0, skipped downstream)
    *reinterpret_cast<float2 *>(means2d + 2 * r) = make_float2(o[it].m2d[0], o[it].m2d[1]);
    depths[r] = o[it].depth;
    conics[3 * r] = o[it].conic[0]; conics[3 * r + 1] = o[it].conic[1]; conics[3 * r + 2] = o[it].conic[2];
    if (comps) comps[r] = o[it].comp;
  }
}

// exclusive scan of n int32 counts into int64 offsets (one workgroup; n workgroup counts of a projection pass)
__global__ void __launch_bounds__(1024)
k_scan_counts(int64_t n, const int32_t *__restrict__ counts, int64_t *__restrict__ offsets, int64_t *__restrict__ total) {
  __shared__ int64_t s[1024];
  const int t = threadIdx.x;
  const int64_t per = (n + 1023) / 1024, b0 = t * per, b1 = (b0 + per < n) ? b0 + per : n;
  int64_t loc = 0;
  for (int64_t b = b0; b < b1; ++b) loc += counts[b];
  s[t] = loc;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int64_t add = t >= d ? s[t - d] : 0;
    __syncthreads();
    s[t] += add;
    __syncthreads();
  }
  int64_t run = s[t] - loc;
  for (int64_t b = b0; b < b1; ++b) { offsets[b] = run; run += counts[b]; }
  if (t == 1023) *total = s[t];
}

// Backward of the packed projection: one lane per packed row; rows of one Gaussian (seen by several cameras) are far
// apart, so the per-Gaussian gradients are float atomics into zero-initialised outputs (no conflict at all when C == 1).
template <bool HAS_COV, bool HAS_VIEW>
__global__ void __launch_bounds__(256)
k_projection_bwd_packed(int C, int N, int64_t nnz, const float *__restrict__ means, const float *__restrict__ covars6,
                        const float *__restrict__ quats, const float *__restrict__ scales, const float *__restrict__ viewmats,
                        const float *__restrict__ Ks, int W, int H, float eps2d, int model,
                        const int64_t *__restrict__ camera_ids, const int64_t *__restrict__ gaussian_ids,
                        const float *__restrict__ v_means2d, const float *__restrict__ v_depths,
                        const float *__restrict__ v_conics, const float *__restrict__ v_comps, float *__restrict__ v_means,
                        float *__restrict__ v_covars6, float *__restrict__ v_quats, float *__restrict__ v_scales,
                        float *__restrict__ v_viewmats) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n_iter = (nnz + stride - 1) / stride;
  for (int64_t it = 0; it < n_iter; ++it) {
    const int64_t r = it * stride + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = r < nnz;
    int c = -1;
    float vR[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, vt[3] = {0, 0, 0};
    if (live) {
      c = (int)camera_ids[r];
      const int64_t n = gaussian_ids[r];
      const CamParams cam = load_cam(viewmats, Ks, c);
      float mean[3] = {means[3 * n], means[3 * n + 1], means[3 * n + 2]};
      float cov6[6] = {1, 0, 0, 1, 0, 1}, q[4] = {1, 0, 0, 0}, s[3] = {1, 1, 1};
      if (HAS_COV) {
#pragma unroll
        for (int k = 0; k < 6; ++k) cov6[k] = covars6[6 * n + k];
      } else {
        const float4 qq = *reinterpret_cast<const float4 *>(quats + 4 * n);
        q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
        s[0] = scales[3 * n]; s[1] = scales[3 * n + 1]; s[2] = scales[3 * n + 2];
      }
      const float2 vm2 = *reinterpret_cast<const float2 *>(v_means2d + 2 * r);
      const float v_m2d[2] = {vm2.x, vm2.y};
      const float v_con[3] = {v_conics[3 * r], v_conics[3 * r + 1], v_conics[3 * r + 2]};
      float vm[3] = {0, 0, 0}, vc6[6] = {0, 0, 0, 0, 0, 0}, vq[4] = {0, 0, 0, 0}, vs[3] = {0, 0, 0};
      project_bwd<float>(mean, HAS_COV ? cov6 : nullptr, q, s, cam.Rw, cam.tw, cam.fx, cam.fy, cam.cx, cam.cy, W, H, eps2d,
                         model, v_m2d, v_depths ? v_depths[r] : 0.f, v_con, v_comps ? v_comps[r] : 0.f, vm, vc6, vq, vs,
                         HAS_VIEW ? vR : nullptr, vt);
#pragma unroll
      for (int k = 0; k < 3; ++k) atomicAdd(v_means + 3 * n + k, vm[k]);
      if (HAS_COV) {
#pragma unroll
        for (int k = 0; k < 6; ++k) atomicAdd(v_covars6 + 6 * n + k, vc6[k]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(v_quats + 4 * n + k, vq[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) atomicAdd(v_scales + 3 * n + k, vs[k]);
      }
    }
    if (HAS_VIEW) {   // rows are camera-major: a wave spans one camera, two at a boundary -- reduce per camera present
      int cmin = live ? c : C, cmax = live ? c : -1;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        cmin = min(cmin, __shfl_xor(cmin, d, 64));
        cmax = max(cmax, __shfl_xor(cmax, d, 64));
      }
      for (int cam = cmin; cam <= cmax; ++cam) {
        const bool mine = live && c == cam;
        float *o = v_viewmats + 16 * cam;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const float rr = wave_reduce_sum(mine ? vR[3 * i + j] : 0.f);
            if (lane_id() == 0 && rr != 0.f) atomicAdd(o + 4 * i + j, rr);
          }
          const float rr = wave_reduce_sum(mine ? vt[i] : 0.f);
          if (lane_id() == 0 && rr != 0.f) atomicAdd(o + 4 * i + 3, rr);
        }
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


extern "C" int64_t so_projection_packed_blocks(int C, int N) {
  return (C <= 0 || N <= 0) ? 0 : so::ceil_div((int64_t)C * N, so::kPackChunk);
}

extern "C" int so_projection_packed(int C, int N, const float *means, const float *covars6, const float *quats,
                                    const float *scales, const float *viewmats, const float *Ks, int width, int height,
                                    float eps2d, float near_plane, float far_plane, float radius_clip, int camera_model,
                                    int32_t *block_counts, uint64_t *vis_masks, int64_t *block_offsets, int64_t *total_dev, int64_t *camera_ids,
                                    int64_t *gaussian_ids, int32_t *radii, float *means2d, float *depths, float *conics,
                                    float *compensations, void *stream) {
  SO_REQUIRE(C >= 0 && N >= 0 && width > 0 && height > 0, "so_projection_packed: bad sizes C=%d N=%d %dx%d", C, N, width, height);
  if (camera_model < 0 || camera_model > SO_CAM_SPHERICAL) {
    so::set_error("so_projection_packed: unsupported camera_model %d", camera_model);
    return SO_ERR_UNSUPPORTED;
  }
  const int64_t nblk = so_projection_packed_blocks(C, N);
  if (nblk == 0) return SO_OK;
  SO_REQUIRE(nblk < ((int64_t)1 << 31), "so_projection_packed: C*N too large");
  SO_REQUIRE(means && viewmats && Ks && (covars6 || (quats && scales)), "so_projection_packed: null pointer");
  hipStream_t st = so::as_stream(stream);
  const bool write = camera_ids != nullptr;
  if (!write) {      // pass 0 + scan
    SO_REQUIRE(block_counts && vis_masks && block_offsets && total_dev, "so_projection_packed: counting pass needs block_counts, vis_masks, block_offsets, total_dev");
    auto kern = covars6 ? so::k_projection_packed<true, false> : so::k_projection_packed<false, false>;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), 0, st, C, N, means, covars6, quats, scales, viewmats, Ks, width,
                       height, eps2d, near_plane, far_plane, radius_clip, camera_model, block_counts, vis_masks, nullptr, nullptr,
                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    hipLaunchKernelGGL(so::k_scan_counts, dim3(1), dim3(1024), 0, st, nblk, block_counts, block_offsets, total_dev);
    return so::check_launch("so_projection_packed (count)");
  }
  SO_REQUIRE(vis_masks && block_offsets && gaussian_ids && radii && means2d && depths && conics, "so_projection_packed: writing pass: null pointer");
  auto kern = covars6 ? so::k_projection_packed<true, true> : so::k_projection_packed<false, true>;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), 0, st, C, N, means, covars6, quats, scales, viewmats, Ks, width,
                     height, eps2d, near_plane, far_plane, radius_clip, camera_model, nullptr, vis_masks, block_offsets, camera_ids,
                     gaussian_ids, radii, means2d, depths, conics, compensations);
  return so::check_launch("so_projection_packed (write)");
}

extern "C" int so_projection_bwd_packed(int C, int N, int64_t nnz, const float *means, const float *covars6, const float *quats,
                                        const float *scales, const float *viewmats, const float *Ks, int width, int height,
                                        float eps2d, int camera_model, const int64_t *camera_ids, const int64_t *gaussian_ids,
                                        const float *v_means2d, const float *v_depths, const float *v_conics,
                                        const float *v_compensations, float *v_means, float *v_covars6, float *v_quats,
                                        float *v_scales, float *v_viewmats, void *stream) {
  SO_REQUIRE(C >= 0 && N >= 0 && nnz >= 0 && width > 0 && height > 0, "so_projection_bwd_packed: bad sizes");
  if (camera_model < 0 || camera_model > SO_CAM_SPHERICAL) {
    so::set_error("so_projection_bwd_packed: unsupported camera_model %d", camera_model);
    return SO_ERR_UNSUPPORTED;
  }
  if (nnz == 0) return SO_OK;
  SO_REQUIRE(means && viewmats && Ks && camera_ids && gaussian_ids && v_means2d && v_conics && v_means, "so_projection_bwd_packed: null pointer");
  SO_REQUIRE(covars6 ? (v_covars6 != nullptr) : (quats && scales && v_quats && v_scales),
             "so_projection_bwd_packed: missing covars6/quats/scales buffers");
  auto kern = covars6 ? (v_viewmats ? so::k_projection_bwd_packed<true, true> : so::k_projection_bwd_packed<true, false>)
                      : (v_viewmats ? so::k_projection_bwd_packed<false, true> : so::k_projection_bwd_packed<false, false>);
  hipLaunchKernelGGL(kern, dim3(so::grid_for(nnz, 256)), dim3(256), 0, so::as_stream(stream), C, N, nnz, means, covars6, quats,
                     scales, viewmats, Ks, width, height, eps2d, camera_model, camera_ids, gaussian_ids, v_means2d, v_depths,
                     v_conics, v_compensations, v_means, v_covars6, v_quats, v_scales, v_viewmats);
  return so::check_launch("so_projection_bwd_packed");
}
