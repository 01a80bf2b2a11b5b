// mcmc.hip -- kernels of the MCMC densification strategy (3DGS-MCMC, Kheradmand et al. 2024) that the
// reference selects with its `mcmc` preset (/root/reference/utils/gsplat_utils/gsplat_trainer.py:975-983,
// driven at :753-761): K13 `compute_relocation` (binomial opacity/scale split) and the position noise
// injection.  One lane per Gaussian, pure streaming.
#include "so_common.hpp"
#include "splat_math.hpp"

namespace so {

__global__ void __launch_bounds__(256)
k_compute_relocation(int64_t N, const float *__restrict__ opacities, const float *__restrict__ scales,
                     const int32_t *__restrict__ ratios, const float *__restrict__ binoms, int n_max,
                     float *__restrict__ new_opacities, float *__restrict__ new_scales) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    int n = ratios[i];
    n = n < 1 ? 1 : (n > n_max ? n_max : n);
    const float op = opacities[i];
    // new opacity: 1 - (1 - o)^(1/n)
    const float new_op = 1.f - powf(1.f - op, 1.f / (float)n);
    // new scale: o / sum_{i=1..n} sum_{k=0..i-1} C(i-1,k) (-1)^k / sqrt(k+1) * new_op^(k+1)
    float denom = 0.f;
    for (int a = 1; a <= n; ++a) {
      float pw = new_op;   // new_op^(k+1)
      for (int k = 0; k <= a - 1; ++k) {
        const float term = ((k & 1) ? -1.f : 1.f) / sqrtf((float)(k + 1)) * pw;
        denom += binoms[(a - 1) * n_max + k] * term;
        pw *= new_op;
      }
    }
    const float coeff = op / denom;
    new_opacities[i] = new_op;
    new_scales[3 * i] = coeff * scales[3 * i];
    new_scales[3 * i + 1] = coeff * scales[3 * i + 1];
    new_scales[3 * i + 2] = coeff * scales[3 * i + 2];
  }
}

// means += Sigma * (noise * sigmoid_k(1 - opacity) * scaler),  Sigma = R diag(s^2) R^T,
// sigmoid_k(x) = 1 / (1 + exp(-100 (x - 0.995)))
__global__ void __launch_bounds__(256)
k_inject_noise(int64_t N, float *__restrict__ means, const float *__restrict__ log_scales,
               const float *__restrict__ quats, const float *__restrict__ logit_opac,
               const float *__restrict__ noise, float scaler) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 qq = *reinterpret_cast<const float4 *>(quats + 4 * i);
    const float q[4] = {qq.x, qq.y, qq.z, qq.w};
    const float s[3] = {expf(log_scales[3 * i]), expf(log_scales[3 * i + 1]), expf(log_scales[3 * i + 2])};
    float cov[9], M[9], Rq[9], qn[4], inv_norm;
    quat_scale_to_covar<float>(q, s, cov, M, Rq, qn, inv_norm);
    const float op = 1.f / (1.f + expf(-logit_opac[i]));
    const float g = 1.f / (1.f + expf(-100.f * ((1.f - op) - 0.995f))) * scaler;
    const float n0 = noise[3 * i] * g, n1 = noise[3 * i + 1] * g, n2 = noise[3 * i + 2] * g;
    means[3 * i] += cov[0] * n0 + cov[1] * n1 + cov[2] * n2;
    means[3 * i + 1] += cov[3] * n0 + cov[4] * n1 + cov[5] * n2;
    means[3 * i + 2] += cov[6] * n0 + cov[7] * n1 + cov[8] * n2;
  }
}

static inline int mc_grid(int64_t n) {
  int64_t g = ceil_div(n, 256);
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace so

extern "C" int so_compute_relocation(int64_t N, const float *opacities, const float *scales, const int32_t *ratios,
                                     const float *binoms, int n_max, float *new_opacities, float *new_scales,
                                     void *stream) {
  SO_REQUIRE(N >= 0 && n_max >= 1, "so_compute_relocation: bad sizes");
  if (N == 0) return SO_OK;
  SO_REQUIRE(opacities && scales && ratios && binoms && new_opacities && new_scales, "so_compute_relocation: null pointer");
  hipLaunchKernelGGL(so::k_compute_relocation, dim3(so::mc_grid(N)), dim3(256), 0, so::as_stream(stream), N, opacities,
                     scales, ratios, binoms, n_max, new_opacities, new_scales);
  return so::check_launch("so_compute_relocation");
}

extern "C" int so_inject_noise(int64_t N, float *means, const float *log_scales, const float *quats,
                               const float *logit_opacities, const float *noise, float scaler, void *stream) {
  SO_REQUIRE(N >= 0, "so_inject_noise: bad sizes");
  if (N == 0) return SO_OK;
  SO_REQUIRE(means && log_scales && quats && logit_opacities && noise, "so_inject_noise: null pointer");
  hipLaunchKernelGGL(so::k_inject_noise, dim3(so::mc_grid(N)), dim3(256), 0, so::as_stream(stream), N, means, log_scales,
                     quats, logit_opacities, noise, scaler);
  return so::check_launch("so_inject_noise");
}
