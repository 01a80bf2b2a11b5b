// tests/host_harness/harness.cpp -- TEST-ONLY CPU build of splat_one_amd/csrc/splat_math.hpp.
// Built with g++ by tests/host_harness/build.py; lets `-m "not gpu"` tests check the product's
// per-Gaussian math (projection fwd/bwd, SH bases + derivatives) against the float64 autograd
// oracle, in double and in float, without a GPU.  Nothing in splat_one_amd/ loads this library.
#include "../../splat_one_amd/csrc/splat_math.hpp"
#include "../../splat_one_amd/csrc/so_rng.hpp"

template <typename T>
static void proj_fwd(int C, int N, const T *means, const T *covars6, const T *quats, const T *scales,
                     const T *viewmats, const T *Ks, int W, int H, T eps2d, T nearp, T farp, T rclip, int model,
                     int32_t *radii, T *means2d, T *depths, T *conics, T *comps) {
  for (int c = 0; c < C; ++c) {
    const T *V = viewmats + 16 * c;
    T Rw[9] = {V[0], V[1], V[2], V[4], V[5], V[6], V[8], V[9], V[10]}, tw[3] = {V[3], V[7], V[11]};
    const T *K = Ks + 9 * c;
    for (int n = 0; n < N; ++n) {
      so::ProjOut<T> o;
      so::project_fwd<T>(means + 3 * n, covars6 ? covars6 + 6 * n : nullptr, quats ? quats + 4 * n : nullptr,
                         scales ? scales + 3 * n : nullptr, Rw, tw, K[0], K[4], K[2], K[5], W, H, eps2d, nearp,
                         farp, rclip, model, o);
      const int64_t i = (int64_t)c * N + n;
      radii[i] = o.radius;
      means2d[2 * i] = o.m2d[0]; means2d[2 * i + 1] = o.m2d[1];
      depths[i] = o.depth;
      conics[3 * i] = o.conic[0]; conics[3 * i + 1] = o.conic[1]; conics[3 * i + 2] = o.conic[2];
      if (comps) comps[i] = o.comp;
    }
  }
}

template <typename T>
static void proj_bwd(int C, int N, const T *means, const T *covars6, const T *quats, const T *scales,
                     const T *viewmats, const T *Ks, int W, int H, T eps2d, int model, const int32_t *radii,
                     const T *v_means2d, const T *v_depths, const T *v_conics, const T *v_comps, T *v_means,
                     T *v_covars6, T *v_quats, T *v_scales, T *v_viewmats) {
  for (int c = 0; c < C; ++c) {
    const T *V = viewmats + 16 * c;
    T Rw[9] = {V[0], V[1], V[2], V[4], V[5], V[6], V[8], V[9], V[10]}, tw[3] = {V[3], V[7], V[11]};
    const T *K = Ks + 9 * c;
    T vR[9] = {0}, vt[3] = {0};
    for (int n = 0; n < N; ++n) {
      const int64_t i = (int64_t)c * N + n;
      if (radii[i] <= 0) continue;
      so::project_bwd<T>(means + 3 * n, covars6 ? covars6 + 6 * n : nullptr, quats ? quats + 4 * n : nullptr,
                         scales ? scales + 3 * n : nullptr, Rw, tw, K[0], K[4], K[2], K[5], W, H, eps2d, model,
                         v_means2d + 2 * i, v_depths[i], v_conics + 3 * i, v_comps ? v_comps[i] : T(0),
                         v_means + 3 * n, v_covars6 ? v_covars6 + 6 * n : nullptr,
                         v_quats ? v_quats + 4 * n : nullptr, v_scales ? v_scales + 3 * n : nullptr,
                         v_viewmats ? vR : nullptr, vt);
    }
    if (v_viewmats) {
      T *o = v_viewmats + 16 * c;
      for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) o[4 * i + j] += vR[3 * i + j];
        o[4 * i + 3] += vt[i];
      }
    }
  }
}

template <typename T>
static void sh_bases_all(int degree, int64_t M, const T *dirs_unit, T *Y, T *dY) {
  const int nb = (degree + 1) * (degree + 1);
  for (int64_t m = 0; m < M; ++m) {
    T y[25], dy[25][3];
    so::sh_bases<T>(degree, dirs_unit[3 * m], dirs_unit[3 * m + 1], dirs_unit[3 * m + 2], y, dy);
    for (int k = 0; k < nb; ++k) {
      Y[m * nb + k] = y[k];
      for (int a = 0; a < 3; ++a) dY[(m * nb + k) * 3 + a] = dy[k][a];
    }
  }
}

#define INST(T, SFX)                                                                                       \
  extern "C" void hh_proj_fwd_##SFX(int C, int N, const T *means, const T *covars6, const T *quats,        \
                                    const T *scales, const T *viewmats, const T *Ks, int W, int H,         \
                                    double eps2d, double nearp, double farp, double rclip, int model,      \
                                    int32_t *radii, T *means2d, T *depths, T *conics, T *comps) {           \
    proj_fwd<T>(C, N, means, covars6, quats, scales, viewmats, Ks, W, H, (T)eps2d, (T)nearp, (T)farp,      \
                (T)rclip, model, radii, means2d, depths, conics, comps);                                   \
  }                                                                                                        \
  extern "C" void hh_proj_bwd_##SFX(int C, int N, const T *means, const T *covars6, const T *quats,        \
                                    const T *scales, const T *viewmats, const T *Ks, int W, int H,         \
                                    double eps2d, int model, const int32_t *radii, const T *v_means2d,     \
                                    const T *v_depths, const T *v_conics, const T *v_comps, T *v_means,    \
                                    T *v_covars6, T *v_quats, T *v_scales, T *v_viewmats) {                 \
    proj_bwd<T>(C, N, means, covars6, quats, scales, viewmats, Ks, W, H, (T)eps2d, model, radii,           \
                v_means2d, v_depths, v_conics, v_comps, v_means, v_covars6, v_quats, v_scales, v_viewmats); \
  }                                                                                                        \
  extern "C" void hh_sh_bases_##SFX(int degree, int64_t M, const T *dirs, T *Y, T *dY) {                   \
    sh_bases_all<T>(degree, M, dirs, Y, dY);                                                               \
  }

INST(double, f64)
INST(float, f32)

// ---- csrc/so_rng.hpp: the counter-based split noise of the device-side densification ----------------
extern "C" void hh_philox4x32_10(int n, const uint32_t *ctr, const uint32_t *key, uint32_t *out) {
  for (int i = 0; i < n; ++i) {
    const so::Philox4 r = so::philox4x32_10(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3], key[0], key[1]);
    for (int k = 0; k < 4; ++k) out[4 * i + k] = r.x[k];
  }
}
extern "C" void hh_split_normals(int n, unsigned long long seed, uint32_t step, const uint32_t *ids, uint32_t child, float *out) {
  for (int i = 0; i < n; ++i) {
    float z[3];
    so::split_normals(seed, step, ids[i], child, z);
    out[3 * i] = z[0]; out[3 * i + 1] = z[1]; out[3 * i + 2] = z[2];
  }
}
