// splat_math.hpp -- per-Gaussian math of the 3DGS path (projection, SH) for gfx950 kernels.
//
// Written from the published algorithm (the reference only *calls* it:
// /root/reference/utils/gsplat_utils/gsplat_trainer.py:477-494).  Everything here is a plain
// inline function of scalars so that
//   * the HIP kernels in projection.hip / sh.hip call it per lane, and
//   * tests/host_harness builds the very same code with g++ (T=double and T=float) and checks
//     it against the float64 autograd oracle on the CPU, before any GPU time is spent.
// No CPU path of the product uses this header: the harness is test-only.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define SO_HD __host__ __device__ __forceinline__
#define SO_UNROLL _Pragma("unroll")
#else
#define SO_HD inline
#define SO_UNROLL
#endif

namespace so {

enum CameraModel : int { CAM_PINHOLE = 0, CAM_ORTHO = 1, CAM_FISHEYE = 2, CAM_SPHERICAL = 3 };
// CAM_SPHERICAL -- the 360-degree (equirectangular) cameras of the reference's data sets
// (utils/datasets/opensfm.py:176-193, 430-436 `spherical` / `equirectangular`; app/camera_models.py).  The fork's
// kernel for it is not part of the reference tree, so this model is DEFINED HERE (parity unpinned):
//   lon = atan2(x, z), lat = atan2(y, sqrt(x^2 + z^2))   (OpenCV axes: +z forward, +y down)
//   u = W (lon / 2pi + 1/2),  v = H (lat / pi + 1/2)       -- K is not used (the reference's spherical "K" is not an
//   intrinsic matrix); depth = |mean_c| (the range: z is negative for half of the sphere), near/far apply to it;
//   the 2D covariance is the EWA one, J = d(u,v)/d(x,y,z); a splat is not wrapped across the +-pi seam.

constexpr double kShC0 = 0.28209479177387814;
constexpr double kShC1 = 0.4886025119029199;
constexpr double kShC2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                             -1.0925484305920792, 0.5462742152960396};
constexpr double kShC3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
                             0.3731763325901154, -0.4570457994644658, 1.445305721320277,
                             -0.5900435899266435};
constexpr double kShC4[9] = {2.5033429417967046, -1.7701307697799304, 0.9461746957575601,
                             -0.6690465435572892, 0.10578554691520431, -0.6690465435572892,
                             0.47308734787878004, -1.7701307697799304, 0.6258357354491761};

template <typename T> SO_HD T tmin(T a, T b) { return a < b ? a : b; }
template <typename T> SO_HD T tmax(T a, T b) { return a > b ? a : b; }

// ---------------------------------------------------------------------------------------------
// small 3x3 helpers, row-major arrays  A[3*i+j]
// ---------------------------------------------------------------------------------------------
template <typename T> SO_HD void mat3_mul(const T *A, const T *B, T *C) {
  SO_UNROLL
  for (int i = 0; i < 3; ++i)
    SO_UNROLL
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
template <typename T> SO_HD void mat3_mul_bt(const T *A, const T *B, T *C) {  // C = A * B^T
  SO_UNROLL
  for (int i = 0; i < 3; ++i)
    SO_UNROLL
    for (int j = 0; j < 3; ++j)
      C[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
template <typename T> SO_HD void mat3_mul_at(const T *A, const T *B, T *C) {  // C = A^T * B
  SO_UNROLL
  for (int i = 0; i < 3; ++i)
    SO_UNROLL
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}

// (w,x,y,z) -> rotation matrix of the normalised quaternion; also returns 1/|q| and q_n.
template <typename T> SO_HD void quat_to_rotmat(const T *q, T *R, T *qn, T &inv_norm) {
  inv_norm = T(1) / std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const T w = q[0] * inv_norm, x = q[1] * inv_norm, y = q[2] * inv_norm, z = q[3] * inv_norm;
  qn[0] = w; qn[1] = x; qn[2] = y; qn[3] = z;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

// Sigma = (R S)(R S)^T ; M = R S is returned for the backward.
template <typename T> SO_HD void quat_scale_to_covar(const T *q, const T *s, T *cov, T *M, T *Rq, T *qn, T &inv_norm) {
  quat_to_rotmat(q, Rq, qn, inv_norm);
  SO_UNROLL
  for (int i = 0; i < 3; ++i)
    SO_UNROLL
    for (int j = 0; j < 3; ++j) M[3 * i + j] = Rq[3 * i + j] * s[j];
  mat3_mul_bt(M, M, cov);
}

// ---------------------------------------------------------------------------------------------
// camera models: mean_c (camera space), cov_c (3x3) -> mean2d, J (2x3), cov2d (a,b,d symmetric)
// ---------------------------------------------------------------------------------------------
template <typename T> struct CamAux {   // what the backward needs again
  T J[6];
  T tx, ty;      // pinhole: clamped numerators
  bool clamp_x, clamp_y;
};

// SPH = false compiles the spherical model out (its reverse mode costs the fused backward kernel registers: the
// launchers pick that variant when no view of the step is spherical).
template <typename T, bool SPH = true>
SO_HD void camera_project(int model, const T *mc, T fx, T fy, T cx, T cy, int W, int H, T *m2d, CamAux<T> &aux) {
  const T x = mc[0], y = mc[1], z = mc[2];
  T *J = aux.J;
  aux.clamp_x = aux.clamp_y = false;
  aux.tx = aux.ty = 0;
  if (model == CAM_PINHOLE) {
    const T tan_fovx = T(0.5) * T(W) / fx, tan_fovy = T(0.5) * T(H) / fy;
    const T lim_x_pos = (T(W) - cx) / fx + T(0.3) * tan_fovx, lim_x_neg = cx / fx + T(0.3) * tan_fovx;
    const T lim_y_pos = (T(H) - cy) / fy + T(0.3) * tan_fovy, lim_y_neg = cy / fy + T(0.3) * tan_fovy;
    const T rz = T(1) / z, rz2 = rz * rz;
    const T xr = x * rz, yr = y * rz;
    aux.clamp_x = !(xr <= lim_x_pos && xr >= -lim_x_neg);
    aux.clamp_y = !(yr <= lim_y_pos && yr >= -lim_y_neg);
    aux.tx = z * tmin(lim_x_pos, tmax(-lim_x_neg, xr));
    aux.ty = z * tmin(lim_y_pos, tmax(-lim_y_neg, yr));
    J[0] = fx * rz; J[1] = 0; J[2] = -fx * aux.tx * rz2;
    J[3] = 0; J[4] = fy * rz; J[5] = -fy * aux.ty * rz2;
    m2d[0] = fx * x * rz + cx;
    m2d[1] = fy * y * rz + cy;
  } else if (model == CAM_ORTHO) {
    J[0] = fx; J[1] = 0; J[2] = 0; J[3] = 0; J[4] = fy; J[5] = 0;
    m2d[0] = fx * x + cx;
    m2d[1] = fy * y + cy;
  } else if (SPH && model == CAM_SPHERICAL) {
    const T fxs = T(W) * T(0.15915494309189535), fys = T(H) * T(0.3183098861837907);   // W / 2pi, H / pi
    const T p2 = x * x + z * z + T(1e-12), p = std::sqrt(p2), r2 = p2 + y * y;
    const T ip2 = T(1) / p2, ir2 = T(1) / r2, k = fys * ir2 / p;
    m2d[0] = fxs * std::atan2(x, z) + T(0.5) * T(W);
    m2d[1] = fys * std::atan2(y, p) + T(0.5) * T(H);
    J[0] = fxs * z * ip2; J[1] = 0; J[2] = -fxs * x * ip2;
    J[3] = -k * x * y; J[4] = fys * p * ir2; J[5] = -k * z * y;
  } else {  // equidistant fisheye
    const T eps = T(1e-7);
    const T xy_len = std::sqrt(x * x + y * y) + eps;
    const T theta = std::atan2(xy_len, z + eps);
    m2d[0] = x * fx * theta / xy_len + cx;
    m2d[1] = y * fy * theta / xy_len + cy;
    const T x2 = x * x + eps, y2 = y * y, xy = x * y, x2y2 = x2 + y2;
    const T inv = T(1) / (x2y2 + z * z);
    const T b = std::atan2(xy_len, z) / xy_len / x2y2;
    const T a = z * inv / x2y2;
    J[0] = fx * (x2 * a + y2 * b); J[1] = fx * xy * (a - b); J[2] = -fx * x * inv;
    J[3] = fy * xy * (a - b); J[4] = fy * (y2 * a + x2 * b); J[5] = -fy * y * inv;
  }
}

// cov2d = J cov_c J^T  -> (a, b, d)
template <typename T> SO_HD void project_cov(const T *J, const T *cc, T &a, T &b, T &d) {
  T JC[6];
  SO_UNROLL
  for (int i = 0; i < 2; ++i)
    SO_UNROLL
    for (int j = 0; j < 3; ++j) JC[3 * i + j] = J[3 * i] * cc[j] + J[3 * i + 1] * cc[3 + j] + J[3 * i + 2] * cc[6 + j];
  a = JC[0] * J[0] + JC[1] * J[1] + JC[2] * J[2];
  b = JC[0] * J[3] + JC[1] * J[4] + JC[2] * J[5];
  d = JC[3] * J[3] + JC[4] * J[4] + JC[5] * J[5];
}

// what near / far apply to and what the tile sort orders by: camera-space z, or the range for the spherical model
template <typename T, bool SPH = true> SO_HD T camera_depth(int model, const T *mc) {
  return (SPH && model == CAM_SPHERICAL) ? std::sqrt(mc[0] * mc[0] + mc[1] * mc[1] + mc[2] * mc[2]) : mc[2];
}

template <typename T> struct ProjOut {
  T m2d[2];
  T depth;
  T conic[3];
  T comp;
  int radius;  // 0 = culled
};

// Full forward for one (camera, Gaussian).  Rw/tw: rotation (row-major 3x3) and translation of
// the world->camera matrix.  `covar6` (xx,xy,xz,yy,yz,zz) is used when non-null, else quat/scale.
template <typename T, bool SPH = true>
SO_HD void project_fwd(const T *mean, const T *covar6, const T *quat, const T *scale, const T *Rw, const T *tw,
                       T fx, T fy, T cx, T cy, int W, int H, T eps2d, T near_plane, T far_plane,
                       T radius_clip, int model, ProjOut<T> &o) {
  o.radius = 0;
  o.m2d[0] = o.m2d[1] = o.depth = o.conic[0] = o.conic[1] = o.conic[2] = o.comp = 0;
  T mc[3];
  SO_UNROLL
  for (int i = 0; i < 3; ++i) mc[i] = Rw[3 * i] * mean[0] + Rw[3 * i + 1] * mean[1] + Rw[3 * i + 2] * mean[2] + tw[i];
  const T depth = camera_depth<T, SPH>(model, mc);
  if (depth < near_plane || depth > far_plane) return;
  T cov[9];
  if (covar6) {
    cov[0] = covar6[0]; cov[1] = covar6[1]; cov[2] = covar6[2];
    cov[3] = covar6[1]; cov[4] = covar6[3]; cov[5] = covar6[4];
    cov[6] = covar6[2]; cov[7] = covar6[4]; cov[8] = covar6[5];
  } else {
    T M[9], Rq[9], qn[4], inv_norm;
    quat_scale_to_covar(quat, scale, cov, M, Rq, qn, inv_norm);
  }
  T tmp[9], cc[9];
  mat3_mul(Rw, cov, tmp);
  mat3_mul_bt(tmp, Rw, cc);
  CamAux<T> aux;
  T m2d[2];
  camera_project<T, SPH>(model, mc, fx, fy, cx, cy, W, H, m2d, aux);
  T a, b, d;
  project_cov(aux.J, cc, a, b, d);
  const T det_orig = a * d - b * b;
  a += eps2d;
  d += eps2d;
  const T det = a * d - b * b;
  if (!(det > 0)) return;
  const T comp = std::sqrt(tmax(T(0), det_orig / det));
  const T hb = T(0.5) * (a + d);
  const T v1 = hb + std::sqrt(tmax(T(0.01), hb * hb - det));
  const T radius = std::ceil(T(3) * std::sqrt(v1));
  if (radius <= radius_clip) return;
  if (m2d[0] + radius <= 0 || m2d[0] - radius >= T(W) || m2d[1] + radius <= 0 || m2d[1] - radius >= T(H)) return;
  const T rdet = T(1) / det;
  o.radius = (int)radius;
  o.m2d[0] = m2d[0]; o.m2d[1] = m2d[1];
  o.depth = depth;
  o.conic[0] = d * rdet; o.conic[1] = -b * rdet; o.conic[2] = a * rdet;
  o.comp = comp;
}

// Backward for one visible (camera, Gaussian): recomputes the forward intermediates.
// Inputs v_m2d[2], v_depth, v_conic[3], v_comp (0 if unused).  Accumulates (+=) into
// v_mean[3], and either v_covar6[6] (covar6 != null) or v_quat[4], v_scale[3]; optionally into
// v_Rw[9], v_tw[3] (world->camera rotation / translation) when v_Rw != null.
template <typename T, bool SPH = true>
SO_HD void project_bwd(const T *mean, const T *covar6, const T *quat, const T *scale, const T *Rw, const T *tw,
                       T fx, T fy, T cx, T cy, int W, int H, T eps2d, int model,
                       const T *v_m2d, T v_depth, const T *v_conic, T v_comp,
                       T *v_mean, T *v_covar6, T *v_quat, T *v_scale, T *v_Rw, T *v_tw) {
  T mc[3];
  SO_UNROLL
  for (int i = 0; i < 3; ++i) mc[i] = Rw[3 * i] * mean[0] + Rw[3 * i + 1] * mean[1] + Rw[3 * i + 2] * mean[2] + tw[i];
  T cov[9], M[9], Rq[9], qn[4], inv_norm = 1;
  if (covar6) {
    cov[0] = covar6[0]; cov[1] = covar6[1]; cov[2] = covar6[2];
    cov[3] = covar6[1]; cov[4] = covar6[3]; cov[5] = covar6[4];
    cov[6] = covar6[2]; cov[7] = covar6[4]; cov[8] = covar6[5];
  } else {
    quat_scale_to_covar(quat, scale, cov, M, Rq, qn, inv_norm);
  }
  T tmp[9], cc[9];
  mat3_mul(Rw, cov, tmp);
  mat3_mul_bt(tmp, Rw, cc);
  CamAux<T> aux;
  T m2d[2];
  camera_project<T, SPH>(model, mc, fx, fy, cx, cy, W, H, m2d, aux);
  const T *J = aux.J;
  T a0, b, d0;
  project_cov(J, cc, a0, b, d0);
  const T a = a0 + eps2d, d = d0 + eps2d;
  const T det = a * d - b * b;
  const T rdet = T(1) / det;
  // Y = X^-1 ; G = sym(v_conic) ; Vx = -Y G Y   (symmetric 2x2: Vxa, Vxb, Vxd)
  const T ya = d * rdet, yb = -b * rdet, yd = a * rdet;
  const T ga = v_conic[0], gb = T(0.5) * v_conic[1], gd = v_conic[2];
  // YG
  const T p00 = ya * ga + yb * gb, p01 = ya * gb + yb * gd, p10 = yb * ga + yd * gb, p11 = yb * gb + yd * gd;
  T Vxa = -(p00 * ya + p01 * yb);
  T Vxb = -(p00 * yb + p01 * yd);
  T Vxd = -(p10 * yb + p11 * yd);
  if (v_comp != 0) {
    const T det_orig = a0 * d0 - b * b;
    const T r = det_orig * rdet;
    if (r > 0) {
      const T comp = std::sqrt(r);
      const T s = v_comp * T(0.5) / (comp * det);
      Vxa += s * (d0 - r * d);
      Vxd += s * (a0 - r * a);
      Vxb += s * (-b) * (1 - r);
    }
  }
  // v_cc = J^T Vx J (3x3 symmetric) ; v_J = 2 Vx J cc (2x3)
  T VJ[6];  // Vx * J
  SO_UNROLL
  for (int j = 0; j < 3; ++j) {
    VJ[j] = Vxa * J[j] + Vxb * J[3 + j];
    VJ[3 + j] = Vxb * J[j] + Vxd * J[3 + j];
  }
  T v_cc[9];
  SO_UNROLL
  for (int i = 0; i < 3; ++i)
    SO_UNROLL
    for (int j = 0; j < 3; ++j) v_cc[3 * i + j] = J[i] * VJ[j] + J[3 + i] * VJ[3 + j];
  T v_J[6];
  SO_UNROLL
  for (int i = 0; i < 2; ++i)
    SO_UNROLL
    for (int j = 0; j < 3; ++j)
      v_J[3 * i + j] = 2 * (VJ[3 * i] * cc[j] + VJ[3 * i + 1] * cc[3 + j] + VJ[3 * i + 2] * cc[6 + j]);
  // camera model VJP -> v_mc
  T v_mc[3] = {0, 0, (SPH && model == CAM_SPHERICAL) ? T(0) : v_depth};
  const T x = mc[0], y = mc[1], z = mc[2];
  if (model == CAM_PINHOLE) {
    const T rz = T(1) / z, rz2 = rz * rz, rz3 = rz2 * rz;
    v_mc[0] += fx * rz * v_m2d[0];
    v_mc[1] += fy * rz * v_m2d[1];
    v_mc[2] += -(fx * x * v_m2d[0] + fy * y * v_m2d[1]) * rz2;
    if (!aux.clamp_x) v_mc[0] += -fx * rz2 * v_J[2]; else v_mc[2] += -fx * rz3 * v_J[2] * aux.tx;
    if (!aux.clamp_y) v_mc[1] += -fy * rz2 * v_J[5]; else v_mc[2] += -fy * rz3 * v_J[5] * aux.ty;
    v_mc[2] += -fx * rz2 * v_J[0] - fy * rz2 * v_J[4] + 2 * fx * aux.tx * rz3 * v_J[2] + 2 * fy * aux.ty * rz3 * v_J[5];
  } else if (model == CAM_ORTHO) {
    v_mc[0] += fx * v_m2d[0];
    v_mc[1] += fy * v_m2d[1];
  } else if (SPH && model == CAM_SPHERICAL) {
    // reverse-mode through the spherical forward sequence of camera_project() (+ depth = sqrt(r2))
    const T fxs = T(W) * T(0.15915494309189535), fys = T(H) * T(0.3183098861837907);
    const T p2 = x * x + z * z + T(1e-12), p = std::sqrt(p2), r2 = p2 + y * y;
    const T ip2 = T(1) / p2, ir2 = T(1) / r2, ip = T(1) / p, k = fys * ir2 * ip;
    T vx = 0, vy = 0, vz = 0, v_ip2 = 0, v_ir2 = 0, v_ip = 0, v_p = 0, v_r2 = 0;
    // J0 = fxs z ip2 ; J2 = -fxs x ip2
    vz += fxs * ip2 * v_J[0]; v_ip2 += fxs * z * v_J[0];
    vx += -fxs * ip2 * v_J[2]; v_ip2 += -fxs * x * v_J[2];
    // J3 = -k x y ; J5 = -k z y ; J4 = fys p ir2
    const T v_k = -x * y * v_J[3] - z * y * v_J[5];
    vx += -k * y * v_J[3]; vy += -k * (x * v_J[3] + z * v_J[5]); vz += -k * y * v_J[5];
    v_p += fys * ir2 * v_J[4]; v_ir2 += fys * p * v_J[4];
    // k = fys ir2 ip
    v_ir2 += fys * ip * v_k; v_ip += fys * ir2 * v_k;
    // u = fxs atan2(x, z) ; v = fys atan2(y, p)
    const T v_lon = fxs * v_m2d[0], v_lat = fys * v_m2d[1];
    vx += z * ip2 * v_lon; vz += -x * ip2 * v_lon;
    vy += p * ir2 * v_lat; v_p += -y * ir2 * v_lat;
    // depth = sqrt(r2)
    v_r2 += T(0.5) * v_depth / std::sqrt(r2);
    // ip = 1/p ; ir2 = 1/r2 ; ip2 = 1/p2 ; p = sqrt(p2) ; r2 = p2 + y^2 ; p2 = x^2 + z^2
    v_p += -ip * ip * v_ip;
    v_r2 += -ir2 * ir2 * v_ir2;
    T v_p2 = -ip2 * ip2 * v_ip2 + T(0.5) * ip * v_p;
    v_p2 += v_r2; vy += 2 * y * v_r2;
    vx += 2 * x * v_p2; vz += 2 * z * v_p2;
    v_mc[0] += vx; v_mc[1] += vy; v_mc[2] += vz;
  } else {
    // reverse-mode through the fisheye forward sequence of camera_project()
    const T eps = T(1e-7);
    const T r2 = x * x + y * y;
    const T sr = std::sqrt(r2);
    const T L = sr + eps;                 // xy_len
    const T ze = z + eps;
    const T theta = std::atan2(L, ze);
    const T x2 = x * x + eps, y2 = y * y, xy = x * y, S = x2 + y2;
    const T q = S + z * z, inv = T(1) / q;
    const T thb = std::atan2(L, z);
    const T bb = thb / L / S;
    const T aa = z * inv / S;
    T vx = 0, vy = 0, vz = 0, vL = 0, vtheta = 0, vthb = 0;
    // mean2d: mx = fx x theta / L + cx
    const T s_ = theta / L;
    vx += fx * s_ * v_m2d[0];
    vy += fy * s_ * v_m2d[1];
    const T v_s = fx * x * v_m2d[0] + fy * y * v_m2d[1];
    vtheta += v_s / L;
    vL += -v_s * theta / (L * L);
    // J entries
    T v_x2 = 0, v_y2 = 0, v_xy = 0, v_a = 0, v_b = 0, v_inv = 0;
    // J00 = fx (x2 a + y2 b)
    v_x2 += fx * aa * v_J[0]; v_y2 += fx * bb * v_J[0]; v_a += fx * x2 * v_J[0]; v_b += fx * y2 * v_J[0];
    // J01 = fx xy (a-b) ; J10 = fy xy (a-b)
    const T vj_od = fx * v_J[1] + fy * v_J[3];
    v_xy += (aa - bb) * vj_od; v_a += xy * vj_od; v_b += -xy * vj_od;
    // J11 = fy (y2 a + x2 b)
    v_y2 += fy * aa * v_J[4]; v_x2 += fy * bb * v_J[4]; v_a += fy * y2 * v_J[4]; v_b += fy * x2 * v_J[4];
    // J02 = -fx x inv ; J12 = -fy y inv
    vx += -fx * inv * v_J[2]; v_inv += -fx * x * v_J[2];
    vy += -fy * inv * v_J[5]; v_inv += -fy * y * v_J[5];
    // a = z inv / S
    T v_S = 0;
    vz += inv / S * v_a; v_inv += z / S * v_a; v_S += -z * inv / (S * S) * v_a;
    // b = thb / L / S
    vthb += v_b / (L * S); vL += -thb / (L * L * S) * v_b; v_S += -thb / (L * S * S) * v_b;
    // inv = 1/q, q = S + z^2
    const T v_q = -inv * inv * v_inv;
    v_S += v_q; vz += 2 * z * v_q;
    // S = x2 + y2
    v_x2 += v_S; v_y2 += v_S;
    // theta = atan2(L, ze): d/dL = ze/(L^2+ze^2), d/dze = -L/(L^2+ze^2)
    { const T den = L * L + ze * ze; vL += ze / den * vtheta; vz += -L / den * vtheta; }
    { const T den = L * L + z * z;   vL += z / den * vthb;    vz += -L / den * vthb; }
    // x2 = x^2+eps, y2 = y^2, xy = x y
    vx += 2 * x * v_x2 + y * v_xy;
    vy += 2 * y * v_y2 + x * v_xy;
    // L = sqrt(x^2+y^2) + eps
    if (sr > 0) { vx += x / sr * vL; vy += y / sr * vL; }
    v_mc[0] += vx; v_mc[1] += vy; v_mc[2] += vz;
  }
  // world -> camera
  SO_UNROLL
  for (int j = 0; j < 3; ++j) v_mean[j] += Rw[j] * v_mc[0] + Rw[3 + j] * v_mc[1] + Rw[6 + j] * v_mc[2];
  T t2[9], v_cov[9];
  mat3_mul_at(Rw, v_cc, t2);   // Rw^T v_cc
  mat3_mul(t2, Rw, v_cov);     // Rw^T v_cc Rw
  if (v_Rw) {
    SO_UNROLL
    for (int i = 0; i < 3; ++i) {
      SO_UNROLL
      for (int j = 0; j < 3; ++j) v_Rw[3 * i + j] += v_mc[i] * mean[j];
      v_tw[i] += v_mc[i];
    }
    // cov_c = Rw cov Rw^T  ->  v_Rw += 2 v_cc Rw cov   (v_cc, cov symmetric)
    T rc[9], add[9];
    mat3_mul(Rw, cov, rc);
    mat3_mul(v_cc, rc, add);
    SO_UNROLL
    for (int i = 0; i < 9; ++i) v_Rw[i] += 2 * add[i];
  }
  if (covar6) {
    v_covar6[0] += v_cov[0]; v_covar6[1] += v_cov[1] + v_cov[3]; v_covar6[2] += v_cov[2] + v_cov[6];
    v_covar6[3] += v_cov[4]; v_covar6[4] += v_cov[5] + v_cov[7]; v_covar6[5] += v_cov[8];
    return;
  }
  // cov = M M^T -> v_M = 2 v_cov M
  T v_M[9];
  mat3_mul(v_cov, M, v_M);
  SO_UNROLL
  for (int i = 0; i < 9; ++i) v_M[i] *= 2;
  T G[9];
  SO_UNROLL
  for (int i = 0; i < 3; ++i)
    SO_UNROLL
    for (int j = 0; j < 3; ++j) G[3 * i + j] = v_M[3 * i + j] * scale[j];   // v_Rq
  SO_UNROLL
  for (int j = 0; j < 3; ++j) v_scale[j] += Rq[j] * v_M[j] + Rq[3 + j] * v_M[3 + j] + Rq[6 + j] * v_M[6 + j];
  const T w = qn[0], qx = qn[1], qy = qn[2], qz = qn[3];
  T vq[4];
  vq[0] = 2 * (-qz * G[1] + qy * G[2] + qz * G[3] - qx * G[5] - qy * G[6] + qx * G[7]);
  vq[1] = 2 * (qy * G[1] + qz * G[2] + qy * G[3] - 2 * qx * G[4] - w * G[5] + qz * G[6] + w * G[7] - 2 * qx * G[8]);
  vq[2] = 2 * (-2 * qy * G[0] + qx * G[1] + w * G[2] + qx * G[3] + qz * G[5] - w * G[6] + qz * G[7] - 2 * qy * G[8]);
  vq[3] = 2 * (-2 * qz * G[0] - w * G[1] + qx * G[2] + w * G[3] - 2 * qz * G[4] + qy * G[5] + qx * G[6] + qy * G[7]);
  const T dot = vq[0] * w + vq[1] * qx + vq[2] * qy + vq[3] * qz;
  SO_UNROLL
  for (int k = 0; k < 4; ++k) v_quat[k] += (vq[k] - dot * qn[k]) * inv_norm;
}

// ---------------------------------------------------------------------------------------------
// spherical harmonics (3DGS sign convention), degree <= 4
// ---------------------------------------------------------------------------------------------
// Calls f(k, Y_k, dY_k/dx, dY_k/dy, dY_k/dz) for every basis k < (degree+1)^2 of the unit vector
// (x,y,z).  The derivatives are those of the polynomial form (their tangential part is what
// survives the normalisation backward); unused values are dead-code-eliminated after inlining.
template <typename T, typename F> SO_HD void sh_eval(int degree, T x, T y, T z, F &&f) {
  f(0, T(kShC0), T(0), T(0), T(0));
  if (degree < 1) return;
  const T c1 = T(kShC1);
  f(1, -c1 * y, T(0), T(-c1), T(0));
  f(2, c1 * z, T(0), T(0), T(c1));
  f(3, -c1 * x, T(-c1), T(0), T(0));
  if (degree < 2) return;
  const T xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  f(4, T(kShC2[0]) * xy, T(T(kShC2[0]) * y), T(T(kShC2[0]) * x), T(0));
  f(5, T(kShC2[1]) * yz, T(0), T(T(kShC2[1]) * z), T(T(kShC2[1]) * y));
  f(6, T(kShC2[2]) * (2 * zz - xx - yy), T(T(kShC2[2]) * (-2 * x)), T(T(kShC2[2]) * (-2 * y)), T(T(kShC2[2]) * (4 * z)));
  f(7, T(kShC2[3]) * xz, T(T(kShC2[3]) * z), T(0), T(T(kShC2[3]) * x));
  f(8, T(kShC2[4]) * (xx - yy), T(T(kShC2[4]) * (2 * x)), T(T(kShC2[4]) * (-2 * y)), T(0));
  if (degree < 3) return;
  f(9, T(kShC3[0]) * y * (3 * xx - yy), T(T(kShC3[0]) * 6 * xy), T(T(kShC3[0]) * (3 * xx - 3 * yy)), T(0));
  f(10, T(kShC3[1]) * xy * z, T(T(kShC3[1]) * yz), T(T(kShC3[1]) * xz), T(T(kShC3[1]) * xy));
  f(11, T(kShC3[2]) * y * (4 * zz - xx - yy), T(T(kShC3[2]) * (-2 * xy)), T(T(kShC3[2]) * (4 * zz - xx - 3 * yy)), T(T(kShC3[2]) * 8 * yz));
  f(12, T(kShC3[3]) * z * (2 * zz - 3 * xx - 3 * yy), T(T(kShC3[3]) * (-6 * xz)), T(T(kShC3[3]) * (-6 * yz)), T(T(kShC3[3]) * (6 * zz - 3 * xx - 3 * yy)));
  f(13, T(kShC3[4]) * x * (4 * zz - xx - yy), T(T(kShC3[4]) * (4 * zz - 3 * xx - yy)), T(T(kShC3[4]) * (-2 * xy)), T(T(kShC3[4]) * 8 * xz));
  f(14, T(kShC3[5]) * z * (xx - yy), T(T(kShC3[5]) * 2 * xz), T(T(kShC3[5]) * (-2 * yz)), T(T(kShC3[5]) * (xx - yy)));
  f(15, T(kShC3[6]) * x * (xx - 3 * yy), T(T(kShC3[6]) * (3 * xx - 3 * yy)), T(T(kShC3[6]) * (-6 * xy)), T(0));
  if (degree < 4) return;
  f(16, T(kShC4[0]) * xy * (xx - yy), T(T(kShC4[0]) * (3 * xx * y - yy * y)), T(T(kShC4[0]) * (xx * x - 3 * x * yy)), T(0));
  f(17, T(kShC4[1]) * yz * (3 * xx - yy), T(T(kShC4[1]) * 6 * xy * z), T(T(kShC4[1]) * z * (3 * xx - 3 * yy)), T(T(kShC4[1]) * y * (3 * xx - yy)));
  f(18, T(kShC4[2]) * xy * (7 * zz - 1), T(T(kShC4[2]) * y * (7 * zz - 1)), T(T(kShC4[2]) * x * (7 * zz - 1)), T(T(kShC4[2]) * 14 * xy * z));
  f(19, T(kShC4[3]) * yz * (7 * zz - 3), T(0), T(T(kShC4[3]) * z * (7 * zz - 3)), T(T(kShC4[3]) * y * (21 * zz - 3)));
  f(20, T(kShC4[4]) * (zz * (35 * zz - 30) + 3), T(0), T(0), T(T(kShC4[4]) * (140 * zz * z - 60 * z)));
  f(21, T(kShC4[5]) * xz * (7 * zz - 3), T(T(kShC4[5]) * z * (7 * zz - 3)), T(0), T(T(kShC4[5]) * x * (21 * zz - 3)));
  f(22, T(kShC4[6]) * (xx - yy) * (7 * zz - 1), T(T(kShC4[6]) * 2 * x * (7 * zz - 1)), T(T(kShC4[6]) * (-2 * y) * (7 * zz - 1)), T(T(kShC4[6]) * 14 * z * (xx - yy)));
  f(23, T(kShC4[7]) * xz * (xx - 3 * yy), T(T(kShC4[7]) * z * (3 * xx - 3 * yy)), T(T(kShC4[7]) * (-6 * xy * z)), T(T(kShC4[7]) * x * (xx - 3 * yy)));
  f(24, T(kShC4[8]) * (xx * (xx - 3 * yy) - yy * (3 * xx - yy)), T(T(kShC4[8]) * (4 * xx * x - 12 * x * yy)), T(T(kShC4[8]) * (4 * yy * y - 12 * xx * y)), T(0));
}

// bases Y[k]; when dY != null also dY[k][3]
template <typename T> SO_HD void sh_bases(int degree, T x, T y, T z, T *Y, T (*dY)[3]) {
  sh_eval<T>(degree, x, y, z, [&](int k, T yk, T dx, T dy, T dz) {
    Y[k] = yk;
    if (dY) { dY[k][0] = dx; dY[k][1] = dy; dY[k][2] = dz; }
  });
}

}  // namespace so
