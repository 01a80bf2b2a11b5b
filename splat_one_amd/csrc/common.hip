// common.hip -- error channel and device queries of libsplat_one_amd.so
#include "so_common.hpp"
#include "rasterize_common.hpp"

namespace so {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace so

extern "C" int so_abi_version(void) { return SO_ABI_VERSION; }
extern "C" const char *so_last_error(void) { return so::g_err; }
extern "C" int so_device_cu_count(void) {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return n;
}

// ---- test hook for the wave-level reduction primitives (tests/test_gpu_ops.py) ----------------
namespace so {
__global__ void __launch_bounds__(64) k_debug_wave_reduce(const float *__restrict__ in, float *__restrict__ out) {
  const int lane = threadIdx.x;
  const float *row = in + ((int64_t)blockIdx.x * 64 + lane) * 9;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = row[k];
  const float e = row_reduce8_transposed(v, lane);
  const float o = row_allreduce_sum(row[8]);
  const float c = wave_reduce_sum(row[8]);
  float *dst = out + (int64_t)blockIdx.x * 10;
  // exactly the store pattern of the rasteriser backward: rows combined, lanes 0..8 add once
  const int l15 = lane & 15;
  const float val = rows_combine(l15 < 8 ? e : o);
  if (lane <= 8) atomicAdd(dst + (lane < 8 ? slot_of_lane(lane) : 8), val);
  if (lane == 17) dst[9] = c;
}
}  // namespace so

namespace so {
__global__ void __launch_bounds__(64) k_debug_wave_reduce9(const float *__restrict__ in, float *__restrict__ out) {
  const int lane = threadIdx.x;
  const float *row = in + ((int64_t)blockIdx.x * 64 + lane) * 9;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = row[k];
  const float w = wave_reduce9_scattered(v, row[8]);
  // exactly the store pattern of the rasteriser backward: nine lanes, nine distinct slots
  if ((kReduce9Lanes >> lane) & 1ull) atomicAdd(out + (int64_t)blockIdx.x * 9 + reduce9_slot_of_lane(lane), w);
}
}  // namespace so

/* in[n_waves*64, 9] -> out[n_waves, 9] (zeroed by the caller): the 64-lane sums of the nine columns (wave_reduce9_scattered) */
extern "C" int so_debug_wave_reduce9(int n_waves, const float *in, float *out, void *stream) {
  SO_REQUIRE(n_waves >= 0 && (n_waves == 0 || (in && out)), "so_debug_wave_reduce9: bad arguments");
  if (n_waves == 0) return SO_OK;
  hipLaunchKernelGGL(so::k_debug_wave_reduce9, dim3(n_waves), dim3(64), 0, so::as_stream(stream), in, out);
  return so::check_launch("so_debug_wave_reduce9");
}

namespace so {
__global__ void k_debug_cull(int64_t n, const float *__restrict__ in, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *p = in + 10 * i;   // mx, my, opacity, ca, cb, cc, x0, x1, y0, y1
  const float tau = cull_tau(p[2], p[3], p[4], p[5]);
  const float4 box = alpha_bound_box(p[0], p[1], p[2], p[3], p[4], p[5]);
  const bool box_hit = !(box.y < p[6] || box.x > p[7] || box.w < p[8] || box.z > p[9]);
  out[2 * i] = box_hit ? 1.f : 0.f;
  out[2 * i + 1] = ellipse_hits_rect(p[0], p[1], tau, p[3], p[4], p[5], p[6], p[7], p[8], p[9]) ? 1.f : 0.f;
}
}  // namespace so

/* test hook for the rasterisers' per-quadrant culling: in[n][10] = {mx, my, opacity, conic a, b, c, rectangle x0, x1, y0, y1}
 * (pixel centres) -> out[n][2] = {the box test, the exact test} as 0 / 1 (tests/test_gpu_ops.py checks both against a
 * float64 minimisation: neither may miss a rectangle in which alpha reaches 1/255) */
extern "C" int so_debug_cull(int64_t n, const float *in, float *out, void *stream) {
  SO_REQUIRE(n >= 0 && (n == 0 || (in && out)), "so_debug_cull: bad arguments");
  if (n == 0) return SO_OK;
  hipLaunchKernelGGL(so::k_debug_cull, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, so::as_stream(stream), n, in, out);
  return so::check_launch("so_debug_cull");
}

namespace so {
__global__ void k_debug_bin_counter_index(int64_t M, int64_t *__restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < M) out[t] = bin_counter_index(t, M);
}
}  // namespace so

/* test hook: out[t] = where the DEVICE code keeps the binned count of tile t of M (so_common.hpp bin_counter_index: a float32
 * quotient estimate put right exactly) -- must equal the host's so_bin_counter_index(t, M) for every t */
extern "C" int so_debug_bin_counter_index(int64_t M, int64_t *out, void *stream) {
  SO_REQUIRE(M >= 0 && (M == 0 || out), "so_debug_bin_counter_index: bad arguments");
  if (M == 0) return SO_OK;
  hipLaunchKernelGGL(so::k_debug_bin_counter_index, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, so::as_stream(stream), M, out);
  return so::check_launch("so_debug_bin_counter_index");
}

/* in[n_waves*64, 9] -> out[n_waves, 10] (zeroed by the caller): slots 0..7 += transposing butterfly
 * row sums of columns 0..7, 8 += row sums of column 8, 9 = wave sum of column 8 (row_bcast DPP form). */
extern "C" int so_debug_wave_reduce(int n_waves, const float *in, float *out, void *stream) {
  SO_REQUIRE(n_waves >= 0 && (n_waves == 0 || (in && out)), "so_debug_wave_reduce: bad arguments");
  if (n_waves == 0) return SO_OK;
  hipLaunchKernelGGL(so::k_debug_wave_reduce, dim3(n_waves), dim3(64), 0, so::as_stream(stream), in, out);
  return so::check_launch("so_debug_wave_reduce");
}

// ---- camera-to-world -> world-to-camera (general 4x4 inverse, one lane per camera) -------------
// Replaces `viewmats=torch.linalg.inv(camtoworlds)` of gsplat_trainer.py:483 (a dozen rocSOLVER
// launches per step for a batch of 4x4 matrices) with one launch.
namespace so {
__global__ void k_camera_inverse(int C, const float *__restrict__ c2w, float *__restrict__ w2c) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  camera_inverse_one(c2w + 16 * c, w2c + 16 * c);
}
}  // namespace so

extern "C" int so_camera_inverse(int C, const float *camtoworlds, float *viewmats, void *stream) {
  SO_REQUIRE(C >= 0 && (C == 0 || (camtoworlds && viewmats)), "so_camera_inverse: bad arguments");
  if (C == 0) return SO_OK;
  hipLaunchKernelGGL(so::k_camera_inverse, dim3((C + 63) / 64), dim3(64), 0, so::as_stream(stream), C, camtoworlds, viewmats);
  return so::check_launch("so_camera_inverse");
}
