// adam.hip -- fused multi-tensor Adam (+ zero_grad, + visibility mask) for gfx950.
//
// Replaces the six torch.optim.Adam steps and `zero_grad(set_to_none=True)` of
// /root/reference/utils/gsplat_utils/gsplat_trainer.py:726-731 (hyper-parameters :266-280) and
// gsplat `SelectiveAdam` (:269-270, :719-728) with ONE launch.  Pure HBM streaming:
// 16 B read + 12 B written per parameter float (+4 B when the gradient is zeroed in place).
#include "so_common.hpp"

extern "C" void so_profile_stage_begin_end(int stage, int begin, void *stream);

namespace so {

struct AdamGroups {
  so_adam_group g[SO_ADAM_MAX_GROUPS];
};

// skip (nullable): a float some collective has summed over the ranks -- non-zero: this iteration is void on every rank,
// parameters and moments stay as they are.  gscale: the gradient is multiplied by it on the fly (1 / world: the
// reduce-scatter delivered the SUM over the ranks' views).
__global__ void __launch_bounds__(256)
k_adam(AdamGroups groups, AdamHyper h, int zero_grad, const float *__restrict__ skip, float gscale) {
  if (skip && *skip != 0.f) return;
  const so_adam_group G = groups.g[blockIdx.y];
  const int64_t n4 = (G.row_len % 4 == 0 || !G.visibility) ? G.numel / 4 : 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float4 *p4 = reinterpret_cast<float4 *>(G.param);
  float4 *g4 = reinterpret_cast<float4 *>(G.grad);
  float4 *m4 = reinterpret_cast<float4 *>(G.exp_avg);
  float4 *v4 = reinterpret_cast<float4 *>(G.exp_avg_sq);
  for (int64_t i = t0; i < n4; i += stride) {
    if (G.visibility && !G.visibility[(i * 4) / G.row_len]) {
      if (zero_grad) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    float4 p = p4[i], g = ld_nt(g4 + i), m = ld_nt(m4 + i), v = ld_nt(v4 + i);
    adam_one(p.x, g.x * gscale, m.x, v.x, h, G.lr_step_size, G.bc2_sqrt);
    adam_one(p.y, g.y * gscale, m.y, v.y, h, G.lr_step_size, G.bc2_sqrt);
    adam_one(p.z, g.z * gscale, m.z, v.z, h, G.lr_step_size, G.bc2_sqrt);
    adam_one(p.w, g.w * gscale, m.w, v.w, h, G.lr_step_size, G.bc2_sqrt);
    p4[i] = p; st_nt(m4 + i, m); st_nt(v4 + i, v);
    if (zero_grad) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int64_t i = n4 * 4 + t0; i < G.numel; i += stride) {
    if (G.visibility && !G.visibility[i / G.row_len]) {
      if (zero_grad) G.grad[i] = 0.f;
      continue;
    }
    float p = G.param[i], m = G.exp_avg[i], v = G.exp_avg_sq[i];
    adam_one(p, G.grad[i] * gscale, m, v, h, G.lr_step_size, G.bc2_sqrt);
    G.param[i] = p; G.exp_avg[i] = m; G.exp_avg_sq[i] = v;
    if (zero_grad) G.grad[i] = 0.f;
  }
}

// ---- device-scheduled variant: step counter, LR schedule and bias corrections live on the device,
// so the launch arguments never change and the whole training step can be replayed as a hipGraph.
// (struct AdamSched and adam_schedule_block live in so_common.hpp: so_step_inputs runs the same schedule)

// one thread per group: evaluates the schedule for the current step, then advances the counter
__global__ void k_adam_prep(AdamSched sch, int n_groups, double beta1, double beta2, int32_t *__restrict__ step_ptr,
                            float2 *__restrict__ hyper) {
  adam_schedule_block(sch.lr0, sch.lr_gamma, n_groups, beta1, beta2, step_ptr, hyper);
}

// float16 shadow of a parameter group inside the attribute rows of attr_rec.hpp: element e of a group whose rows are
// RL elements long lives at arec + (e / RL) * stride + off + 2 * (e % RL).  RL is a template parameter for the row
// lengths of the six 3DGS tensors (division by a constant), 0 = run-time row length, -1 = no shadow.
struct AttrShadowDev {
  uint8_t *arec;
  int32_t stride;
  int32_t off[SO_ADAM_MAX_GROUPS];   // < 0: the group has no float16 copy
};

template <int RL>
__device__ __forceinline__ void shadow_store(uint8_t *arec, int stride, int off, int rl, uint32_t e, float p) {
  const uint32_t r = RL > 0 ? (uint32_t)RL : (uint32_t)rl;
  const uint32_t n = e / r, j = e - n * r;
  *reinterpret_cast<_Float16 *>(arec + (int64_t)n * stride + off + 2 * j) = (_Float16)p;
}

// four consecutive elements of one lane: inside one row they are 8 contiguous bytes of the float16 attribute row, written
// as two dwords, or (row offset 2 mod 4) as half | dword | half -- 2 or 3 store instructions instead of 4 two-byte ones
// (rows are 16-byte aligned: attr_rec.hpp); a group that straddles a row boundary keeps the element-wise stores
template <int RL>
__device__ __forceinline__ void shadow_store4(uint8_t *arec, int stride, int off, int rl, uint32_t e, const float4 p) {
  const uint32_t r = RL > 0 ? (uint32_t)RL : (uint32_t)rl;
  const uint32_t n = e / r, j = e - n * r;
  if (j + 4 <= r) {
    const uint32_t h0 = __builtin_bit_cast(unsigned short, (_Float16)p.x), h1 = __builtin_bit_cast(unsigned short, (_Float16)p.y);
    const uint32_t h2 = __builtin_bit_cast(unsigned short, (_Float16)p.z), h3 = __builtin_bit_cast(unsigned short, (_Float16)p.w);
    uint8_t *a = arec + (int64_t)n * stride + off + 2 * j;
    if (((uint32_t)(off >> 1) + j) & 1u) {
      *reinterpret_cast<unsigned short *>(a) = (unsigned short)h0;
      *reinterpret_cast<uint32_t *>(a + 2) = h1 | (h2 << 16);
      *reinterpret_cast<unsigned short *>(a + 6) = (unsigned short)h3;
    } else {
      *reinterpret_cast<uint32_t *>(a) = h0 | (h1 << 16);
      *reinterpret_cast<uint32_t *>(a + 4) = h2 | (h3 << 16);
    }
    return;
  }
  shadow_store<RL>(arec, stride, off, rl, e, p.x);
  shadow_store<RL>(arec, stride, off, rl, e + 1, p.y);
  shadow_store<RL>(arec, stride, off, rl, e + 2, p.z);
  shadow_store<RL>(arec, stride, off, rl, e + 3, p.w);
}

template <int RL>
__device__ __forceinline__ void adam_dev_loop(const so_adam_group G, const AdamHyper h, float step_size, float bc2_sqrt,
                                              int zero_grad, uint8_t *arec, int stride, int off) {
  const int64_t n4 = G.numel / 4;
  const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float4 *p4 = reinterpret_cast<float4 *>(G.param);
  float4 *g4 = reinterpret_cast<float4 *>(G.grad);
  float4 *m4 = reinterpret_cast<float4 *>(G.exp_avg);
  float4 *v4 = reinterpret_cast<float4 *>(G.exp_avg_sq);
  // gradient and moments are touched once per iteration: non-temporal loads / stores keep them from evicting
  // what the next kernels re-read (measured 737 -> 639 us at 2M Gaussians, 35.5 -> 33.9 us at 100k); the
  // parameters are stored normally, the next forward reads them
  for (int64_t i = t0; i < n4; i += gstride) {
    float4 p = p4[i], g = ld_nt(g4 + i), m = ld_nt(m4 + i), v = ld_nt(v4 + i);
    adam_one(p.x, g.x, m.x, v.x, h, step_size, bc2_sqrt);
    adam_one(p.y, g.y, m.y, v.y, h, step_size, bc2_sqrt);
    adam_one(p.z, g.z, m.z, v.z, h, step_size, bc2_sqrt);
    adam_one(p.w, g.w, m.w, v.w, h, step_size, bc2_sqrt);
    p4[i] = p; st_nt(m4 + i, m); st_nt(v4 + i, v);
    if (zero_grad) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (RL >= 0) {
      const uint32_t e = (uint32_t)i * 4u;
      if (RL == 4) {   // one row per float4: a single 8-byte store
        *reinterpret_cast<uint2 *>(arec + i * stride + off) =
            make_uint2((uint32_t)__builtin_bit_cast(unsigned short, (_Float16)p.x) | ((uint32_t)__builtin_bit_cast(unsigned short, (_Float16)p.y) << 16),
                       (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)p.z) | ((uint32_t)__builtin_bit_cast(unsigned short, (_Float16)p.w) << 16));
      } else {
        shadow_store4<RL>(arec, stride, off, G.row_len, e, p);
      }
    }
  }
  for (int64_t i = n4 * 4 + t0; i < G.numel; i += gstride) {
    float p = G.param[i], m = G.exp_avg[i], v = G.exp_avg_sq[i];
    adam_one(p, G.grad[i], m, v, h, step_size, bc2_sqrt);
    G.param[i] = p; G.exp_avg[i] = m; G.exp_avg_sq[i] = v;
    if (zero_grad) G.grad[i] = 0.f;
    if (RL >= 0) shadow_store<RL>(arec, stride, off, G.row_len, (uint32_t)i, p);
  }
}

__global__ void __launch_bounds__(256)
k_adam_dev(AdamGroups groups, AdamHyper h, const float2 *__restrict__ hyper, int zero_grad,
           const int32_t *__restrict__ skip_i32, const float *__restrict__ skip_f32, AttrShadowDev sh,
           const int32_t *__restrict__ n_rows_dev) {
  // a void iteration (binning overflow on this or, summed through the all-reduce, on any rank) leaves the
  // parameters and moments untouched
  if ((skip_i32 && *skip_i32 != 0) || (skip_f32 && *skip_f32 != 0.f)) return;
  so_adam_group G = groups.g[blockIdx.y];
  if (n_rows_dev) {   // the row count lives on the device (so_refine_default): numel is the capacity
    const int64_t live = (int64_t)*n_rows_dev * G.row_len;
    if (live < G.numel) G.numel = live;
  }
  const float2 hy = hyper[blockIdx.y];
  const int off = sh.arec ? sh.off[blockIdx.y] : -1;
  if (off < 0) { adam_dev_loop<-1>(G, h, hy.x, hy.y, zero_grad, nullptr, 0, 0); return; }
  switch (G.row_len) {   // uniform over the workgroup
    case 3: adam_dev_loop<3>(G, h, hy.x, hy.y, zero_grad, sh.arec, sh.stride, off); break;
    case 4: adam_dev_loop<4>(G, h, hy.x, hy.y, zero_grad, sh.arec, sh.stride, off); break;
    case 45: adam_dev_loop<45>(G, h, hy.x, hy.y, zero_grad, sh.arec, sh.stride, off); break;
    default: adam_dev_loop<0>(G, h, hy.x, hy.y, zero_grad, sh.arec, sh.stride, off); break;
  }
}

}  // namespace so

static int adam_step_dev_impl(int n_groups, const so_adam_group *host_groups, const float *host_lr0,
                              const float *host_lr_gamma, double beta1, double beta2, double eps,
                              int32_t *step_counter, int zero_grad, int schedule_done, const int32_t *skip_i32,
                              const float *skip_f32, const so_attr_shadow *shadow, const int32_t *n_rows_dev, void *stream) {
  SO_REQUIRE(n_groups >= 0 && n_groups <= SO_ADAM_MAX_GROUPS, "so_adam_step_dev: n_groups %d not in [0,%d]", n_groups, SO_ADAM_MAX_GROUPS);
  SO_REQUIRE(step_counter, "so_adam_step_dev: null step counter");
  if (n_groups == 0) return SO_OK;
  SO_REQUIRE(host_groups && host_lr0 && host_lr_gamma, "so_adam_step_dev: null groups");
  so::AdamGroups G;
  so::AdamSched S;
  int64_t max_numel = 0;
  for (int i = 0; i < n_groups; ++i) {
    const so_adam_group &g = host_groups[i];
    SO_REQUIRE(g.numel >= 0 && g.visibility == nullptr, "so_adam_step_dev: group %d bad numel / visibility unsupported", i);
    SO_REQUIRE(g.numel == 0 || (g.param && g.grad && g.exp_avg && g.exp_avg_sq), "so_adam_step_dev: group %d null pointer", i);
    SO_REQUIRE((((uintptr_t)g.param | (uintptr_t)g.grad | (uintptr_t)g.exp_avg | (uintptr_t)g.exp_avg_sq) & 15) == 0,
               "so_adam_step_dev: group %d buffers must be 16-byte aligned", i);
    if (shadow && shadow->arec && shadow->offset_bytes[i] >= 0)
      SO_REQUIRE(g.row_len >= 1 && g.numel % g.row_len == 0 && g.numel < ((int64_t)1 << 31) &&
                     shadow->offset_bytes[i] + 2 * g.row_len <= shadow->stride_bytes &&
                     shadow->offset_bytes[i] % (g.row_len == 4 ? 8 : 2) == 0,
                 "so_adam_step_dev_shadow: group %d does not fit the float16 rows (row_len %d, offset %d, stride %d)", i,
                 g.row_len, shadow->offset_bytes[i], shadow->stride_bytes);
    G.g[i] = g;
    S.lr0[i] = host_lr0[i];
    S.lr_gamma[i] = host_lr_gamma[i];
    if (g.numel > max_numel) max_numel = g.numel;
  }
  hipStream_t st = so::as_stream(stream);
  so_profile_stage_begin_end(8, 1, stream);
  // hyper[] lives right behind the step counter: step_counter[0] = step, [2..2+2*n) = (step_size, bc2_sqrt)
  float2 *hyper = reinterpret_cast<float2 *>(step_counter + 2);
  if (!schedule_done)   // otherwise so_step_inputs has evaluated this step's schedule already
    hipLaunchKernelGGL(so::k_adam_prep, dim3(1), dim3(SO_ADAM_MAX_GROUPS), 0, st, S, n_groups, beta1, beta2, step_counter, hyper);
  if (max_numel > 0) {
    int64_t gx = so::ceil_div(so::ceil_div(max_numel, 4), 256);
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
    const so::AdamHyper H{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps};
    so::AttrShadowDev SH{};
    if (shadow && shadow->arec) {
      SO_REQUIRE((((uintptr_t)shadow->arec) & 15) == 0 && shadow->stride_bytes > 0 && shadow->stride_bytes % 16 == 0,
                 "so_adam_step_dev_shadow: arec must be 16-byte aligned with a stride that is a multiple of 16");
      SH.arec = reinterpret_cast<uint8_t *>(shadow->arec);
      SH.stride = shadow->stride_bytes;
      for (int i = 0; i < SO_ADAM_MAX_GROUPS; ++i) SH.off[i] = i < n_groups ? shadow->offset_bytes[i] : -1;
    }
    hipLaunchKernelGGL(so::k_adam_dev, dim3((unsigned)gx, (unsigned)n_groups), dim3(256), 0, st, G, H, hyper, zero_grad, skip_i32, skip_f32, SH, n_rows_dev);
  }
  so_profile_stage_begin_end(8, 0, stream);
  return so::check_launch("so_adam_step_dev");
}

extern "C" int so_adam_step_dev_shadow(int n_groups, const so_adam_group *host_groups, const float *host_lr0,
                                       const float *host_lr_gamma, double beta1, double beta2, double eps,
                                       int32_t *step_counter, int zero_grad, int schedule_done, const int32_t *skip_i32,
                                       const float *skip_f32, const so_attr_shadow *shadow, void *stream) {
  return adam_step_dev_impl(n_groups, host_groups, host_lr0, host_lr_gamma, beta1, beta2, eps, step_counter, zero_grad,
                            schedule_done, skip_i32, skip_f32, shadow, nullptr, stream);
}

extern "C" int so_adam_step_dev_n(int n_groups, const so_adam_group *host_groups, const float *host_lr0,
                                  const float *host_lr_gamma, double beta1, double beta2, double eps,
                                  int32_t *step_counter, int zero_grad, int schedule_done, const int32_t *skip_i32,
                                  const float *skip_f32, const int32_t *n_rows_dev, void *stream) {
  for (int i = 0; n_rows_dev && i < n_groups && host_groups; ++i)
    SO_REQUIRE(host_groups[i].row_len >= 1 && host_groups[i].numel % host_groups[i].row_len == 0,
               "so_adam_step_dev_n: group %d: numel must be capacity x row_len", i);
  return adam_step_dev_impl(n_groups, host_groups, host_lr0, host_lr_gamma, beta1, beta2, eps, step_counter, zero_grad,
                            schedule_done, skip_i32, skip_f32, nullptr, n_rows_dev, stream);
}

extern "C" int so_adam_step_dev(int n_groups, const so_adam_group *host_groups, const float *host_lr0,
                                const float *host_lr_gamma, double beta1, double beta2, double eps,
                                int32_t *step_counter, int zero_grad, int schedule_done, const int32_t *skip_i32,
                                const float *skip_f32, void *stream) {
  return so_adam_step_dev_shadow(n_groups, host_groups, host_lr0, host_lr_gamma, beta1, beta2, eps, step_counter,
                                 zero_grad, schedule_done, skip_i32, skip_f32, nullptr, stream);
}

static int adam_step_impl(int n_groups, const so_adam_group *host_groups, double beta1, double beta2, double eps, int zero_grad,
                          const float *skip, float gscale, void *stream);
extern "C" int so_adam_step(int n_groups, const so_adam_group *host_groups, double beta1, double beta2, double eps,
                            int zero_grad, void *stream) {
  return adam_step_impl(n_groups, host_groups, beta1, beta2, eps, zero_grad, nullptr, 1.f, stream);
}
extern "C" int so_adam_step_scaled(int n_groups, const so_adam_group *host_groups, double beta1, double beta2, double eps,
                                   int zero_grad, const float *skip_f32, float grad_scale, void *stream) {
  return adam_step_impl(n_groups, host_groups, beta1, beta2, eps, zero_grad, skip_f32, grad_scale, stream);
}
static int adam_step_impl(int n_groups, const so_adam_group *host_groups, double beta1, double beta2, double eps, int zero_grad,
                          const float *skip, float gscale, void *stream) {
  SO_REQUIRE(n_groups >= 0 && n_groups <= SO_ADAM_MAX_GROUPS, "so_adam_step: n_groups %d not in [0,%d]", n_groups, SO_ADAM_MAX_GROUPS);
  if (n_groups == 0) return SO_OK;
  SO_REQUIRE(host_groups, "so_adam_step: null groups");
  so::AdamGroups G;
  int64_t max_numel = 0;
  for (int i = 0; i < n_groups; ++i) {
    const so_adam_group &g = host_groups[i];
    SO_REQUIRE(g.numel >= 0 && g.row_len >= 1, "so_adam_step: group %d bad numel/row_len", i);
    SO_REQUIRE(g.numel == 0 || (g.param && g.grad && g.exp_avg && g.exp_avg_sq), "so_adam_step: group %d null pointer", i);
    SO_REQUIRE((((uintptr_t)g.param | (uintptr_t)g.grad | (uintptr_t)g.exp_avg | (uintptr_t)g.exp_avg_sq) & 15) == 0,
               "so_adam_step: group %d buffers must be 16-byte aligned", i);
    G.g[i] = g;
    if (g.numel > max_numel) max_numel = g.numel;
  }
  if (max_numel == 0) return SO_OK;
  int64_t gx = so::ceil_div(so::ceil_div(max_numel, 4), 256);
  if (gx > 2048) gx = 2048;
  if (gx < 1) gx = 1;
  const so::AdamHyper H{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps};
  hipLaunchKernelGGL(so::k_adam, dim3((unsigned)gx, (unsigned)n_groups), dim3(256), 0, so::as_stream(stream), G, H,
                     zero_grad, skip, gscale);
  return so::check_launch("so_adam_step");
}
