// preprocess.hip -- fused per-Gaussian front end / back end of the training step for gfx950.
//
// The reference does, per iteration, exp(scales), sigmoid(opacities), cat(sh0, shN), inv(camtoworld)
// in `Runner.rasterize_splats` (/root/reference/utils/gsplat_utils/gsplat_trainer.py:456-474) and
// then, inside `rasterization` (:477), projection, SH evaluation (+0.5, clamp) and the first
// binning pass as separate launches with materialised intermediates.  Here ONE forward kernel reads
// the RAW parameters once and writes everything the tile rasteriser needs (and the per-tile
// histogram of the binning pass); ONE backward kernel turns the rasteriser's gradients into
// gradients of the raw parameters (projection bwd + SH bwd + activation bwd + regularisers) and
// accumulates the densification statistics of `DefaultStrategy._update_state` on the way.
// Same math as projection.hip / sh.hip (splat_math.hpp); one lane per (camera, Gaussian) forward,
// one lane per Gaussian backward (no atomics on parameter gradients).
#include "so_common.hpp"
#include "attr_rec.hpp"
#include "splat_math.hpp"
#include "rasterize_common.hpp"   // alpha_bound_box: the rasterisers' cull box, computed once per Gaussian into the record

#include <cstdlib>
#include <type_traits>

#ifndef SO_PP_EARLY_ATOMICS
#define SO_PP_EARLY_ATOMICS 1   // k_preprocess_fwd: the binning atomics of small rectangles leave before the colour is evaluated
#endif

namespace so {

struct CamP {
  float Rw[9], tw[3], fx, fy, cx, cy, pos[3];
};

__device__ __forceinline__ CamP load_camp(const float *__restrict__ viewmats, const float *__restrict__ Ks, int c) {
  CamP p;
  const float *V = viewmats + 16 * c;
  p.Rw[0] = V[0]; p.Rw[1] = V[1]; p.Rw[2] = V[2];
  p.Rw[3] = V[4]; p.Rw[4] = V[5]; p.Rw[5] = V[6];
  p.Rw[6] = V[8]; p.Rw[7] = V[9]; p.Rw[8] = V[10];
  p.tw[0] = V[3]; p.tw[1] = V[7]; p.tw[2] = V[11];
  const float *K = Ks + 9 * c;
  p.fx = K[0]; p.fy = K[4]; p.cx = K[2]; p.cy = K[5];
  // camera centre of a rigid world->camera transform: -R^T t  (== inverse(viewmat)[:3,3])
#pragma unroll
  for (int j = 0; j < 3; ++j) p.pos[j] = -(p.Rw[j] * p.tw[0] + p.Rw[3 + j] * p.tw[1] + p.Rw[6 + j] * p.tw[2]);
  return p;
}

// one model for all views, or SO_CAM_PER_VIEW with 2 bits per view
__device__ __forceinline__ int cam_model_of(int cm, int c) { return (cm & SO_CAM_PER_VIEW) ? ((cm >> (2 * c)) & 3) : cm; }

// does any view use the spherical model?  (the kernels are built with and without it: splat_math.hpp SPH)
static bool camera_model_has_spherical(int cm, int C) {
  if (!(cm & SO_CAM_PER_VIEW)) return cm == SO_CAM_SPHERICAL;
  for (int c = 0; c < C; ++c)
    if (((cm >> (2 * c)) & 3) == SO_CAM_SPHERICAL) return true;
  return false;
}

static bool camera_model_ok(int cm, int C) {
  if (!(cm & SO_CAM_PER_VIEW)) return cm >= 0 && cm <= SO_CAM_SPHERICAL;
  if (C > SO_CAM_PER_VIEW_MAX) return false;
  for (int c = 0; c < C; ++c)
    if (((cm >> (2 * c)) & 3) > SO_CAM_SPHERICAL) return false;
  return true;
}

__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// -DPP_STAMPS (tools/gpu_ppstamps.sh, not the product build): s_memrealtime stamps (100 MHz) of the phases of
// k_preprocess_fwd, one row per wave, read back with so_debug_pp_stamps_read.
#ifdef PP_STAMPS
__device__ unsigned long long g_pp_stamps[16384 * 8];
#define PP_STAMP(k)                                                                                             \
  do {                                                                                                          \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                 \
    if (first_trip && (threadIdx.x & 63) == 0) {                                                                \
      const int wv = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));                                \
      if (wv < 16384) g_pp_stamps[wv * 8 + (k)] = wall_clock64();                                               \
    }                                                                                                           \
  } while (0)
__device__ unsigned long long g_ppb_stamps[16384 * 8];
#define PPB_STAMP(k)                                                                                            \
  do {                                                                                                          \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                 \
    if (first_trip && (threadIdx.x & 63) == 0) {                                                                \
      const int wv = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));                                \
      if (wv < 16384) g_ppb_stamps[wv * 8 + (k)] = wall_clock64();                                              \
    }                                                                                                           \
  } while (0)
#else
#define PP_STAMP(k) do { } while (0)
#define PPB_STAMP(k) do { } while (0)
#endif

// shN rows of a wave read as ONE contiguous run (COOP).  A lane owns one 3 (K - 1)-float row of shN
// (180 bytes at SH degree 3): read lane by lane, every load instruction touches 64 different lines and a wave pulls its 11.5 KB
// through the vector cache in 64-byte crumbs (k_preprocess_bwd had the mirror problem for its WRITES and solved it the same
// way).  Here the wave's 64 rows -- contiguous in memory, row stride == row size -- come in as whole 1 KB pieces by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, lane l of piece i lands at byte 1024 i + 16 l of the wave's LDS region) and
// every lane then reads ITS row from LDS (row stride 45 floats: odd, no bank conflicts).  Only full waves inside one camera
// whose first row is 16-byte aligned take this path; the others (the last wave of the live rows, camera seams of a batch with
// N % 64 != 0) read lane by lane as before.  The coefficients are the same floats either way: results are bit-identical.
__device__ __forceinline__ unsigned lds_byte_address(const void *p) {   // generic -> LDS address (an addrspacecast)
#if defined(__HIP_DEVICE_COMPILE__)
  return (unsigned)(uintptr_t)((const __attribute__((address_space(3))) char *)p);
#else
  return 0u;
#endif
}
struct CoefsLdsRow {
  float c0[3];          // sh0 of this Gaussian, loaded with the other per-Gaussian values (before the binning atomics leave: a
                        // global load behind them would have to wait for them -- the vector memory counter is in order)
  unsigned row;         // byte address of this lane's shN row in LDS
  __device__ __forceinline__ void get(int k, float c[3]) const {
    if (k == 0) { c[0] = c0[0]; c[1] = c0[1]; c[2] = c0[2]; return; }
#if defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((address_space(3))) float *r =
        reinterpret_cast<const __attribute__((address_space(3))) float *>(static_cast<uintptr_t>(row)) + 3 * (k - 1);
    c[0] = r[0]; c[1] = r[1]; c[2] = r[2];
#else
    c[0] = c[1] = c[2] = 0.f;
#endif
  }
};

template <int DEG, class A, bool SPH, bool COOP>
__global__ void __launch_bounds__(256)
k_preprocess_fwd(int C, int N, const float *__restrict__ means, const float *__restrict__ logit_opac, const A attrs,
                 const float *__restrict__ viewmats,
                 const float *__restrict__ Ks, int W, int H, float eps2d, float near_plane, float far_plane,
                 float radius_clip, int model, int antialiased, float tile_size, int tile_w, int tile_h,
                 int32_t *__restrict__ radii, float *__restrict__ means2d, float *__restrict__ depths,
                 float *__restrict__ conics, float *__restrict__ opacities, float *__restrict__ colors,
                 int32_t *__restrict__ tiles_per_gauss, int32_t *__restrict__ tile_counts,
                 float4 *__restrict__ rec, float4 *__restrict__ vrec, int64_t cam_stride,
                 int32_t *__restrict__ tile_slots, int tile_cull, uint64_t *__restrict__ bin_keys, int64_t bin_cap,
                 int32_t *__restrict__ bin_overflow, const int32_t *__restrict__ n_dev, int wrap_ok,
                 int32_t *__restrict__ sub_counts, int replicas) {
  // n_dev (nullable): the number of Gaussians lives in device memory (device-side densification, so_refine_default):
  // N is then the CAPACITY of the buffers, rows n >= *n_dev are idle -- a captured launch follows N without re-capture
  const int n_live = n_dev ? min(*n_dev, N) : N;
  const int64_t total = (int64_t)C * N;
  const int n_tiles = tile_w * tile_h;
  // Replicated bin counters (so_step_desc.bin_replicas > 1: images of few tiles, where the returning atomics of one counter
  // serialise -- tools/census/xcd_atomics.hip: 4.4 G/s on 1024 counters against 17 G/s on eight copies of them): this workgroup
  // bumps copy `rep` of a tile's counter and fills slice `rep` of the tile's bin; k_bins_gather closes the slices up before the sort.
  const int rep = replicas > 1 ? (int)(blockIdx.x % (unsigned)replicas) : 0;
  int32_t *const bcnt = replicas > 1 ? sub_counts + (int64_t)rep * ((int64_t)C * n_tiles) : tile_counts;
  const int64_t bcap = replicas > 1 ? bin_cap / replicas : bin_cap;      // slots of a slice
  const int64_t boff = (int64_t)rep * bcap;
  // the trip count is uniform over the workgroup: the histogram section below needs every lane of a wave
  for (int64_t lin0 = (int64_t)blockIdx.x * blockDim.x; lin0 < total; lin0 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t lin = lin0 + threadIdx.x;
#ifdef PP_STAMPS
    const bool first_trip = lin0 == (int64_t)blockIdx.x * blockDim.x;
#endif
    PP_STAMP(0);
    bool staged = false;          // wave-uniform: this wave's 64 shN rows are (being) staged in its LDS region
    unsigned my_lds_row = 0;
    if constexpr (COOP) {
      extern __shared__ __attribute__((aligned(16))) float s_rows[];
      const int R = 3 * (attrs.K - 1);
      const int wv = threadIdx.x >> 6, lane = lane_id();
      const int64_t lin_w = lin0 + (int64_t)wv * 64;                 // lin of lane 0 (wave-uniform)
      const int64_t c_w = lin_w / N, n_w = lin_w - c_w * N;
      // full wave, one camera, live rows only, and the run 16-byte aligned (R odd: first row a multiple of 4)
      staged = lin_w + 64 <= total && n_w + 64 <= (int64_t)n_live && ((n_w * R) & 3) == 0;
      staged = __builtin_amdgcn_readfirstlane((int)staged) != 0;
      float *mine = s_rows + (size_t)wv * 64 * R;
      if (staged) {
#if defined(__HIP_DEVICE_COMPILE__)
        wave_lds_sync();                                             // the previous trip's row reads are done
        const float4 *src = reinterpret_cast<const float4 *>(attrs.shN + n_w * R);
        const int total4 = 16 * R;                                   // 64 R floats = 16 R float4 (R = 45: 720 = 11.25 pieces)
        for (int i = 0; i * 64 < total4; ++i) {                      // (uniform trip count; the last piece is lane-masked)
          const int q = i * 64 + lane;
          if (q < total4)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) void *>(reinterpret_cast<uintptr_t>(src + q)),
                                             reinterpret_cast<__attribute__((address_space(3))) void *>(static_cast<uintptr_t>(lds_byte_address(mine) + (unsigned)i * 1024u)),
                                             16, 0, 0);
        }
#endif
      }
      my_lds_row = lds_byte_address(mine) + (unsigned)lane * (unsigned)R * 4u;
    }
    int cnt = 0, bx0 = 0, bx1 = 0, by0 = 0, by1 = 0, c = 0;
    float cmx = 0.f, cmy = 0.f, cqa = 0.f, cqb = 0.f, cqc = 0.f, ctau = 0.f;   // what the exact tile test needs
    float cdepth = 0.f;
    int64_t idx = 0;
    int32_t got[SO_TILE_SLOTS];   // binned lists: the slots the returning atomics of a small rectangle hand out (issued EARLY, below)
    bool early = false;
    if (lin < total && (lin - (lin / N) * N) < n_live) {
    c = (int)(lin / N);
    const int64_t n = lin - (int64_t)c * N;
    idx = (int64_t)c * cam_stride + n;   // row of the per-view arrays (cam_stride >= N) = the flatten id of the lists
    const CamP cam = load_camp(viewmats, Ks, c);
    const float mean[3] = {means[3 * n], means[3 * n + 1], means[3 * n + 2]};
    float q[4], ls[3];
    attrs.base(n, q, ls);
    const float s[3] = {A::kActivated ? ls[0] : expf(ls[0]), A::kActivated ? ls[1] : expf(ls[1]), A::kActivated ? ls[2] : expf(ls[2])};
    ProjOut<float> o;
    project_fwd<float, SPH>(mean, nullptr, q, s, cam.Rw, cam.tw, cam.fx, cam.fy, cam.cx, cam.cy, W, H, eps2d, near_plane,
                       far_plane, radius_clip, cam_model_of(model, c), o);
    // radii == NULL ("record-only views", the fused engine): the per-view arrays are not written at all -- the 64-byte
    // record holds everything the rasteriser, the backward and the caller's `info` need (48 B per Gaussian and view less:
    // 173 -> 142 us at 1M Gaussians)
    if (radii) {
      radii[idx] = o.radius;
      *reinterpret_cast<float2 *>(means2d + 2 * idx) = make_float2(o.m2d[0], o.m2d[1]);
      depths[idx] = o.depth;
      conics[3 * idx] = o.conic[0]; conics[3 * idx + 1] = o.conic[1]; conics[3 * idx + 2] = o.conic[2];
    }
    float op = A::kActivated ? logit_opac[n] : sigmoidf(logit_opac[n]);
    if (antialiased) op *= o.comp;
    if (radii) opacities[idx] = op;
    float r = 0.f, g = 0.f, b = 0.f;
    float sh0v[3] = {0.f, 0.f, 0.f};
    if constexpr (COOP) { sh0v[0] = attrs.sh0[3 * n]; sh0v[1] = attrs.sh0[3 * n + 1]; sh0v[2] = attrs.sh0[3 * n + 2]; }
#ifdef PP_STAMPS
    if (o.radius == -12345) r = 1.f;      // (keeps the projection ahead of the stamp)
#endif
    PP_STAMP(1);
    if (o.radius > 0) {
      // first binning pass (same float arithmetic as isect.hip::tile_box)
      const float tile_r = (float)o.radius / tile_size;
      const float tx = o.m2d[0] / tile_size, ty = o.m2d[1] / tile_size;
      int x0 = (int)fminf(fmaxf(floorf(tx - tile_r), 0.f), (float)tile_w);
      int x1 = (int)fminf(fmaxf(ceilf(tx + tile_r), 0.f), (float)tile_w);
      if (SPH && wrap_ok && cam_model_of(model, c) == SO_CAM_SPHERICAL) {
        // a panorama is periodic in x: VIRTUAL tile columns (isect.hip::tile_box), filed under wrapx(x) below
        x0 = (int)fmaxf(floorf(tx - tile_r), (float)-tile_w);
        x1 = (int)fminf(ceilf(tx + tile_r), (float)(2 * tile_w));
        if (x1 - x0 > tile_w) { x0 = (int)ceilf(tx - 0.5f * (float)tile_w - 0.5f); x1 = x0 + tile_w; }   // (isect.hip::tile_box)
      }
      const int y0 = (int)fminf(fmaxf(floorf(ty - tile_r), 0.f), (float)tile_h);
      const int y1 = (int)fminf(fmaxf(ceilf(ty + tile_r), 0.f), (float)tile_h);
      cnt = (x1 - x0) * (y1 - y0);
      bx0 = x0; bx1 = x1; by0 = y0; by1 = y1;
      cmx = o.m2d[0]; cmy = o.m2d[1]; cqa = o.conic[0]; cqb = o.conic[1]; cqc = o.conic[2];
      ctau = cull_tau(op);
      cdepth = o.depth;
    }
    if constexpr (COOP) {
      // the rows must have landed before any lane reads its own -- and before the NEXT trip's DMA may overwrite them, so this
      // wait is taken by every staged wave, whether or not one of its Gaussians is visible (an LDS-DMA is a pending write on
      // the VM counter; staged waves are full, so all 64 lanes are here)
      if (staged) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_sync();
#endif
      }
    }
    if (SO_PP_EARLY_ATOMICS && bin_keys && tile_counts && cnt > 0 && cnt <= SO_TILE_SLOTS) {
      // Binned lists, small rectangle: the returning atomics leave NOW -- behind the last global load this lane needs (the shN rows
      // are in LDS, sh0 in registers) -- and are waited for after the colour has been evaluated and the records stored: the
      // ~12 us a wave spent waiting for its slots (s_memrealtime stamps, round 2) run under ~6 us of its own work (round 5).
      early = true;
      int x = bx0, y = by0;
#pragma unroll
      for (int i = 0; i < SO_TILE_SLOTS; ++i) {
        got[i] = -1;
        if (i < cnt) {
          if (!tile_cull || tile_touches(cmx, cmy, cqa, cqb, cqc, ctau, x, y, tile_size))
            got[i] = atomicAdd(bcnt + bin_counter_index((int64_t)c * n_tiles + y * tile_w + wrapx(x, tile_w), (int64_t)C * n_tiles), 1);
          if (++x == bx1) { x = bx0; ++y; }
        }
      }
    }
    if (o.radius > 0) {
      float dx = mean[0] - cam.pos[0], dy = mean[1] - cam.pos[1], dz = mean[2] - cam.pos[2];
      const float inorm = rsqrtf(dx * dx + dy * dy + dz * dz);
      dx *= inorm; dy *= inorm; dz *= inorm;
      auto colour_from = [&](const auto &coef) {
        sh_eval<float>(DEG, dx, dy, dz, [&](int k, float yk, float, float, float) {
          float cf[3];
          coef.get(k, cf);
          r += yk * cf[0]; g += yk * cf[1]; b += yk * cf[2];
        });
      };
      if constexpr (COOP) {
        if (staged) {      // (wave-uniform)
          colour_from(CoefsLdsRow{{sh0v[0], sh0v[1], sh0v[2]}, my_lds_row});
        } else {
          colour_from(attrs.template coefs<DEG>(n));
        }
      } else {
        colour_from(attrs.template coefs<DEG>(n));
      }
      r = fmaxf(r + 0.5f, 0.f); g = fmaxf(g + 0.5f, 0.f); b = fmaxf(b + 0.5f, 0.f);
    }
    if (radii) {
      colors[3 * idx] = r; colors[3 * idx + 1] = g; colors[3 * idx + 2] = b;
      tiles_per_gauss[idx] = cnt;
    }
#ifdef PP_STAMPS
    if (r == -12345.f) cnt = 0;
#endif
    PP_STAMP(2);
#ifdef PP_NO_REC
    if (rec && o.depth == 12345.678f) {
#else
    if (rec) {   // 64-byte record gathered by the rasteriser: one cache line per Gaussian
#endif
      rec[4 * idx] = make_float4(o.m2d[0], o.m2d[1], o.conic[0], o.conic[1]);
      rec[4 * idx + 1] = make_float4(o.conic[2], op, r, g);
      rec[4 * idx + 2] = make_float4(b, o.depth, __int_as_float(o.radius), cull_tau(op, o.conic[0], o.conic[1], o.conic[2]));
      rec[4 * idx + 3] = alpha_bound_box(o.m2d[0], o.m2d[1], op, o.conic[0], o.conic[1], o.conic[2]);
    }
    if (vrec) {  // gradient record, accumulated atomically by the rasteriser backward
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      vrec[4 * idx] = z; vrec[4 * idx + 1] = z; vrec[4 * idx + 2] = z; vrec[4 * idx + 3] = z;
    }
    PP_STAMP(3);
    }   // lin < total, n < n_live
    else if (n_dev && radii && lin < total) {
      // rows between the live count and the capacity: the kernels that read the per-view arrays over all N rows (the
      // compact lists' scatter pass, so_isect_fill) must see them as invisible -- after a prune they still hold the
      // radii of Gaussians that no longer exist
      const int64_t cdead = lin / N;
      const int64_t idead = cdead * cam_stride + (lin - cdead * N);
      radii[idead] = 0;
      tiles_per_gauss[idead] = 0;
    }
    if (tile_counts) {   // null: the caller bins later (Gaussian-sharded runs bin after the exchange)
      // Histogram of the first binning pass.  A lane walks a small rectangle itself; a large one (the dense
      // init regime: ~70 tiles per Gaussian) is spread over the whole wave in 8x8 tile blocks, so the wave's
      // trip count follows the total work, not its largest rectangle.
      constexpr int kOwn = SO_TILE_SLOTS;
      const bool big = cnt > kOwn;
      if (cnt > 0 && !big) {
        int32_t *row = tile_counts + (int64_t)c * n_tiles;
        if (bin_keys) {
          // binned lists (so_step_desc.bin_capacity): every tile owns bin_cap key slots, the returning atomic IS the
          // slot -- the key goes straight to its place and neither a scan nor a scatter pass exists
          // Two sweeps over the rectangle: ALL returning atomics are issued before the first slot is consumed (one
          // s_waitcnt for the lane instead of one per tile).  s_memrealtime stamps (tools/dbg_ppstamps.py): the binning
          // section is 15.0 -> 11.6 of a wave's 23.5 -> 21.6 us; what remains is the throughput of the memory-side atomic
          // units with every wave of the launch in this section at once (357k returning atomics in ~12 us).
          const uint64_t key = ((uint64_t)__float_as_uint(cdepth) << 32) | (uint64_t)(uint32_t)idx;
          int x = bx0, y = by0;
          if (!early) {
#pragma unroll
            for (int i = 0; i < kOwn; ++i) {
              got[i] = -1;
              if (i < cnt) {
                if (!tile_cull || tile_touches(cmx, cmy, cqa, cqb, cqc, ctau, x, y, tile_size))
                  got[i] = atomicAdd(bcnt + bin_counter_index((int64_t)c * n_tiles + y * tile_w + wrapx(x, tile_w), (int64_t)C * n_tiles), 1);
                if (++x == bx1) { x = bx0; ++y; }
              }
            }
          }
          x = bx0; y = by0;
          bool over = false;
#pragma unroll
          for (int i = 0; i < kOwn; ++i) {
            if (i < cnt) {
              const int32_t s = got[i];
              if (s >= 0) {
                const int64_t t = (int64_t)c * n_tiles + y * tile_w + wrapx(x, tile_w);
                if (s < bcap) bin_keys[t * bin_cap + boff + s] = key;
                else over = true;
              }
              if (++x == bx1) { x = bx0; ++y; }
            }
          }
          if (over) *bin_overflow = 1;
        } else if (tile_slots) {
          // the value an atomic returns IS this Gaussian's slot in that tile's list: keep it, and the scatter pass
          // (k_isect_scatter) places the key at offsets[tile] + slot without a second round of atomics.  All
          // requests are issued before the first result is consumed (one round trip, not cnt of them).
          int32_t got[kOwn] = {};
          int x = bx0, y = by0;
#pragma unroll
          for (int i = 0; i < kOwn; ++i) {
            if (i < cnt) {
              if (!tile_cull || tile_touches(cmx, cmy, cqa, cqb, cqc, ctau, x, y, tile_size)) got[i] = atomicAdd(row + y * tile_w + wrapx(x, tile_w), 1);
              if (++x == bx1) { x = bx0; ++y; }
            }
          }
          int32_t *mine = tile_slots + idx * kOwn;
#pragma unroll
          for (int i = 0; i < kOwn; ++i)
            if (i < cnt) mine[i] = got[i];
        } else {
          for (int y = by0; y < by1; ++y)
            for (int x = bx0; x < bx1; ++x)
              if (!tile_cull || tile_touches(cmx, cmy, cqa, cqb, cqc, ctau, x, y, tile_size)) atomicAdd(row + y * tile_w + wrapx(x, tile_w), 1);
        }
      }
      unsigned long long todo = __ballot(big);
      const int lane = lane_id();
      while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int sx0 = __builtin_amdgcn_readlane(bx0, src), sx1 = __builtin_amdgcn_readlane(bx1, src);
        const int sy0 = __builtin_amdgcn_readlane(by0, src), sy1 = __builtin_amdgcn_readlane(by1, src);
        // with tile_slots the large rectangles are counted apart (second half of tile_counts): a tile's list is then
        // [slotted small entries | large entries], and the scatter fills the second part from the back
        int32_t *row = tile_counts + (int64_t)__builtin_amdgcn_readlane(c, src) * n_tiles + (tile_slots ? (int64_t)C * n_tiles : 0);
        const float smx = readlane_f(cmx, src), smy = readlane_f(cmy, src), sqa = readlane_f(cqa, src),
                    sqb = readlane_f(cqb, src), sqc = readlane_f(cqc, src), stau = readlane_f(ctau, src);
        const uint64_t skey = ((uint64_t)__float_as_uint(readlane_f(cdepth, src)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)idx, src);
        const int64_t srow = (int64_t)__builtin_amdgcn_readlane(c, src) * n_tiles;
        for (int y = sy0 + (lane >> 3); y < sy1; y += 8)
          for (int x = sx0 + (lane & 7); x < sx1; x += 8)
            if (!tile_cull || tile_touches(smx, smy, sqa, sqb, sqc, stau, x, y, tile_size)) {
              if (bin_keys) {
                const int64_t t = srow + y * tile_w + wrapx(x, tile_w);
                const int32_t s = atomicAdd(bcnt + bin_counter_index(t, (int64_t)C * n_tiles), 1);
                if (s < bcap) bin_keys[t * bin_cap + boff + s] = skey;
                else *bin_overflow = 1;
              } else {
                atomicAdd(row + y * tile_w + wrapx(x, tile_w), 1);
              }
            }
      }
    }
    PP_STAMP(4);
  }
}

#ifndef SO_PP_SWEEP_BATCH
#define SO_PP_SWEEP_BATCH 12
#endif
// LDS floats per Gaussian for the float16 row header (quat 4, log-scale 3, sh0 3) when the fused optimiser also re-packs the
// rows; odd, so that lanes writing their own header do not share banks
constexpr int kHalfHdr = 11;

// (One wave per SIMD: 256 VGPRs + AGPRs.  Forcing two with __launch_bounds__(256, 2) spills 49 registers
// and measured slower, 23.1 vs 20.5 us at 100k Gaussians.)
template <int DEG, class A, bool STAGE, bool ADAM, bool SPH>
__global__ void __launch_bounds__(256, (ADAM && A::kHalfRows) ? 2 : 1)
k_preprocess_bwd(int C, int N, int K, const float *__restrict__ means, const float *__restrict__ logit_opac,
                 const A attrs, const float *__restrict__ viewmats,
                 const float *__restrict__ Ks, int W, int H, float eps2d, int model, int antialiased,
                 const int32_t *__restrict__ radii, const float *__restrict__ opacities,
                 const float *__restrict__ colors, const float *__restrict__ v_means2d,
                 const float *__restrict__ v_means2d_abs, const float *__restrict__ v_depths,
                 const float *__restrict__ v_conics, const float *__restrict__ v_colors,
                 const float *__restrict__ v_opacities, float opacity_reg, float scale_reg,
                 float *__restrict__ v_means, float *__restrict__ v_log_scales, float *__restrict__ v_quats,
                 float *__restrict__ v_logit_opac, float *__restrict__ v_sh0, float *__restrict__ v_shN,
                 float *__restrict__ grad2d, float *__restrict__ count, float stat_sx, float stat_sy,
                 const float4 *__restrict__ vrec, int use_abs_stats, int64_t cam_stride,
                 const int32_t *__restrict__ skip_flag, float *__restrict__ skip_out, const AdamFuse af,
                 const int32_t *__restrict__ n_dev, const float4 *__restrict__ rec, int64_t row_begin, int64_t row_end) {
  constexpr int NB = (DEG + 1) * (DEG + 1);
  if (n_dev) N = min(*n_dev, N);   // device-resident Gaussian count (see k_preprocess_fwd); the grid covers the capacity
  // rows [row_begin, row_end) only (row_end <= 0: to the end; row_begin a multiple of 64): data-parallel steps cut the
  // backward into row chunks so that the reduce-scatter of chunk i runs under chunk i + 1 (so_train_step_bwd_rows).
  // N stays the live count (the regularisers are means over all Gaussians).
  const int64_t n_hi = (row_end > 0 && row_end < (int64_t)N) ? row_end : (int64_t)N;
  static_assert(!ADAM || STAGE, "the fused optimiser works on the staged shN rows");
  // skip_flag (nullable): the binning pass overflowed its buffers -> this iteration is void: leave gradients and
  // densification statistics alone (the optimiser step skips too); skip_out (nullable) publishes the flag as a
  // float that a gradient all-reduce can sum across ranks
  const bool skip = skip_flag && *skip_flag != 0;
  if (skip_out && blockIdx.x == 0 && threadIdx.x == 0) *skip_out = skip ? 1.f : 0.f;
  if (skip) return;
  // v_shN staging (stage_rows > 0: dynamic LDS of blockDim.x * stage_rows floats): a lane owns one 180-byte row of
  // v_shN, so direct stores put 16 bytes per lane at a 180-byte stride -- 64 partial-line writes per instruction
  // (170 us at 1M Gaussians, 123 us without these stores).  Each wave transposes its 64 rows through LDS and writes
  // them as one contiguous 11.5 KB run of full lines instead.
  extern __shared__ __attribute__((aligned(16))) float s_stage[];
  const int R = 3 * (K - 1);
  // the trip count is uniform over the workgroup (the staged write-out is cooperative)
  for (int64_t n0 = row_begin + (int64_t)blockIdx.x * blockDim.x; n0 < n_hi; n0 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = n0 + threadIdx.x;
    const bool active = n < n_hi;
#ifdef PP_STAMPS
    const bool first_trip = n0 == row_begin + (int64_t)blockIdx.x * blockDim.x;
#endif
    PPB_STAMP(0);
    float acc[NB][3];
    // staged: the 3 (NB - 1) gradient sums of shN live in this lane's LDS row from the start (45 registers fewer)
    float *const my_row = s_stage + ((size_t)(threadIdx.x >> 6) * 64 + (threadIdx.x & 63)) * R;
    bool first = true;
    if (active) {
    const float mean[3] = {means[3 * n], means[3 * n + 1], means[3 * n + 2]};
    float q[4], ls[3];
    attrs.base(n, q, ls);
    const float s[3] = {A::kActivated ? ls[0] : expf(ls[0]), A::kActivated ? ls[1] : expf(ls[1]), A::kActivated ? ls[2] : expf(ls[2])};
    const float sig = A::kActivated ? logit_opac[n] : sigmoidf(logit_opac[n]);
    const auto coef = attrs.template coefs<DEG>(n);
    float vm[3] = {0.f, 0.f, 0.f}, vq[4] = {0.f, 0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f};
    float v_sig = 0.f, g2 = 0.f, cn = 0.f;
#pragma unroll
    for (int k = 0; k < NB; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.f;
    const bool use_abs = use_abs_stats != 0;
    for (int c = 0; c < C; ++c) {
      const int64_t idx = (int64_t)c * cam_stride + n;
      // rec != NULL (record-only views): radius, blended opacity and clamped colour come from the 64-byte record
      float4 rq1 = make_float4(0.f, 0.f, 0.f, 0.f), rq2 = rq1;
      if (rec) {
        rq2 = rec[4 * idx + 2];
        if (__float_as_int(rq2.z) <= 0) continue;
        rq1 = rec[4 * idx + 1];
      } else if (radii[idx] <= 0) continue;
      const CamP cam = load_camp(viewmats, Ks, c);
      float2 vm2;
      float v_con[3], v_op, vr, vg, vb, abs_x = 0.f, abs_y = 0.f;
      if (vrec) {
        const float4 q0 = vrec[4 * idx], q1 = vrec[4 * idx + 1], q2 = vrec[4 * idx + 2];
        vm2 = make_float2(q0.x, q0.y);
        v_con[0] = q0.z; v_con[1] = q0.w; v_con[2] = q1.x;
        vr = q1.y; vg = q1.z; vb = q1.w;
        v_op = q2.x; abs_x = q2.y; abs_y = q2.z;
      } else {
        vm2 = *reinterpret_cast<const float2 *>(v_means2d + 2 * idx);
        v_con[0] = v_conics[3 * idx]; v_con[1] = v_conics[3 * idx + 1]; v_con[2] = v_conics[3 * idx + 2];
        v_op = v_opacities[idx];
        vr = v_colors[3 * idx]; vg = v_colors[3 * idx + 1]; vb = v_colors[3 * idx + 2];
        if (v_means2d_abs) { abs_x = v_means2d_abs[2 * idx]; abs_y = v_means2d_abs[2 * idx + 1]; }
      }
      const float v_m2d[2] = {vm2.x, vm2.y};
      float v_comp = 0.f;
      if (antialiased) {
        v_comp = v_op * sig;
        v_sig += v_op * ((rec ? rq1.y : opacities[idx]) / sig);
      } else {
        v_sig += v_op;
      }
      project_bwd<float, SPH>(mean, nullptr, q, s, cam.Rw, cam.tw, cam.fx, cam.fy, cam.cx, cam.cy, W, H, eps2d, cam_model_of(model, c),
                         v_m2d, v_depths ? v_depths[idx] : 0.f, v_con, v_comp, vm, nullptr, vq, vs, nullptr, nullptr);
      // SH backward (through +0.5 / clamp: the saved colour is 0 exactly where the clamp cut)
      if (!((rec ? rq1.z : colors[3 * idx]) > 0.f)) vr = 0.f;
      if (!((rec ? rq1.w : colors[3 * idx + 1]) > 0.f)) vg = 0.f;
      if (!((rec ? rq2.x : colors[3 * idx + 2]) > 0.f)) vb = 0.f;
      const float ddx = mean[0] - cam.pos[0], ddy = mean[1] - cam.pos[1], ddz = mean[2] - cam.pos[2];
      const float inorm = rsqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
      const float x = ddx * inorm, y = ddy * inorm, z = ddz * inorm;
      float vdn[3] = {0.f, 0.f, 0.f};
      sh_eval<float>(DEG, x, y, z, [&](int k, float yk, float gx, float gy, float gz) {
        if (STAGE && k > 0) {
          float *rp = my_row + 3 * (k - 1);
          rp[0] = first ? yk * vr : rp[0] + yk * vr;
          rp[1] = first ? yk * vg : rp[1] + yk * vg;
          rp[2] = first ? yk * vb : rp[2] + yk * vb;
        } else {
          acc[k][0] += yk * vr; acc[k][1] += yk * vg; acc[k][2] += yk * vb;
        }
        float cf[3];
        coef.get(k, cf);
        const float w = cf[0] * vr + cf[1] * vg + cf[2] * vb;
        vdn[0] += gx * w; vdn[1] += gy * w; vdn[2] += gz * w;
      });
      first = false;
      const float dot = vdn[0] * x + vdn[1] * y + vdn[2] * z;
      vm[0] += (vdn[0] - dot * x) * inorm;
      vm[1] += (vdn[1] - dot * y) * inorm;
      vm[2] += (vdn[2] - dot * z) * inorm;
      // densification statistics (DefaultStrategy._update_state)
      if (grad2d) {
        float gx = use_abs ? abs_x : vm2.x, gy = use_abs ? abs_y : vm2.y;
        gx *= stat_sx; gy *= stat_sy;
        g2 += sqrtf(gx * gx + gy * gy);
        cn += 1.f;
      }
    }
    PPB_STAMP(1);
    // d/d log s = s * d/ds ; scale regulariser: scale_reg * mean|exp(s)| over 3N entries
    // (AttrAct: the inputs ARE the activated values -- their gradients go out as they are)
    const float sreg = scale_reg / (3.f * (float)N);
    const float gs[3] = {(vs[0] + sreg) * (A::kActivated ? 1.f : s[0]), (vs[1] + sreg) * (A::kActivated ? 1.f : s[1]),
                         (vs[2] + sreg) * (A::kActivated ? 1.f : s[2])};
    const float go = (v_sig + opacity_reg / (float)N) * (A::kActivated ? 1.f : sig * (1.f - sig));
    if (ADAM) {
      // Fused optimiser (so_step_desc.fuse_adam): the gradient of this Gaussian is in registers (and, for shN, in
      // its LDS row) -- apply Adam here instead of writing 236 B of gradient for the Adam kernel to read back.
      // Same arithmetic as adam.hip (adam_one); hyper[g] = (step size, sqrt(bias correction 2)) of group g.
      // All 42 values (parameter + two moments of the five small tensors) are requested BEFORE the first one is used: one
      // memory round trip for the lane instead of one per tensor (s_memrealtime stamps, tools/dbg_ppstamps.py: this
      // section was 9.3 us of a wave's 33 us with load -> update -> store per tensor).
      constexpr int kLen[5] = {3, 3, 4, 1, 3};
      const int64_t at[5] = {3 * n, 3 * n, 4 * n, n, 3 * n};
      float pv[5][4], mv[5][4], vv[5][4];
#pragma unroll
      for (int g = 0; g < 5; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < kLen[g]) { pv[g][j] = af.p[g][at[g] + j]; mv[g][j] = af.m[g][at[g] + j]; vv[g][j] = af.v[g][at[g] + j]; }
      const float *grads[5] = {vm, gs, vq, &go, acc[0]};
#pragma unroll
      for (int g = 0; g < 5; ++g) {
        const float2 hy = af.hyper[g];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < kLen[g]) adam_one(pv[g][j], grads[g][j], mv[g][j], vv[g][j], af.h, hy.x, hy.y);
      }
#pragma unroll
      for (int g = 0; g < 5; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < kLen[g]) { af.p[g][at[g] + j] = pv[g][j]; af.m[g][at[g] + j] = mv[g][j]; af.v[g][at[g] + j] = vv[g][j]; }
      if (A::kHalfRows && af.half_rows) {   // float16 rows: the new quaternion, log-scale and sh0 wait in LDS for the wave's row sweep below
        float *hd = s_stage + (size_t)blockDim.x * R + (size_t)threadIdx.x * kHalfHdr;
#pragma unroll
        for (int j = 0; j < 4; ++j) hd[j] = pv[2][j];
#pragma unroll
        for (int j = 0; j < 3; ++j) { hd[4 + j] = pv[1][j]; hd[7 + j] = pv[4][j]; }
      }
    } else {
      v_means[3 * n] = vm[0]; v_means[3 * n + 1] = vm[1]; v_means[3 * n + 2] = vm[2];
      *reinterpret_cast<float4 *>(v_quats + 4 * n) = make_float4(vq[0], vq[1], vq[2], vq[3]);
      v_log_scales[3 * n] = gs[0]; v_log_scales[3 * n + 1] = gs[1]; v_log_scales[3 * n + 2] = gs[2];
      v_logit_opac[n] = go;
      v_sh0[3 * n] = acc[0][0]; v_sh0[3 * n + 1] = acc[0][1]; v_sh0[3 * n + 2] = acc[0][2];
    }
    if (grad2d) { grad2d[n] += g2; count[n] += cn; }
    PPB_STAMP(2);
    if (!STAGE) {
      float *o = v_shN + n * (int64_t)R;
#pragma unroll
      for (int k = 1; k < NB; ++k) { o[3 * (k - 1)] = acc[k][0]; o[3 * (k - 1) + 1] = acc[k][1]; o[3 * (k - 1) + 2] = acc[k][2]; }
      for (int k = 3 * (NB - 1); k < R; ++k) o[k] = 0.f;
    }
    }   // active
    if (STAGE && R > 0) {
      const int lane = lane_id(), wv = threadIdx.x >> 6;
      float *mine = s_stage + (size_t)wv * 64 * R;        // this wave's 64 rows, row stride R (odd for K = 16: no bank conflicts)
      if (active) {
        if (first)                                        // seen by no camera: the sums were never started
          for (int k = 0; k < 3 * (NB - 1); ++k) my_row[k] = 0.f;
        for (int k = 3 * (NB - 1); k < R; ++k) my_row[k] = 0.f;
      }
      // (a wave sweeps the 64 rows its own lanes wrote: the exchange is inside the wave, so a wave-local LDS fence is all
      // it takes -- __syncthreads() also made every wave wait for the slowest one of the workgroup and for its own stores
      // of the small tensors above to be acknowledged, vmcnt(0))
      wave_lds_sync();
      PPB_STAMP(3);
      const int64_t w0 = n0 + (int64_t)wv * 64;            // first Gaussian of this wave: a multiple of 64 -> 16-byte aligned run
      const int64_t rows = n_hi - w0 < 64 ? n_hi - w0 : 64;
      if (rows > 0) {
        const int total = (int)rows * R;
        const float4 *src4 = reinterpret_cast<const float4 *>(mine);
        float4 *const upd4 = reinterpret_cast<float4 *>(mine);   // float16 rows: the updated shN replaces its gradient in LDS
        if (ADAM) {   // the wave's 64 rows of shN, its moments and (in LDS) its gradient: one coalesced sweep
          const float2 hy = af.hyper[5];
          float4 *p4 = reinterpret_cast<float4 *>(af.p[5] + w0 * R), *m4 = reinterpret_cast<float4 *>(af.m[5] + w0 * R),
                 *v4 = reinterpret_cast<float4 *>(af.v[5] + w0 * R);
          // The sweep in batches of kSweep float4 per lane: ALL loads of a batch (parameter + two moments: 3 kSweep requests per
          // lane, at clamped addresses, no exec-masked blocks) leave before the first value is used, then the updates and the
          // stores follow -- two memory round trips for a wave's 11.25 float4 per lane instead of twelve (round 5; a plain loop
          // waits load -> update -> store per trip at 1.5 waves per SIMD, and `#pragma unroll 4` of it gained nothing in round 2
          // because the stores of trip i may alias the loads of trip i + 1 for all the compiler knows).  The registers are free
          // here: the projection state is dead.
          // (12 float4 per lane = a full wave's rows in ONE round trip: c2 43.6 -> 41.0 us with 12, 41.6 with 6, 43.4 with 3,
          // profiles/r05_experiments.json; the float16-row variant is short of registers already and takes 4)
          constexpr int kSweep = A::kHalfRows ? 4 : SO_PP_SWEEP_BATCH;
          const int n4 = total / 4;
          for (int i0 = 0; i0 < n4; i0 += 64 * kSweep) {   // (uniform)
            float4 pp[kSweep], mm[kSweep], vv[kSweep];
#pragma unroll
            for (int u = 0; u < kSweep; ++u) {
              const int ic = min(i0 + u * 64 + lane, n4 - 1);
              pp[u] = p4[ic]; mm[u] = ld_nt(m4 + ic); vv[u] = ld_nt(v4 + ic);
            }
#pragma unroll
            for (int u = 0; u < kSweep; ++u) {
              const int i = i0 + u * 64 + lane;
              if (i < n4) {
                const float4 g = src4[i];
                adam_one(pp[u].x, g.x, mm[u].x, vv[u].x, af.h, hy.x, hy.y);
                adam_one(pp[u].y, g.y, mm[u].y, vv[u].y, af.h, hy.x, hy.y);
                adam_one(pp[u].z, g.z, mm[u].z, vv[u].z, af.h, hy.x, hy.y);
                adam_one(pp[u].w, g.w, mm[u].w, vv[u].w, af.h, hy.x, hy.y);
                p4[i] = pp[u]; st_nt(m4 + i, mm[u]); st_nt(v4 + i, vv[u]);
                if (A::kHalfRows && af.half_rows) upd4[i] = pp[u];
              }
            }
          }
          for (int i = (total & ~3) + lane; i < total; i += 64) {
            float pj = af.p[5][w0 * R + i], mj = af.m[5][w0 * R + i], vj = af.v[5][w0 * R + i];
            adam_one(pj, mine[i], mj, vj, af.h, hy.x, hy.y);
            af.p[5][w0 * R + i] = pj; af.m[5][w0 * R + i] = mj; af.v[5][w0 * R + i] = vj;
            if (A::kHalfRows && af.half_rows) mine[i] = pj;
          }
          if (A::kHalfRows && af.half_rows) {
            // The wave's 64 float16 rows are one contiguous run (row stride == row size): written as whole 16-byte pieces,
            // eight halves each, from the float32 values now in LDS -- the same round-to-nearest-even as so_attr_pack_f16,
            // so the rows equal masters.half() bit for bit.  (The scatter of 2-byte stores from inside the Adam kernel
            // cost 350 us at 2M Gaussians and the separate re-pack pass re-reads 224 B per Gaussian; this adds no read.)
            wave_lds_sync();
            const int parts = af.half_stride16;
            uint4 *out = reinterpret_cast<uint4 *>(af.half_rows) + w0 * parts;
            const float *hdr = s_stage + (size_t)blockDim.x * R + (size_t)wv * 64 * kHalfHdr;
            for (int q = lane; q < (int)rows * parts; q += 64) {
              const int r = q / parts, part = q - r * parts;
              float hv[8];
              if (part == 0) {
#pragma unroll
                for (int j = 0; j < 7; ++j) hv[j] = hdr[r * kHalfHdr + j];
                hv[7] = 0.f;
              } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  const int e = (part - 1) * 8 + j;
                  hv[j] = e < 3 ? hdr[r * kHalfHdr + 7 + e] : (e < 3 * K ? mine[r * R + (e - 3)] : 0.f);
                }
              }
              out[q] = make_uint4(pack_h2(hv[0], hv[1]), pack_h2(hv[2], hv[3]), pack_h2(hv[4], hv[5]), pack_h2(hv[6], hv[7]));
            }
          }
        } else {
          float4 *dst4 = reinterpret_cast<float4 *>(v_shN + w0 * R);
          for (int i = lane; i < total / 4; i += 64) dst4[i] = src4[i];
          for (int i = (total & ~3) + lane; i < total; i += 64) v_shN[w0 * R + i] = mine[i];
        }
      }
      PPB_STAMP(4);
      wave_lds_sync();   // the rows are rewritten by the next grid-stride trip of the same wave
    }
  }
}

// Gaussian-sharded data parallelism: after the all-to-all a rank holds the 64-byte records of ALL Gaussians
// for its own camera.  Unpack what the binning kernels read (centre, radius, depth) and clear the gradient
// records the rasteriser backward accumulates into.
__global__ void __launch_bounds__(256)
k_rec_unpack(int64_t n, const float4 *__restrict__ rec, float2 *__restrict__ means2d, int32_t *__restrict__ radii,
             float *__restrict__ depths, float4 *__restrict__ vrec) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 q0 = rec[4 * i], q2 = rec[4 * i + 2];
    means2d[i] = make_float2(q0.x, q0.y);
    depths[i] = q2.y;
    radii[i] = __float_as_int(q2.z);
    if (vrec) {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      vrec[4 * i] = z; vrec[4 * i + 1] = z; vrec[4 * i + 2] = z; vrec[4 * i + 3] = z;
    }
  }
}

// Operator-level rasterize_to_pixels (RGB): the separate per-row arrays packed into the 64-byte records the packed
// rasteriser kernels gather (one line per list entry instead of four), and the gradient records spread back out.
__global__ void __launch_bounds__(256)
k_rec_pack(int64_t n, const float2 *__restrict__ means2d, const float *__restrict__ conics,
           const float *__restrict__ colors, const float *__restrict__ opacities, float4 *__restrict__ rec,
           float4 *__restrict__ vrec) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float2 m = means2d[i];
    rec[4 * i] = make_float4(m.x, m.y, conics[3 * i], conics[3 * i + 1]);
    float r = 0.f, g = 0.f, b = 0.f;
    if (colors) { r = colors[3 * i]; g = colors[3 * i + 1]; b = colors[3 * i + 2]; }
    rec[4 * i + 1] = make_float4(conics[3 * i + 2], opacities[i], r, g);
    rec[4 * i + 2] = make_float4(b, 0.f, 0.f, cull_tau(opacities[i], conics[3 * i], conics[3 * i + 1], conics[3 * i + 2]));
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    rec[4 * i + 3] = alpha_bound_box(m.x, m.y, opacities[i], conics[3 * i], conics[3 * i + 1], conics[3 * i + 2]);
    if (vrec) { vrec[4 * i] = z; vrec[4 * i + 1] = z; vrec[4 * i + 2] = z; vrec[4 * i + 3] = z; }
  }
}

__global__ void __launch_bounds__(256)
k_rec_unpack_grads(int64_t n, const float4 *__restrict__ vrec, float2 *__restrict__ v_means2d,
                   float *__restrict__ v_conics, float *__restrict__ v_colors, float *__restrict__ v_opacities,
                   float2 *__restrict__ v_means2d_abs) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 q0 = vrec[4 * i], q1 = vrec[4 * i + 1], q2 = vrec[4 * i + 2];
    v_means2d[i] = make_float2(q0.x, q0.y);
    if (v_conics) {     // (so_rasterization_bwd wants the screen-space gradient -- the densification statistic -- only)
      v_conics[3 * i] = q0.z; v_conics[3 * i + 1] = q0.w; v_conics[3 * i + 2] = q1.x;
      v_colors[3 * i] = q1.y; v_colors[3 * i + 1] = q1.z; v_colors[3 * i + 2] = q1.w;
      v_opacities[i] = q2.x;
    }
    if (v_means2d_abs) v_means2d_abs[i] = make_float2(q2.y, q2.z);
  }
}

// Gaussian-sharded steps: "the binning pass of my view overflowed" has to reach every rank before any optimiser
// runs (the gradients that view sent to the other shards are incomplete).  It travels with the gradient records:
// slot 15 of the first record of every block of vrec_full (the rasteriser's backward accumulates into slots 0..10
// and so_preprocess_bwd reads 0..11), so no collective is added to the step.  One wave; world <= 64.
__global__ void __launch_bounds__(64)
k_shard_flag(int world, int64_t cap, float *__restrict__ vrec, int32_t *__restrict__ flag, int put) {
  const int j = threadIdx.x;
  const int64_t at = (int64_t)j * cap * 16 + 15;
  if (put) {
    if (j < world) vrec[at] = (*flag != 0) ? 1.f : 0.f;
    return;
  }
  const bool any = (j < world) && vrec[at] != 0.f;
  if (__ballot(any) != 0ull && j == 0) *flag = 1;
}

static inline int pp_grid(int64_t total) {
  int64_t g = ceil_div(total, 256);
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// float32 SoA -> float16 attribute rows (attr_rec.hpp): one lane per 16-byte chunk, coalesced stores
__global__ void __launch_bounds__(256)
k_attr_pack_f16(int64_t N, int K, int stride16, const float *__restrict__ log_scales, const float *__restrict__ quats,
                const float *__restrict__ sh0, const float *__restrict__ shN, uint4 *__restrict__ arec,
                const int32_t *__restrict__ n_dev) {
  if (n_dev) N = min((int64_t)*n_dev, N);   // device-resident Gaussian count: the grid covers the capacity
  const int64_t total = N * stride16;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / stride16;
    const int t = (int)(i - n * stride16);
    float v[8];
    if (t == 0) {
      const float4 q = *reinterpret_cast<const float4 *>(quats + 4 * n);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      v[4] = log_scales[3 * n]; v[5] = log_scales[3 * n + 1]; v[6] = log_scales[3 * n + 2]; v[7] = 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int h = 8 * (t - 1) + j;   // half index in the SH part: sh0 (3) then shN (3 (K-1))
        v[j] = h < 3 ? sh0[3 * n + h] : (h < 3 * K ? shN[n * (int64_t)(3 * (K - 1)) + (h - 3)] : 0.f);
      }
    }
    arec[i] = make_uint4(pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7]));
  }
}

template <class A>
static int preprocess_fwd_impl(const char *what, int C, int N, int K, int sh_degree, const float *means,
                               const float *logit_opacities, const A &attrs, const float *viewmats, const float *Ks,
                               int width, int height, float eps2d, float near_plane, float far_plane, float radius_clip,
                               int camera_model, int antialiased, int tile_size, int32_t *radii, float *means2d,
                               float *depths, float *conics, float *opacities, float *colors, int32_t *tiles_per_gauss,
                               int32_t *tile_counts, float *rec, float *vrec, int64_t cam_stride, int32_t *tile_slots,
                               int tile_cull, uint64_t *bin_keys, int64_t bin_cap, int32_t *bin_overflow, void *stream,
                               const int32_t *n_dev = nullptr, int32_t *sub_counts = nullptr, int replicas = 1) {
  SO_REQUIRE(C >= 0 && N >= 0 && K >= 1 && width > 0 && height > 0 && tile_size > 0, "%s: bad sizes", what);
  SO_REQUIRE(replicas <= 1 || (bin_keys && sub_counts && replicas <= 64 && bin_cap % replicas == 0),
             "%s: replicated bin counters need binned lists, the counter copies and bin_cap %% replicas == 0", what);
  SO_REQUIRE(tile_slots == nullptr || tile_counts != nullptr, "%s: tile_slots need the histogram (tile_counts)", what);
  SO_REQUIRE(sh_degree >= 0 && sh_degree <= 4 && (sh_degree + 1) * (sh_degree + 1) <= K,
             "%s: sh_degree %d does not fit K=%d", what, sh_degree, K);
  if (!camera_model_ok(camera_model, C)) {
    set_error("%s: unsupported camera_model 0x%x for %d views", what, camera_model, C);
    return SO_ERR_UNSUPPORTED;
  }
  if ((int64_t)C * N == 0) return SO_OK;
  SO_REQUIRE(means && logit_opacities && viewmats && Ks, "%s: null pointer", what);
  const bool lean = !radii && !means2d && !depths && !conics && !opacities && !colors && !tiles_per_gauss;
  SO_REQUIRE(lean ? (rec != nullptr) : (radii && means2d && depths && conics && opacities && colors && tiles_per_gauss),
             "%s: the per-view arrays are all given, or all NULL together with rec (record-only views)", what);
  if (cam_stride == 0) cam_stride = N;
  SO_REQUIRE(cam_stride >= N, "%s: cam_stride %lld < N %d", what, (long long)cam_stride, N);
  SO_REQUIRE(tile_slots == nullptr || cam_stride == N, "%s: tile_slots need densely packed views (cam_stride == N)", what);
  SO_REQUIRE(bin_keys == nullptr || (tile_counts && bin_overflow && bin_cap > 0 && tile_slots == nullptr &&
                                     (int64_t)C * cam_stride < ((int64_t)1 << 31)),
             "%s: bin_keys need tile_counts, bin_overflow, bin_cap > 0, C * cam_stride < 2^31 and no tile_slots", what);
  const int tile_w = (width + tile_size - 1) / tile_size, tile_h = (height + tile_size - 1) / tile_size;
  const dim3 grid(pp_grid((int64_t)C * N)), block(256);   // (128 / 64 threads measured equal or slower here: tools/gpu_r05_n2.sh)
  hipStream_t st = as_stream(stream);
  const bool sph = camera_model_has_spherical(camera_model, C);
  // shN rows through LDS (COOP, see CoefsLdsRow): float32 SoA attributes, SH degree >= 1, a workgroup's 256 rows within the
  // default 64 KB of dynamic LDS (K <= 22) and a 16-byte aligned tensor.  SPLAT_ONE_AMD_PP_COOP=0 keeps the lane-by-lane reads.
  static const bool coop_env = [] { const char *e = getenv("SPLAT_ONE_AMD_PP_COOP"); return !(e && e[0] == '0'); }();
  size_t coop_bytes = 0;
  if constexpr (std::is_same<A, AttrSoA>::value) {
    const size_t b = (size_t)256 * 3 * (K - 1) * sizeof(float);
    if (coop_env && sh_degree >= 1 && K > 1 && b <= 64 * 1024 && (((uintptr_t)attrs.shN) & 15) == 0) coop_bytes = b;
  }
#define SO_LAUNCH(D)                                                                                               \
  if (sph) SO_LAUNCH_(D, true); else SO_LAUNCH_(D, false)
#define SO_LAUNCH_(D, S)                                                                                           \
  if (coop_bytes) SO_LAUNCH__(D, S, true, coop_bytes); else SO_LAUNCH__(D, S, false, 0)
#define SO_LAUNCH__(D, S, CO, LDS)                                                                                 \
  hipLaunchKernelGGL((k_preprocess_fwd<D, A, S, (CO && std::is_same<A, AttrSoA>::value && D >= 1)>), grid, block, LDS, st, C, N, means, logit_opacities, attrs, viewmats,  \
                     Ks, width, height, eps2d, near_plane, far_plane, radius_clip, camera_model, antialiased,      \
                     (float)tile_size, tile_w, tile_h, radii, means2d, depths, conics, opacities, colors,          \
                     tiles_per_gauss, tile_counts, reinterpret_cast<float4 *>(rec), reinterpret_cast<float4 *>(vrec), \
                     cam_stride, tile_slots, tile_cull, bin_keys, bin_cap, bin_overflow, n_dev, (int)(width % tile_size == 0), sub_counts, replicas)
  switch (sh_degree) {
    case 0: SO_LAUNCH(0); break;
    case 1: SO_LAUNCH(1); break;
    case 2: SO_LAUNCH(2); break;
    case 3: SO_LAUNCH(3); break;
    default: SO_LAUNCH(4); break;
  }
#undef SO_LAUNCH
#undef SO_LAUNCH_
#undef SO_LAUNCH__
  return check_launch(what);
}

template <class A>
static int preprocess_bwd_impl(const char *what, int C, int N, int K, int sh_degree, const float *means,
                               const float *logit_opacities, const A &attrs, const float *viewmats, const float *Ks,
                               int width, int height, float eps2d, int camera_model, int antialiased,
                               const int32_t *radii, const float *opacities, const float *colors,
                               const float *v_means2d, const float *v_means2d_abs, const float *v_depths,
                               const float *v_conics, const float *v_colors, const float *v_opacities,
                               float opacity_reg, float scale_reg, float *v_means, float *v_log_scales, float *v_quats,
                               float *v_logit_opacities, float *v_sh0, float *v_shN, float *grad2d, float *count,
                               const float *vrec, int absgrad_stats, int64_t cam_stride, const int32_t *skip_flag,
                               float *skip_out, void *stream, const AdamFuse *fuse = nullptr, const int32_t *n_dev = nullptr,
                               const float *rec = nullptr, int64_t row_begin = 0, int64_t row_end = 0) {
  SO_REQUIRE(C >= 0 && N >= 0 && K >= 1 && width > 0 && height > 0, "%s: bad sizes", what);
  SO_REQUIRE(sh_degree >= 0 && sh_degree <= 4 && (sh_degree + 1) * (sh_degree + 1) <= K,
             "%s: sh_degree %d does not fit K=%d", what, sh_degree, K);
  if (!camera_model_ok(camera_model, C)) {
    set_error("%s: unsupported camera_model 0x%x for %d views", what, camera_model, C);
    return SO_ERR_UNSUPPORTED;
  }
  if (N == 0) return SO_OK;
  SO_REQUIRE(means && logit_opacities && viewmats && Ks && ((radii && opacities && colors) || rec) &&
                 (vrec || (v_means2d && v_conics && v_colors && v_opacities)) && v_means && v_log_scales && v_quats &&
                 v_logit_opacities && v_sh0 && (v_shN || K == 1), "%s: null pointer", what);
  SO_REQUIRE((((uintptr_t)rec) & 63) == 0, "%s: rec must be 64-byte aligned", what);
  if (radii) rec = nullptr;      // the per-view arrays win when both are given
  SO_REQUIRE((((uintptr_t)vrec) & 63) == 0, "%s: vrec must be 64-byte aligned", what);
  if (cam_stride == 0) cam_stride = N;
  SO_REQUIRE(cam_stride >= N, "%s: cam_stride %lld < N %d", what, (long long)cam_stride, N);
  if (!vrec && v_means2d_abs) absgrad_stats = 1;
  SO_REQUIRE((grad2d == nullptr) == (count == nullptr), "%s: grad2d and count go together", what);
  SO_REQUIRE(row_begin >= 0 && row_begin % 64 == 0 && (row_end <= 0 || row_end >= row_begin) && !(fuse && (row_begin || row_end > 0)),
             "%s: row range [%lld, %lld) must start at a multiple of 64 (and is not available with the fused optimiser)", what,
             (long long)row_begin, (long long)row_end);
  if (row_begin >= N || (row_end > 0 && row_end == row_begin)) return SO_OK;
  // Workgroups of 128 threads (the kernel's LDS exchange is per wave, so any multiple of 64 works; SPLAT_ONE_AMD_PPB_BLOCK = 64 /
  // 128 / 256): tools/gpu_r05_n.sh, _n3.sh -- 1M at 1440p 311 / 298 / 297 us with 256 / 128 / 64 threads, 2M 615 / 607 / 639, 500k 162 /
  // 160, c2 42 all three, float16 rows 669 / 666.
  // The limits on K stay those of 256-thread workgroups within 64 KB (22; 18 with float16 rows), whatever the workgroup size.
  static const int ppb_block = [] { const char *e = getenv("SPLAT_ONE_AMD_PPB_BLOCK"); const int b = e ? atoi(e) : 128; return (b == 64 || b == 256) ? b : 128; }();
  const int64_t ppb_rows = (row_end > 0 && row_end < N ? row_end : (int64_t)N) - row_begin;
  const dim3 grid((unsigned)std::min<int64_t>(std::max<int64_t>(ceil_div(ppb_rows, ppb_block), 1), 4096 * (256 / ppb_block))), block(ppb_block);
  hipStream_t st = as_stream(stream);
  const float sx = 0.5f * (float)width * (float)C, sy = 0.5f * (float)height * (float)C;
  // v_shN rows go through LDS when a workgroup's 256 rows fit the default 64 KB (K <= 22) and the run is 16-byte aligned
  const size_t stage_bytes = (size_t)ppb_block * 3 * (K - 1) * sizeof(float);
  const bool stage = K > 1 && K <= 22 && (((uintptr_t)v_shN) & 15) == 0;
  SO_REQUIRE(!(fuse && A::kActivated), "%s: the fused optimiser needs the raw parameters", what);
  if (fuse) {   // the fused optimiser sweeps the staged rows: same conditions, on the parameter / moment tensors
    SO_REQUIRE(K > 1 && K <= 22, "%s: fused Adam needs 2 <= K <= 22", what);
    uintptr_t bits = 0;
    for (int g = 0; g < 6; ++g) {
      SO_REQUIRE(fuse->p[g] && fuse->m[g] && fuse->v[g], "%s: fused Adam: null parameter / moment pointer (group %d)", what, g);
      bits |= (uintptr_t)fuse->p[g] | (uintptr_t)fuse->m[g] | (uintptr_t)fuse->v[g];
    }
    SO_REQUIRE((bits & 15) == 0 && fuse->hyper, "%s: fused Adam: tensors must be 16-byte aligned, hyper non-null", what);
    SO_REQUIRE(!fuse->half_rows || (A::kHalfRows && (((uintptr_t)fuse->half_rows) & 15) == 0 && fuse->half_stride16 == attr_rec_stride_bytes(K) / 16 &&
                                    K <= 18),
               "%s: fused Adam with float16 rows: rows 16-byte aligned, stride of K, and K <= 18", what);
  }
  // (float16 rows re-packed by the fused optimiser: 11 more floats of LDS per Gaussian for the row header)
  const size_t lds_bytes = stage_bytes + ((fuse && fuse->half_rows) ? (size_t)ppb_block * kHalfHdr * sizeof(float) : 0);
  const bool sph = camera_model_has_spherical(camera_model, C);
#define SO_LAUNCH(D)                                                                                              \
  if (sph) { SO_LAUNCH_(D, true); } else { SO_LAUNCH_(D, false); }
#define SO_BWD_ARGS(FUSE)                                                                                         \
  C, N, K, means, logit_opacities, attrs, viewmats, Ks, width, height, eps2d, camera_model, antialiased, radii,   \
      opacities, colors, v_means2d, v_means2d_abs, v_depths, v_conics, v_colors, v_opacities, opacity_reg,        \
      scale_reg, v_means, v_log_scales, v_quats, v_logit_opacities, v_sh0, v_shN, grad2d, count, sx, sy,          \
      reinterpret_cast<const float4 *>(vrec), absgrad_stats, cam_stride, skip_flag, skip_out, FUSE, n_dev,        \
      reinterpret_cast<const float4 *>(rec), row_begin, row_end
  // (the optimiser-fused variant exists for the raw float32 parameters only: activated inputs belong to a caller whose
  // autograd still has to run the activations' backward)
#define SO_LAUNCH_(D, S)                                                                                          \
  if (fuse) {                                                                                                     \
    if constexpr (!A::kActivated)                                                                                 \
      hipLaunchKernelGGL((k_preprocess_bwd<D, A, true, true, S>), grid, block, lds_bytes, st, SO_BWD_ARGS(*fuse)); \
  } else if (stage)                                                                                               \
    hipLaunchKernelGGL((k_preprocess_bwd<D, A, true, false, S>), grid, block, stage_bytes, st, SO_BWD_ARGS(AdamFuse{})); \
  else                                                                                                            \
    hipLaunchKernelGGL((k_preprocess_bwd<D, A, false, false, S>), grid, block, 0, st, SO_BWD_ARGS(AdamFuse{}))
  switch (sh_degree) {
    case 0: SO_LAUNCH(0); break;
    case 1: SO_LAUNCH(1); break;
    case 2: SO_LAUNCH(2); break;
    case 3: SO_LAUNCH(3); break;
    default: SO_LAUNCH(4); break;
  }
#undef SO_LAUNCH
#undef SO_LAUNCH_
#undef SO_BWD_ARGS
  return check_launch(what);
}

static inline bool attr_rec_ok(const void *arec) { return arec && (((uintptr_t)arec) & 15) == 0; }

}  // namespace so

extern "C" int so_preprocess_fwd(int C, int N, int K, int sh_degree, const float *means, const float *log_scales,
                                 const float *quats, const float *logit_opacities, const float *sh0,
                                 const float *shN, const float *viewmats, const float *Ks, int width, int height,
                                 float eps2d, float near_plane, float far_plane, float radius_clip,
                                 int camera_model, int antialiased, int tile_size, int32_t *radii, float *means2d,
                                 float *depths, float *conics, float *opacities, float *colors,
                                 int32_t *tiles_per_gauss, int32_t *tile_counts, float *rec, float *vrec,
                                 int64_t cam_stride, int32_t *tile_slots, int tile_cull, uint64_t *bin_keys, int64_t bin_cap,
                                 int32_t *bin_overflow, void *stream) {
  SO_REQUIRE((int64_t)C * N == 0 || (log_scales && quats && sh0 && (shN || K == 1)), "so_preprocess_fwd: null pointer");
  const so::AttrSoA attrs{log_scales, quats, sh0, shN, K};
  return so::preprocess_fwd_impl("so_preprocess_fwd", C, N, K, sh_degree, means, logit_opacities, attrs, viewmats, Ks,
                                 width, height, eps2d, near_plane, far_plane, radius_clip, camera_model, antialiased,
                                 tile_size, radii, means2d, depths, conics, opacities, colors, tiles_per_gauss,
                                 tile_counts, rec, vrec, cam_stride, tile_slots, tile_cull, bin_keys, bin_cap, bin_overflow, stream);
}

extern "C" int so_preprocess_fwd_f16(int C, int N, int K, int sh_degree, const float *means,
                                     const float *logit_opacities, const void *arec, const float *viewmats,
                                     const float *Ks, int width, int height, float eps2d, float near_plane,
                                     float far_plane, float radius_clip, int camera_model, int antialiased,
                                     int tile_size, int32_t *radii, float *means2d, float *depths, float *conics,
                                     float *opacities, float *colors, int32_t *tiles_per_gauss, int32_t *tile_counts,
                                     float *rec, float *vrec, int64_t cam_stride, int32_t *tile_slots, int tile_cull, uint64_t *bin_keys, int64_t bin_cap,
                                 int32_t *bin_overflow, void *stream) {
  SO_REQUIRE((int64_t)C * N == 0 || so::attr_rec_ok(arec), "so_preprocess_fwd_f16: arec must be non-null and 16-byte aligned");
  const so::AttrRec attrs{reinterpret_cast<const uint4 *>(arec), so::attr_rec_stride_bytes(K < 1 ? 1 : K) / 16};
  return so::preprocess_fwd_impl("so_preprocess_fwd_f16", C, N, K, sh_degree, means, logit_opacities, attrs, viewmats,
                                 Ks, width, height, eps2d, near_plane, far_plane, radius_clip, camera_model,
                                 antialiased, tile_size, radii, means2d, depths, conics, opacities, colors,
                                 tiles_per_gauss, tile_counts, rec, vrec, cam_stride, tile_slots, tile_cull, bin_keys, bin_cap, bin_overflow, stream);
}

extern "C" int so_preprocess_bwd(int C, int N, int K, int sh_degree, const float *means, const float *log_scales,
                                 const float *quats, const float *logit_opacities, const float *sh0,
                                 const float *shN, const float *viewmats, const float *Ks, int width, int height,
                                 float eps2d, int camera_model, int antialiased, const int32_t *radii,
                                 const float *opacities, const float *colors, const float *v_means2d,
                                 const float *v_means2d_abs, const float *v_depths, const float *v_conics,
                                 const float *v_colors, const float *v_opacities, float opacity_reg,
                                 float scale_reg, float *v_means, float *v_log_scales, float *v_quats,
                                 float *v_logit_opacities, float *v_sh0, float *v_shN, float *grad2d, float *count,
                                 const float *vrec, int absgrad_stats, int64_t cam_stride, const int32_t *skip_flag,
                                 float *skip_out, void *stream) {
  SO_REQUIRE(N == 0 || (log_scales && quats && sh0 && (shN || K == 1)), "so_preprocess_bwd: null pointer");
  const so::AttrSoA attrs{log_scales, quats, sh0, shN, K};
  return so::preprocess_bwd_impl("so_preprocess_bwd", C, N, K, sh_degree, means, logit_opacities, attrs, viewmats, Ks,
                                 width, height, eps2d, camera_model, antialiased, radii, opacities, colors, v_means2d,
                                 v_means2d_abs, v_depths, v_conics, v_colors, v_opacities, opacity_reg, scale_reg,
                                 v_means, v_log_scales, v_quats, v_logit_opacities, v_sh0, v_shN, grad2d, count, vrec,
                                 absgrad_stats, cam_stride, skip_flag, skip_out, stream);
}

extern "C" int so_preprocess_bwd_f16(int C, int N, int K, int sh_degree, const float *means,
                                     const float *logit_opacities, const void *arec, const float *viewmats,
                                     const float *Ks, int width, int height, float eps2d, int camera_model,
                                     int antialiased, const int32_t *radii, const float *opacities,
                                     const float *colors, float opacity_reg, float scale_reg, float *v_means,
                                     float *v_log_scales, float *v_quats, float *v_logit_opacities, float *v_sh0,
                                     float *v_shN, float *grad2d, float *count, const float *vrec, int absgrad_stats,
                                     int64_t cam_stride, const int32_t *skip_flag, float *skip_out, void *stream) {
  SO_REQUIRE(N == 0 || so::attr_rec_ok(arec), "so_preprocess_bwd_f16: arec must be non-null and 16-byte aligned");
  SO_REQUIRE(N == 0 || vrec, "so_preprocess_bwd_f16: the gradient records (vrec) are required");
  const so::AttrRec attrs{reinterpret_cast<const uint4 *>(arec), so::attr_rec_stride_bytes(K < 1 ? 1 : K) / 16};
  return so::preprocess_bwd_impl("so_preprocess_bwd_f16", C, N, K, sh_degree, means, logit_opacities, attrs, viewmats,
                                 Ks, width, height, eps2d, camera_model, antialiased, radii, opacities, colors, nullptr,
                                 nullptr, nullptr, nullptr, nullptr, nullptr, opacity_reg, scale_reg, v_means,
                                 v_log_scales, v_quats, v_logit_opacities, v_sh0, v_shN, grad2d, count, vrec,
                                 absgrad_stats, cam_stride, skip_flag, skip_out, stream);
}

// internal (step.hip): the backward with the optimiser fused in (float32 attributes, or float16 rows that the same
// kernel re-packs: fuse.half_rows)
namespace so {
int preprocess_bwd_fused_adam_f16(int C, int N, int K, int sh_degree, const float *means, const float *logit_opacities,
                                  const void *arec, const float *viewmats, const float *Ks, int width, int height, float eps2d,
                                  int camera_model, int antialiased, const int32_t *radii, const float *opacities,
                                  const float *colors, float opacity_reg, float scale_reg, float *grad2d, float *count,
                                  const float *vrec, int absgrad_stats, const int32_t *skip_flag, float *skip_out,
                                  const AdamFuse &fuse, void *stream, const int32_t *n_dev, const float *rec) {
  SO_REQUIRE(attr_rec_ok(arec) && fuse.half_rows == arec, "so_train_step_fwd_bwd (fused Adam, float16 rows): the rows read and the rows re-packed must be the same 16-byte aligned buffer");
  const AttrRec attrs{reinterpret_cast<const uint4 *>(arec), attr_rec_stride_bytes(K < 1 ? 1 : K) / 16};
  float *dummy = fuse.p[0];
  return preprocess_bwd_impl("so_train_step_fwd_bwd (fused Adam, float16 rows)", C, N, K, sh_degree, means, logit_opacities, attrs,
                             viewmats, Ks, width, height, eps2d, camera_model, antialiased, radii, opacities, colors, nullptr,
                             nullptr, nullptr, nullptr, nullptr, nullptr, opacity_reg, scale_reg, dummy, dummy, dummy, dummy, dummy,
                             fuse.p[5], grad2d, count, vrec, absgrad_stats, 0, skip_flag, skip_out, stream, &fuse, n_dev, rec);
}
int preprocess_bwd_fused_adam(int C, int N, int K, int sh_degree, const float *means, const float *log_scales, const float *quats,
                              const float *logit_opacities, const float *sh0, const float *shN, const float *viewmats,
                              const float *Ks, int width, int height, float eps2d, int camera_model, int antialiased,
                              const int32_t *radii, const float *opacities, const float *colors, float opacity_reg,
                              float scale_reg, float *grad2d, float *count, const float *vrec, int absgrad_stats,
                              const int32_t *skip_flag, float *skip_out, const AdamFuse &fuse, void *stream,
                              const int32_t *n_dev, const float *rec) {
  const AttrSoA attrs{log_scales, quats, sh0, shN, K};
  float *dummy = fuse.p[0];   // the gradient outputs are not written in this mode; any non-null pointer passes the checks
  return preprocess_bwd_impl("so_train_step_fwd_bwd (fused Adam)", C, N, K, sh_degree, means, logit_opacities, attrs, viewmats, Ks,
                             width, height, eps2d, camera_model, antialiased, radii, opacities, colors, nullptr, nullptr,
                             nullptr, nullptr, nullptr, nullptr, opacity_reg, scale_reg, dummy, dummy, dummy, dummy, dummy,
                             fuse.p[5], grad2d, count, vrec, absgrad_stats, 0, skip_flag, skip_out, stream, &fuse, n_dev, rec);
}

// internal (step.hip): so_preprocess_fwd / so_preprocess_bwd with the Gaussian count in device memory (n_dev; N = capacity)
int preprocess_fwd_n(int C, int N, int K, int sh_degree, const float *means, const float *log_scales, const float *quats,
                     const float *logit_opacities, const float *sh0, const float *shN, const float *viewmats, const float *Ks,
                     int width, int height, float eps2d, float near_plane, float far_plane, float radius_clip, int camera_model,
                     int antialiased, int tile_size, int32_t *radii, float *means2d, float *depths, float *conics,
                     float *opacities, float *colors, int32_t *tiles_per_gauss, int32_t *tile_counts, float *rec, float *vrec,
                     int32_t *tile_slots, int tile_cull, uint64_t *bin_keys, int64_t bin_cap, int32_t *bin_overflow,
                     const int32_t *n_dev, void *stream, int32_t *sub_counts, int replicas) {
  SO_REQUIRE((int64_t)C * N == 0 || (log_scales && quats && sh0 && (shN || K == 1)), "so_preprocess_fwd: null pointer");
  const AttrSoA attrs{log_scales, quats, sh0, shN, K};
  return preprocess_fwd_impl("so_preprocess_fwd", C, N, K, sh_degree, means, logit_opacities, attrs, viewmats, Ks, width, height,
                             eps2d, near_plane, far_plane, radius_clip, camera_model, antialiased, tile_size, radii, means2d,
                             depths, conics, opacities, colors, tiles_per_gauss, tile_counts, rec, vrec, 0, tile_slots, tile_cull,
                             bin_keys, bin_cap, bin_overflow, stream, n_dev, sub_counts, replicas);
}
int preprocess_bwd_n(int C, int N, int K, int sh_degree, const float *means, const float *log_scales, const float *quats,
                     const float *logit_opacities, const float *sh0, const float *shN, const float *viewmats, const float *Ks,
                     int width, int height, float eps2d, int camera_model, int antialiased, const int32_t *radii,
                     const float *opacities, const float *colors, float opacity_reg, float scale_reg, float *v_means,
                     float *v_log_scales, float *v_quats, float *v_logit_opacities, float *v_sh0, float *v_shN, float *grad2d,
                     float *count, const float *vrec, int absgrad_stats, const int32_t *skip_flag, float *skip_out,
                     const int32_t *n_dev, const float *rec, void *stream, int64_t row_begin, int64_t row_end) {
  SO_REQUIRE(N == 0 || (log_scales && quats && sh0 && (shN || K == 1)), "so_preprocess_bwd: null pointer");
  const AttrSoA attrs{log_scales, quats, sh0, shN, K};
  return preprocess_bwd_impl("so_preprocess_bwd", C, N, K, sh_degree, means, logit_opacities, attrs, viewmats, Ks, width, height,
                             eps2d, camera_model, antialiased, radii, opacities, colors, nullptr, nullptr, nullptr, nullptr, nullptr,
                             nullptr, opacity_reg, scale_reg, v_means, v_log_scales, v_quats, v_logit_opacities, v_sh0, v_shN,
                             grad2d, count, vrec, absgrad_stats, 0, skip_flag, skip_out, stream, nullptr, n_dev, rec, row_begin, row_end);
}
// internal (raster_op.hip): the POST-ACTIVATION inputs of gsplat's rasterization() call (AttrAct): scales [N,3],
// opacities [N] in (0,1), coeffs [N,K,3]; record-only views; the backward returns the gradients of exactly those
// tensors (v_sh0 [N,3] and v_shN [N,3(K-1)] apart: the caller concatenates)
int preprocess_fwd_act(int C, int N, int K, int sh_degree, const float *means, const float *scales, const float *quats,
                       const float *opacities_in, const float *coeffs, const float *viewmats, const float *Ks, int width,
                       int height, float eps2d, float near_plane, float far_plane, float radius_clip, int camera_model,
                       int antialiased, int tile_size, int32_t *tile_counts, float *rec, float *vrec, int tile_cull,
                       uint64_t *bin_keys, int64_t bin_cap, int32_t *bin_overflow, void *stream, int32_t *sub_counts, int replicas) {
  SO_REQUIRE((int64_t)C * N == 0 || (scales && quats && coeffs), "so_rasterization_fwd: null pointer");
  const AttrAct attrs{scales, quats, coeffs, K};
  return preprocess_fwd_impl("so_rasterization_fwd", C, N, K, sh_degree, means, opacities_in, attrs, viewmats, Ks, width, height,
                             eps2d, near_plane, far_plane, radius_clip, camera_model, antialiased, tile_size, nullptr, nullptr,
                             nullptr, nullptr, nullptr, nullptr, nullptr, tile_counts, rec, vrec, 0, nullptr, tile_cull,
                             bin_keys, bin_cap, bin_overflow, stream, nullptr, sub_counts, replicas);
}
int preprocess_bwd_act(int C, int N, int K, int sh_degree, const float *means, const float *scales, const float *quats,
                       const float *opacities_in, const float *coeffs, const float *viewmats, const float *Ks, int width,
                       int height, float eps2d, int camera_model, int antialiased, float *v_means, float *v_scales,
                       float *v_quats, float *v_opacities, float *v_sh0, float *v_shN, const float *vrec, const float *rec,
                       void *stream) {
  SO_REQUIRE(N == 0 || (scales && quats && coeffs), "so_rasterization_bwd: null pointer");
  const AttrAct attrs{scales, quats, coeffs, K};
  return preprocess_bwd_impl("so_rasterization_bwd", C, N, K, sh_degree, means, opacities_in, attrs, viewmats, Ks, width, height,
                             eps2d, camera_model, antialiased, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                             nullptr, nullptr, 0.f, 0.f, v_means, v_scales, v_quats, v_opacities, v_sh0, v_shN, nullptr, nullptr,
                             vrec, 0, 0, nullptr, nullptr, stream, nullptr, nullptr, rec);
}
// internal (raster_op.hip): the screen-space gradients alone out of the gradient records -- contiguous [n,2] arrays for
// `info["means2d"].grad` / `.absgrad` (v_means2d_abs nullable)
int rec_unpack_means2d(int64_t n, const float *vrec, float *v_means2d, float *v_means2d_abs, void *stream) {
  if (n == 0) return SO_OK;
  SO_REQUIRE(vrec && v_means2d && (((uintptr_t)vrec) & 63) == 0, "so_rasterization_bwd: gradient records missing or misaligned");
  hipLaunchKernelGGL(k_rec_unpack_grads, dim3(pp_grid(n)), dim3(256), 0, as_stream(stream), n, reinterpret_cast<const float4 *>(vrec),
                     reinterpret_cast<float2 *>(v_means2d), (float *)nullptr, (float *)nullptr, (float *)nullptr,
                     reinterpret_cast<float2 *>(v_means2d_abs));
  return check_launch("so_rasterization_bwd (means2d)");
}
// the same with float16 attribute rows (so_preprocess_fwd_f16 / so_preprocess_bwd_f16)
int preprocess_fwd_n_f16(int C, int N, int K, int sh_degree, const float *means, const float *logit_opacities, const void *arec,
                         const float *viewmats, const float *Ks, int width, int height, float eps2d, float near_plane,
                         float far_plane, float radius_clip, int camera_model, int antialiased, int tile_size, int32_t *radii,
                         float *means2d, float *depths, float *conics, float *opacities, float *colors, int32_t *tiles_per_gauss,
                         int32_t *tile_counts, float *rec, float *vrec, int32_t *tile_slots, int tile_cull, uint64_t *bin_keys,
                         int64_t bin_cap, int32_t *bin_overflow, const int32_t *n_dev, void *stream, int32_t *sub_counts, int replicas) {
  SO_REQUIRE((int64_t)C * N == 0 || attr_rec_ok(arec), "so_preprocess_fwd_f16: arec must be non-null and 16-byte aligned");
  const AttrRec attrs{reinterpret_cast<const uint4 *>(arec), attr_rec_stride_bytes(K < 1 ? 1 : K) / 16};
  return preprocess_fwd_impl("so_preprocess_fwd_f16", C, N, K, sh_degree, means, logit_opacities, attrs, viewmats, Ks, width,
                             height, eps2d, near_plane, far_plane, radius_clip, camera_model, antialiased, tile_size, radii,
                             means2d, depths, conics, opacities, colors, tiles_per_gauss, tile_counts, rec, vrec, 0, tile_slots,
                             tile_cull, bin_keys, bin_cap, bin_overflow, stream, n_dev, sub_counts, replicas);
}
int preprocess_bwd_n_f16(int C, int N, int K, int sh_degree, const float *means, const float *logit_opacities, const void *arec,
                         const float *viewmats, const float *Ks, int width, int height, float eps2d, int camera_model,
                         int antialiased, const int32_t *radii, const float *opacities, const float *colors, float opacity_reg,
                         float scale_reg, float *v_means, float *v_log_scales, float *v_quats, float *v_logit_opacities,
                         float *v_sh0, float *v_shN, float *grad2d, float *count, const float *vrec, int absgrad_stats,
                         const int32_t *skip_flag, float *skip_out, const int32_t *n_dev, void *stream, int64_t row_begin,
                         int64_t row_end) {
  SO_REQUIRE(N == 0 || attr_rec_ok(arec), "so_preprocess_bwd_f16: arec must be non-null and 16-byte aligned");
  SO_REQUIRE(N == 0 || vrec, "so_preprocess_bwd_f16: the gradient records (vrec) are required");
  const AttrRec attrs{reinterpret_cast<const uint4 *>(arec), attr_rec_stride_bytes(K < 1 ? 1 : K) / 16};
  return preprocess_bwd_impl("so_preprocess_bwd_f16", C, N, K, sh_degree, means, logit_opacities, attrs, viewmats, Ks, width,
                             height, eps2d, camera_model, antialiased, radii, opacities, colors, nullptr, nullptr, nullptr,
                             nullptr, nullptr, nullptr, opacity_reg, scale_reg, v_means, v_log_scales, v_quats,
                             v_logit_opacities, v_sh0, v_shN, grad2d, count, vrec, absgrad_stats, 0, skip_flag, skip_out, stream,
                             nullptr, n_dev, nullptr, row_begin, row_end);
}
}  // namespace so

extern "C" int64_t so_attr_rec_stride(int K) { return K >= 1 ? so::attr_rec_stride_bytes(K) : 0; }
extern "C" int64_t so_bin_counter_index(int64_t t, int64_t M) { return so::bin_counter_index(t, M); }

extern "C" int so_attr_pack_f16(int64_t N, int K, const float *log_scales, const float *quats, const float *sh0,
                                const float *shN, void *arec, void *stream) {
  return so_attr_pack_f16_n(N, K, log_scales, quats, sh0, shN, arec, nullptr, stream);
}

extern "C" int so_attr_pack_f16_n(int64_t N, int K, const float *log_scales, const float *quats, const float *sh0,
                                  const float *shN, void *arec, const int32_t *n_dev, void *stream) {
  SO_REQUIRE(N >= 0 && K >= 1, "so_attr_pack_f16: bad sizes");
  if (N == 0) return SO_OK;
  SO_REQUIRE(log_scales && quats && sh0 && (shN || K == 1), "so_attr_pack_f16: null pointer");
  SO_REQUIRE(so::attr_rec_ok(arec), "so_attr_pack_f16: arec must be non-null and 16-byte aligned");
  const int stride16 = so::attr_rec_stride_bytes(K) / 16;
  hipLaunchKernelGGL(so::k_attr_pack_f16, dim3(so::pp_grid(N * stride16)), dim3(256), 0, so::as_stream(stream), N, K,
                     stride16, log_scales, quats, sh0, shN, reinterpret_cast<uint4 *>(arec), n_dev);
  return so::check_launch("so_attr_pack_f16");
}

extern "C" int so_rec_unpack(int64_t n, const float *rec, float *means2d, int32_t *radii, float *depths, float *vrec,
                             void *stream) {
  SO_REQUIRE(n >= 0, "so_rec_unpack: bad size");
  if (n == 0) return SO_OK;
  SO_REQUIRE(rec && means2d && radii && depths, "so_rec_unpack: null pointer");
  SO_REQUIRE(((((uintptr_t)rec) | ((uintptr_t)vrec)) & 63) == 0, "so_rec_unpack: records must be 64-byte aligned");
  hipLaunchKernelGGL(so::k_rec_unpack, dim3(so::pp_grid(n)), dim3(256), 0, so::as_stream(stream), n,
                     reinterpret_cast<const float4 *>(rec), reinterpret_cast<float2 *>(means2d), radii, depths,
                     reinterpret_cast<float4 *>(vrec));
  return so::check_launch("so_rec_unpack");
}

#ifdef PP_STAMPS
extern "C" int so_debug_pp_stamps_read(unsigned long long *host, int n_words) {
  if (hipDeviceSynchronize() != hipSuccess) return SO_ERR_LAUNCH;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(so::g_pp_stamps), (size_t)n_words * 8) == hipSuccess ? SO_OK : SO_ERR_LAUNCH;
}
extern "C" int so_debug_ppb_stamps_read(unsigned long long *host, int n_words) {
  if (hipDeviceSynchronize() != hipSuccess) return SO_ERR_LAUNCH;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(so::g_ppb_stamps), (size_t)n_words * 8) == hipSuccess ? SO_OK : SO_ERR_LAUNCH;
}
#endif

extern "C" int so_rec_pack(int64_t n, const float *means2d, const float *conics, const float *colors,
                           const float *opacities, float *rec, float *vrec, void *stream) {
  SO_REQUIRE(n >= 0, "so_rec_pack: bad size");
  if (n == 0) return SO_OK;
  SO_REQUIRE(means2d && conics && opacities && rec, "so_rec_pack: null pointer");
  SO_REQUIRE(((((uintptr_t)rec) | ((uintptr_t)vrec)) & 63) == 0, "so_rec_pack: records must be 64-byte aligned");
  hipLaunchKernelGGL(so::k_rec_pack, dim3(so::pp_grid(n)), dim3(256), 0, so::as_stream(stream), n,
                     reinterpret_cast<const float2 *>(means2d), conics, colors, opacities,
                     reinterpret_cast<float4 *>(rec), reinterpret_cast<float4 *>(vrec));
  return so::check_launch("so_rec_pack");
}

extern "C" int so_rec_unpack_grads(int64_t n, const float *vrec, float *v_means2d, float *v_conics, float *v_colors,
                                   float *v_opacities, float *v_means2d_abs, void *stream) {
  SO_REQUIRE(n >= 0, "so_rec_unpack_grads: bad size");
  if (n == 0) return SO_OK;
  SO_REQUIRE(vrec && v_means2d && v_conics && v_colors && v_opacities, "so_rec_unpack_grads: null pointer");
  SO_REQUIRE((((uintptr_t)vrec) & 63) == 0, "so_rec_unpack_grads: records must be 64-byte aligned");
  hipLaunchKernelGGL(so::k_rec_unpack_grads, dim3(so::pp_grid(n)), dim3(256), 0, so::as_stream(stream), n,
                     reinterpret_cast<const float4 *>(vrec), reinterpret_cast<float2 *>(v_means2d), v_conics, v_colors,
                     v_opacities, reinterpret_cast<float2 *>(v_means2d_abs));
  return so::check_launch("so_rec_unpack_grads");
}

extern "C" int so_shard_flag_put(int world, int64_t cap, const int32_t *overflow, float *vrec_full, void *stream) {
  SO_REQUIRE(world >= 1 && world <= 64 && cap >= 0, "so_shard_flag_put: world %d not in 1..64 or cap < 0", world);
  if (cap == 0) return SO_OK;
  SO_REQUIRE(overflow && vrec_full, "so_shard_flag_put: null pointer");
  hipLaunchKernelGGL(so::k_shard_flag, dim3(1), dim3(64), 0, so::as_stream(stream), world, cap, vrec_full,
                     const_cast<int32_t *>(overflow), 1);
  return so::check_launch("so_shard_flag_put");
}

extern "C" int so_shard_flag_get(int world, int64_t cap, const float *vrec_shard, int32_t *overflow, void *stream) {
  SO_REQUIRE(world >= 1 && world <= 64 && cap >= 0, "so_shard_flag_get: world %d not in 1..64 or cap < 0", world);
  if (cap == 0) return SO_OK;
  SO_REQUIRE(overflow && vrec_shard, "so_shard_flag_get: null pointer");
  hipLaunchKernelGGL(so::k_shard_flag, dim3(1), dim3(64), 0, so::as_stream(stream), world, cap,
                     const_cast<float *>(vrec_shard), overflow, 0);
  return so::check_launch("so_shard_flag_get");
}
