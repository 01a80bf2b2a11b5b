// loss.hip -- fused photometric loss for gfx950:  (1-lambda) * L1 + lambda * (1 - SSIM), forward and
// backward, directly on the rasteriser's channel-last [B,H,W,CH] output.
//
// Replaces `F.l1_loss` + the CUDA-only `fused_ssim(..., padding="valid")` of
// /root/reference/utils/gsplat_utils/gsplat_trainer.py:624-628.  SSIM: 11x11
// Gaussian window (sigma 1.5), C1=0.01^2, C2=0.03^2, zero "same" padding, optional 5-pixel "valid"
// crop -- the published fused-SSIM formulation (separable window; the backward convolves three
// derivative maps).  HBM-bound: forward reads 2 images and writes 3 maps, backward reads 3 maps +
// 2 images and writes 1 image; all window arithmetic runs out of LDS.
//
// Row-streaming kernels: see the comment above kT.
#include "so_common.hpp"

namespace so {

constexpr int kWin = 11, kHalf = 5;
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

struct Window {
  float w[kWin];
};

// ---------------------------------------------------------------------------------------------
// Row-streaming formulation.  A workgroup owns a strip of kT floats of the channel-last rows
// (thread = one float = one (column, channel), so global loads/stores are contiguous over the
// thread index) and walks kRows output rows top to bottom.  Per input row: the row
// (+5 halo columns each side) is staged through a double-buffered LDS line, every thread does its
// 11-tap horizontal sums and pushes them into an 11-deep ring of REGISTERS; the vertical 11-tap
// sum for the row 5 above comes straight from that ring.  No 2-D intermediate ever exists in LDS,
// the next-but-one row is prefetched into registers while the current one is computed, and each
// input element is read from HBM ~1.3x instead of 1.7-3x.
//
// The kernels are bound by instruction issue (measured on MI355X, tools/probes/valu_rate.hip: a
// wave-instruction costs a SIMD ~1.2 ns for v_fma_f32, ~2.1 ns for v_pk_fma_f32 / 64-bit integer
// ops, ~3.5 ns for v_rcp_f32 / a 4- or 8-byte LDS read, ~6.9 ns for a 16-byte LDS read, 13 ns for
// ds_read2_b64, and scalar instructions are not free either), so the inner loop is written to
// issue as little as possible:
//   * staging is branch-free: row / column indices are clamped into the image and the loaded
//     values multiplied by 0/1 masks; row pointers are wave-uniform, lane offsets 32-bit;
//   * one LDS record per staged element holds everything a tap needs, as one 16-byte read:
//     forward (x, y, x*x + y*y, x*y) -- the products are formed once per element, not once per
//     tap -- backward (three derivative maps, pad);
//   * window sums are two-wide (v_pk_fma_f32 on (mu1,mu2) and (E[x^2+y^2],E[xy]));
//   * the SSIM map uses v_rcp_f32 (1 ulp) instead of three IEEE divisions;
//   * the first ring fill (no output rows yet) is a separate instantiation, so the steady state
//     carries no "is there an output row" test.
// ---------------------------------------------------------------------------------------------
#ifndef SO_SSIM_THREADS
#define SO_SSIM_THREADS 256
#endif
#ifndef SO_SSIM_ROWS
#define SO_SSIM_ROWS 36
#endif
// A workgroup is kT threads = kT consecutive FLOATS of a channel-last row (not a whole number of
// pixels for CH = 3: the horizontal neighbours of a float are simply CH floats away), four waves,
// one per SIMD, so workgroups pack a CU exactly.  kRows = 36: 23 x 30 = 690 workgroups for a
// 1080p RGB frame, all resident at once at three waves per SIMD (<= 168 VGPRs).
constexpr int kT = SO_SSIM_THREADS;   // floats of a row per workgroup
constexpr int kRows = SO_SSIM_ROWS;   // output rows per workgroup
#ifndef SO_SSIM_TAPGROUP
#define SO_SSIM_TAPGROUP 4
#endif
constexpr int kTapGroup = SO_SSIM_TAPGROUP;

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v2f pk_fma(float w, v2f x, v2f acc) {
  const v2f ww = {w, w};
  return __builtin_elementwise_fma(ww, x, acc);
}
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// A load through a pointer that was itself read from memory (the target image's slot, so_step_inputs) is a FLAT load for the
// compiler -- it cannot know the address space -- and a flat load counts in lgkmcnt as well as vmcnt: the first
// `s_waitcnt lgkmcnt(0)` for an LDS read after it also waits for the global round trip.  Every load of the target goes
// through this cast (round 3: the row prefetch two steps ahead was being waited for in the step that issued it).
__device__ __forceinline__ float gload(const float *p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *reinterpret_cast<const __attribute__((address_space(1))) float *>(reinterpret_cast<uintptr_t>(p));
#else
  return *p;
#endif
}

// Per-thread staging geometry: a staged line is the workgroup's kT floats plus kHalf*CH halo floats on
// each side.  Every thread stages element `tid` of the line and (the first 2*kHalf*CH threads only)
// element `tid + kT`; the others re-read their first element and park the copy in an LDS slot nobody
// reads, so that no lane-dependent branch is needed.
template <int CH, int T = kT>
struct StageGeom {
  unsigned off0, off1;   // offset inside an image row, clamped into it
  float cm0, cm1;        // 1 if the float is inside the row
  int slot1;             // LDS slot of the second element
  __device__ __forceinline__ StageGeom(int tid, int f0, int W) {
    constexpr int E = T + 2 * kHalf * CH;
    const int e1 = tid + T;
    const int ff0 = f0 - kHalf * CH + tid, ff1 = f0 - kHalf * CH + e1;
    const bool ok0 = ff0 >= 0 && ff0 < W * CH, ok1 = e1 < E && ff1 >= 0 && ff1 < W * CH;
    off0 = ok0 ? (unsigned)ff0 : 0u;
    off1 = ok1 ? (unsigned)ff1 : off0;
    cm0 = ok0 ? 1.f : 0.f;
    cm1 = ok1 ? 1.f : 0.f;
    slot1 = e1 < E ? e1 : E + tid;   // dummy slots start at E
  }
};
template <int CH, int T = kT>
constexpr int lds_line_slots() { return 2 * T + 2 * kHalf * CH; }

// wave-uniform pointer to row y (clamped into the image) of ONE image (`img` already points at batch element b;
// one image is < 2^31 floats, checked on the host, so the row offset is 32-bit scalar arithmetic); rmask: 0/1
__device__ __forceinline__ const float *row_ptr(const float *img, int H, int stride, int y, float &rmask) {
#ifdef SO_SSIM_DBG_SAMEROW
  const int yc = (y < 0 ? 0 : (y >= H ? H - 1 : y)) & ~31;
#else
  const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
#endif
  rmask = (y == yc) ? 1.f : 0.f;
  return img + (unsigned)(yc * stride);
}

// ------------------------------------------------------------------------------------ forward
template <int CH>
struct FwdStage {
  v2f p0, p1;   // (img1, img2) of the two staged elements, as loaded (not yet masked)
};

template <int CH, class G>
__device__ __forceinline__ void fwd_gload(FwdStage<CH> &st, const G &g, const float *img1, const float *img2,
                                          int H, int stride, int y) {
  float rm;
  const float *r1 = row_ptr(img1, H, stride, y, rm);
  const float *r2 = row_ptr(img2, H, stride, y, rm);
#ifdef SO_SSIM_DBG_NOGLOAD
  st.p0 = v2f{(float)g.off0, rm};
  st.p1 = v2f{(float)g.off1, (float)y};
  (void)r1; (void)r2;
#else
  st.p0 = v2f{r1[g.off0], gload(r2 + g.off0)};
  st.p1 = v2f{r1[g.off1], gload(r2 + g.off1)};
#endif
}

__device__ __forceinline__ v4f fwd_record(v2f p, float mask) {   // (x, y, x*x + y*y, x*y), zero outside the image
  p *= v2f{mask, mask};
  return v4f{p.x, p.y, fmaf(p.x, p.x, p.y * p.y), p.x * p.y};
}

// The masks are applied here, when the row goes to LDS, not when it is loaded: touching a loaded value
// is what makes the wave wait for it, and the loads are issued two steps before this point.
template <int CH, class G>
__device__ __forceinline__ void fwd_lstore(const FwdStage<CH> &st, const G &g, v4f *line, int tid, int y, int H) {
  const float rm = (y >= 0 && y < H) ? 1.f : 0.f;
  line[tid] = fwd_record(st.p0, rm * g.cm0);
  line[g.slot1] = fwd_record(st.p1, rm * g.cm1);
}

template <int CH>
struct SsimFwdState {
  v2f ring[kWin][2];   // per input row: (blur_x mu1, mu2), (blur_x E[x^2+y^2], E[xy])
  float l1_acc, ss_acc;
};

struct FwdOut {        // per-thread output geometry
  unsigned off;        // x * CH + ch = the float's index in its row
  bool in_image;       // inside the row
  float cmask;         // x counted in the SSIM mean (valid crop)
  float l1mask;        // x < W as float
};

// One input row.  P = ring slot (compile time), OUT = an output row is completed by this input row.
template <int CH, int P, bool OUT>
__device__ __forceinline__ void ssim_fwd_step(SsimFwdState<CH> &S, FwdStage<CH> &preA, FwdStage<CH> &preB,
                                              v4f (*rows)[lds_line_slots<CH>()],
                                              const StageGeom<CH> &g, const FwdOut &o, const float *img1, const float *img2,
                                              int it, int n_out, int b, int H, int W, int y0, int tid, int valid,
                                              const Window &win, float *__restrict__ dmaps, int64_t map_stride) {
  __builtin_amdgcn_sched_barrier(0);   // keep the unrolled steps apart: overlapping them only costs registers
  lds_barrier();
  {   // stage row it+1 (loaded two steps ago), fetch row it+3; both harmless past the end
    fwd_lstore<CH>(preA, g, rows[(it + 1) & 1], tid, y0 - kHalf + it + 1, H);
    preA = preB;
    fwd_gload<CH>(preB, g, img1, img2, H, W * CH, y0 - kHalf + it + 3);
  }
  const v4f *R = rows[it & 1] + tid;   // tap k of this thread's column/channel sits CH slots further per k
  v2f m = {0.f, 0.f}, q = {0.f, 0.f};
  float ctr_abs = 0.f;
  // taps in groups of kTapGroup: at most that many 16-byte records are in flight, which bounds the
  // registers the scheduler may spend on hoisted LDS reads
#pragma unroll
  for (int k0 = 0; k0 < kWin; k0 += kTapGroup) {
    v4f t[kTapGroup];
#pragma unroll
    for (int j = 0; j < kTapGroup; ++j)
#ifdef SO_SSIM_DBG_NOLDSREAD
      if (k0 + j < kWin) t[j] = v4f{(float)tid, (float)it, (float)j, 1.f};
#else
      if (k0 + j < kWin) t[j] = R[(k0 + j) * CH];
#endif
#pragma unroll
    for (int j = 0; j < kTapGroup; ++j)
      if (k0 + j < kWin) {
        m = pk_fma(win.w[k0 + j], v2f{t[j].x, t[j].y}, m);
        q = pk_fma(win.w[k0 + j], v2f{t[j].z, t[j].w}, q);
        if (k0 + j == kHalf) ctr_abs = fabsf(t[j].x - t[j].y);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  S.ring[P][0] = m; S.ring[P][1] = q;
  // L1 term of the centre pixel of this input row, if the row belongs to this workgroup's outputs
  const float l1m = (it >= kHalf && it < n_out + kHalf) ? o.l1mask : 0.f;
  S.l1_acc = fmaf(ctr_abs, l1m, S.l1_acc);
  if constexpr (OUT) {
    const int y = y0 + it - 2 * kHalf;   // output row
    v2f mu = {0.f, 0.f}, sq = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < kWin; ++k) {
      const int slot = (P + 1 + k) % kWin;   // oldest row first; compile-time after unrolling
      mu = pk_fma(win.w[k], S.ring[slot][0], mu);
      sq = pk_fma(win.w[k], S.ring[slot][1], sq);
    }
    const float mu1 = mu.x, mu2 = mu.y;
    const float musq = fmaf(mu1, mu1, mu2 * mu2), m12 = mu1 * mu2;
    const float Av = fmaf(2.f, m12, kC1), Bv = fmaf(2.f, sq.y - m12, kC2);
    const float D = musq + kC1, E = (sq.x - musq) + kC2;
#ifdef SO_SSIM_DBG_NOEPI
    const float rD = D, rE = E, rDE = rD * rE;
#else
    const float rD = fast_rcp(D), rE = fast_rcp(E), rDE = rD * rE;
#endif
    const float ssim = Av * Bv * rDE;
    const float rowc = (!valid || (y >= kHalf && y < H - kHalf)) ? o.cmask : 0.f;
    S.ss_acc = fmaf(ssim, rowc, S.ss_acc);
    if (dmaps) {
      const float dm_dA = Bv * rDE * rowc, dm_dB2 = 2.f * Av * rDE * rowc;
      const float dm_dD = -ssim * rD * rowc, dm_dE = -ssim * rE * rowc;
      const float g_mu = 2.f * (mu2 * dm_dA + mu1 * (dm_dD - dm_dE)) - mu2 * dm_dB2;
#ifdef SO_SSIM_DBG_NOSTORE
      if (o.in_image && g_mu == 123.f) {
#else
      if (o.in_image) {
#endif
        float *d = dmaps + (unsigned)(y * (W * CH));
        d[o.off] = g_mu;
        d[map_stride + o.off] = dm_dE;
        d[2 * map_stride + o.off] = dm_dB2;
      }
    }
  }
}

#ifndef SO_SSIM_WAVES
#define SO_SSIM_WAVES 3
#endif
// SO_SSIM_WAVES waves per SIMD = workgroups per CU: every workgroup of a 1080p frame must be resident
// at once, or the launch runs in two rounds of half-empty CUs.
template <int CH>
__global__ void __launch_bounds__(kT, SO_SSIM_WAVES)
k_ssim_l1_fwd(int B, int H, int W, const float *__restrict__ img1, const float *__restrict__ img2_direct,
              const float *const *__restrict__ img2_slot, int valid, Window win, float *__restrict__ sums,
              float *__restrict__ dmaps) {
  const float *__restrict__ img2 = img2_slot ? *img2_slot : img2_direct;   // target read in place (so_step_inputs)
  __shared__ v4f rows[2][lds_line_slots<CH>()];
  __shared__ float red[2][kT / 64];
  const int tid = threadIdx.x;
  const int f0 = blockIdx.x * kT, y0 = blockIdx.y * kRows, b = blockIdx.z;
  const int n_out = (H - y0) < kRows ? (H - y0) : kRows;
  const int n_in = n_out + 2 * kHalf;
  const int64_t map_stride = (int64_t)B * H * W * CH;
  const StageGeom<CH> g(tid, f0, W);
  const int f = f0 + tid, x = f / CH;
  FwdOut o;
  o.in_image = f < W * CH;
  o.off = o.in_image ? (unsigned)f : 0u;
  o.l1mask = o.in_image ? 1.f : 0.f;
  o.cmask = (o.in_image && (!valid || (x >= kHalf && x < W - kHalf))) ? 1.f : 0.f;
  {   // from here on every pointer addresses batch element b, rows are 32-bit offsets from it
    const int64_t ob = (int64_t)b * H * ((int64_t)W * CH);
    img1 += ob; img2 += ob;
    if (dmaps) dmaps += ob;
  }
  FwdStage<CH> preA, preB;
  fwd_gload<CH>(preA, g, img1, img2, H, W * CH, y0 - kHalf);
  fwd_lstore<CH>(preA, g, rows[0], tid, y0 - kHalf, H);
  fwd_gload<CH>(preA, g, img1, img2, H, W * CH, y0 - kHalf + 1);
  fwd_gload<CH>(preB, g, img1, img2, H, W * CH, y0 - kHalf + 2);
  SsimFwdState<CH> S;
  S.l1_acc = S.ss_acc = 0.f;
#pragma unroll
  for (int i = 0; i < kWin; ++i) S.ring[i][0] = S.ring[i][1] = v2f{0.f, 0.f};
#define SO_STEP(P, OUT) ssim_fwd_step<CH, P, OUT>(S, preA, preB, rows, g, o, img1, img2, base + P, n_out, b, H, W, y0, tid, valid, win, dmaps, map_stride)
  {   // ring fill: n_in >= 11 always, the first ten input rows complete no output row
    const int base = 0;
    SO_STEP(0, false); SO_STEP(1, false); SO_STEP(2, false); SO_STEP(3, false); SO_STEP(4, false);
    SO_STEP(5, false); SO_STEP(6, false); SO_STEP(7, false); SO_STEP(8, false); SO_STEP(9, false);
    SO_STEP(10, true);
  }
#define SO_STEPC(P) if (base + P >= n_in) break; SO_STEP(P, true)
#pragma unroll 1
  for (int base = kWin; base < n_in; base += kWin) {
    SO_STEPC(0); SO_STEPC(1); SO_STEPC(2); SO_STEPC(3); SO_STEPC(4); SO_STEPC(5);
    SO_STEPC(6); SO_STEPC(7); SO_STEPC(8); SO_STEPC(9); SO_STEPC(10);
  }
#undef SO_STEPC
#undef SO_STEP
  const float l1 = wave_reduce_sum(S.l1_acc), ss = wave_reduce_sum(S.ss_acc);
  if ((tid & 63) == 0) { red[0][tid >> 6] = l1; red[1][tid >> 6] = ss; }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int w = 0; w < kT / 64; ++w) { a += red[0][w]; c += red[1][w]; }
    atomicAdd(sums, a);
    atomicAdd(sums + 1, c);
  }
}

// ------------------------------------------------------------------------------------ backward
template <int CH>
struct BwdStage {
  float a0, b0, c0, a1, b1, c1;   // the three maps at the two staged elements, as loaded
};

template <int CH>
__device__ __forceinline__ void bwd_gload(BwdStage<CH> &st, const StageGeom<CH> &g, const float *dmaps, int64_t map_stride,
                                          int H, int stride, int y) {
  float rm;
  const float *r0 = row_ptr(dmaps, H, stride, y, rm);
  const float *r1 = r0 + map_stride, *r2 = r1 + map_stride;
#ifdef SO_SSIM_DBG_BWD_NOGLOAD      // ablation (tools/gpu_ssim.sh): the backward without its derivative-map reads
  st.a0 = (float)g.off0; st.b0 = rm; st.c0 = (float)y; st.a1 = (float)g.off1; st.b1 = rm; st.c1 = (float)y;
  (void)r1; (void)r2;
#else
  st.a0 = r0[g.off0]; st.b0 = r1[g.off0]; st.c0 = r2[g.off0];
  st.a1 = r0[g.off1]; st.b1 = r1[g.off1]; st.c1 = r2[g.off1];
#endif
}

template <int CH>
__device__ __forceinline__ void bwd_lstore(const BwdStage<CH> &st, const StageGeom<CH> &g, v4f *line, int tid, int y, int H) {
  const float rm = (y >= 0 && y < H) ? 1.f : 0.f;
  const float m0 = rm * g.cm0, m1 = rm * g.cm1;
  line[tid] = v4f{st.a0 * m0, st.b0 * m0, st.c0 * m0, 0.f};
  line[g.slot1] = v4f{st.a1 * m1, st.b1 * m1, st.c1 * m1, 0.f};
}

template <int CH>
struct SsimBwdState {
  v2f ring01[kWin];
  float ring2[kWin];
};

template <int CH, int P, bool OUT>
__device__ __forceinline__ void ssim_bwd_step(SsimBwdState<CH> &S, BwdStage<CH> &preA, BwdStage<CH> &preB,
                                              v4f (*rows)[lds_line_slots<CH>()],
                                              const StageGeom<CH> &g, const FwdOut &o, const float *dmaps, int64_t map_stride,
                                              const float *__restrict__ img1, const float *__restrict__ img2, int it, int b,
                                              int H, int W, int y0, int tid, const Window &win, float wl1, float wss,
                                              float *__restrict__ v_img1) {
  __builtin_amdgcn_sched_barrier(0);
  lds_barrier();
  {
    bwd_lstore<CH>(preA, g, rows[(it + 1) & 1], tid, y0 - kHalf + it + 1, H);
    preA = preB;
    bwd_gload<CH>(preB, g, dmaps, map_stride, H, W * CH, y0 - kHalf + it + 3);
  }
  float xv = 0.f, yv = 0.f;
  unsigned orow = 0;
  if constexpr (OUT) {   // issue the two pixel loads early; consumed after the vertical sums
    orow = (unsigned)((y0 + it - 2 * kHalf) * (W * CH));
    xv = (img1 + orow)[o.off];
    yv = gload(img2 + orow + o.off);
  }
  const v4f *R = rows[it & 1] + tid;
  v2f ac = {0.f, 0.f};
  float d = 0.f;
#pragma unroll
  for (int k0 = 0; k0 < kWin; k0 += kTapGroup) {
    v4f t[kTapGroup];
#pragma unroll
    for (int j = 0; j < kTapGroup; ++j)
      if (k0 + j < kWin) t[j] = R[(k0 + j) * CH];
#pragma unroll
    for (int j = 0; j < kTapGroup; ++j)
      if (k0 + j < kWin) {
        ac = pk_fma(win.w[k0 + j], v2f{t[j].x, t[j].y}, ac);
        d = fmaf(win.w[k0 + j], t[j].z, d);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  S.ring01[P] = ac; S.ring2[P] = d;
  if constexpr (OUT) {
    v2f vac = {0.f, 0.f};
    float vd = 0.f;
#pragma unroll
    for (int k = 0; k < kWin; ++k) {
      const int slot = (P + 1 + k) % kWin;
      vac = pk_fma(win.w[k], S.ring01[slot], vac);
      vd = fmaf(win.w[k], S.ring2[slot], vd);
    }
    if (o.in_image) {
      const float diff = xv - yv;
      const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
      (v_img1 + orow)[o.off] = wl1 * sgn + wss * (vac.x + 2.f * xv * vac.y + yv * vd);
    }
  }
}

template <int CH>
__global__ void __launch_bounds__(kT, SO_SSIM_WAVES)
k_ssim_l1_bwd(int B, int H, int W, const float *__restrict__ img1, const float *__restrict__ img2_direct,
              const float *const *__restrict__ img2_slot, const float *__restrict__ dmaps, Window win, float w_l1, float w_ssim,
              const float *__restrict__ v_loss, float *__restrict__ v_img1, const float *__restrict__ sums,
              float *__restrict__ loss_out, float a_l1, float b_ssim, float c_const) {
  const float *__restrict__ img2 = img2_slot ? *img2_slot : img2_direct;
  __shared__ v4f rows[2][lds_line_slots<CH>()];
  const int tid = threadIdx.x;
  const int f0 = blockIdx.x * kT, y0 = blockIdx.y * kRows, b = blockIdx.z;
  const int n_out = (H - y0) < kRows ? (H - y0) : kRows;
  const int n_in = n_out + 2 * kHalf;
  const int64_t map_stride = (int64_t)B * H * W * CH;
  const float up = v_loss ? *v_loss : 1.f;
  const float wl1 = w_l1 * up, wss = w_ssim * up;
  if (loss_out && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
    // the forward kernel has completed: finalise the scalars here instead of a separate launch
    const float l1 = sums[0] * a_l1, ss = sums[1] * b_ssim;   // mean |x-y| , mean SSIM
    loss_out[0] = w_l1 / a_l1 * l1 + w_ssim / b_ssim * ss + c_const;
    loss_out[1] = l1;
    loss_out[2] = 1.f - ss;
  }
  const StageGeom<CH> g(tid, f0, W);
  const int f = f0 + tid;
  FwdOut o;
  o.in_image = f < W * CH;
  o.off = o.in_image ? (unsigned)f : 0u;
  o.l1mask = o.cmask = 0.f;
  {
    const int64_t ob = (int64_t)b * H * ((int64_t)W * CH);
    img1 += ob; img2 += ob; dmaps += ob; v_img1 += ob;
  }
  BwdStage<CH> preA, preB;
  bwd_gload<CH>(preA, g, dmaps, map_stride, H, W * CH, y0 - kHalf);
  bwd_lstore<CH>(preA, g, rows[0], tid, y0 - kHalf, H);
  bwd_gload<CH>(preA, g, dmaps, map_stride, H, W * CH, y0 - kHalf + 1);
  bwd_gload<CH>(preB, g, dmaps, map_stride, H, W * CH, y0 - kHalf + 2);
  SsimBwdState<CH> S;
#pragma unroll
  for (int i = 0; i < kWin; ++i) { S.ring01[i] = v2f{0.f, 0.f}; S.ring2[i] = 0.f; }
#define SO_STEP(P, OUT) ssim_bwd_step<CH, P, OUT>(S, preA, preB, rows, g, o, dmaps, map_stride, img1, img2, base + P, b, H, W, y0, tid, win, wl1, wss, v_img1)
  {
    const int base = 0;
    SO_STEP(0, false); SO_STEP(1, false); SO_STEP(2, false); SO_STEP(3, false); SO_STEP(4, false);
    SO_STEP(5, false); SO_STEP(6, false); SO_STEP(7, false); SO_STEP(8, false); SO_STEP(9, false);
    SO_STEP(10, true);
  }
#define SO_STEPC(P) if (base + P >= n_in) break; SO_STEP(P, true)
#pragma unroll 1
  for (int base = kWin; base < n_in; base += kWin) {
    SO_STEPC(0); SO_STEPC(1); SO_STEPC(2); SO_STEPC(3); SO_STEPC(4); SO_STEPC(5);
    SO_STEPC(6); SO_STEPC(7); SO_STEPC(8); SO_STEPC(9); SO_STEPC(10);
  }
#undef SO_STEPC
#undef SO_STEP
}

// ------------------------------------------------------------------------------------ fused: loss and gradient in one kernel
// Two chained row-streaming stages in one workgroup, no derivative maps in HBM:
//   stage 1 = k_ssim_l1_fwd's step: input row -> LDS line A -> 11 horizontal taps -> register ring -> vertical taps ->
//             SSIM value and the three derivative values of the row 5 above; they go to LDS line B;
//   stage 2 = k_ssim_l1_bwd's step on line B (one step later, so that one barrier per step serves both lines):
//             11 horizontal taps -> second register ring -> vertical taps -> gradient of the row 5 further up.
// Cost of fusing: an output row needs derivative rows +-5, which need input rows +-10, and the same in columns: a
// workgroup of kFusedT = 512 floats x `rows` rows (eight waves: one workgroup per CU at two waves per SIMD) stages rows+20
// input rows and runs stage 1 over rows+10 of them, but emits only 512 - 10*CH floats x rows; both rings live in registers
// (77 values, 244 VGPRs).  `rows` is a run-time argument chosen so that the whole grid is resident at once (52 at 1080p
// RGB: 12 x 21 workgroups on 256 CUs).  Measured at 1080p RGB as single launches (tools/probes/ssim_bench.hip,
// profiles/r02_experiments.json): 87.6 us (256-thread workgroups at 57 rows: 89.8 us) against 52 + 47 us for the pair;
// with rows that put the grid into a second round (45) or leave it short of waves (108): 111 / 131 us.
#ifndef SO_FUSED_THREADS
#define SO_FUSED_THREADS 512
#endif
#ifndef SO_FUSED_WAVES
#define SO_FUSED_WAVES 2
#endif
#ifndef SO_FUSED_TAPGROUP
#define SO_FUSED_TAPGROUP 6
#endif
constexpr int kFusedT = SO_FUSED_THREADS;   // floats of a row per workgroup of the fused kernel (8 waves: one workgroup per CU)
constexpr int kFusedWaves = SO_FUSED_WAVES;
constexpr int kFusedTapGroup = SO_FUSED_TAPGROUP;
// make_window()'s weights as compile-time constants (the launcher checks them against the computed ones): eleven scalar
// registers less to keep alive through a loop that is short of them -- constants are re-materialised, arguments spilled
constexpr float kFusedW[kWin] = {0x1.0d956cp-10f, 0x1.f1fe02p-8f, 0x1.26eb18p-5f, 0x1.bff0fep-4f, 0x1.b43c4p-3f, 0x1.10656p-2f,
                                 0x1.b43c4p-3f,   0x1.bff0fep-4f, 0x1.26eb18p-5f, 0x1.f1fe02p-8f, 0x1.0d956cp-10f};

// SO_FUSED_XY8 (round 3, default): line A holds (x, y) only -- 8-byte records, half the LDS bytes of stage 1's eleven
// taps -- and the two products are formed per tap (one packed multiply + one fma) instead of once per staged element.
// The step was within 8 % of its instruction-issue bound and STILL gained 3.6 us (74.3 -> 70.7): eight waves reading
// 22 x 1 KB per step is most of what a CU's LDS delivers, so bytes count there, not only instructions.
#ifndef SO_FUSED_XY8
#define SO_FUSED_XY8 1
#endif
#if SO_FUSED_XY8
typedef v2f fusedA_t;
#else
typedef v4f fusedA_t;
#endif
template <int CH, class G>
__device__ __forceinline__ void fused_lstore(const FwdStage<CH> &st, const G &g, fusedA_t *line, int tid, int y, int H) {
#if SO_FUSED_XY8
  const float rm = (y >= 0 && y < H) ? 1.f : 0.f;
  const float m0 = rm * g.cm0, m1 = rm * g.cm1;
  line[tid] = st.p0 * v2f{m0, m0};
  line[g.slot1] = st.p1 * v2f{m1, m1};
#else
  fwd_lstore<CH>(st, g, line, tid, y, H);
#endif
}
template <int CH>
constexpr int fused_out_floats() { return kFusedT - 2 * kHalf * CH; }
template <int CH>
constexpr int line_b_slots() { return kFusedT + 2 * kHalf * CH; }

template <int CH>
struct FusedState {
  v2f r1[kWin][2];   // stage 1: (blur_x mu1, mu2), (blur_x E[x^2+y^2], E[xy]) per input row
  v2f r2a[kWin];     // stage 2: blur_x of (g_mu, dm_dE) per derivative row
  float r2b[kWin];   //          blur_x of dm_dB2
  float l1_acc, ss_acc;
  float xo, yo;      // the two images at this thread's float of the row the CURRENT step emits, loaded during the previous step
};

struct FusedOut {
  unsigned off;      // the thread's float inside an image row (clamped)
  float cmask;       // float inside the row (and inside the crop): stage-1 derivative values are kept
  float sum_l1;      // float counted in the L1 sum  (inside the row, inner thread)
  float sum_ss;      // float counted in the SSIM sum (cmask, inner thread)
  bool store;        // stage 2 writes this float
};

// One step = one input row.  MODE 0: stage-1 horizontal only; 1: + stage-1 vertical, line B written; 2: + stage-2
// horizontal; 3: + stage-2 vertical and the output row.  Step `it` handles input row y0-10+it, derivative row
// y0-15+it (written to line B), reads derivative row y0-16+it from line B, and emits output row y0-21+it.
template <int CH, int P, int MODE>
__device__ __forceinline__ void fused_step(FusedState<CH> &S, FwdStage<CH> &preA, FwdStage<CH> &preB,
                                           fusedA_t (*rowsA)[lds_line_slots<CH, kFusedT>()], v4f (*rowsB)[line_b_slots<CH>()],
                                           const StageGeom<CH, kFusedT> &g, const FusedOut &o, const float *img1, const float *img2,
                                           int it, int n_out, int H, int W, int y0, int tid, int valid,
                                           float wl1, float wss, float *__restrict__ v_img1) {
  __builtin_amdgcn_sched_barrier(0);
  lds_barrier();
  {
    fused_lstore<CH>(preA, g, rowsA[(it + 1) & 1], tid, y0 - 2 * kHalf + it + 1, H);
    preA = preB;
    fwd_gload<CH>(preB, g, img1, img2, H, W * CH, y0 - 2 * kHalf + it + 3);
  }
  // The output row's own pixel values (the gradient needs x and y at the centre) are requested ONE STEP AHEAD and carried
  // in the state: requested in the step that uses them, hipcc sinks the loads into the `if (o.store)` block at the end of
  // the step, right in front of their use, and every step then waits a full L2 round trip there (round 3: that wait was
  // most of the 1.03 us a wave needs per step with a SIMD to itself).  A load cannot sink past the next step's barrier.
  float xv = 0.f, yv = 0.f, xn = 0.f, yn = 0.f;
  unsigned orow = 0;
  if constexpr (MODE == 3) {
    orow = (unsigned)((y0 + it - 21) * (W * CH));
    xv = S.xo;
    yv = S.yo;
  }
  if constexpr (MODE >= 2) {
    int yr = y0 + it - 20;                        // the row the NEXT step emits, clamped into the image
    yr = yr < 0 ? 0 : (yr >= H ? H - 1 : yr);
    const unsigned orow_n = (unsigned)(yr * (W * CH));
    xn = (img1 + orow_n)[o.off];
    yn = gload(img2 + orow_n + o.off);
  }
  // ---- stage 1, horizontal
  {
    const fusedA_t *R = rowsA[it & 1] + tid;
    v2f m = {0.f, 0.f}, q = {0.f, 0.f};
    float ctr_abs = 0.f;
#pragma unroll
    for (int k0 = 0; k0 < kWin; k0 += kFusedTapGroup) {
      fusedA_t t[kFusedTapGroup];
#pragma unroll
      for (int j = 0; j < kFusedTapGroup; ++j)
        if (k0 + j < kWin) t[j] = R[(k0 + j) * CH];
#pragma unroll
      for (int j = 0; j < kFusedTapGroup; ++j)
        if (k0 + j < kWin) {
          m = pk_fma(kFusedW[k0 + j], v2f{t[j].x, t[j].y}, m);
#if SO_FUSED_XY8
          const v2f xx = v2f{t[j].x, t[j].x} * v2f{t[j].x, t[j].y};          // (x^2, x y)
          q = pk_fma(kFusedW[k0 + j], v2f{fmaf(t[j].y, t[j].y, xx.x), xx.y}, q);
#else
          q = pk_fma(kFusedW[k0 + j], v2f{t[j].z, t[j].w}, q);
#endif
          if (k0 + j == kHalf) ctr_abs = fabsf(t[j].x - t[j].y);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    S.r1[P][0] = m; S.r1[P][1] = q;
    const float l1m = (it >= 2 * kHalf && it < n_out + 2 * kHalf) ? o.sum_l1 : 0.f;
    S.l1_acc = fmaf(ctr_abs, l1m, S.l1_acc);
  }
  // ---- stage 1, vertical: SSIM and its derivatives at row y0-15+it -> line B
  if constexpr (MODE >= 1) {
    const int y = y0 + it - 3 * kHalf;
    v2f mu = {0.f, 0.f}, sq = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < kWin; ++k) {
      const int slot = (P + 1 + k) % kWin;
      mu = pk_fma(kFusedW[k], S.r1[slot][0], mu);
      sq = pk_fma(kFusedW[k], S.r1[slot][1], sq);
    }
    const float mu1 = mu.x, mu2 = mu.y;
    const float musq = fmaf(mu1, mu1, mu2 * mu2), m12 = mu1 * mu2;
    const float Av = fmaf(2.f, m12, kC1), Bv = fmaf(2.f, sq.y - m12, kC2);
    const float D = musq + kC1, E = (sq.x - musq) + kC2;
    const float rD = fast_rcp(D), rE = fast_rcp(E), rDE = rD * rE;
    const float ssim = Av * Bv * rDE;
    const bool row_ok = y >= 0 && y < H && (!valid || (y >= kHalf && y < H - kHalf));
    const float rowc = row_ok ? o.cmask : 0.f;
    const float cnt = (row_ok && it >= 3 * kHalf && it < n_out + 3 * kHalf) ? o.sum_ss : 0.f;
    S.ss_acc = fmaf(ssim, cnt, S.ss_acc);
    const float dm_dA = Bv * rDE * rowc, dm_dB2 = 2.f * Av * rDE * rowc;
    const float dm_dD = -ssim * rD * rowc, dm_dE = -ssim * rE * rowc;
    const float g_mu = 2.f * (mu2 * dm_dA + mu1 * (dm_dD - dm_dE)) - mu2 * dm_dB2;
    rowsB[it & 1][tid + kHalf * CH] = v4f{g_mu, dm_dE, dm_dB2, 0.f};
  }
  // ---- stage 2, horizontal over the derivative row written in the previous step
  if constexpr (MODE >= 2) {
    const v4f *R = rowsB[(it - 1) & 1] + tid;
    v2f ac = {0.f, 0.f};
    float d = 0.f;
#pragma unroll
    for (int k0 = 0; k0 < kWin; k0 += kFusedTapGroup) {
      v4f t[kFusedTapGroup];
#pragma unroll
      for (int j = 0; j < kFusedTapGroup; ++j)
        if (k0 + j < kWin) t[j] = R[(k0 + j) * CH];
#pragma unroll
      for (int j = 0; j < kFusedTapGroup; ++j)
        if (k0 + j < kWin) {
          ac = pk_fma(kFusedW[k0 + j], v2f{t[j].x, t[j].y}, ac);
          d = fmaf(kFusedW[k0 + j], t[j].z, d);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    S.r2a[P] = ac; S.r2b[P] = d;
  }
  // ---- stage 2, vertical: gradient of output row y0-21+it
  if constexpr (MODE == 3) {
    v2f vac = {0.f, 0.f};
    float vd = 0.f;
#pragma unroll
    for (int k = 0; k < kWin; ++k) {
      const int slot = (P + 1 + k) % kWin;
      vac = pk_fma(kFusedW[k], S.r2a[slot], vac);
      vd = fmaf(kFusedW[k], S.r2b[slot], vd);
    }
    if (o.store) {
      const float diff = xv - yv;
      const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
      (v_img1 + orow)[o.off] = wl1 * sgn + wss * (vac.x + 2.f * xv * vac.y + yv * vd);
    }
  }
  if constexpr (MODE >= 2) { S.xo = xn; S.yo = yn; }
}

template <int CH>
__global__ void __launch_bounds__(kFusedT, kFusedWaves)
k_ssim_l1_fused(int B, int H, int W, int rows, const float *__restrict__ img1, const float *__restrict__ img2_direct,
                const float *const *__restrict__ img2_slot, int valid, float w_l1, float w_ssim,
                const float *__restrict__ v_loss, float *__restrict__ sums, float *__restrict__ v_img1,
                float *__restrict__ loss_out, int32_t *__restrict__ ticket, float c_const) {
#ifdef SO_FUSED_DBG_DIRECT
  const float *__restrict__ img2 = img2_direct;
#else
  const float *__restrict__ img2 = img2_slot ? *img2_slot : img2_direct;
#endif
  const float up = v_loss ? *v_loss : 1.f;
  const float wl1 = w_l1 * up, wss = w_ssim * up;
  __shared__ fusedA_t rowsA[2][lds_line_slots<CH, kFusedT>()];
  __shared__ v4f rowsB[2][line_b_slots<CH>()];
  __shared__ float red[2][kFusedT / 64];
  const int tid = threadIdx.x;
  const int f0 = blockIdx.x * fused_out_floats<CH>() - kHalf * CH;   // float of thread 0 (negative in the first strip)
  const int y0 = blockIdx.y * rows, b = blockIdx.z;
  const int n_out = (H - y0) < rows ? (H - y0) : rows;
  const int n_steps = n_out + 4 * kHalf + 1;
  const StageGeom<CH, kFusedT> g(tid, f0, W);
  const int f = f0 + tid;
  const bool in_row = f >= 0 && f < W * CH, inner = tid >= kHalf * CH && tid < kFusedT - kHalf * CH;
  const int x = in_row ? f / CH : 0;
  FusedOut o;
  o.off = in_row ? (unsigned)f : 0u;
  o.cmask = (in_row && (!valid || (x >= kHalf && x < W - kHalf))) ? 1.f : 0.f;
  o.sum_l1 = (in_row && inner) ? 1.f : 0.f;
  o.sum_ss = inner ? o.cmask : 0.f;
  o.store = in_row && inner;
  {
    const int64_t ob = (int64_t)b * H * ((int64_t)W * CH);
    img1 += ob; img2 += ob; v_img1 += ob;
  }
  for (int i = tid; i < 2 * line_b_slots<CH>(); i += kFusedT) (&rowsB[0][0])[i] = v4f{0.f, 0.f, 0.f, 0.f};
  FwdStage<CH> preA, preB;
  fwd_gload<CH>(preA, g, img1, img2, H, W * CH, y0 - 2 * kHalf);
  fused_lstore<CH>(preA, g, rowsA[0], tid, y0 - 2 * kHalf, H);
  fwd_gload<CH>(preA, g, img1, img2, H, W * CH, y0 - 2 * kHalf + 1);
  fwd_gload<CH>(preB, g, img1, img2, H, W * CH, y0 - 2 * kHalf + 2);
  FusedState<CH> S;
  S.l1_acc = S.ss_acc = 0.f;
  S.xo = S.yo = 0.f;
#pragma unroll
  for (int i = 0; i < kWin; ++i) { S.r1[i][0] = S.r1[i][1] = S.r2a[i] = v2f{0.f, 0.f}; S.r2b[i] = 0.f; }
#define SO_STEP(P, MODE) fused_step<CH, P, MODE>(S, preA, preB, rowsA, rowsB, g, o, img1, img2, base + P, n_out, H, W, y0, tid, valid, wl1, wss, v_img1)
  {
    const int base = 0;
    SO_STEP(0, 0); SO_STEP(1, 0); SO_STEP(2, 0); SO_STEP(3, 0); SO_STEP(4, 0);
    SO_STEP(5, 0); SO_STEP(6, 0); SO_STEP(7, 0); SO_STEP(8, 0); SO_STEP(9, 0);
    SO_STEP(10, 1);
  }
  {
    const int base = kWin;
    SO_STEP(0, 2); SO_STEP(1, 2); SO_STEP(2, 2); SO_STEP(3, 2); SO_STEP(4, 2);
    SO_STEP(5, 2); SO_STEP(6, 2); SO_STEP(7, 2); SO_STEP(8, 2); SO_STEP(9, 2);
    SO_STEP(10, 3);
  }
#define SO_STEPC(P) if (base + P >= n_steps) break; SO_STEP(P, 3)
#pragma unroll 1
  for (int base = 2 * kWin; base < n_steps; base += kWin) {
    SO_STEPC(0); SO_STEPC(1); SO_STEPC(2); SO_STEPC(3); SO_STEPC(4); SO_STEPC(5);
    SO_STEPC(6); SO_STEPC(7); SO_STEPC(8); SO_STEPC(9); SO_STEPC(10);
  }
#undef SO_STEPC
#undef SO_STEP
  const float l1 = wave_reduce_sum(S.l1_acc), ss = wave_reduce_sum(S.ss_acc);
  if ((tid & 63) == 0) { red[0][tid >> 6] = l1; red[1][tid >> 6] = ss; }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int w = 0; w < kFusedT / 64; ++w) { a += red[0][w]; c += red[1][w]; }
    atomicAdd(sums, a);
    atomicAdd(sums + 1, c);
    if (loss_out) {
      // the workgroup that draws the last ticket sees every other workgroup's sums (fence before the ticket, sums read
      // back through the same memory-side atomics) and writes the scalars; it leaves the ticket at zero for the next launch
      __threadfence();
      const int n_wg = (int)(gridDim.x * gridDim.y * gridDim.z);
      if (atomicAdd(ticket, 1) == n_wg - 1) {
        __threadfence();
        const float s0 = atomicAdd(sums, 0.f), s1 = atomicAdd(sums + 1, 0.f);
        const float a_l1 = 1.f / ((float)B * H * W * CH);
        const float b_ss = 1.f / ((float)B * CH * (valid ? (float)(H - 10) * (float)(W - 10) : (float)H * (float)W));
        const float l1m = s0 * a_l1, ssm = s1 * b_ss;   // mean |x-y| , mean SSIM
        loss_out[0] = w_l1 / a_l1 * l1m + w_ssim / b_ss * ssm + c_const;
        loss_out[1] = l1m;
        loss_out[2] = 1.f - ssm;
        atomicExch(ticket, 0);
      }
    }
  }
}


static Window make_window() {
  Window w;
  double g[kWin], s = 0.0;
  for (int i = 0; i < kWin; ++i) {
    const double x = i - kHalf;
    g[i] = exp(-(x * x) / (2.0 * 1.5 * 1.5));
    s += g[i];
  }
  for (int i = 0; i < kWin; ++i) w.w[i] = (float)(g[i] / s);
  return w;
}

}  // namespace so

/* img1 (rendered, receives the gradient) / img2 (target): [B,H,W,CH] f32, CH in {1,3,4}.
 * sums[2] (device, zeroed by the caller): sums[0] += sum |img1-img2|, sums[1] += sum of the SSIM map
 * over the counted pixels (all, or the interior when padding_valid).  dmaps[3,B,H,W,CH] nullable. */
namespace so {
int ssim_l1_fwd_launch(int B, int H, int W, int CH, const float *img1, const float *img2, const float *const *img2_slot,
                       int padding_valid, float *sums, float *dmaps, void *stream);
int ssim_l1_bwd_launch(int B, int H, int W, int CH, const float *img1, const float *img2, const float *const *img2_slot,
                       const float *dmaps, float w_l1, float w_ssim, const float *v_loss, float *v_img1, const float *sums,
                       float *loss_out, int padding_valid, float loss_const, void *stream);
}
extern "C" int so_ssim_l1_fwd(int B, int H, int W, int CH, const float *img1, const float *img2,
                              int padding_valid, float *sums, float *dmaps, void *stream) {
  return so::ssim_l1_fwd_launch(B, H, W, CH, img1, img2, nullptr, padding_valid, sums, dmaps, stream);
}
int so::ssim_l1_fwd_launch(int B, int H, int W, int CH, const float *img1, const float *img2, const float *const *img2_slot,
                           int padding_valid, float *sums, float *dmaps, void *stream) {
  SO_REQUIRE(B >= 0 && H > 0 && W > 0, "so_ssim_l1_fwd: bad sizes");
  SO_REQUIRE(CH == 1 || CH == 3 || CH == 4, "so_ssim_l1_fwd: CH=%d not in {1,3,4}", CH);
  if (B == 0) return SO_OK;
  SO_REQUIRE(img1 && (img2 || img2_slot) && sums, "so_ssim_l1_fwd: null pointer");
  SO_REQUIRE((int64_t)H * W * CH < (int64_t)INT32_MAX, "so_ssim_l1_fwd: one image must hold fewer than 2^31 values");
  const so::Window win = so::make_window();
  const dim3 grid((W * CH + so::kT - 1) / so::kT, (H + so::kRows - 1) / so::kRows, B), block(so::kT);
  hipStream_t st = so::as_stream(stream);
  if (CH == 1) hipLaunchKernelGGL(so::k_ssim_l1_fwd<1>, grid, block, 0, st, B, H, W, img1, img2, img2_slot, padding_valid, win, sums, dmaps);
  else if (CH == 3) hipLaunchKernelGGL(so::k_ssim_l1_fwd<3>, grid, block, 0, st, B, H, W, img1, img2, img2_slot, padding_valid, win, sums, dmaps);
  else hipLaunchKernelGGL(so::k_ssim_l1_fwd<4>, grid, block, 0, st, B, H, W, img1, img2, img2_slot, padding_valid, win, sums, dmaps);
  return so::check_launch("so_ssim_l1_fwd");
}

/* v_img1[B,H,W,CH] = v_loss * ( w_l1 * sign(img1-img2) + w_ssim * d(sum of SSIM map)/d img1 ).
 * For loss = (1-l)*mean|.| + l*(1-mean SSIM): w_l1 = (1-l)/(B*H*W*CH), w_ssim = -l/n_counted.
 * v_loss: device scalar (nullable = 1).  If loss_out[3] is given (with the forward's `sums`), thread 0
 * also writes (w_l1*sum|.| + w_ssim*sum SSIM + loss_const, mean|.|, 1 - mean SSIM). */
extern "C" int so_ssim_l1_bwd(int B, int H, int W, int CH, const float *img1, const float *img2,
                              const float *dmaps, float w_l1, float w_ssim, const float *v_loss, float *v_img1,
                              const float *sums, float *loss_out, int padding_valid, float loss_const,
                              void *stream) {
  return so::ssim_l1_bwd_launch(B, H, W, CH, img1, img2, nullptr, dmaps, w_l1, w_ssim, v_loss, v_img1, sums, loss_out,
                                padding_valid, loss_const, stream);
}
int so::ssim_l1_bwd_launch(int B, int H, int W, int CH, const float *img1, const float *img2, const float *const *img2_slot,
                           const float *dmaps, float w_l1, float w_ssim, const float *v_loss, float *v_img1, const float *sums,
                           float *loss_out, int padding_valid, float loss_const, void *stream) {
  SO_REQUIRE(B >= 0 && H > 0 && W > 0, "so_ssim_l1_bwd: bad sizes");
  SO_REQUIRE(CH == 1 || CH == 3 || CH == 4, "so_ssim_l1_bwd: CH=%d not in {1,3,4}", CH);
  if (B == 0) return SO_OK;
  SO_REQUIRE(img1 && (img2 || img2_slot) && dmaps && v_img1, "so_ssim_l1_bwd: null pointer");
  SO_REQUIRE((int64_t)H * W * CH < (int64_t)INT32_MAX, "so_ssim_l1_bwd: one image must hold fewer than 2^31 values");
  SO_REQUIRE(loss_out == nullptr || sums != nullptr, "so_ssim_l1_bwd: loss_out needs sums");
  const float a_l1 = 1.f / ((float)B * H * W * CH);
  const float b_ss = 1.f / ((float)B * CH * (padding_valid ? (float)(H - 10) * (float)(W - 10) : (float)H * (float)W));
  const so::Window win = so::make_window();
  const dim3 grid((W * CH + so::kT - 1) / so::kT, (H + so::kRows - 1) / so::kRows, B), block(so::kT);
  hipStream_t st = so::as_stream(stream);
  if (CH == 1) hipLaunchKernelGGL(so::k_ssim_l1_bwd<1>, grid, block, 0, st, B, H, W, img1, img2, img2_slot, dmaps, win, w_l1, w_ssim, v_loss, v_img1, sums, loss_out, a_l1, b_ss, loss_const);
  else if (CH == 3) hipLaunchKernelGGL(so::k_ssim_l1_bwd<3>, grid, block, 0, st, B, H, W, img1, img2, img2_slot, dmaps, win, w_l1, w_ssim, v_loss, v_img1, sums, loss_out, a_l1, b_ss, loss_const);
  else hipLaunchKernelGGL(so::k_ssim_l1_bwd<4>, grid, block, 0, st, B, H, W, img1, img2, img2_slot, dmaps, win, w_l1, w_ssim, v_loss, v_img1, sums, loss_out, a_l1, b_ss, loss_const);
  return so::check_launch("so_ssim_l1_bwd");
}

/* The loss and its gradient in ONE launch (no derivative maps):  v_img1 as so_ssim_l1_bwd writes it, sums[2] as
 * so_ssim_l1_fwd accumulates them (zeroed by the caller).  loss_out[3] (nullable) receives (loss, mean|.|, 1 - mean SSIM)
 * from the workgroup that finishes last; it needs `ticket`, one int32 that is zero before the first launch (the kernel
 * returns it to zero) and is not shared by launches that may overlap.  rows: output rows per workgroup, 0 = chosen so that
 * the grid is resident at once. */
extern "C" int so_device_cu_count(void);
namespace so {
int ssim_l1_fused_launch(int B, int H, int W, int CH, const float *img1, const float *img2, const float *const *img2_slot,
                         int padding_valid, float w_l1, float w_ssim, const float *v_loss, float *sums, float *v_img1,
                         float *loss_out, int32_t *ticket, float loss_const, int rows, void *stream);
}
extern "C" int so_ssim_l1_fused(int B, int H, int W, int CH, const float *img1, const float *img2, int padding_valid,
                                float w_l1, float w_ssim, const float *v_loss, float *sums, float *v_img1, float *loss_out,
                                int32_t *ticket, float loss_const, int rows, void *stream) {
  return so::ssim_l1_fused_launch(B, H, W, CH, img1, img2, nullptr, padding_valid, w_l1, w_ssim, v_loss, sums, v_img1, loss_out,
                                  ticket, loss_const, rows, stream);
}
int so::ssim_l1_fused_launch(int B, int H, int W, int CH, const float *img1, const float *img2, const float *const *img2_slot,
                             int padding_valid, float w_l1, float w_ssim, const float *v_loss, float *sums, float *v_img1,
                             float *loss_out, int32_t *ticket, float loss_const, int rows, void *stream) {
  SO_REQUIRE(B >= 0 && H > 0 && W > 0 && rows >= 0, "so_ssim_l1_fused: bad sizes");
  SO_REQUIRE(CH == 1 || CH == 3 || CH == 4, "so_ssim_l1_fused: CH=%d not in {1,3,4}", CH);
  if (B == 0) return SO_OK;
  SO_REQUIRE(img1 && (img2 || img2_slot) && sums && v_img1, "so_ssim_l1_fused: null pointer");
  SO_REQUIRE(loss_out == nullptr || ticket != nullptr, "so_ssim_l1_fused: loss_out needs a ticket");
  SO_REQUIRE((int64_t)H * W * CH < (int64_t)INT32_MAX, "so_ssim_l1_fused: one image must hold fewer than 2^31 values");
  {
    const so::Window win = so::make_window();
    for (int i = 0; i < so::kWin; ++i)
      SO_REQUIRE(win.w[i] == so::kFusedW[i], "so_ssim_l1_fused: window constant %d differs from the computed weight", i);
  }
  const int out_t = so::kFusedT - 2 * so::kHalf * CH, nx = (W * CH + out_t - 1) / out_t;
  if (rows == 0) {
    static const int cus = so_device_cu_count() > 0 ? so_device_cu_count() : 256;   // 256 on an unpartitioned MI355X
    const int64_t slots = (int64_t)cus * so::kFusedWaves * 4 / (so::kFusedT / 64);   // workgroups resident at once
    const int64_t ny_max = slots / ((int64_t)nx * B) > 0 ? slots / ((int64_t)nx * B) : 1;
    rows = (int)((H + ny_max - 1) / ny_max);
    // (a floor of 8 rows per workgroup -- 28 row steps with the halo; it was 24 until round 5, which left small images on a third of
    // the CUs: 512 x 512 40.5 -> 32.1 us, 960 x 540 42.5 -> 36.5, 256 x 256 39.2 -> 25.8; from 1440 x 720 on the floor is not reached,
    // tools/gpu_r05_aa.sh)
    if (rows < 8) rows = 8;
  }
  const int ny = (H + rows - 1) / rows;
  SO_REQUIRE(ny <= 65535 && B <= 65535, "so_ssim_l1_fused: grid too large");
  const dim3 grid(nx, ny, B), block(so::kFusedT);
  hipStream_t st = so::as_stream(stream);
#define SO_LAUNCH_FUSED(CHV) hipLaunchKernelGGL(so::k_ssim_l1_fused<CHV>, grid, block, 0, st, B, H, W, rows, img1, img2, img2_slot, \
                                                 padding_valid, w_l1, w_ssim, v_loss, sums, v_img1, loss_out, ticket, loss_const)
  if (CH == 1) SO_LAUNCH_FUSED(1);
  else if (CH == 3) SO_LAUNCH_FUSED(3);
  else SO_LAUNCH_FUSED(4);
#undef SO_LAUNCH_FUSED
  return so::check_launch("so_ssim_l1_fused");
}
