// refine.hip -- device-side densification: gsplat `DefaultStrategy` refinement (duplicate / split / prune /
// opacity reset) as a stream compaction on the device, with the Gaussian count N in device memory.
//
// The reference drives it every `refine_every` steps from its training loop
// (/root/reference/utils/gsplat_utils/gsplat_trainer.py:744-763, hooks at :616-622; defaults SURVEY.md section 8
// a11 / B.3 [upstream-memory]): `_grow_gs` duplicates small and splits large high-gradient Gaussians with
// torch.where / torch.cat on the parameters AND the Adam moments, `_prune_gs` removes transparent / oversized ones.
// Each of those reads a count back to the host to size the new tensors.  Here the whole refinement is five launches
// on capacity-preallocated buffers and reads nothing back:
//
//   k_refine_classify   one lane per source Gaussian: the three decisions, packed into a flag byte; per-workgroup
//                       counts of the rows each output segment receives (wave ballots + popcounts)
//   k_refine_scan       one workgroup: exclusive scan of the per-workgroup counts, new N, the report {duplicated, split,
//                       pruned, N, overflow, old N, refinements, rows needed}; a refined set larger than the capacity
//                       turns the whole refinement into the identity copy (overflow = 1, statistics kept)
//   k_refine_map        destination of every surviving row (ballot prefix inside the workgroup + scanned offset):
//                       src_of[dst] = source row | kind << 30
//   k_refine_gather     one lane per OUTPUT element: parameters, exp_avg, exp_avg_sq of all six tensors copied from the
//                       source set into the destination set (moments of new rows zeroed, split children displaced
//                       by R diag(s) z and shrunk by 1.6) -- writes fully coalesced, reads in row-sized runs
//   (+ the densification statistics zeroed)
//
// Output order = gsplat's: [surviving originals that were not split | surviving duplicates | first split children |
// second split children], each in source order, so the result equals the host-side torch formulation row for row
// (tests/test_gpu_refine.py), the split samples being a counter-based function of (seed, step, source id, child)
// instead of a draw from a host generator (so_rng.hpp).
#include "so_common.hpp"
#include "so_rng.hpp"
#include "splat_math.hpp"

namespace so {

constexpr int kRefBlock = 256;
constexpr int kRefItems = 4;
constexpr int kRefChunk = kRefBlock * kRefItems;   // source rows per workgroup
constexpr uint32_t kSrcMask = 0x3FFFFFFFu;

enum : uint32_t { RF_A = 1u, RF_B = 2u, RF_C = 4u, RF_DUP = 8u, RF_SPLIT = 16u };

struct RefineDev {
  float grow_grad2d, grow_scale3d, prune_opa, prune_scale3d;
  int prune_big, revised_opacity;
};

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ int64_t imin64(int64_t a, int64_t b) { return a < b ? a : b; }

__device__ __forceinline__ uint32_t classify_one(int64_t i, const float *__restrict__ ls, const float *__restrict__ lo,
                                                 const float *__restrict__ grad2d, const float *__restrict__ count,
                                                 const RefineDev P) {
  const float s_max = fmaxf(fmaxf(expf(ls[3 * i]), expf(ls[3 * i + 1])), expf(ls[3 * i + 2]));
  const float avg = grad2d[i] / fmaxf(count[i], 1.f);
  const bool high = avg > P.grow_grad2d;
  const bool small = s_max <= P.grow_scale3d;
  const bool dup = high && small, split = high && !small;
  const float op = sigmoid_f(lo[i]);
  const bool prune_self = (op < P.prune_opa) || (P.prune_big && s_max > P.prune_scale3d);
  // revised opacity 1 - sqrt(1 - op) = op / (1 + sqrt(1 - op)), with 1 - op = sigmoid(-x): no cancellation at either end
  const float child_op = P.revised_opacity ? op / (1.f + sqrtf(sigmoid_f(-lo[i]))) : op;
  const bool prune_child = (child_op < P.prune_opa) || (P.prune_big && s_max / 1.6f > P.prune_scale3d);
  uint32_t f = 0;
  if (!split && !prune_self) f |= RF_A;
  if (dup && !prune_self) f |= RF_B;
  if (split && !prune_child) f |= RF_C;
  if (dup) f |= RF_DUP;
  if (split) f |= RF_SPLIT;
  return f;
}

__global__ void __launch_bounds__(kRefBlock)
k_refine_classify(int64_t cap, const int32_t *__restrict__ n_src, const float *__restrict__ ls, const float *__restrict__ lo,
                  const float *__restrict__ grad2d, const float *__restrict__ count, const RefineDev P,
                  uint8_t *__restrict__ flags, int32_t *__restrict__ block_counts) {
  __shared__ int32_t s_cnt[kRefBlock / 64][5];
  const int64_t N = imin64((int64_t)*n_src, cap);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int32_t c[5] = {0, 0, 0, 0, 0};
#pragma unroll
  for (int it = 0; it < kRefItems; ++it) {
    const int64_t i = (int64_t)blockIdx.x * kRefChunk + it * kRefBlock + threadIdx.x;
    uint32_t f = 0;
    if (i < N) {
      f = classify_one(i, ls, lo, grad2d, count, P);
      flags[i] = (uint8_t)f;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) c[k] += __popcll(__ballot((f >> k) & 1u));   // wave-uniform
  }
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < 5; ++k) s_cnt[wv][k] = c[k];
  __syncthreads();
  if (threadIdx.x < 5) {
    int32_t t = 0;
#pragma unroll
    for (int w = 0; w < kRefBlock / 64; ++w) t += s_cnt[w][threadIdx.x];
    block_counts[(int64_t)blockIdx.x * 5 + threadIdx.x] = t;
  }
}

// One workgroup of 1024: exclusive scan of the per-workgroup counts of the three segments, totals, new N, report.
__global__ void __launch_bounds__(1024)
k_refine_scan(int nblk, int64_t cap, const int32_t *__restrict__ block_counts, int32_t *__restrict__ block_offsets,
              int32_t *__restrict__ totals, const int32_t *__restrict__ n_src, int32_t *__restrict__ n_dst,
              int32_t *__restrict__ report) {
  __shared__ int32_t s[5][1024];
  const int t = threadIdx.x;
  const int per = (nblk + 1023) / 1024;
  const int b0 = t * per, b1 = min(nblk, b0 + per);
  int32_t loc[5] = {0, 0, 0, 0, 0};
  for (int b = b0; b < b1; ++b)
#pragma unroll
    for (int k = 0; k < 5; ++k) loc[k] += block_counts[(int64_t)b * 5 + k];
#pragma unroll
  for (int k = 0; k < 5; ++k) s[k][t] = loc[k];
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {   // inclusive Hillis-Steele over the 1024 partial sums
    int32_t add[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) add[k] = t >= d ? s[k][t - d] : 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 5; ++k) s[k][t] += add[k];
    __syncthreads();
  }
  int32_t run[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) run[k] = s[k][t] - loc[k];   // exclusive prefix of this thread's first workgroup
  for (int b = b0; b < b1; ++b)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      block_offsets[(int64_t)b * 3 + k] = run[k];
      run[k] += block_counts[(int64_t)b * 5 + k];
    }
  if (t == 1023) {
    const int64_t A = s[0][t], B = s[1][t], Cc = s[2][t], n_dup = s[3][t], n_split = s[4][t];
    const int64_t n_old = imin64((int64_t)*n_src, cap);
    int64_t n_new = A + B + 2 * Cc;
    const int over = n_new > cap;
    // A refined set that does not fit the capacity is NOT written (round 2 truncated it: the tail of the output order --
    // split children, duplicates -- was lost while their parents were already gone): the refinement becomes the identity
    // copy src -> dst with the statistics kept, report[4] tells the host, which enlarges the buffers and refines again.
    report[7] = (int32_t)imin64(n_new, 0x7fffffff);                      // rows the refined set needs
    if (over) n_new = n_old;
    totals[0] = (int32_t)(over ? n_old : A); totals[1] = (int32_t)(over ? 0 : B); totals[2] = (int32_t)(over ? 0 : Cc);
    totals[3] = over;
    *n_dst = (int32_t)n_new;
    report[0] = (int32_t)n_dup; report[1] = (int32_t)n_split;
    report[2] = (int32_t)(n_old + n_dup + n_split - (A + B + 2 * Cc));   // pruned from the grown set
    report[3] = (int32_t)n_new; report[4] = over; report[5] = (int32_t)n_old;
    // the report lives in host-mapped memory and is polled without a synchronisation (FusedEngine.poll_refine_report): the
    // sequence word goes out LAST, behind a system-scope fence, so that a host that sees it sees report[4] / [7] too
    __threadfence_system();
    report[6] += 1;                 // refinements done on these buffers (the host's view of N is stale when it differs)
  }
}

__global__ void __launch_bounds__(kRefBlock)
k_refine_map(int64_t cap, const int32_t *__restrict__ n_src, const uint8_t *__restrict__ flags,
             const int32_t *__restrict__ block_offsets, const int32_t *__restrict__ totals, uint32_t *__restrict__ src_of) {
  __shared__ int32_t s_cnt[kRefItems * (kRefBlock / 64)][3];
  __shared__ int32_t s_base[kRefItems * (kRefBlock / 64)][3];
  const int64_t N = imin64((int64_t)*n_src, cap);
  if (totals[3]) {   // the refined set would not fit (k_refine_scan): identity
#pragma unroll
    for (int it = 0; it < kRefItems; ++it) {
      const int64_t i = (int64_t)blockIdx.x * kRefChunk + it * kRefBlock + threadIdx.x;
      if (i < N) src_of[i] = (uint32_t)i;
    }
    return;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
  uint32_t f[kRefItems];
  int32_t rank[kRefItems][3];
#pragma unroll
  for (int it = 0; it < kRefItems; ++it) {
    const int64_t i = (int64_t)blockIdx.x * kRefChunk + it * kRefBlock + threadIdx.x;
    f[it] = i < N ? flags[i] : 0u;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const unsigned long long m = __ballot((f[it] >> k) & 1u);
      rank[it][k] = __popcll(m & lt);
      if (lane == 0) s_cnt[it * (kRefBlock / 64) + wv][k] = __popcll(m);
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    int32_t run = block_offsets[(int64_t)blockIdx.x * 3 + threadIdx.x];
    for (int e = 0; e < kRefItems * (kRefBlock / 64); ++e) {
      s_base[e][threadIdx.x] = run;
      run += s_cnt[e][threadIdx.x];
    }
  }
  __syncthreads();
  const int64_t A = totals[0], B = totals[1], Cc = totals[2];
#pragma unroll
  for (int it = 0; it < kRefItems; ++it) {
    const uint32_t i = (uint32_t)((int64_t)blockIdx.x * kRefChunk + it * kRefBlock + threadIdx.x);
    const int e = it * (kRefBlock / 64) + wv;
    if (f[it] & RF_A) {
      const int64_t d = s_base[e][0] + rank[it][0];
      if (d < cap) src_of[d] = i;
    }
    if (f[it] & RF_B) {
      const int64_t d = A + s_base[e][1] + rank[it][1];
      if (d < cap) src_of[d] = i | (1u << 30);
    }
    if (f[it] & RF_C) {
      const int64_t d = A + B + s_base[e][2] + rank[it][2];
      if (d < cap) src_of[d] = i | (2u << 30);
      if (d + Cc < cap) src_of[d + Cc] = i | (3u << 30);
    }
  }
}

struct ModelSet {
  float *p[6], *m[6], *v[6];
};

// grid.y: 0..5 the six parameter tensors (means, log-scales, quats, opacity logits, sh0, shN), 6: the statistics
__global__ void __launch_bounds__(256)
k_refine_gather(int64_t cap, int K, const ModelSet src, const ModelSet dst, const int32_t *__restrict__ n_dst,
                const uint32_t *__restrict__ src_of, float *__restrict__ grad2d, float *__restrict__ count,
                int revised_opacity, uint64_t seed, uint32_t step, const int32_t *__restrict__ totals) {
  const int g = blockIdx.y;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g == 6) {   // the statistics start over after every refinement (gsplat zeroes them)
    if (totals[3]) return;   // ... unless this one was put off (capacity): it will run again on the same statistics
    for (int64_t i = t0; i < cap; i += stride) { grad2d[i] = 0.f; count[i] = 0.f; }
    return;
  }
  const int L = (g == 0 || g == 1 || g == 4) ? 3 : (g == 2 ? 4 : (g == 3 ? 1 : 3 * (K - 1)));
  if (L == 0) return;
  const int64_t total = imin64((int64_t)*n_dst, cap) * L;
  const float *__restrict__ P = src.p[g], *__restrict__ M = src.m[g], *__restrict__ V = src.v[g];
  float *__restrict__ Pd = dst.p[g], *__restrict__ Md = dst.m[g], *__restrict__ Vd = dst.v[g];
  for (int64_t e = t0; e < total; e += stride) {
    const int64_t j = e / L;
    const int c = (int)(e - j * L);
    const uint32_t s = src_of[j];
    const int64_t i = s & kSrcMask;
    const uint32_t kind = s >> 30;
    float val = P[i * L + c], mo = 0.f, vo = 0.f;
    if (kind == 0) { mo = M[i * L + c]; vo = V[i * L + c]; }
    if (kind >= 2) {
      if (g == 0) {          // mean + R diag(exp(log s)) z
        const float4 qq = *reinterpret_cast<const float4 *>(src.p[2] + 4 * i);
        const float q[4] = {qq.x, qq.y, qq.z, qq.w};
        float R[9], qn[4], inv_norm;
        quat_to_rotmat<float>(q, R, qn, inv_norm);
        float z[3];
        split_normals(seed, step, (uint32_t)i, kind - 2u, z);
        const float *lsp = src.p[1] + 3 * i;
        val += R[3 * c] * (expf(lsp[0]) * z[0]) + R[3 * c + 1] * (expf(lsp[1]) * z[1]) + R[3 * c + 2] * (expf(lsp[2]) * z[2]);
      } else if (g == 1) {   // log(exp(log s) / 1.6)
        val = logf(expf(val) / 1.6f);
      } else if (g == 3 && revised_opacity) {
        // o' = 1 - sqrt(1 - o) = o / (1 + s) and 1 - o' = s with s = sqrt(1 - o) = sqrt(sigmoid(-x)): stable at both ends
        const float op = sigmoid_f(val), sq = sqrtf(sigmoid_f(-val));
        val = logf(op / ((1.f + sq) * sq));
      }
    }
    Pd[e] = val; Md[e] = mo; Vd[e] = vo;
  }
}

__global__ void __launch_bounds__(256)
k_reset_opacity(int64_t cap, const int32_t *__restrict__ n_dev, float *__restrict__ lo, float *__restrict__ m,
                float *__restrict__ v, float max_logit) {
  const int64_t N = n_dev ? imin64((int64_t)*n_dev, cap) : cap;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    lo[i] = fminf(lo[i], max_logit);
    m[i] = 0.f;
    v[i] = 0.f;
  }
}

// DefaultStrategy._update_state on the dense [C,N] layout (the operator-level path; the fused engine accumulates the same
// statistic inside k_preprocess_bwd): one lane per Gaussian sums over the cameras that see it -- no atomics, no
// intermediate [C,N] tensors (the torch formulation is a dozen launches).
__global__ void __launch_bounds__(256)
k_strategy_update(int C, int64_t N, const float2 *__restrict__ v_means2d, const int32_t *__restrict__ radii, int64_t rstride,
                  float sx, float sy, float inv_max_wh, float *__restrict__ grad2d, float *__restrict__ count,
                  float *__restrict__ radii_state) {
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
    float g = 0.f, cn = 0.f, rmax = 0.f;
    for (int c = 0; c < C; ++c) {
      const int32_t r = radii[((int64_t)c * N + n) * rstride];
      if (r > 0) {
        const float2 v = v_means2d[(int64_t)c * N + n];
        const float gx = v.x * sx, gy = v.y * sy;
        g += sqrtf(gx * gx + gy * gy);
        cn += 1.f;
        rmax = fmaxf(rmax, (float)r * inv_max_wh);
      }
    }
    if (cn > 0.f) {
      grad2d[n] += g;
      count[n] += cn;
      if (radii_state) radii_state[n] = fmaxf(radii_state[n], rmax);
    }
  }
}

static inline int ref_nblk(int64_t cap) { return (int)ceil_div(cap, kRefChunk); }

}  // namespace so

extern "C" int64_t so_refine_scratch_words(int64_t capacity) {
  if (capacity <= 0) return 0;
  const int64_t nblk = so::ref_nblk(capacity);
  return (capacity + 3) / 4 + capacity + 8 * nblk + 16;
}

extern "C" int so_refine_default(int64_t capacity, int K, const so_model_set *src, const int32_t *n_src_dev,
                                 const so_model_set *dst, int32_t *n_dst_dev, float *grad2d, float *count,
                                 const so_refine_params *prm, int32_t *scratch, int32_t *report_dev, void *stream) {
  SO_REQUIRE(capacity > 0 && capacity < (int64_t)so::kSrcMask && K >= 1, "so_refine_default: capacity %lld / K %d out of range",
             (long long)capacity, K);
  SO_REQUIRE(src && dst && n_src_dev && n_dst_dev && grad2d && count && prm && scratch && report_dev,
             "so_refine_default: null pointer");
  SO_REQUIRE(n_src_dev != n_dst_dev, "so_refine_default: source and destination need their own N");
  so::ModelSet S{}, D{};
  for (int g = 0; g < 6; ++g) {
    const bool empty = g == 5 && K == 1;
    SO_REQUIRE(empty || (src->p[g] && src->m[g] && src->v[g] && dst->p[g] && dst->m[g] && dst->v[g]),
               "so_refine_default: null tensor in group %d", g);
    SO_REQUIRE(empty || (src->p[g] != dst->p[g] && src->m[g] != dst->m[g] && src->v[g] != dst->v[g]),
               "so_refine_default: group %d: the compaction is out of place (source == destination)", g);
    S.p[g] = src->p[g]; S.m[g] = src->m[g]; S.v[g] = src->v[g];
    D.p[g] = dst->p[g]; D.m[g] = dst->m[g]; D.v[g] = dst->v[g];
  }
  SO_REQUIRE((((uintptr_t)src->p[2]) & 15) == 0, "so_refine_default: quaternions must be 16-byte aligned");
  const int nblk = so::ref_nblk(capacity);
  uint8_t *flags = reinterpret_cast<uint8_t *>(scratch);
  uint32_t *src_of = reinterpret_cast<uint32_t *>(scratch + (capacity + 3) / 4);
  int32_t *block_counts = scratch + (capacity + 3) / 4 + capacity;
  int32_t *block_offsets = block_counts + 5 * (int64_t)nblk;
  int32_t *totals = block_offsets + 3 * (int64_t)nblk;
  const so::RefineDev P{prm->grow_grad2d, prm->grow_scale3d, prm->prune_opa, prm->prune_scale3d, prm->prune_big,
                        prm->revised_opacity};
  hipStream_t st = so::as_stream(stream);
  hipLaunchKernelGGL(so::k_refine_classify, dim3(nblk), dim3(so::kRefBlock), 0, st, capacity, n_src_dev, src->p[1], src->p[3],
                     grad2d, count, P, flags, block_counts);
  hipLaunchKernelGGL(so::k_refine_scan, dim3(1), dim3(1024), 0, st, nblk, capacity, block_counts, block_offsets, totals,
                     n_src_dev, n_dst_dev, report_dev);
  hipLaunchKernelGGL(so::k_refine_map, dim3(nblk), dim3(so::kRefBlock), 0, st, capacity, n_src_dev, flags, block_offsets, totals,
                     src_of);
  int64_t gx = so::ceil_div(capacity * 3, 256);
  if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(so::k_refine_gather, dim3((unsigned)gx, 7), dim3(256), 0, st, capacity, K, S, D, n_dst_dev, src_of, grad2d,
                     count, prm->revised_opacity, prm->seed, (uint32_t)prm->step, totals);
  return so::check_launch("so_refine_default");
}

extern "C" int so_reset_opacity(int64_t capacity, const int32_t *n_dev, float *logit_opacities, float *exp_avg,
                                float *exp_avg_sq, float max_logit, void *stream) {
  SO_REQUIRE(capacity >= 0, "so_reset_opacity: bad capacity");
  if (capacity == 0) return SO_OK;
  SO_REQUIRE(logit_opacities && exp_avg && exp_avg_sq, "so_reset_opacity: null pointer");
  int64_t gx = so::ceil_div(capacity, 256);
  if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(so::k_reset_opacity, dim3((unsigned)gx), dim3(256), 0, so::as_stream(stream), capacity, n_dev,
                     logit_opacities, exp_avg, exp_avg_sq, max_logit);
  return so::check_launch("so_reset_opacity");
}

extern "C" int so_strategy_update_state(int C, int64_t N, const float *v_means2d, const int32_t *radii, int64_t radii_stride,
                                        float sx, float sy, float inv_max_wh, float *grad2d, float *count, float *radii_state,
                                        void *stream) {
  SO_REQUIRE(C >= 0 && N >= 0, "so_strategy_update_state: bad sizes");
  if (C == 0 || N == 0) return SO_OK;
  SO_REQUIRE(v_means2d && radii && grad2d && count, "so_strategy_update_state: null pointer");
  int64_t gx = so::ceil_div(N, 256);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(so::k_strategy_update, dim3((unsigned)gx), dim3(256), 0, so::as_stream(stream), C, N,
                     reinterpret_cast<const float2 *>(v_means2d), radii, radii_stride > 0 ? radii_stride : (int64_t)1, sx, sy,
                     inv_max_wh, grad2d, count, radii_state);
  return so::check_launch("so_strategy_update_state");
}
