// isect.hip -- K6/K7/K8: tile binning, per-tile depth sort and tile offsets for gfx950.
//
// Replaces gsplat `isect_tiles` (count + emit kernels and a global cub radix sort of 64-bit
// camera|tile|depth keys) and `isect_offset_encode`, reached inside `rasterization`
// (/root/reference/utils/gsplat_utils/gsplat_trainer.py:477).
//
// MI355X design (HBM-bound integer work, no dense contraction):
//   1. k_isect_count   one lane per (camera, Gaussian): AABB in tiles, tiles_per_gauss, and a
//                      histogram over (camera, tile) with fire-and-forget int atomics.
//   2. k_scan_tiles    exclusive scan of the histogram = `isect_offsets` (K8 comes for free) and
//                      the total n_isects, all on the device: no host round trip, capturable.
//   3. k_isect_scatter counting-sort scatter: slot = offsets[tile] + atomic cursor; writes one
//                      64-bit key (fp32 depth bits << 32 | flatten id) per intersection.
//   4. k_tile_sort_*   one workgroup per tile sorts its keys in LDS (bitonic network, all
//                      comparisons ascending so ragged sizes need no padding) and writes
//                      flatten_ids / isect_ids.  Ties in depth resolve by ascending flatten id, which
//                      is exactly what the reference's stable radix sort of Gaussian-major keys gives.
// Traffic per intersection: 8 B written + 8 B read + 4 (+8) B written, versus >=6 radix passes of
// 24 B for the global 64-bit sort.
#include "so_common.hpp"
#include "wave_sort.hpp"   // lane_xor_u32 / shfl_xor_u64 / wave_bitonic_sort<E>: shared with the rasteriser's prologue sort

namespace so {

struct TileBox {
  int x0, x1, y0, y1;
};

// float32 AABB arithmetic exactly as published
// wrap: the image is periodic in x -- the columns are VIRTUAL (x0 may be negative, x1 beyond tile_w; at most one image
// width of them), every user files column x under wrapx(x, tile_w)
__device__ __forceinline__ TileBox tile_box(float mx, float my, float radius, float tile_size, int tile_w, int tile_h,
                                            bool wrap = false) {
  const float tile_r = radius / tile_size;
  const float tx = mx / tile_size, ty = my / tile_size;
  TileBox b;
  if (wrap) {
    b.x0 = (int)fmaxf(floorf(tx - tile_r), (float)-tile_w);
    b.x1 = (int)fminf(ceilf(tx + tile_r), (float)(2 * tile_w));
    // wider than the image (splats near the poles of a panorama): every column once -- the tile_w VIRTUAL columns centred
    // on the splat, so that each is within half an image of it and the exact tile test / the rasteriser's nearest copy
    // mean the same copy (round 2: columns 0 .. tile_w-1 tested the far ones against the wrong copy and culled them)
    // (the columns whose CENTRES x + 0.5 lie within half an image of the splat: what "nearest copy" means to the rasteriser)
    if (b.x1 - b.x0 > tile_w) { b.x0 = (int)ceilf(tx - 0.5f * (float)tile_w - 0.5f); b.x1 = b.x0 + tile_w; }
  } else {
  b.x0 = (int)fminf(fmaxf(floorf(tx - tile_r), 0.f), (float)tile_w);
  b.x1 = (int)fminf(fmaxf(ceilf(tx + tile_r), 0.f), (float)tile_w);
  }
  b.y0 = (int)fminf(fmaxf(floorf(ty - tile_r), 0.f), (float)tile_h);
  b.y1 = (int)fminf(fmaxf(ceilf(ty + tile_r), 0.f), (float)tile_h);
  return b;
}

__global__ void __launch_bounds__(256)
k_isect_count(int C, int N, const float *__restrict__ means2d, const int32_t *__restrict__ radii, float tile_size,
              int tile_w, int tile_h, int32_t *__restrict__ tiles_per_gauss, int32_t *__restrict__ tile_counts,
              const float4 *__restrict__ cull_rec, int wrap_flags) {
  const int64_t total = (int64_t)C * N;
  const int n_tiles = tile_w * tile_h;
  // uniform trip count: the cooperative part needs every lane of a wave (same scheme as k_preprocess_fwd)
  for (int64_t idx0 = (int64_t)blockIdx.x * blockDim.x; idx0 < total; idx0 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t idx = idx0 + threadIdx.x;
    int cnt = 0, c = 0;
    TileBox b{0, 0, 0, 0};
    float mx = 0.f, my = 0.f, qa = 0.f, qb = 0.f, qc = 0.f, tau = 0.f;   // exact tile culling (cull_rec: see so_isect_fill)
    if (idx < total) {
      const int r = radii[idx];
      if (r > 0) {
        const float2 m = *reinterpret_cast<const float2 *>(means2d + 2 * idx);
        c = (int)(idx / N);
        b = tile_box(m.x, m.y, (float)r, tile_size, tile_w, tile_h, wrap_for(wrap_flags, c));
        cnt = (b.x1 - b.x0) * (b.y1 - b.y0);
        mx = m.x; my = m.y;
        if (cull_rec) {
          const float4 q0 = cull_rec[4 * idx], q1 = cull_rec[4 * idx + 1];
          qa = q0.z; qb = q0.w; qc = q1.x;
          tau = cull_tau(q1.y);
        }
      }
      tiles_per_gauss[idx] = cnt;
    }
    // small rectangles: the owning lane; large ones: the whole wave in 8x8 tile blocks
    constexpr int kOwn = 12;
    const bool big = cnt > kOwn;
    if (cnt > 0 && !big) {
      int32_t *row = tile_counts + (int64_t)c * n_tiles;
      for (int y = b.y0; y < b.y1; ++y)
        for (int x = b.x0; x < b.x1; ++x)
          if (!cull_rec || tile_touches(mx, my, qa, qb, qc, tau, x, y, tile_size)) atomicAdd(row + y * tile_w + wrapx(x, tile_w), 1);
    }
    unsigned long long todo = __ballot(big);
    const int lane = lane_id();
    auto rl = [](float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int sx0 = __builtin_amdgcn_readlane(b.x0, src), sx1 = __builtin_amdgcn_readlane(b.x1, src);
      const int sy0 = __builtin_amdgcn_readlane(b.y0, src), sy1 = __builtin_amdgcn_readlane(b.y1, src);
      const float smx = rl(mx, src), smy = rl(my, src), sqa = rl(qa, src), sqb = rl(qb, src), sqc = rl(qc, src), stau = rl(tau, src);
      int32_t *row = tile_counts + (int64_t)__builtin_amdgcn_readlane(c, src) * n_tiles;
      for (int y = sy0 + (lane >> 3); y < sy1; y += 8)
        for (int x = sx0 + (lane & 7); x < sx1; x += 8)
          if (!cull_rec || tile_touches(smx, smy, sqa, sqb, sqc, stau, x, y, tile_size)) atomicAdd(row + y * tile_w + wrapx(x, tile_w), 1);
    }
  }
}

// single-workgroup exclusive scan over M = C*n_tiles counters (M is at most a few 100k).  Each thread owns E
// consecutive counters; E = 8 covers the 8160 tiles of a 1080p view in ONE trip (one load round, three barriers).
template <int E>
__global__ void __launch_bounds__(1024)
k_scan_tiles(int64_t M, const int32_t *__restrict__ counts, const int32_t *__restrict__ counts2,
             int32_t *__restrict__ offsets, int32_t *__restrict__ total) {
  __shared__ int32_t wave_sums[16];
  __shared__ int32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < M; base += 1024 * E) {
    const int64_t i0 = base + (int64_t)tid * E;
    int32_t v[E];
    if (i0 + E <= M) {   // whole run in range: 16-byte loads (counters are 16-byte aligned, E is a multiple of 4)
#pragma unroll
      for (int k = 0; k < E; k += 4) {
        const int4 a = *reinterpret_cast<const int4 *>(counts + i0 + k);
        v[k] = a.x; v[k + 1] = a.y; v[k + 2] = a.z; v[k + 3] = a.w;
        if (counts2) {
          const int4 c = *reinterpret_cast<const int4 *>(counts2 + i0 + k);
          v[k] += c.x; v[k + 1] += c.y; v[k + 2] += c.z; v[k + 3] += c.w;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < E; ++k) v[k] = (i0 + k < M) ? counts[i0 + k] + (counts2 ? counts2[i0 + k] : 0) : 0;
    }
    int32_t mine = 0;
#pragma unroll
    for (int k = 0; k < E; ++k) mine += v[k];
    // inclusive wave scan
    int32_t s = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int32_t o = __shfl_up(s, d, 64);
      if (lane >= d) s += o;
    }
    if (lane == 63) wave_sums[wid] = s;
    __syncthreads();
    int32_t wave_off = 0, block_total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int32_t ws = wave_sums[w];
      wave_off += w < wid ? ws : 0;
      block_total += ws;
    }
    int32_t run = carry_s + wave_off + s - mine;
#pragma unroll
    for (int k = 0; k < E; ++k) {
      if (i0 + k < M) offsets[i0 + k] = run;
      run += v[k];
    }
    __syncthreads();
    if (tid == 0) carry_s += block_total;
    __syncthreads();
  }
  if (tid == 0) *total = carry_s;
}

static inline void launch_scan(int64_t M, const int32_t *counts, const int32_t *counts2, int32_t *offsets, int32_t *total,
                               hipStream_t st) {
  const bool aligned = ((((uintptr_t)counts) | ((uintptr_t)counts2)) & 15) == 0 && (counts2 == nullptr || ((counts2 - counts) & 3) == 0);
  if (aligned && M > 4096)
    hipLaunchKernelGGL(k_scan_tiles<8>, dim3(1), dim3(1024), 0, st, M, counts, counts2, offsets, total);
  else if (aligned)
    hipLaunchKernelGGL(k_scan_tiles<4>, dim3(1), dim3(1024), 0, st, M, counts, counts2, offsets, total);
  else
    hipLaunchKernelGGL(k_scan_tiles<1>, dim3(1), dim3(1024), 0, st, M, counts, counts2, offsets, total);
}

// 16 lanes cooperate on one (camera, Gaussian): lane s takes tiles s, s+16, ... of its AABB, so the
// returning atomics of one Gaussian are in flight together instead of one after the other
// (the per-Gaussian loop was latency-bound: ~7 dependent atomic round trips per lane).
// With tile_slots most keys need no atomic at all and at large N 16 lanes per Gaussian only multiply the thread count
// (16M threads at 1M Gaussians): so_isect_fill picks 2..16 lanes from C*N.
template <int kScatterLanes>
__global__ void __launch_bounds__(256)
k_isect_scatter(int C, int N, const float *__restrict__ means2d, const int32_t *__restrict__ radii,
                const float *__restrict__ depths, float tile_size, int tile_w, int tile_h,
                const int32_t *__restrict__ offsets, int32_t *__restrict__ cursor, int64_t capacity,
                uint64_t *__restrict__ key_buf, int32_t *__restrict__ overflow,
                const int32_t *__restrict__ tile_slots, const int32_t *__restrict__ n_isects,
                const float4 *__restrict__ cull_rec, int wrap_flags) {
  const int64_t total = (int64_t)C * N;
  const int n_tiles = tile_w * tile_h;
  const int sub = threadIdx.x & (kScatterLanes - 1);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x / kScatterLanes;
  for (int64_t idx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kScatterLanes; idx < total; idx += stride) {
    const int r = radii[idx];
    if (r <= 0) continue;
    const float2 m = *reinterpret_cast<const float2 *>(means2d + 2 * idx);
    const TileBox b = tile_box(m.x, m.y, (float)r, tile_size, tile_w, tile_h, wrap_for(wrap_flags, (int)(idx / N)));
    const int nx = b.x1 - b.x0, cnt = nx * (b.y1 - b.y0);
    const uint64_t key = ((uint64_t)__float_as_uint(depths[idx]) << 32) | (uint64_t)(uint32_t)idx;
    const int64_t row = (idx / N) * n_tiles;
    // tile_slots: the histogram pass (so_preprocess_fwd) kept the slot each of its returning atomics handed out for
    // rectangles of <= SO_TILE_SLOTS tiles (row-major) -- those keys are placed without atomics.  It counted larger
    // rectangles apart, in `cursor`; they fill the tail of the tile's list from the back, counting `cursor` down to 0.
    const bool slotted = tile_slots != nullptr;
    const bool have = slotted && cnt <= SO_TILE_SLOTS;
    // exact tile culling (so_common.hpp tile_touches): the same test, on the same float32 values, as the histogram pass
    float qa = 0.f, qb = 0.f, qc = 0.f, tau = 0.f;
    if (cull_rec) {
      const float4 q0 = cull_rec[4 * idx], q1 = cull_rec[4 * idx + 1];
      qa = q0.z; qb = q0.w; qc = q1.x;
      tau = cull_tau(q1.y);
    }
    for (int k = sub; k < cnt; k += kScatterLanes) {
      const int y = b.y0 + k / nx, x = b.x0 + k % nx;
      if (cull_rec && !tile_touches(m.x, m.y, qa, qb, qc, tau, x, y, tile_size)) continue;
      const int64_t t = row + y * tile_w + wrapx(x, tile_w);
      int64_t pos;
      if (have) {
        pos = (int64_t)offsets[t] + tile_slots[idx * SO_TILE_SLOTS + k];
      } else if (slotted) {
        const int64_t end = (t == (int64_t)C * n_tiles - 1) ? (int64_t)*n_isects : (int64_t)offsets[t + 1];
        pos = end - atomicSub(cursor + t, 1);
      } else {
        pos = (int64_t)offsets[t] + atomicAdd(cursor + t, 1);
      }
      if (pos < capacity) key_buf[pos] = key;
      else if (overflow) *overflow = 1;
    }
  }
}

// The merge network used for lists beyond the register sort (bitonic_merge_runs_shared below) has every comparison ascending
// ("flip" then "disperse" steps): elements past n behave as +inf and never move, so ragged sizes need no padding.  Index
// arithmetic in shifts and masks: with `i / hk`, `i / j`, `i % j` on run-time values the compiler emitted three integer
// divisions per compare-exchange, ~100 instructions for 10 of work (round 4).
__device__ __forceinline__ void tile_range(int64_t t, int64_t M, const int32_t *offsets, const int32_t *n_isects,
                                           int64_t capacity, int64_t &lo, int64_t &hi) {
  if (!n_isects && capacity < 0) {   // binned lists: cap = -capacity slots per tile, `offsets` holds the per-tile counts
    const int64_t cap = -capacity, cnt = offsets[bin_counter_index(t, M)];
    lo = t * cap;
    hi = lo + (cnt < cap ? cnt : cap);
    return;
  }
  lo = offsets[t];
  hi = (t == M - 1) ? (int64_t)*n_isects : (int64_t)offsets[t + 1];
  if (hi > capacity) hi = capacity;
  if (lo > hi) lo = hi;
}

// The last eight steps of a merge level (partner distances 128 ... 1, every comparison ascending: the flip-form network)
// on 256 consecutive keys held by one wave, element index = lane + 64 e: two steps between registers
// of the same lane, six on shuffles.
__device__ __forceinline__ void wave_merge_tail_256(uint64_t (&v)[4], int lane) {
#pragma unroll
  for (int je = 2; je >= 1; je >>= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if ((e & je) == 0) {
        const uint64_t x = v[e], y = v[e | je];
        v[e] = x < y ? x : y;
        v[e | je] = x < y ? y : x;
      }
    }
  }
#pragma unroll
  for (int j = 32; j >= 1; j >>= 1) {
    const bool upper = (lane & j) != 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint64_t o = shfl_xor_u64(v[e], j);
      v[e] = upper ? (v[e] < o ? o : v[e]) : (v[e] < o ? v[e] : o);
    }
  }
}

// Merge levels 512 ... of the flip-form network over n keys in LDS (n a multiple of 256, the keys standing in sorted runs
// of 256): per level the flip step and the steps with partner distance >= 256 run in LDS, one barrier each, and the eight
// steps below that in registers, each wave taking whole 256-key blocks -- 21 LDS steps + 6 register tails for 16384 keys
// instead of 69 LDS steps (of which the 30 with partner distances below 32 were bank-conflicted).
template <int THREADS>
__device__ __forceinline__ void bitonic_merge_runs_shared(uint64_t *keys, int n) {
  int lnp2 = 0;
  while ((1 << lnp2) < n) ++lnp2;
  const int half = (1 << lnp2) >> 1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int lk = 9; lk <= lnp2; ++lk) {
    const int k = 1 << lk, hk = k >> 1;
    for (int i = threadIdx.x; i < half; i += THREADS) {  // flip
      const int blk = i >> (lk - 1), off = i & (hk - 1);
      const int a = (blk << lk) + off, b = (blk << lk) + k - 1 - off;
      if (b < n) {
        const uint64_t ka = keys[a], kb = keys[b];
        if (ka > kb) { keys[a] = kb; keys[b] = ka; }
      }
    }
    __syncthreads();
    for (int lj = lk - 2; lj >= 8; --lj) {  // disperse, partner distance >= 256
      const int j = 1 << lj;
      for (int i = threadIdx.x; i < half; i += THREADS) {
        const int a = ((i >> lj) << (lj + 1)) + (i & (j - 1)), b = a + j;
        if (b < n) {
          const uint64_t ka = keys[a], kb = keys[b];
          if (ka > kb) { keys[a] = kb; keys[b] = ka; }
        }
      }
      __syncthreads();
    }
    for (int c = wave; c < (n >> 8); c += THREADS / 64) {
      uint64_t v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = keys[256 * c + lane + 64 * e];
      wave_merge_tail_256(v, lane);
#pragma unroll
      for (int e = 0; e < 4; ++e) keys[256 * c + lane + 64 * e] = v[e];
    }
    __syncthreads();
  }
}

// Four INDEPENDENT sequences of up to 64 keys, one per register element (element e of lane l = key l of sequence e), sorted
// side by side: the 21 shuffle steps of the 64-key network, each issued for the four sequences back to back so that their
// ds_bpermute round trips overlap (one sequence at a time, a wave waits out every round trip on its own).
__device__ __forceinline__ void wave_bitonic_sort_4x64(uint64_t (&v)[4], int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j >= 1; j >>= 1) {
      const bool upper = (lane & j) != 0;
      const bool asc = (lane & k) == 0 || k == 64;
      const bool take_min = (asc != upper);
      uint64_t o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = shfl_xor_u64(v[e], j);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = take_min ? (v[e] < o[e] ? v[e] : o[e]) : (v[e] < o[e] ? o[e] : v[e]);
    }
  }
}

// The work list of long tiles: *long_count = number of entries in bits 0..29; bit 30 = "some tile holds more than one section
// of kLongSection keys" -- only then has the rank-merge pass of k_tile_sort_long anything to do, and it costs 28 us just to
// walk the list and find that out.
constexpr int kLongSection = 16384;
constexpr int32_t kLongMerge = 1 << 30;
__device__ __forceinline__ void push_long_tile(int32_t *long_list, int32_t *long_count, int64_t t, int64_t L) {
  long_list[atomicAdd(long_count, 1) & (kLongMerge - 1)] = (int32_t)t;
  if (L > kLongSection) atomicOr(long_count, kLongMerge);
}

// LDS sort for lists with L <= CAP (CAP keys of 8 B in LDS); longer lists are appended to the
// work list (long_list[0..*long_count)) for k_tile_sort_long.
template <int THREADS, int CAP>
__global__ void __launch_bounds__(THREADS)
k_tile_sort_lds(int64_t M, int n_tiles, int tile_bits, const int32_t *__restrict__ offsets,
                const int32_t *__restrict__ n_isects, int64_t capacity, const uint64_t *__restrict__ key_buf,
                int32_t *__restrict__ flatten_ids, int64_t *__restrict__ isect_ids, int32_t *__restrict__ long_list,
                int32_t *__restrict__ long_count) {
  extern __shared__ __attribute__((aligned(16))) uint64_t s_keys[];
  for (int64_t t = blockIdx.x; t < M; t += gridDim.x) {
    int64_t lo, hi;
    tile_range(t, M, offsets, n_isects, capacity, lo, hi);
    const int64_t L = hi - lo;
    if (L > CAP) {
      if (threadIdx.x == 0) push_long_tile(long_list, long_count, t, L);
      continue;
    }
    if (L <= 0) continue;
    if (L <= 256) {   // one wave, registers only, no barrier; the other waves of the block move on (8 keys per lane,
                      // 512-key lists, measured no faster than the LDS network)
      if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (L <= 64) wave_sort_list<1>(key_buf, lo, (int)L, lane, t, n_tiles, tile_bits, flatten_ids, isect_ids);
        else if (L <= 128) wave_sort_list<2>(key_buf, lo, (int)L, lane, t, n_tiles, tile_bits, flatten_ids, isect_ids);
        else wave_sort_list<4>(key_buf, lo, (int)L, lane, t, n_tiles, tile_bits, flatten_ids, isect_ids);
      }
      continue;
    }
    sort_mid_chunks<THREADS>(s_keys, key_buf, lo, (int)L, t, n_tiles, tile_bits, flatten_ids, isect_ids);
  }
}

// The same with one tile per WAVE (sparse scenes: nearly every list fits the register sort, so no wave idles) -- 
// LDS sort for lists with L <= CAP (CAP keys of 8 B in LDS); longer lists are appended to the
// work list (long_list[0..*long_count)) for k_tile_sort_long.
template <int THREADS, int CAP>
__global__ void __launch_bounds__(THREADS)
k_tile_sort_waves(int64_t M, int n_tiles, int tile_bits, const int32_t *__restrict__ offsets,
                const int32_t *__restrict__ n_isects, int64_t capacity, const uint64_t *__restrict__ key_buf,
                int32_t *__restrict__ flatten_ids, int64_t *__restrict__ isect_ids, int32_t *__restrict__ long_list,
                int32_t *__restrict__ long_count) {
  extern __shared__ __attribute__((aligned(16))) uint64_t s_keys[];
  constexpr int WAVES = THREADS / 64;
  __shared__ int s_mid[WAVES];   // tiles of this group that need the whole workgroup (256 < L <= CAP), -1 = none
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // A workgroup takes WAVES consecutive tiles at a time: every wave sorts its own tile in registers (the common case:
  // no LDS array, no barrier, no idle wave), then the workgroup does the few medium lists of the group together.
  for (int64_t base = (int64_t)blockIdx.x * WAVES; base < M; base += (int64_t)gridDim.x * WAVES) {
    const int64_t t = base + wave;
    int64_t lo = 0, hi = 0;
    if (t < M) tile_range(t, M, offsets, n_isects, capacity, lo, hi);
    const int64_t L = hi - lo;
    if (lane == 0) s_mid[wave] = (L > 256 && L <= CAP) ? 1 : 0;
    if (L > CAP) {
      if (lane == 0) push_long_tile(long_list, long_count, t, L);
    } else if (L > 0 && L <= 256) {
      if (L <= 64) wave_sort_list<1>(key_buf, lo, (int)L, lane, t, n_tiles, tile_bits, flatten_ids, isect_ids);
      else if (L <= 128) wave_sort_list<2>(key_buf, lo, (int)L, lane, t, n_tiles, tile_bits, flatten_ids, isect_ids);
      else wave_sort_list<4>(key_buf, lo, (int)L, lane, t, n_tiles, tile_bits, flatten_ids, isect_ids);   // (8 keys per lane,
                                                  // 512-key lists, measured no faster than the LDS network)
    }
    __syncthreads();
    for (int w = 0; w < WAVES; ++w) {
      if (!s_mid[w]) continue;                    // uniform over the workgroup
      const int64_t tm = base + w;
      int64_t mlo, mhi;
      tile_range(tm, M, offsets, n_isects, capacity, mlo, mhi);
      const int64_t ML = mhi - mlo;
      sort_mid_chunks<THREADS>(s_keys, key_buf, mlo, (int)ML, tm, n_tiles, tile_bits, flatten_ids, isect_ids);
    }
    __syncthreads();                              // s_mid is rewritten by the next group
  }
}

// The flip-form network on n keys of an LDS sub-array, in place, any n (keys past n behave as +inf): the fall-back for one
// oversized bucket of sort_section_buckets (every thread of the workgroup calls it).
template <int THREADS>
__device__ __forceinline__ void bitonic_sort_ragged_shared(uint64_t *keys, int n) {
  int lnp2 = 0;
  while ((1 << lnp2) < n) ++lnp2;
  const int half = (1 << lnp2) >> 1;
  for (int lk = 1; lk <= lnp2; ++lk) {
    const int k = 1 << lk, hk = k >> 1;
    for (int i = threadIdx.x; i < half; i += THREADS) {  // flip
      const int blk = i >> (lk - 1), off = i & (hk - 1);
      const int a = (blk << lk) + off, b = (blk << lk) + k - 1 - off;
      if (b < n) {
        const uint64_t ka = keys[a], kb = keys[b];
        if (ka > kb) { keys[a] = kb; keys[b] = ka; }
      }
    }
    __syncthreads();
    for (int lj = lk - 2; lj >= 0; --lj) {  // disperse
      const int j = 1 << lj;
      for (int i = threadIdx.x; i < half; i += THREADS) {
        const int a = ((i >> lj) << (lj + 1)) + (i & (j - 1)), b = a + j;
        if (b < n) {
          const uint64_t ka = keys[a], kb = keys[b];
          if (ka > kb) { keys[a] = kb; keys[b] = ka; }
        }
      }
      __syncthreads();
    }
  }
}

// One section (<= 16384 keys) of a long list, sorted by DEPTH BUCKETS (end of round 4).  The merge network spends n log^2 n
// 64-bit compare-exchanges (a 12 500-key section: ~125 us, issue-bound); here the keys are dealt into up to 448 buckets by a
// monotone map of the depth word (same depth -> same bucket, larger -> same or later bucket), each bucket -- ~42 keys -- is
// sorted by ONE wave in registers on the full 64-bit key, and the buckets are already in order: ~5x fewer instructions.
//   A  minimum / maximum depth of the section                       (keys read from global memory; they are L2-resident)
//   B  histogram of the buckets (LDS atomics)          C  exclusive scan (one wave)
//   D  keys re-read and scattered to their bucket's range of s_keys (returning LDS atomics: order inside a bucket is arbitrary)
//   E  every wave sorts four buckets at a time side by side in registers (a bucket of 65 .. 256 keys: on its own, two or four
//      registers per lane) and emits them at bucket base + rank
//   F  a bucket with more than 256 keys (many equal depths, a depth distribution with a spike): the ragged network on its
//      range of s_keys, by the whole workgroup, then emitted
// Same result as any correct sort of the distinct 64-bit keys.  emit(position in the section, key).
constexpr int kSortBuckets = 512;
template <int THREADS, class Emit>
__device__ __forceinline__ void sort_section_buckets(uint64_t *s_keys, const uint64_t *keys, int sn, Emit emit) {
  __shared__ int32_t s_cnt[kSortBuckets], s_base[kSortBuckets], s_cur[kSortBuckets];
  __shared__ uint32_t s_lohi[2][THREADS / 64];
  __shared__ int32_t s_big[kSortBuckets], s_nbig;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // ~42 keys per bucket on average (a multiple of 64 buckets, 448 at most): the 64-lane sequences of the register sort are
  // two thirds full and a Poisson-like fill passes 64 in one bucket of 10^3
  const int nb = 64 * ((sn + 64 * 42 - 1) / (64 * 42));
  // A (the section's keys stay in registers for B and D: sixteen loads per thread in flight at once, not three walks over
  // global memory with a round trip per key)
  static_assert(THREADS * 16 >= 16384, "sixteen keys per thread cover a section");
  uint64_t kreg[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) kreg[j] = tid + THREADS * j < sn ? keys[tid + THREADS * j] : ~0ull;
  uint32_t dlo = 0xffffffffu, dhi = 0u;
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (tid + THREADS * j < sn) {
      const uint32_t d = (uint32_t)(kreg[j] >> 32);
      dlo = d < dlo ? d : dlo;
      dhi = d > dhi ? d : dhi;
    }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const uint32_t ol = (uint32_t)__shfl_xor((int)dlo, m, 64), oh = (uint32_t)__shfl_xor((int)dhi, m, 64);
    dlo = ol < dlo ? ol : dlo;
    dhi = oh > dhi ? oh : dhi;
  }
  if (lane == 0) { s_lohi[0][wave] = dlo; s_lohi[1][wave] = dhi; }
  for (int i = tid; i < kSortBuckets; i += THREADS) s_cnt[i] = 0;
  if (tid == 0) s_nbig = 0;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < THREADS / 64; ++w) {
    dlo = s_lohi[0][w] < dlo ? s_lohi[0][w] : dlo;
    dhi = s_lohi[1][w] > dhi ? s_lohi[1][w] : dhi;
  }
  // the order wanted is that of the 64-bit keys, i.e. of the depth's BIT PATTERN as an unsigned integer: the map works on
  // that integer (difference to the minimum -> float -> scaled -> truncated: every step monotone, whatever its rounding)
  const float scale = dhi > dlo ? (float)nb / (float)(dhi - dlo) : 0.f;
  auto bucket_of = [&](uint64_t key) {
    const int b = (int)((float)((uint32_t)(key >> 32) - dlo) * scale);
    return b >= nb ? nb - 1 : b;
  };
  // B
#if defined(SO_SORT_ABL) && SO_SORT_ABL >= 3
  if (false)
#endif
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (tid + THREADS * j < sn) atomicAdd(&s_cnt[bucket_of(kreg[j])], 1);
  __syncthreads();
  // C (one wave: nb / 64 counts per lane)
  if (wave == 0) {
    const int per = nb >> 6;                       // 1 .. 6
    int mine = 0;
    for (int j = 0; j < per; ++j) mine += s_cnt[lane * per + j];
    int sc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(sc, d, 64);
      if (lane >= d) sc += o;
    }
    int run = sc - mine;
    for (int j = 0; j < per; ++j) {
      const int b = lane * per + j;
      s_base[b] = run;
      s_cur[b] = run;
      run += s_cnt[b];
    }
  }
  __syncthreads();
  // D
#if defined(SO_SORT_ABL) && SO_SORT_ABL >= 2
  if (false)
#endif
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (tid + THREADS * j < sn) s_keys[atomicAdd(&s_cur[bucket_of(kreg[j])], 1)] = kreg[j];
  __syncthreads();
  // E (a wave takes four neighbouring buckets at a time: all four small -- the rule -- go through the side-by-side sort.
  // Joining neighbouring buckets into fuller 64-key segments first -- buckets are in key order, so that is allowed -- was
  // tried: the greedy cut is a serial walk over the bucket counts, 25 us per section for one lane, more than it saves.)
  const int nseg = nb;
  const int32_t *s_seg = s_base, *s_seglen = s_cnt;
#if defined(SO_SORT_ABL) && SO_SORT_ABL >= 1   // ablation builds (timing only, WRONG lists): 1 = no bucket sorts, 2 = no scatter either, 3 = nothing
  if (false)
#endif
  for (int g0 = 4 * wave; g0 < nseg; g0 += 4 * (THREADS / 64)) {
    int cnt4[4], base4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { cnt4[e] = g0 + e < nseg ? s_seglen[g0 + e] : 0; base4[e] = g0 + e < nseg ? s_seg[g0 + e] : 0; }
    if (cnt4[0] <= 64 && cnt4[1] <= 64 && cnt4[2] <= 64 && cnt4[3] <= 64) {
      uint64_t v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = lane < cnt4[e] ? s_keys[base4[e] + lane] : ~0ull;
#if !defined(SO_SORT_SKIP_SORT4)
      wave_bitonic_sort_4x64(v, lane);
#endif
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (lane < cnt4[e]) emit(base4[e] + lane, v[e]);
      continue;
    }
    for (int e4 = 0; e4 < 4; ++e4) {
      const int cnt = cnt4[e4], base = base4[e4];
      if (cnt == 0) continue;
      if (cnt <= 64) {
        uint64_t v[1] = {lane < cnt ? s_keys[base + lane] : ~0ull};
        wave_bitonic_sort<1>(v, lane);
        if (lane < cnt) emit(base + lane, v[0]);
      } else if (cnt <= 128) {
        uint64_t v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) v[e] = lane + 64 * e < cnt ? s_keys[base + lane + 64 * e] : ~0ull;
        wave_bitonic_sort<2>(v, lane);
#pragma unroll
        for (int e = 0; e < 2; ++e)
          if (lane + 64 * e < cnt) emit(base + lane + 64 * e, v[e]);
      } else if (cnt <= 256) {
        uint64_t v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = lane + 64 * e < cnt ? s_keys[base + lane + 64 * e] : ~0ull;
        wave_bitonic_sort<4>(v, lane);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (lane + 64 * e < cnt) emit(base + lane + 64 * e, v[e]);
      } else if (lane == 0) {
        s_big[atomicAdd(&s_nbig, 1)] = g0 + e4;
      }
    }
  }
  __syncthreads();
  // F
#if defined(SO_SORT_SKIP_F)
  const int nbig = 0;
#else
  const int nbig = s_nbig;
#endif
  for (int k = 0; k < nbig; ++k) {
    const int g = s_big[k], cnt = s_seglen[g], base = s_seg[g];
    bitonic_sort_ragged_shared<THREADS>(s_keys + base, cnt);
    for (int i = tid; i < cnt; i += THREADS) emit(base + i, s_keys[base + i]);
  }
  __syncthreads();                                  // s_keys and the bucket tables are rewritten by the next section
}

// Long lists (work list built by k_tile_sort_lds): a fixed grid walks the list, so the launches cost next to nothing when no
// tile is long.  The unit of work is a SECTION of CAP keys of a long tile, sections dealt round-robin over the workgroups
// (round 4: until then a workgroup took a whole tile, and the one tile with 40k keys of a gathered cloud kept one CU busy
// for 400 us after all others had finished):
//   pass 1 (MERGE = false)  a section's runs of 256 keys are sorted in registers, merged by the network in 128 KiB of LDS;
//                           the only section of a tile (<= CAP keys) goes straight to flatten_ids, otherwise the sorted
//                           keys are written back in place;
//   pass 2 (MERGE = true)   tiles of two or more sections: every key of a section finds its final place by itself -- its
//                           index in its own section plus, for each other section, the number of keys below it (a
//                           branch-free binary search, the sections being L2-resident; until round 4 the whole network ran
//                           on the global key buffer, ~120 barrier-separated passes for 30k keys).  Keys are distinct (the
//                           id is the low word), so that is the rank: the same order as any correct sort.
// The kernel boundary is the only synchronisation between the passes.
template <int THREADS, int CAP, bool MERGE>
__global__ void __launch_bounds__(THREADS)
k_tile_sort_long(int64_t M, int n_tiles, int tile_bits, const int32_t *__restrict__ offsets,
                 const int32_t *__restrict__ n_isects, int64_t capacity, uint64_t *__restrict__ key_buf,
                 int32_t *__restrict__ flatten_ids, int64_t *__restrict__ isect_ids,
                 const int32_t *__restrict__ long_list, const int32_t *__restrict__ long_count) {
  extern __shared__ __attribute__((aligned(16))) uint64_t s_keys[];
  static_assert(CAP % 256 == 0 && THREADS % 64 == 0 && THREADS <= 1024, "whole runs, whole waves");
  __shared__ int32_t s_wave_sums[THREADS / 64];
  __shared__ int32_t s_pick[2];
  static_assert(CAP == kLongSection, "the merge flag is set against kLongSection");
  const int32_t packed = *long_count;
  const int n_long = packed & (kLongMerge - 1);
  if (MERGE && !(packed & kLongMerge)) return;     // no tile of several sections: nothing to merge
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int first = (int)blockIdx.x;          // this workgroup's first section of the current batch of THREADS long tiles
  for (int w0 = 0; w0 < n_long; w0 += THREADS) {
    // every thread looks at one long tile: how many sections?  then a block-wide exclusive scan numbers the sections
    int32_t nsec = 0;
    if (w0 + tid < n_long) {
      int64_t lo, hi;
      tile_range(long_list[w0 + tid], M, offsets, n_isects, capacity, lo, hi);
      nsec = (int32_t)((hi - lo + CAP - 1) / CAP);
      if (MERGE && nsec == 1) nsec = 0;
    }
    int32_t sc = nsec;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int32_t o = __shfl_up(sc, d, 64);
      if (lane >= d) sc += o;
    }
    if (lane == 63) s_wave_sums[wave] = sc;
    __syncthreads();
    int32_t wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) {
      const int32_t ws = s_wave_sums[w];
      wave_off += w < wave ? ws : 0;
      total += ws;
    }
    const int32_t p0 = wave_off + sc - nsec;
    int q = first;
    for (; q < total; q += (int)gridDim.x) {
      if (p0 <= q && q < p0 + nsec) { s_pick[0] = w0 + tid; s_pick[1] = q - p0; }
      __syncthreads();
      const int64_t t = long_list[s_pick[0]];
      const int sec = s_pick[1];
      __syncthreads();
      int64_t lo, hi;
      tile_range(t, M, offsets, n_isects, capacity, lo, hi);
      const int64_t n = hi - lo;
      const int nsec_t = (int)((n + CAP - 1) / CAP);
      uint64_t *keys = key_buf + lo;
      const int64_t s0 = (int64_t)sec * CAP;
      const int sn = (int)(n - s0 < CAP ? n - s0 : CAP);
      if constexpr (!MERGE) {
        // (until the end of round 4: runs of 256 keys sorted in registers + bitonic_merge_runs_shared over the section)
        // (ONE instantiation for both destinations: the function's bucket tables are static LDS arrays)
        const bool only_section = nsec_t == 1;
        sort_section_buckets<THREADS>(s_keys, keys + s0, sn, [&](int pos, uint64_t key) {
          if (only_section) write_sorted(key, lo + pos, t, n_tiles, tile_bits, flatten_ids, isect_ids);
          else keys[s0 + pos] = key;
        });
      } else {
        constexpr int U = 4;                               // searches in flight per thread
        for (int i0 = tid * U; i0 < sn; i0 += THREADS * U) {
          uint64_t key[U];
          int64_t rank[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            key[u] = i0 + u < sn ? keys[s0 + i0 + u] : ~0ull;
            rank[u] = i0 + u;
          }
          for (int other = 0; other < nsec_t; ++other) {
            if (other == sec) continue;
            const uint64_t *ch = keys + (int64_t)other * CAP;
            const int len = (int)(n - (int64_t)other * CAP < CAP ? n - (int64_t)other * CAP : CAP);
            int pos[U];
#pragma unroll
            for (int u = 0; u < U; ++u) pos[u] = 0;
            for (int step = CAP; step >= 1; step >>= 1) {
#pragma unroll
              for (int u = 0; u < U; ++u) {
                const int p = pos[u] + step;
                if (p <= len && ch[p - 1] < key[u]) pos[u] = p;
              }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) rank[u] += pos[u];
          }
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (i0 + u < sn) write_sorted(key[u], lo + rank[u], t, n_tiles, tile_bits, flatten_ids, isect_ids);
        }
      }
    }
    first = q - total;                   // where this workgroup's round-robin turn falls in the next batch
    __syncthreads();                     // s_wave_sums is rewritten by the next batch
  }
}

__global__ void __launch_bounds__(256)
k_isect_emit_unsorted(int C, int N, const float *__restrict__ means2d, const int32_t *__restrict__ radii,
                      const float *__restrict__ depths, const int64_t *__restrict__ cum_tiles, float tile_size,
                      int tile_w, int tile_h, int tile_bits, int64_t *__restrict__ isect_ids,
                      int32_t *__restrict__ flatten_ids, int wrap_flags) {
  const int64_t total = (int64_t)C * N;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int r = radii[idx];
    if (r <= 0) continue;
    const float2 m = *reinterpret_cast<const float2 *>(means2d + 2 * idx);
    const TileBox b = tile_box(m.x, m.y, (float)r, tile_size, tile_w, tile_h, wrap_for(wrap_flags, (int)(idx / N)));
    int64_t cur = (idx == 0) ? 0 : cum_tiles[idx - 1];
    const int64_t cam_enc = (idx / N) << (32 + tile_bits);
    const int64_t dbits = (int64_t)__float_as_uint(depths[idx]);
    for (int y = b.y0; y < b.y1; ++y)
      for (int x = b.x0; x < b.x1; ++x) {
        isect_ids[cur] = cam_enc | ((int64_t)(y * tile_w + wrapx(x, tile_w)) << 32) | dbits;
        flatten_ids[cur] = (int32_t)idx;
        ++cur;
      }
  }
}

__global__ void __launch_bounds__(256)
k_isect_offset_encode(int64_t n_isects, const int64_t *__restrict__ isect_ids, int C, int n_tiles, int tile_bits,
                      int32_t *__restrict__ offsets) {
  const int64_t M = (int64_t)C * n_tiles;
  const int64_t tile_mask = ((int64_t)1 << tile_bits) - 1;
  // offsets[t] = lower_bound of t in the sorted (camera, tile) sequence
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < M; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = n_isects;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      const int64_t key = isect_ids[mid] >> 32;
      const int64_t lin = (key >> tile_bits) * n_tiles + (key & tile_mask);
      if (lin < t) lo = mid + 1; else hi = mid;
    }
    offsets[t] = (int32_t)lo;
  }
}

static inline int tile_bits_of(int n_tiles) {
  int b = 0;
  while ((1 << b) <= n_tiles) ++b;  // floor(log2(n)) + 1
  return n_tiles > 0 ? b : 0;
}

static inline int grid_1d(int64_t total, int block, int cap = 4096) {
  int64_t g = ceil_div(total, block);
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace so

namespace so {
// the per-tile sorts over lists given either compactly (offsets / n_isects / capacity) or binned (n_isects NULL,
// capacity = -slots per tile, offsets = per-tile counts): see tile_range
static void launch_tile_sorts(int64_t M, int n_tiles, int tb, const int32_t *isect_offsets, const int32_t *n_isects,
                              int64_t capacity, uint64_t *key_buf, int32_t *flatten_ids, int64_t *isect_ids,
                              int32_t *long_list, int32_t *long_count, hipStream_t st) {
  const int gridM = (int)(M < 65535 * 8 ? M : 65535 * 8);
  // sparse scenes (few slots per tile on average): one tile per wave; dense ones (most lists 257..2048 keys): one tile
  // per workgroup -- measured 14.7 -> 11.7 us on c2 and 137 -> 176 us on the dense init regime with the former
  // (all the host knows is the capacity: bins hold ~8x the fullest tile, a compact buffer ~1-2x the total)
  const bool binned = !n_isects && capacity < 0;
  const bool per_wave = binned ? -capacity <= 4096 : (M > 0 && capacity / M <= 192);
  const int64_t groups = (M + 3) / 4;
  const int gridW = (int)(groups < 65535 * 8 ? groups : 65535 * 8);
  static bool lds_attr_set = false;  // 128 KiB of dynamic LDS needs an explicit opt-in
  if (!lds_attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_sort_long<1024, 16384, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8) != hipSuccess) {
      (void)hipGetLastError();
    }
    lds_attr_set = true;
  }
  // lists up to 2048 keys: 256 threads, 16 KiB LDS (4096 keys / 32 KiB costs the sparse regime 14 us of occupancy; most lists of a trained scene take the
  // one-wave register path anyway, and in the dense init regime -- ~800 keys per tile -- this kernel does the bulk)
  if (per_wave)
    hipLaunchKernelGGL((k_tile_sort_waves<256, 2048>), dim3(gridW), dim3(256), 2048 * 8, st, M, n_tiles, tb,
                       isect_offsets, n_isects, capacity, key_buf, flatten_ids, isect_ids, long_list, long_count);
  else
    hipLaunchKernelGGL((k_tile_sort_lds<256, 2048>), dim3(gridM), dim3(256), 2048 * 8, st, M, n_tiles, tb,
                       isect_offsets, n_isects, capacity, key_buf, flatten_ids, isect_ids, long_list, long_count);
  // longer lists: one workgroup per CU (128 KiB of LDS each) over the sections of the work list -- the grid is fixed at
  // launch, the list length is only known on the device
  // (binned lists whose bins hold no more than the first kernel sorts cannot have a long tile: nothing to launch; bins of
  // no more than one section need no merge pass)
  if (binned && -capacity <= 2048) return;
  hipLaunchKernelGGL((k_tile_sort_long<1024, 16384, false>), dim3(256), dim3(1024), 16384 * 8, st, M, n_tiles, tb,
                     isect_offsets, n_isects, capacity, key_buf, flatten_ids, isect_ids, long_list, long_count);
  if (binned && -capacity <= 16384) return;
  hipLaunchKernelGGL((k_tile_sort_long<1024, 16384, true>), dim3(256), dim3(1024), 0, st, M, n_tiles, tb,
                     isect_offsets, n_isects, capacity, key_buf, flatten_ids, isect_ids, long_list, long_count);
}
}  // namespace so

extern "C" int so_isect_count(int C, int N, const float *means2d, const int32_t *radii, int tile_size,
                              int tile_width, int tile_height, int32_t *tiles_per_gauss, int32_t *tile_counts,
                              int32_t *isect_offsets, int32_t *n_isects, const float *cull_rec, void *stream) {
  const int wrap_flags = tile_size & ~0xFF;
  tile_size = so::tile_size_of(tile_size);
  SO_REQUIRE(C >= 0 && N >= 0 && tile_size > 0 && tile_width > 0 && tile_height > 0, "so_isect_count: bad sizes");
  SO_REQUIRE(tile_counts && isect_offsets && n_isects, "so_isect_count: null pointer");
  SO_REQUIRE((int64_t)C * N < ((int64_t)1 << 31), "so_isect_count: C*N must fit int32 flatten ids");
  hipStream_t st = so::as_stream(stream);
  const int64_t M = (int64_t)C * tile_width * tile_height;
  if ((int64_t)C * N > 0) {
    SO_REQUIRE(means2d && radii && tiles_per_gauss, "so_isect_count: null pointer");
    hipLaunchKernelGGL(so::k_isect_count, dim3(so::grid_1d((int64_t)C * N, 256)), dim3(256), 0, st, C, N, means2d,
                       radii, (float)tile_size, tile_width, tile_height, tiles_per_gauss, tile_counts,
                       reinterpret_cast<const float4 *>(cull_rec), wrap_flags);
  }
  so::launch_scan(M, tile_counts, nullptr, isect_offsets, n_isects, st);
  return so::check_launch("so_isect_count");
}

extern "C" int so_isect_scan(int C, int tile_width, int tile_height, const int32_t *tile_counts,
                             const int32_t *tile_counts_big, int32_t *isect_offsets, int32_t *n_isects, void *stream) {
  SO_REQUIRE(C >= 0 && tile_width > 0 && tile_height > 0, "so_isect_scan: bad sizes");
  SO_REQUIRE(tile_counts && isect_offsets && n_isects, "so_isect_scan: null pointer");
  so::launch_scan((int64_t)C * tile_width * tile_height, tile_counts, tile_counts_big, isect_offsets, n_isects, so::as_stream(stream));
  return so::check_launch("so_isect_scan");
}

extern "C" int so_isect_fill(int C, int N, const float *means2d, const int32_t *radii, const float *depths,
                             int tile_size, int tile_width, int tile_height, const int32_t *isect_offsets,
                             const int32_t *n_isects, int32_t *tile_cursor, int64_t capacity, uint64_t *key_buf,
                             int32_t *flatten_ids, int64_t *isect_ids, int32_t *overflow, const int32_t *tile_slots,
                             const float *cull_rec, void *stream) {
  const int wrap_flags = tile_size & ~0xFF;
  tile_size = so::tile_size_of(tile_size);
  SO_REQUIRE(C >= 0 && N >= 0 && tile_size > 0 && tile_width > 0 && tile_height > 0 && capacity >= 0,
             "so_isect_fill: bad sizes");
  if ((int64_t)C * N == 0 || capacity == 0) return SO_OK;
  SO_REQUIRE(means2d && radii && depths && isect_offsets && n_isects && tile_cursor && key_buf && flatten_ids,
             "so_isect_fill: null pointer");
  hipStream_t st = so::as_stream(stream);
  const int n_tiles = tile_width * tile_height;
  const int64_t M = (int64_t)C * n_tiles;
  const int tb = so::tile_bits_of(n_tiles);
  // lanes per Gaussian of the slotted scatter: few Gaussians want many lanes (parallelism: 100k Gaussians 32 us with
  // 16 lanes, 46 with 4), many Gaussians want few (thread count: 1M Gaussians 98 us with 4 lanes, 154 with 16)
  const int64_t CN = (int64_t)C * N;
  const int slotted_lanes = CN <= 300000 ? 16 : (CN <= 600000 ? 8 : (CN <= 1500000 ? 4 : 2));
#define SO_SCATTER(L)                                                                                                         \
    hipLaunchKernelGGL(so::k_isect_scatter<L>, dim3(so::grid_1d((int64_t)C * N * L, 256, 16384)), dim3(256), 0, st, C, N, means2d, \
                       radii, depths, (float)tile_size, tile_width, tile_height, isect_offsets, tile_cursor, capacity,        \
                       key_buf, overflow, tile_slots, n_isects, reinterpret_cast<const float4 *>(cull_rec), wrap_flags)
  if (tile_slots && slotted_lanes == 2) SO_SCATTER(2);
  else if (tile_slots && slotted_lanes == 4) SO_SCATTER(4);
  else if (tile_slots && slotted_lanes == 8) SO_SCATTER(8);
  else SO_SCATTER(16);
#undef SO_SCATTER
  // after the scatter the cursor array is dead: it becomes the work list of long tiles, and the
  // (caller-zeroed) element behind it is the list length
  so::launch_tile_sorts(M, n_tiles, tb, isect_offsets, n_isects, capacity, key_buf, flatten_ids, isect_ids, tile_cursor, tile_cursor + M, st);
  return so::check_launch("so_isect_fill");
}

extern "C" int so_isect_emit_unsorted(int C, int N, const float *means2d, const int32_t *radii,
                                      const float *depths, const int64_t *cum_tiles, int tile_size,
                                      int tile_width, int tile_height, int64_t *isect_ids, int32_t *flatten_ids,
                                      void *stream) {
  const int wrap_flags = tile_size & ~0xFF;
  tile_size = so::tile_size_of(tile_size);
  SO_REQUIRE(C >= 0 && N >= 0 && tile_size > 0 && tile_width > 0 && tile_height > 0, "so_isect_emit_unsorted: bad sizes");
  if ((int64_t)C * N == 0) return SO_OK;
  SO_REQUIRE(means2d && radii && depths && cum_tiles && isect_ids && flatten_ids, "so_isect_emit_unsorted: null pointer");
  const int tb = so::tile_bits_of(tile_width * tile_height);
  hipLaunchKernelGGL(so::k_isect_emit_unsorted, dim3(so::grid_1d((int64_t)C * N, 256)), dim3(256), 0,
                     so::as_stream(stream), C, N, means2d, radii, depths, cum_tiles, (float)tile_size, tile_width,
                     tile_height, tb, isect_ids, flatten_ids, wrap_flags);
  return so::check_launch("so_isect_emit_unsorted");
}

extern "C" int so_isect_offset_encode(int64_t n_isects, const int64_t *isect_ids, int C, int tile_width,
                                      int tile_height, int32_t *isect_offsets, void *stream) {
  SO_REQUIRE(n_isects >= 0 && C >= 0 && tile_width > 0 && tile_height > 0, "so_isect_offset_encode: bad sizes");
  if (C == 0) return SO_OK;
  SO_REQUIRE(isect_offsets && (n_isects == 0 || isect_ids), "so_isect_offset_encode: null pointer");
  const int n_tiles = tile_width * tile_height;
  hipLaunchKernelGGL(so::k_isect_offset_encode, dim3(so::grid_1d((int64_t)C * n_tiles, 256)), dim3(256), 0,
                     so::as_stream(stream), n_isects, isect_ids, C, n_tiles, so::tile_bits_of(n_tiles), isect_offsets);
  return so::check_launch("so_isect_offset_encode");
}

// Replicated bin counters (so_step_desc.bin_replicas = R > 1): k_preprocess_fwd filled R slices of every tile's bin, slice r
// with sub_counts[r * M + bin_counter_index(t)] keys from slot r * bin_cap / R on.  One wave per tile closes the slices up into one
// run from slot 0 (slice r moves DOWN to the sum of the clamped counts before it: its destination never reaches the source of a
// later slice, and inside a slice a chunk is read whole before it is written), writes the run's length where every consumer of
// binned lists reads it (tile_counts[bin_counter_index(t)]), zeroes the tile's sub-counters for the next iteration, and -- only if a
// slice overflowed -- raises `eff_fullest` to R x the fullest slice: the bin capacity this tile would have needed.
namespace so {
__global__ void __launch_bounds__(256)
k_bins_gather(int M, int R, int32_t *__restrict__ sub_counts, int32_t *__restrict__ tile_counts, uint64_t *__restrict__ bin_keys,
              int64_t bin_cap, int32_t *__restrict__ eff_fullest) {
  const int lane = threadIdx.x & 63;
  const int t = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (t >= M) return;                       // (wave-uniform)
  const int64_t at = bin_counter_index(t, M);
  const int cap_r = (int)(bin_cap / R);
  int raw = 0;
  if (lane < R) {
    raw = sub_counts[(int64_t)lane * M + at];
    sub_counts[(int64_t)lane * M + at] = 0;
  }
  const int cnt = raw < cap_r ? raw : cap_r;
  int inc = cnt;                            // inclusive scan over the R <= 64 slices
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  const int total = __shfl(inc, 63, 64);
  const int fullest = wave_max_i32(raw);
  if (lane == 0) {
    tile_counts[at] = total;
    if (fullest > cap_r) atomicMax(eff_fullest, fullest * R);
  }
  uint64_t *const keys = bin_keys + (int64_t)t * bin_cap;
  if (R <= 8 && (fullest < cap_r ? fullest : cap_r) <= 128) {
    // every slice holds <= 128 keys (the rule on small images): ALL slices' keys are read (two per lane and slice) before the first
    // one is written -- one memory round trip for the tile instead of one per slice (7 of them: 11 -> ~5 us at 512 x 512)
    uint64_t k[7][2];
#pragma unroll
    for (int r = 1; r < 8; ++r) {
      const int n = r < R ? __shfl(cnt, r, 64) : 0, src = r * cap_r;
#pragma unroll
      for (int u = 0; u < 2; ++u) k[r - 1][u] = 64 * u + lane < n ? keys[src + 64 * u + lane] : 0ull;
    }
#pragma unroll
    for (int r = 1; r < 8; ++r) {
      const int n = r < R ? __shfl(cnt, r, 64) : 0, dst = r < R ? __shfl(inc - cnt, r, 64) : 0;
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (64 * u + lane < n) keys[dst + 64 * u + lane] = k[r - 1][u];
    }
    return;
  }
  for (int r = 1; r < R; ++r) {             // (wave-uniform trip counts: the counts come from lane r)
    const int n = __shfl(cnt, r, 64), dst = __shfl(inc - cnt, r, 64), src = r * cap_r;
    if (dst == src) continue;
    // four chunks of 64 keys per round trip: all four loads leave before the first store (a store of chunk k lands below every
    // key the later chunks still have to read: dst + i <= src + i < src + i' for i' > i) -- 13 -> ~6 us at 512 x 512 with 520 keys per tile
    for (int i0 = 0; i0 < n; i0 += 256) {
      uint64_t k[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) k[u] = i0 + 64 * u + lane < n ? keys[src + i0 + 64 * u + lane] : 0ull;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + 64 * u + lane < n) keys[dst + i0 + 64 * u + lane] = k[u];
    }
  }
}

int bins_gather_launch(int64_t M, int R, int32_t *sub_counts, int32_t *tile_counts, uint64_t *bin_keys, int64_t bin_cap,
                       int32_t *eff_fullest, hipStream_t st) {
  SO_REQUIRE(M > 0 && M < ((int64_t)1 << 25) && R > 1 && R <= 64 && bin_cap % R == 0 && sub_counts && tile_counts && bin_keys && eff_fullest,
             "so_train_step_fwd_bwd: bad replicated-counter arguments");
  hipLaunchKernelGGL(k_bins_gather, dim3((unsigned)((M * 64 + 255) / 256)), dim3(256), 0, st, (int)M, R, sub_counts, tile_counts, bin_keys,
                     bin_cap, eff_fullest);
  return check_launch("so_train_step_fwd_bwd (bins gather)");
}
}  // namespace so

// Binned lists (so_preprocess_fwd bin_keys): sorts every tile's min(tile_counts[t], bin_cap) keys in place and writes
// flatten_ids[t * bin_cap + i].  long_list: int32[M + 1] scratch whose LAST element is zero on entry.
extern "C" int so_isect_sort_bins(int C, int tile_width, int tile_height, const int32_t *tile_counts, int64_t bin_cap,
                                  uint64_t *bin_keys, int32_t *flatten_ids, int32_t *long_list, void *stream) {
  SO_REQUIRE(C >= 0 && tile_width > 0 && tile_height > 0 && bin_cap > 0, "so_isect_sort_bins: bad sizes");
  if (C == 0) return SO_OK;
  SO_REQUIRE(tile_counts && bin_keys && flatten_ids && long_list, "so_isect_sort_bins: null pointer");
  const int n_tiles = tile_width * tile_height;
  const int64_t M = (int64_t)C * n_tiles;
  SO_REQUIRE(M * bin_cap < ((int64_t)1 << 31), "so_isect_sort_bins: C*tiles*bin_cap = %lld does not fit 31 bits", (long long)(M * bin_cap));
  so::launch_tile_sorts(M, n_tiles, so::tile_bits_of(n_tiles), tile_counts, nullptr, -bin_cap, bin_keys, flatten_ids, nullptr,
                        long_list, long_list + M, so::as_stream(stream));
  return so::check_launch("so_isect_sort_bins");
}
