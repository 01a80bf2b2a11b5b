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

__device__ __forceinline__ void write_sorted(uint64_t key, int64_t pos, int64_t t, int n_tiles, int tile_bits,
                                             int32_t *flatten_ids, int64_t *isect_ids) {
  flatten_ids[pos] = (int32_t)(uint32_t)(key & 0xffffffffull);
  if (isect_ids) {
    const int64_t cam = t / n_tiles, tile = t - cam * n_tiles;
    isect_ids[pos] = (cam << (32 + tile_bits)) | (tile << 32) | (int64_t)(key >> 32);
  }
}

// Register sort of up to 256 keys by ONE wave: lane l holds elements l, l+64, ... (UINT64_MAX pads the
// tail), the classic xor-partner bitonic network runs on ds_bpermute shuffles -- no LDS array, no
// barrier.  Short lists dominate trained-like scenes (mean ~40 keys per tile on the c2 workload, ~150 at
// 1M Gaussians / 1440p).
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
  const uint32_t lo = __shfl_xor((uint32_t)v, m, 64), hi = __shfl_xor((uint32_t)(v >> 32), m, 64);
  return ((uint64_t)hi << 32) | lo;
}

// E elements per lane (element index = lane + 64 e): up to 64 E keys sorted by ONE wave in registers.  Steps with
// j >= 64 pair two registers of the same lane (no shuffle); the sort direction of an element depends on lane bits
// only for k < 64, so everything else folds at compile time.
template <int E>
__device__ __forceinline__ void wave_bitonic_sort(uint64_t (&v)[E], int lane) {
  constexpr int NE = 64 * E;
#pragma unroll
  for (int k = 2; k <= NE; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j >= 1; j >>= 1) {
      if (j >= 64) {
        const int je = j >> 6;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          if ((e & je) == 0) {
            const bool asc = (((64 * e) & k) == 0);            // k >= 128 here: the lane bits do not reach bit k
            const uint64_t x = v[e], y = v[e | je];
            const uint64_t lo = x < y ? x : y, hi = x < y ? y : x;
            v[e] = asc ? lo : hi;
            v[e | je] = asc ? hi : lo;
          }
        }
      } else {
        const bool upper = (lane & j) != 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const bool asc = (((lane + 64 * e) & k) == 0);
          const uint64_t o = shfl_xor_u64(v[e], j);
          const bool take_min = (asc != upper);
          v[e] = take_min ? (v[e] < o ? v[e] : o) : (v[e] < o ? o : v[e]);
        }
      }
    }
  }
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

template <int E>
__device__ __forceinline__ void wave_sort_list(const uint64_t *__restrict__ key_buf, int64_t lo, int L, int lane, int64_t t,
                                               int n_tiles, int tile_bits, int32_t *flatten_ids, int64_t *isect_ids) {
  uint64_t v[E];
#pragma unroll
  for (int e = 0; e < E; ++e) v[e] = (lane + 64 * e < L) ? key_buf[lo + lane + 64 * e] : ~0ull;
  wave_bitonic_sort<E>(v, lane);
#pragma unroll
  for (int e = 0; e < E; ++e)
    if (lane + 64 * e < L) write_sorted(v[e], lo + lane + 64 * e, t, n_tiles, tile_bits, flatten_ids, isect_ids);
}

// Medium lists (256 < L <= CAP, CAP a multiple of 256) by the whole workgroup (round 4; until then the barrier-per-step
// bitonic network in LDS: 45 barriers for 512 keys, 66 for 2048 -- 188 us per iteration on the 1M-Gaussian / 1440p run, where
// most tiles hold 300-500 keys).  Chunks of 256 keys are sorted by ONE wave each in registers (the short-list path), parked
// in LDS, and after ONE barrier every key finds its final place by itself: its index in its own chunk plus, for every other
// chunk, the number of keys below it (a branch-free 8-step binary search in LDS) -- keys are distinct (the id is in the low
// word), so that is the rank.  512 keys: two register sorts side by side + 8 LDS reads per key; 2048: 8 sorts on 4 waves +
// 56 reads per key.  Same order as any correct sort of the distinct 64-bit keys: bit-identical lists.
template <int THREADS>
__device__ __forceinline__ void sort_mid_chunks(uint64_t *s_keys, const uint64_t *__restrict__ key_buf, int64_t lo, int L, int64_t t,
                                                int n_tiles, int tile_bits, int32_t *flatten_ids, int64_t *isect_ids) {
  constexpr int WAVES = THREADS / 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nch = (L + 255) >> 8;
  for (int c = wave; c < nch; c += WAVES) {
    uint64_t v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = 256 * c + lane + 64 * e;
      v[e] = i < L ? key_buf[lo + i] : ~0ull;          // the last chunk's tail: +inf, sorts behind every key
    }
    wave_bitonic_sort<4>(v, lane);
#pragma unroll
    for (int e = 0; e < 4; ++e) s_keys[256 * c + lane + 64 * e] = v[e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 256 * nch; i += THREADS) {
    const uint64_t key = s_keys[i];
    if (key == ~0ull) continue;                          // padding
    const int ci = i >> 8;
    int rank = i & 255;
    for (int c = 0; c < nch; ++c) {
      if (c == ci) continue;
      const uint64_t *ch = s_keys + 256 * c;
      int pos = 0;
#pragma unroll
      for (int step = 128; step >= 1; step >>= 1) pos += (ch[pos + step - 1] < key) ? step : 0;
      rank += pos;                                        // keys of chunk c below this one (255 at most counted: the 256th
      rank += (pos == 255 && ch[255] < key) ? 1 : 0;     // needs its own look)
    }
    write_sorted(key, lo + rank, t, n_tiles, tile_bits, flatten_ids, isect_ids);
  }
  __syncthreads();
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
      if (threadIdx.x == 0) long_list[atomicAdd(long_count, 1)] = (int32_t)t;
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
      if (lane == 0) long_list[atomicAdd(long_count, 1)] = (int32_t)t;
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
  const int n_long = *long_count;
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
        const int nch = (sn + 255) >> 8;
        for (int c = wave; c < nch; c += THREADS / 64) {
          uint64_t v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = 256 * c + lane + 64 * e;
            v[e] = i < sn ? keys[s0 + i] : ~0ull;          // the last run's tail: +inf, sorts behind every key
          }
          wave_bitonic_sort<4>(v, lane);
#pragma unroll
          for (int e = 0; e < 4; ++e) s_keys[256 * c + lane + 64 * e] = v[e];
        }
        __syncthreads();
        bitonic_merge_runs_shared<THREADS>(s_keys, 256 * nch);
        if (nsec_t == 1) {
          for (int i = tid; i < sn; i += THREADS) write_sorted(s_keys[i], lo + i, t, n_tiles, tile_bits, flatten_ids, isect_ids);
        } else {
          for (int i = tid; i < sn; i += THREADS) keys[s0 + i] = s_keys[i];
        }
        __syncthreads();
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
