// wave_sort.hpp -- the register sort of up to 256 64-bit keys by ONE wave (gfx950, wave64).
//
// Shared by the per-tile sort kernels (isect.hip) and by the forward rasteriser's prologue, which sorts the list of its own tile
// when the lists are short everywhere (rasterize_fwd.hip, so_step_desc.sort_in_rasteriser: one launch and one pass over the keys
// fewer per iteration).  Replaces the cub radix sort of gsplat `isect_tiles`
// (/root/reference/utils/gsplat_utils/gsplat_trainer.py:477); keys are (float32 depth bits << 32 | flatten id), distinct, so
// any correct sort gives the stable radix sort's order.
#pragma once
#include "so_common.hpp"

namespace so {

// Register sort of up to 256 keys by ONE wave: lane l holds elements l, l+64, ... (UINT64_MAX pads the
// tail), the classic xor-partner bitonic network runs on ds_bpermute shuffles -- no LDS array, no
// barrier.  Short lists dominate trained-like scenes (mean ~40 keys per tile on the c2 workload, ~150 at
// 1M Gaussians / 1440p).
// The value lane (l ^ J) holds, without the LDS crossbar (end of round 4; until then __shfl_xor = ds_bpermute: 1 200 of them in
// the long-list kernel, each a round trip through LDS that a wave sorting ONE list has nothing to hide behind).  J = 1, 2, 8:
// one DPP move (quad_perm, row_ror:8 -- a rotation by half a row IS the exchange of its halves); J = 4: two DPP moves with
// bank masks (quads 0, 2 take lane + 4, quads 1, 3 take lane - 4); J = 16, 32: v_permlane16/32_swap of two copies leaves the
// even rows (lower half-wave) of the value in one register and the odd ones in the other, twice each -- `upper` = (l & J) != 0
// picks.  Inline asm for the swaps (so_common.hpp rows_combine: the builtin's struct return is miscompiled; the s_nop's are
// the wait states around a cross-lane read).  EXEC must be all ones: every caller runs whole waves.
template <int J>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t x, bool upper) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (J == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xf, 0xf, true);          // quad_perm:[1,0,3,2]
  else if constexpr (J == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xf, 0xf, true);     // quad_perm:[2,3,0,1]
  else if constexpr (J == 4) {
    const int r = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x104, 0xf, 0x5, false);                       // row_shl:4 into quads 0, 2
    return (uint32_t)__builtin_amdgcn_update_dpp(r, (int)x, 0x114, 0xf, 0xa, false);                         // row_shr:4 into quads 1, 3
  } else if constexpr (J == 8) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x128, 0xf, 0xf, true);   // row_ror:8
  else {
    uint32_t a = x, b = x;
    if constexpr (J == 16) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    else asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return upper ? a : b;
  }
#else
  (void)upper;
  return x;
#endif
}
// m in {1, 2, 4, 8, 16, 32}, a constant wherever the sort networks call it (their loops are fully unrolled)
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
  const int lane = lane_id();
  const bool upper = (lane & m) != 0;
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  uint32_t olo, ohi;
  switch (m) {
    case 1: olo = lane_xor_u32<1>(lo, upper); ohi = lane_xor_u32<1>(hi, upper); break;
    case 2: olo = lane_xor_u32<2>(lo, upper); ohi = lane_xor_u32<2>(hi, upper); break;
    case 4: olo = lane_xor_u32<4>(lo, upper); ohi = lane_xor_u32<4>(hi, upper); break;
    case 8: olo = lane_xor_u32<8>(lo, upper); ohi = lane_xor_u32<8>(hi, upper); break;
#if defined(SO_SORT_PERMLANE)
    case 16: olo = lane_xor_u32<16>(lo, upper); ohi = lane_xor_u32<16>(hi, upper); break;
    default: olo = lane_xor_u32<32>(lo, upper); ohi = lane_xor_u32<32>(hi, upper); break;
#else
    // across rows the crossbar stays: the swap form costs two copies, the swap, its wait states and a select per half --
    // five vector instructions where ds_bpermute is one LDS instruction (measured: dense regime sort 72 -> 75.5 us with it)
    default: olo = (uint32_t)__shfl_xor((int)lo, m, 64); ohi = (uint32_t)__shfl_xor((int)hi, m, 64); break;
#endif
  }
  return ((uint64_t)ohi << 32) | olo;
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

__device__ __forceinline__ void write_sorted(uint64_t key, int64_t pos, int64_t t, int n_tiles, int tile_bits,
                                             int32_t *flatten_ids, int64_t *isect_ids) {
  flatten_ids[pos] = (int32_t)(uint32_t)(key & 0xffffffffull);
  if (isect_ids) {
    const int64_t cam = t / n_tiles, tile = t - cam * n_tiles;
    isect_ids[pos] = (cam << (32 + tile_bits)) | (tile << 32) | (int64_t)(key >> 32);
  }
}

// lds_ids (nullable): the sorted ids also go to this LDS array (the forward rasteriser stages its first batch from there)
template <int E>
__device__ __forceinline__ void wave_sort_list(const uint64_t *__restrict__ key_buf, int64_t lo, int L, int lane, int64_t t,
                                               int n_tiles, int tile_bits, int32_t *flatten_ids, int64_t *isect_ids,
                                               int32_t *lds_ids = nullptr) {
  uint64_t v[E];
#pragma unroll
  for (int e = 0; e < E; ++e) v[e] = (lane + 64 * e < L) ? key_buf[lo + lane + 64 * e] : ~0ull;
  wave_bitonic_sort<E>(v, lane);
#pragma unroll
  for (int e = 0; e < E; ++e)
    if (lane + 64 * e < L) {
      write_sorted(v[e], lo + lane + 64 * e, t, n_tiles, tile_bits, flatten_ids, isect_ids);
      if (lds_ids) lds_ids[lane + 64 * e] = (int32_t)(uint32_t)(v[e] & 0xffffffffull);
    }
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

// A list of any length L > 2048 by 256 threads WITHOUT scratch: every key ranks itself against all others straight from global
// memory (keys are distinct).  O(L^2 / 256) loads per thread: the forward rasteriser's safety net for a tile that outgrew the
// in-kernel sorts between two looks of the host at the list statistics -- correct, slow, and rare (list_policy switches the
// step back to the sort kernels as soon as it sees such a list).
__device__ __forceinline__ void rank_sort_global(const uint64_t *__restrict__ key_buf, int64_t lo, int L, int64_t t, int n_tiles,
                                                 int tile_bits, int32_t *flatten_ids, int64_t *isect_ids) {
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const uint64_t key = key_buf[lo + i];
    int rank = 0;
    for (int j = 0; j < L; ++j) rank += key_buf[lo + j] < key ? 1 : 0;
    write_sorted(key, lo + rank, t, n_tiles, tile_bits, flatten_ids, isect_ids);
  }
}

}  // namespace so
