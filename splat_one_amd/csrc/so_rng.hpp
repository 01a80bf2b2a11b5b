// so_rng.hpp -- counter-based random numbers for the device-side densification (refine.hip).
//
// gsplat's `split` draws `torch.randn(2, n_split, 3)` from a stateful generator on the host side of the step
// ([upstream-memory]).  A device-resident refinement has no host in the loop, so the normals are a pure
// FUNCTION of (seed, step, source Gaussian, child, component): Philox4x32-10 (Salmon et al., "Parallel random numbers:
// as easy as 1, 2, 3", SC'11 -- the generator torch itself uses on GPUs) keyed by the seed, counter = (source id,
// child, step, stream tag), followed by Box-Muller.  Every replica of a data-parallel run therefore draws identical
// samples without communication, and the test-side CPU restatement evaluates the same function in numpy (integers
// bit-exact, the float transform within 1e-6).
#pragma once
#include <cmath>
#include <cstdint>

namespace so {

#if defined(__HIPCC__)
#define SO_RNG_HD __host__ __device__ __forceinline__
#else
#define SO_RNG_HD inline
#endif

struct Philox4 {
  uint32_t x[4];
};

SO_RNG_HD void philox_mulhilo(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo) {
  const uint64_t p = (uint64_t)a * (uint64_t)b;
  hi = (uint32_t)(p >> 32);
  lo = (uint32_t)p;
}

// Philox4x32 with 10 rounds; known answers (Random123 kat_vectors) are checked in tests/test_refine_oracle.py.
SO_RNG_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0, lo0, hi1, lo1;
    philox_mulhilo(0xD2511F53u, c0, hi0, lo0);
    philox_mulhilo(0xCD9E8D57u, c2, hi1, lo1);
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox4{{c0, c1, c2, c3}};
}

constexpr uint32_t kSplitStream = 0x53504C54u;   // "SPLT": the stream of the split samples

// 24-bit uniform in (0, 1]: ((x >> 8) + 0.5) / 2^24  (never 0, so the logarithm below is finite)
SO_RNG_HD float u24(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// three standard normals for child `child` (0 / 1) of source Gaussian `id` at training step `step`
SO_RNG_HD void split_normals(uint64_t seed, uint32_t step, uint32_t id, uint32_t child, float (&z)[3]) {
  const Philox4 r = philox4x32_10(id, child, step, kSplitStream, (uint32_t)seed, (uint32_t)(seed >> 32));
  const float two_pi = 6.283185307179586f;
  const float r0 = sqrtf(-2.f * logf(u24(r.x[0]))), r1 = sqrtf(-2.f * logf(u24(r.x[2])));
  z[0] = r0 * cosf(two_pi * u24(r.x[1]));
  z[1] = r0 * sinf(two_pi * u24(r.x[1]));
  z[2] = r1 * cosf(two_pi * u24(r.x[3]));
}

}  // namespace so
