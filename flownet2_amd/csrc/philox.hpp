// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123 reference constants),
// a COUNTER-BASED generator: output = f(key, counter), no state to carry, any element of any launch can be produced independently --
// which is what a device kernel needs, and what makes a run reproducible from (seed, iteration, sample, pixel) alone.
// The reference draws the noise effect from cuRAND's XORWOW stream (caffe_gpu_rng_gaussian, src/caffe/util/math_functions.cu:405-409) and
// its coefficients from boost::mt19937 (src/caffe/common.cpp / util/rng.hpp): neither stream can be reproduced; the DISTRIBUTIONS are pinned
// (tests/test_augmentation_random.py).  Shared by csrc/data_augmentation.hip and restated in oracle/fn2_oracle.c.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define FN2_HD __host__ __device__ __forceinline__
#else
#define FN2_HD inline
#endif

namespace fn2 {

struct Philox4 { uint32_t v[4]; };

FN2_HD void philox_mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
  const uint64_t p = (uint64_t)a * (uint64_t)b;
  hi = (uint32_t)(p >> 32); lo = (uint32_t)p;
}

FN2_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0, lo0, hi1, lo1;
    philox_mulhilo(M0, c0, hi0, lo0);
    philox_mulhilo(M1, c2, hi1, lo1);
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  return Philox4{{c0, c1, c2, c3}};
}

// (x + 1) * 2^-32 in (0, 1]: never 0, so the logarithm of the Box-Muller transform is finite
FN2_HD float philox_unit(uint32_t x) { return ((float)x + 1.0f) * 2.3283064365386963e-10f; }

}  // namespace fn2
