// Philox4x32-10 counter-based generator, usable from host C++ and HIP device code.
// Streams of the synthetic-population generator are addressed by counter, so the
// device kernel and its host twin produce identical bits for any sub-block of the population.
#ifndef KGX_PHILOX_H
#define KGX_PHILOX_H

#include <stdint.h>

#if defined(__HIPCC__)
#define KGX_HD __host__ __device__ __forceinline__
#else
#define KGX_HD inline
#endif

struct kgx_u32x4 { uint32_t v[4]; };

KGX_HD void kgx_mulhilo32(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
  const uint64_t p = static_cast<uint64_t>(a) * static_cast<uint64_t>(b);
  hi = static_cast<uint32_t>(p >> 32);
  lo = static_cast<uint32_t>(p);
}

KGX_HD kgx_u32x4 kgx_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                   uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int round = 0; round < 10; ++round) {
    uint32_t hi0, lo0, hi1, lo1;
    kgx_mulhilo32(M0, c0, hi0, lo0);
    kgx_mulhilo32(M1, c2, hi1, lo1);
    const uint32_t n0 = hi1 ^ c1 ^ k0;
    const uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  kgx_u32x4 r;
  r.v[0] = c0; r.v[1] = c1; r.v[2] = c2; r.v[3] = c3;
  return r;
}

// Uniform double in (0,1) from 32 random bits: (r + 0.5) * 2^-32, exact in fp64.
KGX_HD double kgx_u01(uint32_t r) {
  return (static_cast<double>(r) + 0.5) * (1.0 / 4294967296.0);
}

#endif  // KGX_PHILOX_H
