// Philox-4x32-10 counter-based generator: one complex normal deviate per 64-bit counter (shared by the vector
// kernels and the layout-aware fills, which key it by the element's reference index).
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

namespace dnm {

__device__ __forceinline__ void philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3,
                                             uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
  uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
  uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

// N(0,1) + i N(0,1) for counter `ctr` under `seed`
__device__ __forceinline__ double2 philox_normal(uint64_t ctr, uint64_t seed) {
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x243F6A88u, c3 = 0x85A308D3u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  // two uniforms in (0,1] and [0,1) from 53 bits each -> Box-Muller pair
  double u1 = ((double)((((uint64_t)c0 << 32) | c1) >> 11) + 1.0) * (1.0 / 9007199254740992.0);
  double u2 = (double)((((uint64_t)c2 << 32) | c3) >> 11) * (1.0 / 9007199254740992.0);
  double rad = sqrt(-2.0 * log(u1));
  double s, c;
  sincospi(2.0 * u2, &s, &c);
  return make_double2(rad * c, rad * s);
}

}  // namespace dnm
