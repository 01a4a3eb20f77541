// Device-side helpers shared by the SpinConserve internal-layout kernels (sc3_kernels.hip: chain operators and the
// layout's vector utilities; sc3g_kernels.hip: operators on any bond graph).
#pragma once

#include "sc3.h"

namespace dnm {

typedef double2 c128;
typedef double d2v __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ int64_t rl_i64(int64_t v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(v & 0xffffffff), l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l);
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double rl_f64(double v, int l) { return __longlong_as_double(rl_i64(__double_as_longlong(v), l)); }
__device__ __forceinline__ int rl_i32(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ void store_nt(c128 *p, double re, double im) {
  d2v v = {re, im};
  __builtin_nontemporal_store(v, reinterpret_cast<d2v *>(p));
}
__device__ __forceinline__ c128 load_nt(const c128 *p) {
  d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(p));
  return make_double2(v.x, v.y);
}
__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double flip(double c, uint32_t parity_bit) {
  int hi = __double2hiint(c) ^ (int)(parity_bit << 31);
  return __hiloint2double(hi, __double2loint(c));
}

// wave priority by phase (round 4, as in tile_pass_kernel): raised while a workgroup is in a memory phase, lowered for
// its LDS bond loops, so that of the two workgroups of a CU the one asking for memory wins the issue slots
// (SpinConserve(32,16): window pass 5.38 -> 5.25 ms, lo pass level; a rank of config 5 25.7 -> 24.8 ms;
// profiles/r04_sc3_spans.txt; -DDNM_SC3_PRIO=0 builds without)
// the layout's and the operator's tables are read-only: through the constant address space their wave-uniform reads
// stay scalar loads whatever else the kernel contains (see CQuad in matvec_kernels.hip)
#define SC3_CP(T, p) ((const __attribute__((address_space(4))) T *)(p))
#ifndef DNM_SC3_PRIO
#define DNM_SC3_PRIO 1
#endif
#if DNM_SC3_PRIO == 1
#define SC3_PRIO_MEM() asm volatile("s_setprio 3")
#define SC3_PRIO_LDS() asm volatile("s_setprio 0")
#else
#define SC3_PRIO_MEM()
#define SC3_PRIO_LDS()
#endif

constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v >> 1); }
constexpr uint32_t SC3_NOROW = 1u << 29;       // lo pass: a sub-group slot without a row

constexpr int cbinom(int n, int k) {
  long long r = 1;
  for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return (int)r;
}

// row (T, W) of a workgroup: everything the kernels derive from the perm entry
struct RowId {
  uint32_t T, W;
  int cw, kr, kl, nrows, pitch;
  int64_t tb, base;        // internal offset of the T block / of the row
};
__device__ __forceinline__ RowId decode_row(uint32_t e, const Sc3Tab &S) {
  RowId r;
  r.T = e >> S.w;
  r.W = e & ((1u << S.w) - 1u);
  r.cw = __popc(r.W);
  r.kr = S.k - __popc(r.T);
  r.kl = r.kr - r.cw;
  r.nrows = S.nl[r.kl];
  r.pitch = S.pitch[r.kl];
  r.tb = SC3_CP(int64_t, S.ibase)[r.T];
  r.base = r.tb + SC3_CP(int64_t, S.icoff)[r.kr * (S.w + 1) + r.cw] + (int64_t)SC3_CP(uint16_t, S.w_rank)[r.W] * r.pitch;
  return r;
}

// entries of the lo pass's LDS tile: what the workgroup's threads hold (RPT entries each), at least the longest row
constexpr int sc3_lo_cap(int a, int nt) { return ((cbinom(a, a / 2) + nt - 1) / nt) * nt; }

// waves per SIMD a tiled pass can have: what its LDS tile lets be resident (two workgroups per CU for the 64 KB
// tiles), at most 8 -- the register budget follows from it (128 registers at 512 threads, 64 at 1024)
constexpr int sc3_win_waves(int nt, int tile_kb) {
  int wgs = 160 / (tile_kb > 0 ? tile_kb : 1);
  if (wgs > 2048 / nt) wgs = 2048 / nt;
  if (wgs < 1) wgs = 1;
  const int w = wgs * nt / 256;
  return w > 8 ? 8 : (w < 1 ? 1 : w);
}

// shape of the real lo pass: NTR threads with PPT pairs each (DNM_SC3R_SHAPE 0: as many threads as the complex pass and
// twice its entries per thread -- 1024 x 8 entries, a 64 KB tile, two workgroups per CU; 1: half the threads, 512 x 8, a
// 32 KB tile, four workgroups per CU; 2: 1024 x 4, 32 KB, two per CU)
#ifndef DNM_SC3R_SHAPE
#define DNM_SC3R_SHAPE 0
#endif
constexpr int sc3r_threads(int nt) { return (DNM_SC3R_SHAPE == 1 && nt >= 512) ? nt / 2 : nt; }
constexpr int sc3r_pairs(int a, int nt) {
  return (DNM_SC3R_SHAPE == 2 && nt >= 512) ? (cbinom(a, a / 2) / 2 + nt - 1) / nt : (cbinom(a, a / 2) + nt - 1) / nt;
}
constexpr int sc3_lo_cap_r(int a, int nt) { return 2 * sc3r_pairs(a, nt) * sc3r_threads(nt); }

}  // namespace
}  // namespace dnm
