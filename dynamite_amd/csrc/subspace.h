// Basis-index <-> spin-configuration maps, usable on host and device.
//
// Semantics follow the reference's header-only maps
// (src/dynamite/_backend/bsubspace_impl.h): Full :57-83, Parity :112-143,
// SpinConserve (colex combinadic) :187-245, Explicit (sorted table + binary
// search) :302-347.  Where the reference's CUDA twins deviate from the CPU
// header (S2I_CUDA_Explicit starts its search at `dim`, bcuda_impl.cu:159) we
// follow the CPU header.  Results must be bit-exact.
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

namespace dnm {

// POD view of one subspace; table pointers are host pointers in a host copy
// and device pointers in a device copy.
struct SubView {
  int32_t type;
  int32_t L;
  int32_t space;
  int32_t k;
  int32_t ld;                // L+1
  int64_t dim;
  const int64_t *nchoosek;   // (k+1) x (L+1)
  const int64_t *state_map;
  const int64_t *rmap_indices;
  const int64_t *rmap_states;
  // Explicit: optional bucket table over the sorted rmap_states -- the states with (state >> bucket_shift) == b
  // are rmap_states[bucket[b] .. bucket[b+1]); the search of S2I starts inside one bucket
  const int64_t *bucket;
  int32_t bucket_shift;
  int32_t swz;               // vector layout of Full / Parity (dnm_subspace::vec_swizzle): XOR-swizzle shift
  int32_t sc3;               // vector layout of SpinConserve: a | w << 8 (sc3.h), 0 = reference order
};

// position of element i of a vector in the XOR-swizzled layout (S = 0: index order)
__host__ __device__ __forceinline__ int64_t vec_pos(int64_t i, int S) {
  return S ? (i ^ (((i >> S) & (((int64_t)1 << (S - 4)) - 1)) << 4)) : i;
}

#define DNM_HD __host__ __device__ __forceinline__

DNM_HD int hd_popc(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popcll(v);
#else
  return __builtin_popcountll(v);
#endif
}
DNM_HD int hd_ctz(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __ffsll((long long)v) - 1;
#else
  return __builtin_ctzll(v);
#endif
}
DNM_HD int hd_par(uint64_t v) { return hd_popc(v) & 1; }

template <int TYPE>
struct Sub;

template <>
struct Sub<DNM_FULL> {
  static DNM_HD int64_t dim(const SubView &s) { return (int64_t)1 << s.L; }
  static DNM_HD int64_t i2s(int64_t idx, const SubView &) { return idx; }
  static DNM_HD int64_t s2i(int64_t st, const SubView &) { return st; }
  static DNM_HD bool contains(int64_t, const SubView &) { return true; }     // membership without the index
};

template <>
struct Sub<DNM_PARITY> {
  static DNM_HD int64_t dim(const SubView &s) { return (int64_t)1 << (s.L - 1); }
  static DNM_HD int64_t i2s(int64_t idx, const SubView &s) {
    return (idx << 1) | (int64_t)(hd_par((uint64_t)idx) ^ s.space);
  }
  static DNM_HD int64_t s2i(int64_t st, const SubView &s) {
    return hd_par((uint64_t)st) == s.space ? (st >> 1) : (int64_t)-1;
  }
  static DNM_HD bool contains(int64_t st, const SubView &s) { return hd_par((uint64_t)st) == s.space; }
};

template <>
struct Sub<DNM_SPIN_CONSERVE> {
  static DNM_HD int64_t dim(const SubView &s) { return s.nchoosek[(int64_t)s.k * s.ld + s.L]; }
  // greedy unranking from the top bit down
  static DNM_HD int64_t i2s(int64_t idx, const SubView &s) {
    int64_t st = 0;
    int k = s.k;
    for (int n = s.L; n > 0; --n) {
      int64_t here = (k > n - 1) ? 0 : s.nchoosek[(int64_t)k * s.ld + (n - 1)];
      st <<= 1;
      if (idx >= here) {
        idx -= here;
        --k;
        st |= 1;
      }
    }
    return st;
  }
  // colex rank: j-th set bit (from 1) at position n contributes C(n, j) if j <= n
  static DNM_HD int64_t rank(int64_t st, const SubView &s) {
    uint64_t v = (uint64_t)st;
    int64_t idx = 0;
    int j = 0;
    while (v) {
      int n = hd_ctz(v);
      ++j;
      if (j <= n) idx += s.nchoosek[(int64_t)j * s.ld + n];
      v &= v - 1;
    }
    return idx;
  }
  static DNM_HD int64_t s2i(int64_t st, const SubView &s) {
    if (hd_popc((uint64_t)st) != s.k) return -1;
    return rank(st, s);
  }
  static DNM_HD bool contains(int64_t st, const SubView &s) { return hd_popc((uint64_t)st) == s.k; }
};

template <>
struct Sub<DNM_EXPLICIT> {
  static DNM_HD int64_t dim(const SubView &s) { return s.dim; }
  static DNM_HD int64_t i2s(int64_t idx, const SubView &s) { return s.state_map[idx]; }
  static DNM_HD int64_t s2i(int64_t st, const SubView &s) {
    // binary search in the sorted reverse map (bsubspace_impl.h:306-338), narrowed to the state's bucket first
    int64_t lo = 0, hi = s.dim - 1;
    if (s.bucket) {
      const int64_t b = st >> s.bucket_shift;
      lo = s.bucket[b];
      hi = s.bucket[b + 1] - 1;
    }
    while (lo <= hi) {
      int64_t mid = (lo + hi) / 2;
      int64_t v = s.rmap_states[mid];
      if (v == st) return s.rmap_indices ? s.rmap_indices[mid] : mid;
      if (v < st) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
  }
  static DNM_HD bool contains(int64_t st, const SubView &s) { return s2i(st, s) >= 0; }
};

// run-time dispatch (host side of the C ABI)
static inline int64_t sub_dim(const SubView &s) {
  switch (s.type) {
    case DNM_FULL: return Sub<DNM_FULL>::dim(s);
    case DNM_PARITY: return Sub<DNM_PARITY>::dim(s);
    case DNM_SPIN_CONSERVE: return Sub<DNM_SPIN_CONSERVE>::dim(s);
    case DNM_EXPLICIT: return Sub<DNM_EXPLICIT>::dim(s);
  }
  return -1;
}
static inline int64_t sub_i2s(int64_t i, const SubView &s) {
  switch (s.type) {
    case DNM_FULL: return Sub<DNM_FULL>::i2s(i, s);
    case DNM_PARITY: return Sub<DNM_PARITY>::i2s(i, s);
    case DNM_SPIN_CONSERVE: return Sub<DNM_SPIN_CONSERVE>::i2s(i, s);
    case DNM_EXPLICIT: return Sub<DNM_EXPLICIT>::i2s(i, s);
  }
  return -1;
}
static inline int64_t sub_s2i(int64_t st, const SubView &s) {
  switch (s.type) {
    case DNM_FULL: return Sub<DNM_FULL>::s2i(st, s);
    case DNM_PARITY: return Sub<DNM_PARITY>::s2i(st, s);
    case DNM_SPIN_CONSERVE: return Sub<DNM_SPIN_CONSERVE>::s2i(st, s);
    case DNM_EXPLICIT: return Sub<DNM_EXPLICIT>::s2i(st, s);
  }
  return -1;
}

}  // namespace dnm
