// tile_pass2_kernel -- second generation of the tiled hypercube pass (Full / Parity index spaces), gfx950.
//
// Same tables as tile_pass_kernel (plan.h: DevPass + DevQuad records, built by mat.cpp:build_pass), same
// arithmetic per record, different execution:
//   * addresses are  base(SGPR pair) + 32-bit VGPR byte offset.  The element index of row k of a thread is
//     phys(row) = pt ^ pk[k]  with pt the thread's part (lane / wave bits of the tile coordinate, constant for
//     the whole kernel) and pk[k] wave-uniform; the partner of a gathered mask is  phys(row ^ mask) =
//     phys(row) ^ phys(mask)  -- one v_xor per load, everything else on the scalar unit;
//   * phys() is the XOR-swizzled vector layout (DevPass::swz_shift = S): index bits [S, 2S-4) are folded onto
//     bits [4, S), so the far-apart 256 B runs of a window tile and of its XCD-group siblings spread over the L2
//     sets instead of aliasing; S = 0 is the natural layout;
//   * gathered records whose coefficient does not depend on the lane (chain bonds above the tile: signs on block
//     and k bits only) are evaluated per row on the scalar unit: rows with a vanishing matrix element are neither
//     loaded nor multiplied (the XX+YY bond across the tile boundary is dead on half of the rows), the others
//     take the coefficient as a scalar operand;
//   * gathered records are software-pipelined: the loads of the next live record are in flight while the
//     current one is multiplied.
// Reference semantics replaced: MatMult_GPU / device_MatMult (src/dynamite/_backend/bcuda_template_2.cu:141-273),
// MatMult_CPU_Fast (bpetsc_template_2.c:713-889).
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "kernels.h"

namespace dnm {

typedef double2 c128;
typedef double d2v __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ double flip_sign(double c, uint32_t parity_bit) {
  int hi = __double2hiint(c) ^ (int)(parity_bit << 31);
  return __hiloint2double(hi, __double2loint(c));
}
__device__ __forceinline__ double flip_sign_bits(double c, uint32_t signbit_in_place) {
  int hi = __double2hiint(c) ^ (int)signbit_in_place;
  return __hiloint2double(hi, __double2loint(c));
}

template <int MAXS>
__device__ __forceinline__ uint32_t deposit(uint32_t v, int nseg, const int32_t *off, const int32_t *len,
                                            const int32_t *pos) {
  uint32_t r = 0;
#pragma unroll
  for (int j = 0; j < MAXS; ++j)
    if (j < nseg) r |= ((v >> off[j]) & ((1u << len[j]) - 1u)) << pos[j];
  return r;
}

// wave-uniform 64-bit pattern of a double held in a VGPR
__device__ __forceinline__ uint64_t uniform_bits(double v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)__double2loint(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)__double2hiint(v));
  return ((uint64_t)hi << 32) | lo;
}

// The swizzled layout: element `i` of a vector lives at i ^ sw(i).
struct Swz {
  uint32_t sh, msk;   // sw(v) = ((v >> sh) & msk) << 4
  __device__ __forceinline__ uint32_t sw(uint32_t v) const { return ((v >> sh) & msk) << 4; }
  __device__ __forceinline__ uint32_t phys(uint32_t v) const { return v ^ sw(v); }
};

// Address of the element whose index is (thread part) ^ (scalar part).  The thread part has no bit at or above
// bit 28 (checked on the host), so the bits above go to the scalar base and the rest is a 32-bit byte offset.
__device__ __forceinline__ const d2v *elem(const c128 *vec, uint32_t pt4, uint32_t ps) {
  const char *b = reinterpret_cast<const char *>(vec) + ((uint64_t)(ps >> 28) << 32);
  return reinterpret_cast<const d2v *>(b + (pt4 ^ ((ps & 0x0FFFFFFFu) << 4)));
}
__device__ __forceinline__ d2v *elem(c128 *vec, uint32_t pt4, uint32_t ps) {
  char *b = reinterpret_cast<char *>(vec) + ((uint64_t)(ps >> 28) << 32);
  return reinterpret_cast<d2v *>(b + (pt4 ^ ((ps & 0x0FFFFFFFu) << 4)));
}

// signed amplitude of slot j of a record for this thread (thread-constant part of the row)
__device__ __forceinline__ double slot_amp(const DevQuad &q, int j, uint32_t tt, uint64_t sbase) {
  uint32_t p = (uint32_t)(__popc(tt & q.sign_tile[j]) + __popcll(sbase & q.sign_ext[j])) & 1u;
  return flip_sign(q.coeff[j], p);
}

// ---- LDS records (as tile_pass_kernel; sign-variant rows pick one of two sums) ----------------------------
template <int R, int LOGNT, bool KVAR, bool CPLX, bool K0>
__device__ __forceinline__ void lds_records(const DevQuad *__restrict__ quads, uint32_t b, uint32_t e,
                                            double (&ar)[R], double (&ai)[R], const c128 *tile, uint32_t tid,
                                            uint64_t sbase) {
  constexpr uint32_t NT = 1u << LOGNT;
  for (uint32_t qi = b; qi < e; ++qi) {
    const DevQuad &Q = quads[qi];
    const double a0 = slot_amp(Q, 0, tid, sbase);
    const double a1 = slot_amp(Q, 1, tid, sbase);
    double a2 = 0.0, a3 = 0.0;
    if constexpr (CPLX) {
      a2 = slot_amp(Q, 2, tid, sbase);
      a3 = slot_amp(Q, 3, tid, sbase);
    }
    c128 xv[R];
    if constexpr (K0) {
      const c128 *p = tile + (tid ^ Q.mask_tile);
#pragma unroll
      for (int k = 0; k < R; ++k) xv[k] = p[k * NT];
    } else {
      const uint32_t mt = Q.mask_tile;
      const uint32_t p_lo = tid ^ (mt & (NT - 1u));
      const uint32_t mk = mt >> LOGNT;
#pragma unroll
      for (int k = 0; k < R; ++k) xv[k] = tile[p_lo + (((uint32_t)k ^ mk) << LOGNT)];
    }
    if constexpr (!KVAR) {
      const double cre = a0 + a1;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        ar[k] = fma(cre, xv[k].x, ar[k]);
        ai[k] = fma(cre, xv[k].y, ai[k]);
      }
      if constexpr (CPLX) {
        const double cim = a2 + a3;
#pragma unroll
        for (int k = 0; k < R; ++k) {
          ar[k] = fma(-cim, xv[k].y, ar[k]);
          ai[k] = fma(cim, xv[k].x, ai[k]);
        }
      }
    } else {
      // row k: c = s0(k) a0 + s1(k) a1 = s0(k) (a0 + s0 s1 a1): one of two sums, then one sign
      const uint32_t s0 = Q.sign_tile[0] >> LOGNT, s1 = Q.sign_tile[1] >> LOGNT;
      const double sp = a0 + a1, sm = a0 - a1;
      double ip = 0.0, im = 0.0;
      uint32_t s2 = 0, s3 = 0;
      if constexpr (CPLX) {
        s2 = Q.sign_tile[2] >> LOGNT;
        s3 = Q.sign_tile[3] >> LOGNT;
        ip = a2 + a3;
        im = a2 - a3;
      }
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const uint32_t f0 = (uint32_t)__popc(k & s0) & 1u, f1 = (uint32_t)__popc(k & s1) & 1u;
        const double cre = flip_sign((f0 ^ f1) ? sm : sp, f0);
        ar[k] = fma(cre, xv[k].x, ar[k]);
        ai[k] = fma(cre, xv[k].y, ai[k]);
        if constexpr (CPLX) {
          const uint32_t f2 = (uint32_t)__popc(k & s2) & 1u, f3 = (uint32_t)__popc(k & s3) & 1u;
          const double cim = flip_sign((f2 ^ f3) ? im : ip, f2);
          ar[k] = fma(-cim, xv[k].y, ar[k]);
          ai[k] = fma(cim, xv[k].x, ai[k]);
        }
      }
    }
  }
}

// ---- gathered records ----------------------------------------------------------------------------------
// What a real record needs between the issue of its loads and its multiply.
template <int R>
struct GatherState {
  d2v xv[R];
  uint32_t live;        // rows whose loads were issued (bit k)
  uint32_t kind;        // 1: wave-uniform coefficients (scalar unit), 2: per-thread coefficients
  uint64_t sp_bits, sm_bits;   // kind 1: c0 + c1, c0 - c1 as bit patterns (wave-uniform)
  double sp, sm;               // kind 2: a0 + a1, a0 - a1 of this thread
  uint32_t s0, s1, p0, p1;     // per-row sign selectors (k bits of the sign masks), block parities (kind 1)
};

__device__ __forceinline__ bool record_has_imag(const DevQuad &Q) {
  return ((((uint64_t)__double_as_longlong(Q.coeff[2]) | (uint64_t)__double_as_longlong(Q.coeff[3])) << 1) != 0);
}

struct GatherCtx {
  const c128 *__restrict__ x;
  const c128 *__restrict__ xr;
  uint32_t xr_xor;
  Swz z;
  uint32_t pt4, tid;
  uint64_t sbase;
};

// Issue the loads of the real record Q for the rows that need them.  Returns false when the whole wavefront has
// nothing to do for this record (nothing issued).
template <int R, int LOGNT>
__device__ __forceinline__ bool gather_issue(const DevQuad &Q, GatherState<R> &g, const GatherCtx &c,
                                             const uint32_t (&pk)[R]) {
  constexpr uint32_t NT = 1u << LOGNT;
  const uint32_t lane_signs = (Q.sign_tile[0] | Q.sign_tile[1]) & (NT - 1u);
  const c128 *__restrict__ src = Q.src ? c.xr : c.x;
  const uint32_t xm = c.z.phys(Q.mask_loc) ^ (Q.src ? c.xr_xor : 0u);   // phys(row ^ mask) = phys(row) ^ phys(mask)
  g.s0 = Q.sign_tile[0] >> LOGNT;
  g.s1 = Q.sign_tile[1] >> LOGNT;
  if (lane_signs == 0) {
    // coefficient independent of the lane: per-row scalars
    g.kind = 1;
    g.p0 = (uint32_t)__popcll(c.sbase & Q.sign_ext[0]) & 1u;
    g.p1 = (uint32_t)__popcll(c.sbase & Q.sign_ext[1]) & 1u;
    g.sp_bits = uniform_bits(Q.coeff[0] + Q.coeff[1]);
    g.sm_bits = uniform_bits(Q.coeff[0] - Q.coeff[1]);
    const bool zp = (g.sp_bits << 1) == 0, zm = (g.sm_bits << 1) == 0;
    uint32_t live = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const uint32_t f0 = (g.p0 ^ (uint32_t)__popc(k & g.s0)) & 1u, f1 = (g.p1 ^ (uint32_t)__popc(k & g.s1)) & 1u;
      if (!((f0 ^ f1) ? zm : zp)) live |= 1u << k;
    }
    live = __builtin_amdgcn_readfirstlane(live);
    g.live = live;
    if (live == 0) return false;
#pragma unroll
    for (int k = 0; k < R; ++k)
      if ((live >> k) & 1u) g.xv[k] = *elem(src, c.pt4, pk[k] ^ xm);
    return true;
  }
  // per-thread coefficients
  g.kind = 2;
  const double a0 = slot_amp(Q, 0, c.tid, c.sbase), a1 = slot_amp(Q, 1, c.tid, c.sbase);
  g.sp = a0 + a1;
  g.sm = a0 - a1;
  const bool kvar = (g.s0 | g.s1) != 0;
  const bool live = kvar ? (g.sp != 0.0 || g.sm != 0.0) : (g.sp != 0.0);
  if (!__any(live)) return false;
  g.live = (1u << R) - 1u;
  if (live) {
#pragma unroll
    for (int k = 0; k < R; ++k) g.xv[k] = *elem(src, c.pt4, pk[k] ^ xm);
  } else {
#pragma unroll
    for (int k = 0; k < R; ++k) g.xv[k] = d2v{0.0, 0.0};
  }
  return true;
}

template <int R>
__device__ __forceinline__ void gather_apply(const GatherState<R> &g, double (&ar)[R], double (&ai)[R]) {
  if (g.kind == 1) {
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if ((g.live >> k) & 1u) {
        const uint32_t f0 = (g.p0 ^ (uint32_t)__popc(k & g.s0)) & 1u, f1 = (g.p1 ^ (uint32_t)__popc(k & g.s1)) & 1u;
        const uint64_t cb = ((f0 ^ f1) ? g.sm_bits : g.sp_bits) ^ ((uint64_t)f0 << 63);
        const double cre = __longlong_as_double((long long)cb);
        ar[k] = fma(cre, g.xv[k].x, ar[k]);
        ai[k] = fma(cre, g.xv[k].y, ai[k]);
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const uint32_t f0 = (uint32_t)__popc(k & g.s0) & 1u, f1 = (uint32_t)__popc(k & g.s1) & 1u;
      const double cre = flip_sign((f0 ^ f1) ? g.sm : g.sp, f0);
      ar[k] = fma(cre, g.xv[k].x, ar[k]);
      ai[k] = fma(cre, g.xv[k].y, ai[k]);
    }
  }
}

// Records with an imaginary part (slots 2, 3): issue, wait, multiply -- not pipelined.
template <int R, int LOGNT>
__device__ __forceinline__ void gather_complex(const DevQuad &Q, const GatherCtx &c, const uint32_t (&pk)[R],
                                               double (&ar)[R], double (&ai)[R]) {
  const c128 *__restrict__ src = Q.src ? c.xr : c.x;
  const uint32_t xm = c.z.phys(Q.mask_loc) ^ (Q.src ? c.xr_xor : 0u);
  const double a0 = slot_amp(Q, 0, c.tid, c.sbase), a1 = slot_amp(Q, 1, c.tid, c.sbase);
  const double a2 = slot_amp(Q, 2, c.tid, c.sbase), a3 = slot_amp(Q, 3, c.tid, c.sbase);
  const uint32_t s0 = Q.sign_tile[0] >> LOGNT, s1 = Q.sign_tile[1] >> LOGNT;
  const uint32_t s2 = Q.sign_tile[2] >> LOGNT, s3 = Q.sign_tile[3] >> LOGNT;
  const bool kvar = (s0 | s1 | s2 | s3) != 0;
  const bool live = kvar || (a0 + a1 != 0.0) || (a2 + a3 != 0.0);
  if (!__any(live)) return;
  d2v xv[R];
  if (live) {
#pragma unroll
    for (int k = 0; k < R; ++k) xv[k] = *elem(src, c.pt4, pk[k] ^ xm);
  } else {
#pragma unroll
    for (int k = 0; k < R; ++k) xv[k] = d2v{0.0, 0.0};
  }
  const double sp = a0 + a1, sm = a0 - a1, ip = a2 + a3, im = a2 - a3;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const uint32_t f0 = (uint32_t)__popc(k & s0) & 1u, f1 = (uint32_t)__popc(k & s1) & 1u;
    const uint32_t f2 = (uint32_t)__popc(k & s2) & 1u, f3 = (uint32_t)__popc(k & s3) & 1u;
    const double cre = flip_sign((f0 ^ f1) ? sm : sp, f0);
    const double cim = flip_sign((f2 ^ f3) ? im : ip, f2);
    ar[k] = fma(cre, xv[k].x, ar[k]);
    ai[k] = fma(cre, xv[k].y, ai[k]);
    ar[k] = fma(-cim, xv[k].y, ar[k]);
    ai[k] = fma(cim, xv[k].x, ai[k]);
  }
}

// Next real record with work at or after qi; its loads are issued into g.
template <int R, int LOGNT>
__device__ __forceinline__ bool gather_next(const DevQuad *__restrict__ quads, uint32_t &qi, uint32_t ge,
                                            GatherState<R> &g, const GatherCtx &c, const uint32_t (&pk)[R]) {
  // every quantity steering this loop is wave-uniform; readfirstlane states it where the compiler cannot see it
  while (qi < ge) {
    const uint32_t q = __builtin_amdgcn_readfirstlane(qi);
    qi = q + 1;
    if (__builtin_amdgcn_readfirstlane((uint32_t)gather_issue<R, LOGNT>(quads[q], g, c, pk))) return true;
  }
  return false;
}

constexpr int waves_per_simd2(int B, int LOGR) {
  int nt = 1 << (B - LOGR);
  int blocks = (160 * 1024) / (16 << B);
  if (blocks < 1) blocks = 1;
  int w = blocks * nt / 256;
  return w < 1 ? 1 : (w > 4 ? 4 : w);
}

}  // namespace

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// GF: gathered records in flight (1: issue, wait, multiply; 2: the next live record's loads are issued first)
// DMA: the tile goes straight from global memory to LDS (global_load_lds, no staging registers)
template <int B, int LOGR, int GF, bool DMA>
__global__ void __launch_bounds__(1 << (B - LOGR), waves_per_simd2(B, LOGR))
tile_pass2_kernel(const DevPass P, const c128 *__restrict__ x, c128 *__restrict__ y,
                  const c128 *__restrict__ xr) {
  constexpr int R = 1 << LOGR;
  constexpr int LOGNT = B - LOGR;
  constexpr uint32_t NT = 1u << LOGNT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  c128 *tile = reinterpret_cast<c128 *>(smem);

  const uint32_t tid = threadIdx.x;
  const uint32_t base = deposit<MAXBSEG>(blockIdx.x, P.nbseg, P.bseg_off, P.bseg_len, P.bseg_pos);
  const uint64_t sbase = P.sign_base | (uint64_t)base;
  Swz z;
  z.sh = P.swz_shift ? (uint32_t)P.swz_shift : 31u;
  z.msk = P.swz_shift ? ((1u << (P.swz_shift - 4)) - 1u) : 0u;

  // element index of row k of this thread: pt ^ pk[k]
  const uint32_t pt4 = z.phys(deposit<MAXSEG>(tid, P.nseg, P.seg_off, P.seg_len, P.seg_pos)) << 4;
  uint32_t pk[R];
#pragma unroll
  for (int k = 0; k < R; ++k)
    pk[k] = z.phys(base | deposit<MAXSEG>((uint32_t)k << LOGNT, P.nseg, P.seg_off, P.seg_len, P.seg_pos));
  const uint32_t yx = P.swz_xor_y;      // sub-block passes: the block's own offset enters the swizzle

  // ---- accumulator start values (streamed: read once)
  double ar[R], ai[R];
  if (P.accumulate) {
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const d2v v = __builtin_nontemporal_load(elem((const c128 *)y, pt4, pk[k] ^ yx));
      ar[k] = v.x;
      ai[k] = v.y;
    }
  } else if (P.zinit) {
    const c128 *__restrict__ zv = (const c128 *)P.zinit;
    const double zs = -P.zscale;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const d2v v = __builtin_nontemporal_load(elem(zv, pt4, pk[k] ^ yx));
      ar[k] = zs * v.x;
      ai[k] = zs * v.y;
    }
    if (P.zinit2) {
      const c128 *__restrict__ z2 = (const c128 *)P.zinit2;
      const double cr = P.z2re, ci = P.z2im;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const d2v v = __builtin_nontemporal_load(elem(z2, pt4, pk[k] ^ yx));
        ar[k] = fma(cr, v.x, ar[k]);
        ar[k] = fma(-ci, v.y, ar[k]);
        ai[k] = fma(cr, v.y, ai[k]);
        ai[k] = fma(ci, v.x, ai[k]);
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < R; ++k) ar[k] = ai[k] = 0.0;
  }

  const DevQuad *__restrict__ quads = P.quads;
  GatherCtx gc;
  gc.x = x;
  gc.xr = xr;
  gc.xr_xor = P.swz_xor_src;
  gc.z = z;
  gc.pt4 = pt4;
  gc.tid = tid;
  gc.sbase = sbase;
  uint32_t gq = P.loop[LP_GATHER_REAL];
  const uint32_t ge = P.loop[LP_GATHER_KVAR_REAL + 1];

  // ---- the tile's loads, the first gathered records' loads right behind them (sibling workgroups of an XCD
  // group ask for the same lines within the same microsecond: the L2 merges the requests)
  GatherState<R> ga, gb;
  bool have_a;
  if constexpr (DMA) {
    if (P.need_tile) {
#pragma unroll
      for (int k = 0; k < R; ++k)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)elem(x, pt4, pk[k]),
                                         (LDS_AS void *)(tile + (k * NT + (tid & ~63u))), 16, 0, 0);
    }
    have_a = gather_next<R, LOGNT>(quads, gq, ge, ga, gc, pk);
  } else {
    if (P.need_tile) {
      d2v v[R];
#pragma unroll
      for (int k = 0; k < R; ++k) v[k] = *elem(x, pt4, pk[k]);
      have_a = gather_next<R, LOGNT>(quads, gq, ge, ga, gc, pk);
#pragma unroll
      for (int k = 0; k < R; ++k) *reinterpret_cast<d2v *>(tile + (tid + k * NT)) = v[k];
    } else {
      have_a = gather_next<R, LOGNT>(quads, gq, ge, ga, gc, pk);
    }
  }
  if constexpr (GF == 1) {
    while (have_a) {
      gather_apply<R>(ga, ar, ai);
      have_a = gather_next<R, LOGNT>(quads, gq, ge, ga, gc, pk);
    }
  } else {
    while (have_a) {
      const bool have_b = gather_next<R, LOGNT>(quads, gq, ge, gb, gc, pk);
      gather_apply<R>(ga, ar, ai);
      have_a = false;
      if (have_b) {
        have_a = gather_next<R, LOGNT>(quads, gq, ge, ga, gc, pk);
        gather_apply<R>(gb, ar, ai);
      }
    }
  }
  // records with an imaginary part: one at a time
  for (uint32_t qi = P.loop[LP_GATHER_CPLX]; qi < P.loop[LP_GATHER_KVAR_CPLX + 1]; ++qi)
    gather_complex<R, LOGNT>(quads[qi], gc, pk, ar, ai);

  // ---- diagonal, part 1 (under the tile loads): terms whose sign mask lies outside the tile
  double dext = 0.0;
  if (P.has_diag) {
    const uint32_t lane = tid & 63u;
    const uint32_t nterm = (P.dext_end - P.dext_begin) * 4u;
    for (uint32_t t0 = 0; t0 < nterm; t0 += 64u) {
      const uint32_t t = t0 + lane;
      double v = 0.0;
      if (t < nterm) {
        const DevQuad &Q = quads[P.dext_begin + (t >> 2)];
        const uint32_t j = t & 3u;
        const uint32_t p = (uint32_t)__popcll(sbase & Q.sign_ext[j]) & 1u;
        v = flip_sign(Q.coeff[j], p);
      }
      dext += v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dext += __shfl_xor(dext, off, 64);
  }
  if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the tile have landed
  __syncthreads();

  // ---- diagonal, part 2: terms that see the tile, bucketed by their k-bit pattern, Walsh-Hadamard over k
  if (P.has_diag) {
    double D[R];
#pragma unroll
    for (int j = 0; j < R; ++j) D[j] = 0.0;
    D[0] = dext;
#pragma unroll
    for (int j = 0; j < R; ++j)
      for (uint32_t q = P.dbucket[j]; q < P.dbucket[j + 1]; ++q) {
        const uint32_t ns = quads[q].nslots;
        double v = slot_amp(quads[q], 0, tid, sbase);
        if (ns > 1) v += slot_amp(quads[q], 1, tid, sbase);
        if (ns > 2) v += slot_amp(quads[q], 2, tid, sbase);
        if (ns > 3) v += slot_amp(quads[q], 3, tid, sbase);
        D[j] += v;
      }
#pragma unroll
    for (int h = 1; h < R; h <<= 1) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        if ((i & h) == 0) {
          double a = D[i], b = D[i | h];
          D[i] = a + b;
          D[i | h] = a - b;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      c128 xs = tile[tid + k * NT];
      ar[k] = fma(D[k], xs.x, ar[k]);
      ai[k] = fma(D[k], xs.y, ai[k]);
    }
  }

  // ---- masks inside the tile
  lds_records<R, LOGNT, false, false, true>(quads, P.loop[LP_TILE_REAL_K0], P.loop[LP_TILE_REAL_K0 + 1], ar, ai, tile, tid, sbase);
  lds_records<R, LOGNT, false, false, false>(quads, P.loop[LP_TILE_REAL], P.loop[LP_TILE_REAL + 1], ar, ai, tile, tid, sbase);
  lds_records<R, LOGNT, false, true, false>(quads, P.loop[LP_TILE_CPLX], P.loop[LP_TILE_CPLX + 1], ar, ai, tile, tid, sbase);
  lds_records<R, LOGNT, true, false, false>(quads, P.loop[LP_TILE_KVAR_REAL], P.loop[LP_TILE_KVAR_REAL + 1], ar, ai, tile, tid, sbase);
  lds_records<R, LOGNT, true, true, false>(quads, P.loop[LP_TILE_KVAR_CPLX], P.loop[LP_TILE_KVAR_CPLX + 1], ar, ai, tile, tid, sbase);

#pragma unroll
  for (int k = 0; k < R; ++k) __builtin_nontemporal_store(d2v{ar[k], ai[k]}, elem(y, pt4, pk[k] ^ yx));

  // ---- fused <x, y> (Lanczos alpha) and |y|^2: the rows' own x values are still in the tile
  if (P.dot_out) {
    double dr = 0.0, di = 0.0, dn = 0.0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const c128 xs = tile[tid + k * NT];
      dr = fma(xs.x, ar[k], dr);
      dr = fma(xs.y, ai[k], dr);
      di = fma(xs.x, ai[k], di);
      di = fma(-xs.y, ar[k], di);
      dn = fma(ar[k], ar[k], dn);
      dn = fma(ai[k], ai[k], dn);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      dr += __shfl_xor(dr, off, 64);
      di += __shfl_xor(di, off, 64);
      dn += __shfl_xor(dn, off, 64);
    }
    __shared__ double dred[3 * (NT / 64 > 0 ? NT / 64 : 1)];
    if ((tid & 63u) == 0) {
      dred[3 * (tid >> 6)] = dr;
      dred[3 * (tid >> 6) + 1] = di;
      dred[3 * (tid >> 6) + 2] = dn;
    }
    __syncthreads();
    if (tid == 0) {
      double sr = 0.0, si = 0.0, sn = 0.0;
      for (uint32_t w = 0; w < NT / 64; ++w) {
        sr += dred[3 * w];
        si += dred[3 * w + 1];
        sn += dred[3 * w + 2];
      }
      P.dot_out[3 * (size_t)blockIdx.x] = sr;
      P.dot_out[3 * (size_t)blockIdx.x + 1] = si;
      P.dot_out[3 * (size_t)blockIdx.x + 2] = sn;
    }
  }
}

// ---------------------------------------------------------------------------
template <int B, int LOGR>
static int launch_cfg2(const DevPass &P, int n_loc, const void *x, void *y, const void *xr, hipStream_t st) {
  constexpr int NT = 1 << (B - LOGR);
  // DNM_GATHER_INFLIGHT = 1 | 2 (default 2), DNM_TILE_DMA = 0 | 1 (default 1): A/B switches of the instance
  const int variant = []() {
    const char *e = getenv("DNM_GATHER_INFLIGHT");
    const char *d = getenv("DNM_TILE_DMA");
    const int gf = e ? atoi(e) : 2, dma = d ? atoi(d) : 1;
    return (gf <= 1 ? 0 : 1) + (dma ? 2 : 0);
  }();
  const size_t lds = (size_t)16 << B;
  const unsigned grid = 1u << (n_loc - B);
  using kern_t = void (*)(const DevPass, const c128 *, c128 *, const c128 *);
  static const kern_t kerns[4] = {tile_pass2_kernel<B, LOGR, 1, false>, tile_pass2_kernel<B, LOGR, 2, false>,
                                  tile_pass2_kernel<B, LOGR, 1, true>, tile_pass2_kernel<B, LOGR, 2, true>};
  kern_t k = kerns[variant];
  static bool attr_done[4] = {false, false, false, false};
  if (!attr_done[variant]) {
    DNM_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done[variant] = true;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, P, (const c128 *)x, (c128 *)y, (const c128 *)xr);
  DNM_HIP(hipGetLastError());
  return 0;
}

bool tile2_config_supported(int B, int logR) {
  switch (B * 16 + logR) {
    case 8 * 16 + 2: case 10 * 16 + 2: case 10 * 16 + 3: case 11 * 16 + 3: case 12 * 16 + 3: case 12 * 16 + 4:
      return true;
  }
  return false;
}

int launch_tile_pass2(const DevPass &P, int B, int logR, int n_loc, const void *x, void *y, const void *xr,
                      hipStream_t st) {
  DNM_CHECK(n_loc >= B, "tile larger than the local vector");
  switch (B * 16 + logR) {
    case 8 * 16 + 2: return launch_cfg2<8, 2>(P, n_loc, x, y, xr, st);
    case 10 * 16 + 2: return launch_cfg2<10, 2>(P, n_loc, x, y, xr, st);
    case 10 * 16 + 3: return launch_cfg2<10, 3>(P, n_loc, x, y, xr, st);
    case 11 * 16 + 3: return launch_cfg2<11, 3>(P, n_loc, x, y, xr, st);
    case 12 * 16 + 3: return launch_cfg2<12, 3>(P, n_loc, x, y, xr, st);
    case 12 * 16 + 4: return launch_cfg2<12, 4>(P, n_loc, x, y, xr, st);
  }
  set_error("unsupported tile configuration B=%d logR=%d (second-generation kernel)", B, logR);
  return 1;
}

}  // namespace dnm
