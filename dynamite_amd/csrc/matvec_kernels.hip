// HIP kernels for the matrix-free H|psi> of an MSC (mask, sign, coefficient)
// Pauli-string operator, written for gfx950 (MI355X): 64-wide wavefronts,
// 160 KB LDS per CU, HBM-bound.  No MFMA: there is no dense contraction here.
//
// tile_pass_kernel  -- hypercube index spaces (Full, Parity): a workgroup
//   stages a 2^B-amplitude tile of x in LDS (direct global->LDS DMA), every
//   wavefront owns contiguous 64-amplitude (1 KB) runs of it, coupled indices
//   are XORs of LDS addresses, signs are popcounts, and each thread keeps
//   2^LOGR output amplitudes in registers.  Term tables are wave-uniform and
//   arrive through the scalar cache.
// gather_matvec_kernel -- any subspace pair (SpinConserve, Explicit, mixed):
//   one thread per row with the index maps of subspace.h.
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "kernels.h"

namespace dnm {

typedef double2 c128;

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

typedef double d2v __attribute__((ext_vector_type(2)));

// The record tables are read-only for the life of a launch: addressed through the constant address space their
// (wave-uniform) reads are scalar loads whatever else the kernel does -- the compiler otherwise falls back to vector
// loads as soon as anything it takes for a store (an s_setprio, an inline-asm load) precedes them
typedef const __attribute__((address_space(4))) DevQuad CQuad;

// y store that does not keep the line in this XCD's L2 (write-through, sc1)
__device__ __forceinline__ void store_through(c128 *p, double re, double im) {
  d2v v = {re, im};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_streaming(c128 *p, double re, double im) {
  d2v v = {re, im};
  __builtin_nontemporal_store(v, reinterpret_cast<d2v *>(p));
}
__device__ __forceinline__ c128 load_streaming(const c128 *p) {
  d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(p));
  return make_double2(v.x, v.y);
}

// Phase stamps of a workgroup (diagnostic builds only: DNM_HIPCC_EXTRA=-DDNM_PHASE_TIMING, tools/phase_times.py):
// thread 0 writes the 100 MHz wall clock at eight points of tile_pass_kernel into a buffer set by
// dnm_debug_phase_buffer; the waits that make a stamp mean "landed" slow the kernel down (L=30: 16.5 -> 25.8 ms)
#ifdef DNM_PHASE_TIMING
__device__ unsigned long long *g_phase_buf = nullptr;
#define DNM_PH(i, waitvm)                                                                                  \
  do {                                                                                                     \
    if (waitvm) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
    if (g_phase_buf && threadIdx.x == 0)                                                                   \
      g_phase_buf[((size_t)(P.accumulate ? gridDim.x : 0) + blockIdx.x) * 8 + (i)] = wall_clock64();        \
    asm volatile("" ::: "memory");                                                                         \
  } while (0)
#else
#define DNM_PH(i, waitvm)
#endif

// ---- wave priority by phase (round 4; profiles/r04_exp1_overlap.txt): s_setprio 3 while the
// workgroup is in a memory phase (tile loads, early gathers, late y, stores), 0 in its LDS record loops, so that of the
// two workgroups resident on a CU the one that can put requests on the memory pipeline wins the issue slots: -0.7 % at
// L=30 (same box, three alternating runs).  The variants measured against it and rejected -- the reverse, priority
// around the issue of memory instructions only, a split barrier, a touch of the successor's tile, a rotated vector
// layout -- are kept as a patch against this file: tools/experiments/r04_overlap_variants.patch.
// (inline asm, not the builtin: see the CQuad note below on what either does to the table loads)
#ifndef DNM_XP_PRIO
#define DNM_XP_PRIO 1
#endif
#if DNM_XP_PRIO == 1
#define DNM_PRIO_MEM() asm volatile("s_setprio 3")
#define DNM_PRIO_LDS() asm volatile("s_setprio 0")
#else
#define DNM_PRIO_MEM()
#define DNM_PRIO_LDS()
#endif

// +-c by a parity bit: flips the IEEE sign bit (v_xor on the high dword)
__device__ __forceinline__ double flip_sign(double c, uint32_t parity_bit) {
  int hi = __double2hiint(c) ^ (int)(parity_bit << 31);
  return __hiloint2double(hi, __double2loint(c));
}

// signed amplitude of slot j of a record for this thread: coeff * (-1)^popcount(row & sign)
// restricted to the thread-constant part of the row (tile coordinate `tt`, block part `sbase`)
__device__ __forceinline__ double slot_amp(CQuad &q, int j, uint32_t tt, uint64_t sbase) {
  uint32_t p = (uint32_t)(__popc(tt & q.sign_tile[j]) + __popcll(sbase & q.sign_ext[j])) & 1u;
  return flip_sign(q.coeff[j], p);
}

template <int MAXS>
__device__ __forceinline__ uint32_t deposit(uint32_t v, int nseg, const int32_t *off,
                                            const int32_t *len, const int32_t *pos) {
  uint32_t r = 0;
#pragma unroll
  for (int j = 0; j < MAXS; ++j)
    if (j < nseg) r |= ((v >> off[j]) & ((1u << len[j]) - 1u)) << pos[j];
  return r;
}

// Where the R rows of a thread live.  position(row k) = upos ^ tpos ^ kpos[k]: the block part (wave-uniform), the
// thread's tile coordinate, and the k bits (uniform per k).  Every bit the THREAD part can reach lies in tmask
// (< 2^28, DevPass::pos_tmask), so the access to row k of a vector p, with a uniform XOR uxor on top, is
//   (p + 16 * (u_k & ~tmask))  +  (t4 ^ 16 * (u_k & tmask)),     u_k = upos ^ uxor ^ kpos[k]:
// a scalar base and a 32-bit byte offset -- one v_xor per 16-byte load or store, one address register per thread.
template <int R>
struct RowAddr {
  uint32_t t4;        // 16 * tpos (per thread)
  uint32_t upos;      // position of the block part
  uint32_t tmask;
  uint32_t kpos[R];
  // (pointer arithmetic on the argument itself: a round trip through an integer would lose the global address space)
  __device__ __forceinline__ const c128 *at(const c128 *p, int k, uint32_t uxor) const {
    const uint32_t u = upos ^ uxor ^ kpos[k];
    const char *b = reinterpret_cast<const char *>(p) + ((uint64_t)(u & ~tmask) << 4);
    return reinterpret_cast<const c128 *>(b + (t4 ^ ((u & tmask) << 4)));
  }
  __device__ __forceinline__ c128 *at(c128 *p, int k, uint32_t uxor) const {
    const uint32_t u = upos ^ uxor ^ kpos[k];
    char *b = reinterpret_cast<char *>(p) + ((uint64_t)(u & ~tmask) << 4);
    return reinterpret_cast<c128 *>(b + (t4 ^ ((u & tmask) << 4)));
  }
};

// y += (coefficients of one record) * (partner amplitudes xv) for the R rows of this thread
// PACK: real-packed operators (plan.h, RowMask::pack_flip): an element holds two REAL amplitudes, slots 0,1 are the
// coefficient of its first lane (.x), slots 2,3 of its second (.y), and Q.nslots says whether a lane reads the
// partner element's other lane
template <int R, int LOGNT, bool KVAR, bool CPLX, bool PACK>
__device__ __forceinline__ void accum_record(CQuad &Q, double a0, double a1, double a2, double a3,
                                             const c128 (&xv)[R], double (&ar)[R], double (&ai)[R]) {
  if constexpr (PACK && CPLX) {
    const bool fl = Q.nslots != 0u;
    const uint32_t s0 = Q.sign_tile[0] >> LOGNT, s1 = Q.sign_tile[1] >> LOGNT;
    const uint32_t s2 = Q.sign_tile[2] >> LOGNT, s3 = Q.sign_tile[3] >> LOGNT;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      double c0 = a0 + a1, c1 = a2 + a3;
      if constexpr (KVAR) {
        c0 = flip_sign(a0, (uint32_t)__popc(k & s0) & 1u) + flip_sign(a1, (uint32_t)__popc(k & s1) & 1u);
        c1 = flip_sign(a2, (uint32_t)__popc(k & s2) & 1u) + flip_sign(a3, (uint32_t)__popc(k & s3) & 1u);
      }
      ar[k] = fma(c0, fl ? xv[k].y : xv[k].x, ar[k]);
      ai[k] = fma(c1, fl ? xv[k].x : xv[k].y, ai[k]);
    }
  } else if constexpr (!KVAR) {
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
    // per-row sign of each slot: one scalar parity, one v_xor on the high dword
    const uint32_t s0 = Q.sign_tile[0] >> LOGNT, s1 = Q.sign_tile[1] >> LOGNT;
    const uint32_t s2 = Q.sign_tile[2] >> LOGNT, s3 = Q.sign_tile[3] >> LOGNT;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const double cre = flip_sign(a0, (uint32_t)__popc(k & s0) & 1u) + flip_sign(a1, (uint32_t)__popc(k & s1) & 1u);
      ar[k] = fma(cre, xv[k].x, ar[k]);
      ai[k] = fma(cre, xv[k].y, ai[k]);
      if constexpr (CPLX) {
        const double cim = flip_sign(a2, (uint32_t)__popc(k & s2) & 1u) + flip_sign(a3, (uint32_t)__popc(k & s3) & 1u);
        ar[k] = fma(-cim, xv[k].y, ar[k]);
        ai[k] = fma(cim, xv[k].x, ai[k]);
      }
    }
  }
}

// One loop of the off-diagonal part: records [b, e) all share the compile-time
// traits, so the accumulators stay in place and the body has no branches.
//   KVAR  : some sign mask reaches this thread's k bits -> the coefficient is
//           rebuilt per owned row from the per-thread slot amplitudes
//   CPLX  : slots 2,3 (imaginary part) are populated
//   GATHER: partner amplitudes come from global memory instead of the LDS tile
//   K0    : (LDS, !KVAR) the mask does not touch the k bits: the R partner rows
//           sit at fixed LDS offsets from one address
template <int R, int LOGNT, bool KVAR, bool CPLX, bool GATHER, bool K0, bool PACK = false>
__device__ __forceinline__ void apply_records(CQuad *__restrict__ quads, uint32_t b, uint32_t e,
                                              double (&ar)[R], double (&ai)[R], const c128 *tile,
                                              const RowAddr<R> &RA, const c128 *__restrict__ x,
                                              const c128 *__restrict__ xr, uint32_t tid, uint64_t sbase,
                                              uint32_t skw = 0, uint32_t xrx = 0) {
  constexpr uint32_t NT = 1u << LOGNT;
  for (uint32_t qi = b; qi < e; ++qi) {
    CQuad &Q = quads[qi];
    const double a0 = slot_amp(Q, 0, tid, sbase);
    const double a1 = slot_amp(Q, 1, tid, sbase);
    double a2 = 0.0, a3 = 0.0;
    if constexpr (CPLX) {
      a2 = slot_amp(Q, 2, tid, sbase);
      a3 = slot_amp(Q, 3, tid, sbase);
    }
    c128 xv[R];
    if constexpr (GATHER) {
      // lanes whose coefficient vanishes for every owned amplitude fetch nothing;
      // a wavefront with no live lane skips the record
      const bool live = KVAR || (a0 + a1 != 0.0) || (CPLX && (a2 + a3 != 0.0));
      if (!__any(live)) continue;
      const c128 *__restrict__ src = Q.src ? xr : x;
      // the layout map is XOR-linear, so the partner of a mask sits at position(row) ^ position(mask)
      const uint32_t mloc = Q.mask_loc;
      const uint32_t xm = (skw ? (mloc ^ (((mloc >> skw) & ((1u << (skw - 4)) - 1u)) << 4)) : mloc) ^ (Q.src ? xrx : 0u);
      if (live) {
        // the partner sits at position(row) ^ position(mask)
#pragma unroll
        for (int k = 0; k < R; ++k) xv[k] = *RA.at(src, k, xm);
      } else {
#pragma unroll
        for (int k = 0; k < R; ++k) xv[k] = make_double2(0.0, 0.0);
      }
    } else if constexpr (K0) {
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
    accum_record<R, LOGNT, KVAR, CPLX, PACK>(Q, a0, a1, a2, a3, xv, ar, ai);
  }
}

// Table records (plan.h: DevTab): one per mask (or group of a mask's terms) -- the coefficient of a row is a parity times
// a table entry picked by the row's bits at the flipped positions.  Thread part of the index from tid (bits that lie in
// the block part or in the k bits are scalar), one 16-byte load of the entry (R of them when a flipped bit is a k bit).
typedef const __attribute__((address_space(4))) DevTab CTab;

// The tables of TABB records at a time are staged in LDS (they lie one after the other in DevPass::tabvals): the look-ups
// then are LDS reads -- through the texture path they were half of the loads of a multiply that waits for exactly that path
// (SYK L=24: 2.7e9 of 5.3e9 wave loads, data return 91 % busy; profiles/r06_syk_table_records.txt).
constexpr uint32_t TABB = 32;
constexpr uint32_t TABL_ENTRIES = TABB << MAXTABBITS;

template <int R, int LOGNT, bool GATHER>
__device__ __forceinline__ void apply_tabs(CTab *__restrict__ tabs, const c128 *__restrict__ vals, uint32_t b, uint32_t e,
                                           double (&ar)[R], double (&ai)[R], const c128 *tile, const RowAddr<R> &RA,
                                           const c128 *__restrict__ x, const c128 *__restrict__ xr, uint32_t tid,
                                           uint64_t sbase, uint32_t skw, uint32_t xrx, c128 *tabl) {
  constexpr uint32_t NT = 1u << LOGNT;
  c128 xv[R];
  bool fresh = true;        // the first record of a mask: fetch the partner amplitudes
  for (uint32_t b0 = b; b0 < e; b0 += TABB) {
  const uint32_t b1 = (b0 + TABB < e) ? b0 + TABB : e;
  const uint32_t f0 = tabs[b0].first, f1 = tabs[b1 - 1].first + (1u << tabs[b1 - 1].nbits);
  __syncthreads();          // (the readers of the previous batch are done)
  for (uint32_t i = tid; i < f1 - f0; i += NT) tabl[i] = vals[f0 + i];
  __syncthreads();
  for (uint32_t qi = b0; qi < b1; ++qi) {
    CTab &T = tabs[qi];
    if (fresh) {
      if constexpr (GATHER) {
        const c128 *__restrict__ src = T.src ? xr : x;
        const uint32_t mloc = T.mask_loc;
        const uint32_t xm = (skw ? (mloc ^ (((mloc >> skw) & ((1u << (skw - 4)) - 1u)) << 4)) : mloc) ^ (T.src ? xrx : 0u);
#pragma unroll
        for (int k = 0; k < R; ++k) xv[k] = *RA.at(src, k, xm);
      } else {
        const uint32_t mt = T.mask_tile;
        const uint32_t p_lo = tid ^ (mt & (NT - 1u));
        const uint32_t mk = mt >> LOGNT;
#pragma unroll
        for (int k = 0; k < R; ++k) xv[k] = tile[p_lo + (((uint32_t)k ^ mk) << LOGNT)];
      }
    }
    // table index: thread part by bit-field extracts of tid (width 0: not a thread bit), block part by scalar shifts
    const uint32_t tp = T.tpos, tw = T.twid, ep = T.epos, ew = T.ewid;
    uint32_t it = 0, is = 0;
#pragma unroll
    for (int q = 0; q < MAXTABBITS; ++q) {
      it |= __builtin_amdgcn_ubfe(tid, (tp >> (8 * q)) & 0xffu, (tw >> (8 * q)) & 0xffu) << q;
      is |= ((uint32_t)(sbase >> ((ep >> (8 * q)) & 0xffu)) & ((ew >> (8 * q)) & 0xffu)) << q;
    }
    it |= is;
    const c128 *tv = tabl + (T.first - f0);
    // (-1)^popcount(row & z): the thread-constant part once, the k part is a bit per row (DevTab::ksign)
    const uint32_t psign = ((uint32_t)(__popc(tid & T.z_tile) + __popcll(sbase & T.z_ext)) & 1u) << 31;
    const uint32_t ks = T.ksign;
    const uint64_t ikp = T.ik;
    if (T.flags & 1u) {
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const c128 cf = tv[it | ((uint32_t)(ikp >> (4 * k)) & 0xfu)];
        const uint32_t sg = psign ^ (((ks >> k) & 1u) << 31);
        const double cr = __hiloint2double(__double2hiint(cf.x) ^ (int)sg, __double2loint(cf.x));
        const double ci = __hiloint2double(__double2hiint(cf.y) ^ (int)sg, __double2loint(cf.y));
        ar[k] = fma(cr, xv[k].x, ar[k]);
        ar[k] = fma(-ci, xv[k].y, ar[k]);
        ai[k] = fma(cr, xv[k].y, ai[k]);
        ai[k] = fma(ci, xv[k].x, ai[k]);
      }
    } else {
      const c128 c0 = tv[it];
      const double c0r = __hiloint2double(__double2hiint(c0.x) ^ (int)psign, __double2loint(c0.x));
      const double c0i = __hiloint2double(__double2hiint(c0.y) ^ (int)psign, __double2loint(c0.y));
      if (ks == 0u) {       // the common sign mask misses the k bits: one coefficient for all rows of the thread
#pragma unroll
        for (int k = 0; k < R; ++k) {
          ar[k] = fma(c0r, xv[k].x, ar[k]);
          ar[k] = fma(-c0i, xv[k].y, ar[k]);
          ai[k] = fma(c0r, xv[k].y, ai[k]);
          ai[k] = fma(c0i, xv[k].x, ai[k]);
        }
      } else
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const uint32_t sg = ((ks >> k) & 1u) << 31;       // uniform
        const double cr = __hiloint2double(__double2hiint(c0r) ^ (int)sg, __double2loint(c0r));
        const double ci = __hiloint2double(__double2hiint(c0i) ^ (int)sg, __double2loint(c0i));
        ar[k] = fma(cr, xv[k].x, ar[k]);
        ar[k] = fma(-ci, xv[k].y, ar[k]);
        ai[k] = fma(cr, xv[k].y, ai[k]);
        ai[k] = fma(ci, xv[k].x, ai[k]);
      }
    }
    fresh = (T.flags & 2u) != 0u;       // the groups of one mask share the partner amplitudes: one fetch per MASK
  }
  }
}

// 4 rows per thread need 76 registers as compiled for 4 waves per SIMD -- one 1024-thread workgroup per CU at B=12
// (19.1 ms at L=30); asked for 8 waves they fit in 64 and two workgroups run: 17.35 ms, level with 8 rows per thread
// at half the waves (17.45; profiles/r02_exp37_waves8.txt) -- occupancy is not what the passes wait for
#ifndef DNM_WAVES_4ROWS
#define DNM_WAVES_4ROWS 8
#endif
// waves per SIMD the launch bounds ask for: what LDS lets be resident, capped at 4
constexpr int tile_waves_per_simd(int B, int LOGR, bool TAB = false) {
  if (TAB) {      // the instance with table records: 128 registers (64 spilled 63 of them: 853 against 199 ms for SYK at L=24)
    int nt = 1 << (B - LOGR);
    int blocks = (160 * 1024) / (16 << B);
    if (blocks < 1) blocks = 1;
    int w = blocks * nt / 256;
    const int cap = LOGR <= 2 ? 8 : 4;       // (4 rows: 66 registers uncapped -- one short of two workgroups per CU)
    return w < 1 ? 1 : (w > cap ? cap : w);
  }
  int nt = 1 << (B - LOGR);
  int blocks = (160 * 1024) / (16 << B);
  if (blocks < 1) blocks = 1;
  int w = blocks * nt / 256;
  const int cap = LOGR <= 2 ? DNM_WAVES_4ROWS : 4;
  return w < 1 ? 1 : (w > cap ? cap : w);
}

// GV (gather variant): 0 = gathers after the LDS masks; 1 = right behind the tile loads, before the barrier
// (default: sibling workgroups then ask for the same lines within the same microsecond and the L2 merges the
// requests).  (Two records in flight was tried: spills at 8 rows per thread, no gain at 16.)
// TAB: the instance that also knows table records (DevPass::tabs; passes without any run on the plain one, which this
// parameter leaves as it was)
template <int B, int LOGR, bool GLDS, int GV, bool PACK = false, bool TAB = false>
__global__ void __launch_bounds__(1 << (B - LOGR), tile_waves_per_simd(B, LOGR, TAB))
tile_pass_kernel(const DevPass P, const c128 *__restrict__ x, c128 *__restrict__ y,
                 const c128 *__restrict__ xr) {
  constexpr int R = 1 << LOGR;
  constexpr int LOGNT = B - LOGR;
  constexpr uint32_t NT = 1u << LOGNT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  c128 *tile = reinterpret_cast<c128 *>(smem);

  const uint32_t tid = threadIdx.x;
  DNM_PH(0, 0);
  const uint32_t base = deposit<MAXBSEG>(blockIdx.x + P.block_offset, P.nbseg, P.bseg_off, P.bseg_len, P.bseg_pos);
  const uint32_t dep_t = deposit<MAXSEG>(tid, P.nseg, P.seg_off, P.seg_len, P.seg_pos);
  const uint64_t sbase = P.sign_base | (uint64_t)base;

  // positions of the rows this thread owns (tile coordinate tid + k*NT): see RowAddr.  XOR-swizzled vector layout
  // (DESIGN.md section 3): index bits [s, 2s-4) are folded onto bits [4, s); the map is linear, so it applies to the
  // block part, the thread part and the k part separately.  Sub-block passes add the swizzle of the block's own
  // offset (swz_xor_y for y, swz_xor_src for the partner's amplitudes)
  const uint32_t skw = (uint32_t)P.swz_shift;
  auto lay = [skw](uint32_t v) -> uint32_t { return skw ? (v ^ (((v >> skw) & ((1u << (skw - 4)) - 1u)) << 4)) : v; };
  RowAddr<R> RA;
  RA.t4 = lay(dep_t) << 4;
  RA.upos = lay(base);
  RA.tmask = P.pos_tmask;
#pragma unroll
  for (int k = 0; k < R; ++k) RA.kpos[k] = lay(deposit<MAXSEG>((uint32_t)k << LOGNT, P.nseg, P.seg_off, P.seg_len, P.seg_pos));
  const uint32_t yx = P.swz_xor_y;

  // ---- stage the tile: each wavefront moves 1 KB runs, lane = low 6 tile bits
  DNM_PRIO_MEM();
  if (!P.need_tile) {
    // pure gather pass (remote partner vector): nothing to stage
  } else if constexpr (GLDS) {
#pragma unroll
    for (int k = 0; k < R; ++k)
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)RA.at(x, k, 0u),
                                       (LDS_AS void *)(tile + (k * NT + (tid & ~63u))), 16, 0, 0);
  } else {
    c128 v[R];
    if (P.cache_policy & 4) {
#pragma unroll
      for (int k = 0; k < R; ++k) v[k] = load_streaming(RA.at(x, k, 0u));
    } else {
#pragma unroll
      for (int k = 0; k < R; ++k) v[k] = *RA.at(x, k, 0u);
    }
#pragma unroll
    for (int k = 0; k < R; ++k) tile[tid + k * NT] = v[k];
  }
  DNM_PH(1, 1);

  double ar[R], ai[R];
  // cache_policy bit 7: an accumulating pass adds its y at the END (loaded right before the stores) instead of
  // starting from it -- y then crosses the L2 after the gathers of the workgroup, not before them
  const bool late_y = P.accumulate && (P.cache_policy & 128);
  if (P.accumulate && !late_y) {
    if (P.cache_policy & 2) {
#pragma unroll
      for (int k = 0; k < R; ++k) {
        c128 v = load_streaming(RA.at((const c128 *)y, k, yx));
        ar[k] = v.x;
        ai[k] = v.y;
      }
    } else {
#pragma unroll
      for (int k = 0; k < R; ++k) {
        c128 v = *RA.at((const c128 *)y, k, yx);
        ar[k] = v.x;
        ai[k] = v.y;
      }
    }
  } else if (P.zinit) {
    const c128 *__restrict__ z = (const c128 *)P.zinit;
    const double zs = -P.zscale;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const c128 v = load_streaming(RA.at(z, k, yx));      // read once
      ar[k] = zs * v.x;
      ai[k] = zs * v.y;
    }
    if (P.zinit2) {
      const c128 *__restrict__ z2 = (const c128 *)P.zinit2;
      const double cr = P.z2re, ci = P.z2im;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const c128 v = load_streaming(RA.at(z2, k, yx));
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

  CQuad *__restrict__ quads = (CQuad *)P.quads;

#define DNM_LOOP(LP, KV, CX, GA, KZ) \
  apply_records<R, LOGNT, KV, CX, GA, KZ, PACK>(quads, P.loop[LP], P.loop[LP + 1], ar, ai, tile, RA, x, xr, tid, sbase, skw, P.swz_xor_src)
  if constexpr (GV >= 1) {
    // (two live records in flight at a time -- half the L2 round trips of this phase -- changed nothing:
    // profiles/r03_exp9_gather_pairs.txt; the passes are not bound by the life of a workgroup)
    DNM_LOOP(LP_GATHER_REAL, false, false, true, false);
    DNM_LOOP(LP_GATHER_KVAR_REAL, true, false, true, false);
    DNM_LOOP(LP_GATHER_CPLX, false, true, true, false);
    DNM_LOOP(LP_GATHER_KVAR_CPLX, true, true, true, false);
  }
#undef DNM_LOOP
  c128 *tabl = nullptr;
  if constexpr (TAB) {
    __shared__ __attribute__((aligned(16))) unsigned char tabl_mem[TABL_ENTRIES * 16];
    tabl = reinterpret_cast<c128 *>(tabl_mem);
  }
  if constexpr (TAB)
    apply_tabs<R, LOGNT, true>((CTab *)P.tabs, (const c128 *)P.tabvals, P.tab_loop[1], P.tab_loop[2], ar, ai, tile, RA, x, xr,
                               tid, sbase, skw, P.swz_xor_src, tabl);
  DNM_PH(2, 1);

  // ---- diagonal, part 1 (before the barrier, under the tile loads): the terms
  // whose sign mask lies outside the tile are the same for the whole workgroup.
  // Each lane evaluates one term, a butterfly sums them across the wavefront.
  double dext = 0.0;
  if (P.has_diag) {
    const uint32_t lane = tid & 63u;
    const uint32_t nterm = (P.dext_end - P.dext_begin) * 4u;
    for (uint32_t t0 = 0; t0 < nterm; t0 += 64u) {
      const uint32_t t = t0 + lane;
      double v = 0.0;
      if (t < nterm) {
        CQuad &Q = quads[P.dext_begin + (t >> 2)];
        const uint32_t j = t & 3u;
        const uint32_t p = (uint32_t)__popcll(sbase & Q.sign_ext[j]) & 1u;
        v = flip_sign(Q.coeff[j], p);
      }
      dext += v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dext += __shfl_xor(dext, off, 64);
  }
  // grouped diagonal terms (DevPass::gbucket): every group's sum over the bits outside the tile, once per workgroup
  double *cg = nullptr;
  if constexpr (TAB) {
    __shared__ double cg_mem[MAXDGROUPS];
    cg = cg_mem;
    if (P.has_diag) {
      const uint32_t lane = tid & 63u, g0 = P.gbucket[0], g1 = P.gbucket[R];
      // (a group per wavefront at a time: the workgroup's waves share the groups out)
      for (uint32_t g = g0 + (tid >> 6); g < g1; g += (NT >> 6)) {
        const uint32_t q0 = quads[g].mask_loc, nterm = quads[g].src * 4u;
        double v = 0.0;
        for (uint32_t t0 = 0; t0 < nterm; t0 += 64u) {
          const uint32_t t = t0 + lane;
          if (t < nterm) {
            CQuad &Q = quads[q0 + (t >> 2)];
            const uint32_t j = t & 3u;
            v += flip_sign(Q.coeff[j], (uint32_t)__popcll(sbase & Q.sign_ext[j]) & 1u);
          }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) cg[g - g0] = v;
      }
    }
  }
  // ---- diagonal, part 2: sum_t c_t chi_t(row) for the terms that see the tile.
  // Terms are bucketed by the part of their sign mask that falls on this
  // thread's k bits; a length-R Walsh-Hadamard butterfly then yields all R row
  // values at once.  (Reads only the thread's own amplitudes of the tile.)
  auto diag_part2 = [&]() {
  if (P.has_diag) {
    double D[R];
#pragma unroll
    for (int j = 0; j < R; ++j) D[j] = 0.0;
    D[0] = dext;
    double dtab[R];
    if (P.dtile) {      // tile-only terms, tabulated per tile coordinate (L2-resident)
#pragma unroll
      for (int k = 0; k < R; ++k) dtab[k] = P.dtile[tid + k * NT];
    }
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
    if constexpr (TAB) {
      const uint32_t g0 = P.gbucket[0];
#pragma unroll
      for (int j = 0; j < R; ++j)
        for (uint32_t q = P.gbucket[j]; q < P.gbucket[j + 1]; ++q)
          D[j] += flip_sign(cg[q - g0], (uint32_t)__popc(tid & quads[q].sign_tile[0]) & 1u);
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
    if (P.dtile) {
#pragma unroll
      for (int k = 0; k < R; ++k) D[k] += dtab[k];
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      c128 xs = tile[tid + k * NT];
      ar[k] = fma(D[k], xs.x, ar[k]);
      ai[k] = fma(D[k], xs.y, ai[k]);
    }
  }
  };
  __syncthreads();
  DNM_PRIO_LDS();
  DNM_PH(3, 0);
  diag_part2();

  DNM_PH(4, 0);
  // ---- off-diagonal masks, one branch-free loop per record class
#define DNM_LOOP(LP, KV, CX, GA, KZ) \
  apply_records<R, LOGNT, KV, CX, GA, KZ, PACK>(quads, P.loop[LP], P.loop[LP + 1], ar, ai, tile, RA, x, xr, tid, sbase, skw, P.swz_xor_src)
  DNM_LOOP(LP_TILE_REAL_K0, false, false, false, true);
  DNM_LOOP(LP_TILE_REAL, false, false, false, false);
  DNM_LOOP(LP_TILE_CPLX, false, true, false, false);
  DNM_LOOP(LP_TILE_KVAR_REAL, true, false, false, false);
  DNM_LOOP(LP_TILE_KVAR_CPLX, true, true, false, false);
  if constexpr (GV == 0) {
    DNM_LOOP(LP_GATHER_REAL, false, false, true, false);
    DNM_LOOP(LP_GATHER_KVAR_REAL, true, false, true, false);
    DNM_LOOP(LP_GATHER_CPLX, false, true, true, false);
    DNM_LOOP(LP_GATHER_KVAR_CPLX, true, true, true, false);
  }
#undef DNM_LOOP
  if constexpr (TAB)
    apply_tabs<R, LOGNT, false>((CTab *)P.tabs, (const c128 *)P.tabvals, P.tab_loop[0], P.tab_loop[1], ar, ai, tile, RA, x, xr,
                                tid, sbase, skw, P.swz_xor_src, tabl);

  DNM_PH(5, 0);
  DNM_PRIO_MEM();
  if (late_y) {
    c128 w[R];
#pragma unroll
    for (int k = 0; k < R; ++k) w[k] = load_streaming(RA.at((const c128 *)y, k, yx));
#pragma unroll
    for (int k = 0; k < R; ++k) {
      ar[k] += w[k].x;
      ai[k] += w[k].y;
    }
  }
  DNM_PH(6, 1);
  if (P.cache_policy & 64) {
#pragma unroll
    for (int k = 0; k < R; ++k) store_streaming(RA.at(y, k, yx), ar[k], ai[k]);
  } else if (P.cache_policy & 1) {
#pragma unroll
    for (int k = 0; k < R; ++k) store_through(RA.at(y, k, yx), ar[k], ai[k]);
  } else {
#pragma unroll
    for (int k = 0; k < R; ++k) *RA.at(y, k, yx) = make_double2(ar[k], ai[k]);
  }

  DNM_PH(7, 1);
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

#ifdef DNM_PHASE_TIMING
}  // namespace dnm
extern "C" int dnm_debug_phase_buffer(void *buf) {
  unsigned long long *p = (unsigned long long *)buf;
  return hipMemcpyToSymbol(HIP_SYMBOL(dnm::g_phase_buf), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
namespace dnm {
#endif

// ---------------------------------------------------------------------------
template <int B, int LOGR>
static int launch_cfg(const DevPass &P, bool glds, int n_loc, const void *x, void *y,
                      const void *xr, hipStream_t st, unsigned nparts) {
  constexpr int NT = 1 << (B - LOGR);
  // DNM_LDS_KB (experiments): request more LDS than the tile needs to cap the
  // number of resident workgroups per CU
  static const size_t lds_req = []() {
    const char *e = knob("DNM_LDS_KB");
    return e ? (size_t)atoi(e) * 1024 : (size_t)0;
  }();
  const size_t lds = std::max(((size_t)16 << B), lds_req);
  const unsigned grid = (1u << (n_loc - B)) / nparts;       // nparts > 1: the range starting at P.block_offset
  // gather variant from the plan's cache policy: bit5 = early (default)
  const int gv = (P.cache_policy & 32) ? 1 : 0;
  using kern_t = void (*)(const DevPass, const c128 *, c128 *, const c128 *);
  kern_t k = nullptr;
  const bool pack = (P.cache_policy & 256) != 0;          // real-packed records: their own instance (early gathers only)
  // table records / grouped diagonal terms: their own instance (early gathers, plain tile loads)
  const bool tab = P.tab_loop[2] > 0 || P.gbucket[MAXR] > P.gbucket[0];
  if (pack) k = tile_pass_kernel<B, LOGR, false, 1, true>;
  else if (tab) k = tile_pass_kernel<B, LOGR, false, 1, false, true>;
  else if (glds) k = tile_pass_kernel<B, LOGR, true, 1>;
  else if (gv == 0) k = tile_pass_kernel<B, LOGR, false, 0>;
  else k = tile_pass_kernel<B, LOGR, false, 1>;
  static size_t attr_done[6] = {0, 0, 0, 0, 0, 0};
  const int slot = pack ? 4 : (tab ? 5 : (glds ? 3 : gv));
  if (attr_done[slot] < lds) {
    DNM_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done[slot] = lds;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, P, (const c128 *)x, (c128 *)y, (const c128 *)xr);
  DNM_HIP(hipGetLastError());
  return 0;
}

bool tile_config_supported(int B, int logR) {
  switch (B * 16 + logR) {
    case 10 * 16 + 2: case 11 * 16 + 2: case 12 * 16 + 2:
    case 8 * 16 + 2: case 10 * 16 + 3: case 10 * 16 + 4: case 11 * 16 + 3: case 11 * 16 + 4:
    case 12 * 16 + 3: case 12 * 16 + 4: case 13 * 16 + 3: case 13 * 16 + 4:
      return true;
  }
  return false;
}

int launch_tile_pass(const DevPass &P, int B, int logR, bool glds, int n_loc, const void *x,
                     void *y, const void *xr, hipStream_t st, unsigned nparts) {
  DNM_CHECK(n_loc >= B, "tile larger than the local vector");
  switch (B * 16 + logR) {
    case 8 * 16 + 2: return launch_cfg<8, 2>(P, glds, n_loc, x, y, xr, st, nparts);
    case 10 * 16 + 2: return launch_cfg<10, 2>(P, glds, n_loc, x, y, xr, st, nparts);
    case 11 * 16 + 2: return launch_cfg<11, 2>(P, glds, n_loc, x, y, xr, st, nparts);
    case 12 * 16 + 2: return launch_cfg<12, 2>(P, glds, n_loc, x, y, xr, st, nparts);
    case 10 * 16 + 3: return launch_cfg<10, 3>(P, glds, n_loc, x, y, xr, st, nparts);
    case 10 * 16 + 4: return launch_cfg<10, 4>(P, glds, n_loc, x, y, xr, st, nparts);
    case 11 * 16 + 3: return launch_cfg<11, 3>(P, glds, n_loc, x, y, xr, st, nparts);
    case 11 * 16 + 4: return launch_cfg<11, 4>(P, glds, n_loc, x, y, xr, st, nparts);
    case 12 * 16 + 3: return launch_cfg<12, 3>(P, glds, n_loc, x, y, xr, st, nparts);
    case 12 * 16 + 4: return launch_cfg<12, 4>(P, glds, n_loc, x, y, xr, st, nparts);
    case 13 * 16 + 3: return launch_cfg<13, 3>(P, glds, n_loc, x, y, xr, st, nparts);
    case 13 * 16 + 4: return launch_cfg<13, 4>(P, glds, n_loc, x, y, xr, st, nparts);
  }
  set_error("unsupported tile configuration B=%d logR=%d", B, logR);
  return 1;
}

// ===========================================================================
// Generic row-gather kernels (any subspace pair)
// ===========================================================================

// SpinConserve binomial tables are staged in LDS once per workgroup.
constexpr int NCK_LDS_MAX = 2 * 1024;   // int64 entries (16 KB)

template <int T>
__device__ __forceinline__ SubView stage_sub(const SubView &s, int64_t *lds_tab, int &used) {
  SubView r = s;
  if constexpr (T == DNM_SPIN_CONSERVE) {
    int n = (s.k + 1) * s.ld;
    if (used + n <= NCK_LDS_MAX) {
      for (int i = threadIdx.x; i < n; i += blockDim.x) lds_tab[used + i] = s.nchoosek[i];
      r.nchoosek = lds_tab + used;
      used += n;
    }
  }
  return r;
}

constexpr int GATHER_NT = 256;

// column-range sweeps: note the chunk a column lies in (one store per wavefront when the lanes agree)
__device__ __forceinline__ void mark_column(const ColMark &cm, int64_t col) {
  if (!cm.map) return;
  const int64_t c = (col >> cm.shift) - cm.first;
  const int c32 = (int)c;
  if (__all(c32 == __builtin_amdgcn_readfirstlane(c32))) {
    if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) cm.map[c] = 1;
  } else {
    cm.map[c] = 1;
  }
}

// Rows [row0, row0 + M) of the matrix (a rank's block); x holds the columns [win_start, ...) in the layout xswz
// (one rank: the whole right vector in its own layout; partitioned: the rank's column window in index order);
// y and diag are local (index row - row0).  colrange != nullptr: no multiply, only the min / max column each
// workgroup touches (2 int64 per workgroup) -- how a partition finds its column window.
template <int LT, int RT>
__global__ void __launch_bounds__(GATHER_NT)
gather_matvec_kernel(const DevMsc msc, const SubView left_g, const SubView right_g, int64_t M, int64_t row0,
                     int64_t win_start, int xswz, const double *__restrict__ diag, const c128 *__restrict__ x,
                     c128 *__restrict__ y, int64_t *__restrict__ colrange, const ColMark mark) {
  __shared__ int64_t nck[NCK_LDS_MAX];
  int used = 0;
  const SubView left = stage_sub<LT>(left_g, nck, used);
  const SubView right = stage_sub<RT>(right_g, nck, used);
  if (used) __syncthreads();

  const int64_t lrow = (int64_t)blockIdx.x * GATHER_NT + threadIdx.x;
  const bool active = lrow < M;
  if (!active && !colrange) return;
  const int64_t row = row0 + lrow;
  int64_t cmin = INT64_MAX, cmax = INT64_MIN;
  if (active) {
    const int64_t ket = Sub<LT>::i2s(row, left);
    double accr = 0.0, acci = 0.0;
    int m0 = 0;
    if (diag && !colrange) {   // bcuda_template_2.cu:230-236
      c128 xs = x[vec_pos(row, xswz) - win_start];
      accr = diag[lrow] * xs.x;
      acci = diag[lrow] * xs.y;
      m0 = 1;
    }
    for (int m = m0; m < msc.nmasks; ++m) {
      const int64_t mask = msc.masks[m];
      const int64_t bra = ket ^ mask;
      const int64_t col = Sub<RT>::s2i(bra, right);
      if (col < 0) continue;   // projection semantics
      if (colrange) {
        cmin = col < cmin ? col : cmin;
        cmax = col > cmax ? col : cmax;
        mark_column(mark, col);
        continue;
      }
      double cre = 0.0, cim = 0.0;
      for (int64_t t = msc.mask_offsets[m]; t < msc.mask_offsets[m + 1]; ++t) {
        const int64_t sg = msc.signs[t];
        const double c = flip_sign(msc.real_coeffs[t], (uint32_t)__popcll((uint64_t)(bra & sg)) & 1u);
        if (__popcll((uint64_t)(mask & sg)) & 1) cim += c; else cre += c;   // TERM_REAL
      }
      const c128 xv = x[vec_pos(col, xswz) - win_start];
      accr = fma(cre, xv.x, accr);
      acci = fma(cre, xv.y, acci);
      accr = fma(-cim, xv.y, accr);
      acci = fma(cim, xv.x, acci);
    }
    if (!colrange) y[vec_pos(lrow, left.swz)] = make_double2(accr, acci);
  }
  if (colrange) {
    __shared__ int64_t smin[GATHER_NT / 64], smax[GATHER_NT / 64];
    for (int off = 32; off > 0; off >>= 1) {
      const int64_t a = __shfl_xor(cmin, off, 64), b2 = __shfl_xor(cmax, off, 64);
      cmin = a < cmin ? a : cmin;
      cmax = b2 > cmax ? b2 : cmax;
    }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = cmin; smax[threadIdx.x >> 6] = cmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int i = 1; i < GATHER_NT / 64; ++i) {
        cmin = smin[i] < cmin ? smin[i] : cmin;
        cmax = smax[i] > cmax ? smax[i] : cmax;
      }
      colrange[2 * (int64_t)blockIdx.x] = cmin;
      colrange[2 * (int64_t)blockIdx.x + 1] = cmax;
    }
  }
}

// ---------------------------------------------------------------------------
// SpinConserve / SpinConserve (same L, k): consecutive basis indices are
// consecutive k-subsets in colex order, so the column of a coupled state is
// row + delta with delta a difference of a few binomials:
//   rank(s) = sum_j C(p_j, j)  (bsubspace_impl.h:191-202); flipping the bits of
//   `mask` keeps every one outside the mask's span [lo, hi] at its position and
//   ordinal (the popcount is conserved), so only the ones inside the span
//   contribute to rank(bra) - rank(ket).
// For the two-bit bond masks that is one table lookup.  Lanes hold consecutive
// rows, so x[row + delta] is a contiguous (shifted) run wherever the rows of a
// wavefront share the bits above the bond -- long runs for the high bonds.
// Each lane unranks its own row (L steps on the LDS-resident binomial table).
// ---------------------------------------------------------------------------
constexpr int SC_NT = 256;

template <bool IN_LDS>
__global__ void __launch_bounds__(SC_NT)
sc_matvec_kernel(const DevMsc msc, const ScMask *__restrict__ scm, const ScLow low, const SubView sub_g, int64_t M,
                 int64_t row0, int64_t win_start, const double *__restrict__ diag,
                 const c128 *__restrict__ xw, c128 *__restrict__ y, int64_t *__restrict__ colrange,
                 const ColMark mark) {
  // rows [row0, row0 + M) of the matrix; xw holds columns [win_start, ...); y and diag are
  // local (index row - row0).  colrange != nullptr: only record min/max column per workgroup.
  __shared__ int64_t nck[NCK_LDS_MAX];
  const int ld = sub_g.ld, kk = sub_g.k, Lb = sub_g.L;
  const int ntab = (kk + 1) * ld;
  if (IN_LDS) {
    for (int i = threadIdx.x; i < ntab; i += SC_NT) nck[i] = sub_g.nchoosek[i];
    __syncthreads();
  }
  // the binomial table: LDS reads (ds_read_b64) when it fits -- a pointer that may be either LDS or global
  // compiles to FLAT loads, which go through the texture addresser and saturate it
#define SC_TAB(i) (IN_LDS ? nck[(i)] : sub_g.nchoosek[(i)])

  const int64_t lrow = (int64_t)blockIdx.x * SC_NT + threadIdx.x;
  const int64_t row = row0 + lrow;
  const c128 *__restrict__ x = xw - win_start;
  int64_t cmin = row, cmax = row;
  const bool active = lrow < M;
  if (!active && !colrange) return;
  if (active) {
  // I2S_SpinConserve (bsubspace_impl.h:210-228): the greedy walk over the positions >= 16, then the
  // remaining (index, number of ones) selects the low 16 bits from a table -- the walk over the low
  // positions would unrank exactly that pair among the 16-bit patterns
  uint64_t ket = 0;
  {
    int64_t idx = row;
    int k = kk;
    for (int n = Lb; n > 16; --n) {
      const int64_t here = (k > n - 1) ? 0 : SC_TAB(k * ld + (n - 1));
      ket <<= 1;
      if (idx >= here) { idx -= here; --k; ket |= 1; }
    }
    const uint64_t lowbits = low.tab[low.off[k] + (int32_t)idx];
    ket = Lb > 16 ? ((ket << 16) | lowbits) : lowbits;
  }
  double accr = 0.0, acci = 0.0;
  int m0 = 0;
  if (diag && !colrange) {
    const c128 xs = x[row];
    const double dg = __builtin_nontemporal_load(diag + lrow);     // read once: keep it out of the caches' way
    accr = dg * xs.x;
    acci = dg * xs.y;
    m0 = 1;
  }
  for (int m = m0; m < msc.nmasks; ++m) {
    if (scm[m].fast) {
      // adjacent bond with local signs: one lookup, two possible coefficients
      const int lo = scm[m].lo;
      const uint32_t pair = (uint32_t)(ket >> lo) & 3u;
      if (pair == 1u || pair == 2u) {
        const bool up = pair == 1u;
        const int ord0 = __popcll(ket & ((1ull << lo) - 1));
        const int64_t d = SC_TAB(ord0 * ld + lo);       // C(lo, ord0)
        if (colrange) {
          const int64_t c = up ? row + d : row - d;
          cmin = c < cmin ? c : cmin;
          cmax = c > cmax ? c : cmax;
          mark_column(mark, c);
          continue;
        }
        const c128 xv = x[up ? row + d : row - d];
        const double cre = up ? scm[m].up_re : scm[m].dn_re;
        const double cim = up ? scm[m].up_im : scm[m].dn_im;
        accr = fma(cre, xv.x, accr);
        acci = fma(cre, xv.y, acci);
        accr = fma(-cim, xv.y, accr);
        acci = fma(cim, xv.x, acci);
      }
      continue;
    }
    const uint64_t mask = (uint64_t)msc.masks[m];
    const uint64_t bra = ket ^ mask;
    int64_t delta = 0;
    if (mask && (mask & (mask + 1)) == 0) {
      // the mask flips every spin below c (XParity's complemented terms): among the c-bit patterns with the same
      // number of ones the complement reverses the order, so with hc the rank contribution of the ones at or
      // above c:  row = hc + r,  col = hc + C(c, n1) - 1 - r
      if (__popcll(bra) != kk) continue;
      const int c = 64 - __clzll((long long)mask);
      const int n1 = __popcll(ket & mask);
      int64_t hc = 0;
      uint64_t hb = ket >> c;
      int o = n1;
      while (hb) {
        const int p = c + __ffsll((long long)hb) - 1;
        ++o;
        hc += SC_TAB(o * ld + p);
        hb &= hb - 1;
      }
      delta = SC_TAB(n1 * ld + c) - 1 - 2 * (row - hc);
    } else if (mask) {
      if (__popcll(bra) != kk) continue;             // leaves the subspace: projection semantics
      const int lo = __ffsll((long long)mask) - 1;    // wave-uniform
      const int hi = 63 - __clzll((long long)mask);
      const uint64_t span = (hi >= 63 ? ~0ull : ((2ull << hi) - 1)) & ~((1ull << lo) - 1);
      const int ord0 = __popcll(ket & ((1ull << lo) - 1));
      uint64_t bb = bra & span, kb = ket & span;
      int o = ord0;
      while (bb) {
        const int p = __ffsll((long long)bb) - 1;
        ++o;
        if (o <= p) delta += SC_TAB(o * ld + p);
        bb &= bb - 1;
      }
      o = ord0;
      while (kb) {
        const int p = __ffsll((long long)kb) - 1;
        ++o;
        if (o <= p) delta -= SC_TAB(o * ld + p);
        kb &= kb - 1;
      }
    }
    if (colrange) {
      const int64_t c = row + delta;
      cmin = c < cmin ? c : cmin;
      cmax = c > cmax ? c : cmax;
      mark_column(mark, c);
      continue;
    }
    double cre = 0.0, cim = 0.0;
    for (int64_t t = msc.mask_offsets[m]; t < msc.mask_offsets[m + 1]; ++t) {
      const uint64_t sg = (uint64_t)msc.signs[t];
      const double c = flip_sign(msc.real_coeffs[t], (uint32_t)__popcll(bra & sg) & 1u);
      if (__popcll(mask & sg) & 1) cim += c; else cre += c;   // TERM_REAL
    }
    const c128 xv = x[row + delta];
    accr = fma(cre, xv.x, accr);
    acci = fma(cre, xv.y, acci);
    accr = fma(-cim, xv.y, accr);
    acci = fma(cim, xv.x, acci);
  }
  if (!colrange) store_streaming(y + lrow, accr, acci);
  }  // active
  if (colrange) {
    // workgroup min / max of the columns touched (one-time sweep when a partition is set up)
    __shared__ int64_t smin[SC_NT / 64], smax[SC_NT / 64];
    if (!active) { cmin = INT64_MAX; cmax = INT64_MIN; }
    for (int off = 32; off > 0; off >>= 1) {
      const int64_t a = __shfl_xor(cmin, off, 64), b = __shfl_xor(cmax, off, 64);
      cmin = a < cmin ? a : cmin;
      cmax = b > cmax ? b : cmax;
    }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = cmin; smax[threadIdx.x >> 6] = cmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int i = 1; i < SC_NT / 64; ++i) {
        cmin = smin[i] < cmin ? smin[i] : cmin;
        cmax = smax[i] > cmax ? smax[i] : cmax;
      }
      colrange[2 * (int64_t)blockIdx.x] = cmin;
      colrange[2 * (int64_t)blockIdx.x + 1] = cmax;
    }
  }
}

#undef SC_TAB

int sc_num_blocks(int64_t M) { return (int)((M + SC_NT - 1) / SC_NT); }
int sc_rows_per_block() { return SC_NT; }

int launch_sc_matvec(const DevMsc &msc, const ScMask *scm, const ScLow &low, const SubView &sub, int64_t M, int64_t row0,
                     int64_t win_start, const double *diag, const void *xw, void *y, int64_t *colrange,
                     hipStream_t st, ColMark mark) {
  DNM_CHECK(M > 0 && M + SC_NT < (int64_t)1 << 32, "row count out of range (one thread per row, fewer than 2^32 per launch)");
  const dim3 grid((unsigned)sc_num_blocks(M)), blk(SC_NT);
  if ((sub.k + 1) * sub.ld <= NCK_LDS_MAX)
    hipLaunchKernelGGL(sc_matvec_kernel<true>, grid, blk, 0, st, msc, scm, low, sub, M, row0, win_start, diag,
                       (const c128 *)xw, (c128 *)y, colrange, mark);
  else
    hipLaunchKernelGGL(sc_matvec_kernel<false>, grid, blk, 0, st, msc, scm, low, sub, M, row0, win_start, diag,
                       (const c128 *)xw, (c128 *)y, colrange, mark);
  DNM_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// SpinConserve, block form.  Rows are ordered by the numeric value of the state, so all rows that share the
// high part H = state >> LB are consecutive: a block of C(LB, k - |H|) rows whose low parts are the LB-bit
// patterns with k - |H| ones in ascending order.  One workgroup owns one block:
//   * the block of x is staged in LDS; a bond inside the low LB bits couples rows of the same block
//     (local index r -> r +- C(lo, ones below lo)), read from LDS;
//   * a bond inside the high part couples the block to the block of H ^ bond at the same local index: one
//     coalesced run x[base(H') + r], with a coefficient and a skip decision that are uniform over the block;
//   * the row's configuration is one table lookup (local index -> low pattern), no unranking walk;
//   * any other mask takes the per-row path of sc_matvec_kernel.
// Workgroup order: [3 XCD selector | 6 low bits of H | rest], so the blocks resident on one XCD are partners
// under the six lowest high bonds and their requests meet in that XCD's L2.
// ---------------------------------------------------------------------------
constexpr int sc_binom(int n, int k) {
  long long r = 1;
  for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return (int)r;
}
constexpr int SCB_MAXM = 64;    // masks per operator: one lane of a wavefront each
#ifndef DNM_SC_NB
#define DNM_SC_NB 2
#endif
#ifndef DNM_SC_NT
#define DNM_SC_NT 512
#endif
constexpr int SC_NB = DNM_SC_NB;   // high bonds whose partner blocks are requested together

__device__ __forceinline__ int64_t rl_i64(int64_t v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(v & 0xffffffff), l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l);
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double rl_f64(double v, int l) { return __longlong_as_double(rl_i64(__double_as_longlong(v), l)); }

template <int LB, int NT>
__global__ void __launch_bounds__(NT, NT >= 1024 ? 8 : 4)
sc_block_kernel(const DevMsc msc, const ScMask *__restrict__ scm, const ScBlock blk, const SubView sub_g, int64_t M,
                int64_t row0, int64_t win_start, int64_t win_len, const double *__restrict__ diag,
                const c128 *__restrict__ xw, c128 *__restrict__ y, const c128 *__restrict__ zinit, double zscale,
                double *__restrict__ dot_out, const c128 *__restrict__ zinit2, double z2re, double z2im) {
  constexpr int MAXROWS = sc_binom(LB, LB / 2);
  constexpr int RPT = (MAXROWS + NT - 1) / NT;
  __shared__ c128 xs[MAXROWS];
  __shared__ int32_t cl[LB * (LB + 1)];      // cl[lo * (LB+1) + o] = C(lo, o)
  __shared__ int32_t lf_lo[SCB_MAXM];        // bonds that touch the low part (written by the first wavefront)
  __shared__ double lf_c[SCB_MAXM][4];       // up_re, up_im, dn_re, dn_im
  __shared__ int32_t gen_m[SCB_MAXM];        // masks on the per-row path
  __shared__ int32_t cnt[2];
  const int ld = sub_g.ld, kk = sub_g.k;
  const int64_t *__restrict__ gtab = sub_g.nchoosek;
  const int lane = threadIdx.x & 63;

  // block id -> high part
  int64_t hrel;
  {
    const uint32_t b = blockIdx.x;
    if (blk.perm) {
      const uint32_t e = blk.perm[b];
      if (e == 0xffffffffu) return;
      hrel = e;
    } else if (blk.swizzle) {
      const uint32_t seq = b >> 3, xcd = b & 7u;
      hrel = ((int64_t)(seq >> 6) << 9) | (xcd << 6) | (seq & 63u);
    } else {
      hrel = b;
    }
  }
  const uint64_t H = (uint64_t)(blk.h_first + hrel);
  if (H > (uint64_t)blk.h_last) return;
  const int kl = kk - __popcll(H);
  if (kl < 0 || kl > LB) return;
  // first row of the block: colex rank of (H << LB) | (kl lowest bits set)  (bsubspace_impl.h:191-202);
  // one lane per set bit of H, summed over the wavefront
  int64_t base;
  {
    int64_t term = 0;
    if ((H >> lane) & 1ull) term = gtab[(kl + __popcll(H & ((2ull << lane) - 1))) * ld + LB + lane];
    for (int off = 32; off > 0; off >>= 1) term += __shfl_xor(term, off, 64);
    base = rl_i64(term, 0);
  }
  const int nrows = (int)gtab[kl * ld + LB];
  const int64_t lo_row = row0 - base, hi_row = row0 + M - base;     // local rows of this rank: [lo_row, hi_row)
  if (hi_row <= 0 || lo_row >= nrows) return;
  const int r_lo = lo_row > 0 ? (int)lo_row : 0;
  const int r_hi = hi_row < nrows ? (int)hi_row : nrows;
  const c128 *__restrict__ x = xw - win_start;

  // this block of x and the low patterns of its rows
  uint32_t lowb[RPT];
  c128 xv[RPT];
  const uint16_t *__restrict__ pat = blk.lowtab + blk.off[kl];
  const bool whole = (base >= win_start) && (base + nrows <= win_start + win_len);
  uint32_t live = 0;       // bit i: row i of this thread belongs to this rank
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = threadIdx.x + i * NT;
    lowb[i] = 0;
    xv[i] = make_double2(0.0, 0.0);
    if (r < nrows) {
      lowb[i] = pat[r];
      if (whole || (base + r >= win_start && base + r < win_start + win_len)) xv[i] = x[base + r];
      if (r >= r_lo && r < r_hi) live |= 1u << i;
    }
  }

  // every wavefront classifies the masks for this block, one mask per lane (ascending mask order in the
  // ballots: deterministic sums): 0 = bond inside the high part that acts on the block, 1 = bond touching
  // the low part, 2 = per-row path
  const int m0 = diag ? 1 : 0;
  int cls = -1;
  int64_t delta_v = 0;
  double c0 = 0.0, c1 = 0.0;
  if (lane >= m0 && lane < msc.nmasks) {
    const ScMask sm = scm[lane];
    if (sm.fast && sm.lo >= LB) {
      const uint32_t pair = (uint32_t)(H >> (sm.lo - LB)) & 3u;
      if (pair == 1u || pair == 2u) {
        const bool up = pair == 1u;
        const int ord0 = kl + __popcll(H & ((1ull << (sm.lo - LB)) - 1));
        const int64_t d = gtab[ord0 * ld + sm.lo];
        delta_v = up ? d : -d;
        c0 = up ? sm.up_re : sm.dn_re;
        c1 = up ? sm.up_im : sm.dn_im;
        cls = 0;
      }
    } else {
      cls = sm.fast ? 1 : 2;
    }
  }
  uint64_t hb = __ballot(cls == 0);
  {
    const uint64_t lfb = __ballot(cls == 1), genb = __ballot(cls == 2);
    if (threadIdx.x < 64) {
      const uint64_t below = (1ull << lane) - 1;
      if (cls == 1) {
        const ScMask sm = scm[lane];
        const int p = __popcll(lfb & below);
        lf_lo[p] = sm.lo;
        lf_c[p][0] = sm.up_re; lf_c[p][1] = sm.up_im; lf_c[p][2] = sm.dn_re; lf_c[p][3] = sm.dn_im;
      } else if (cls == 2) {
        gen_m[__popcll(genb & below)] = lane;
      }
      if (lane == 0) { cnt[0] = __popcll(lfb); cnt[1] = __popcll(genb); }
    }
  }

  for (int t = threadIdx.x; t < LB * (LB + 1); t += NT) {
    const int lo = t / (LB + 1), o = t % (LB + 1);
    cl[t] = (o <= lo && o <= kk) ? (int32_t)gtab[o * ld + lo] : 0;
  }
  double accr[RPT], acci[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = threadIdx.x + i * NT;
    accr[i] = 0.0;
    acci[i] = 0.0;
    if (r < nrows) {
      xs[r] = xv[i];
      if (diag && ((live >> i) & 1u)) {
        const double dg = __builtin_nontemporal_load(diag + (base + r - row0));
        accr[i] = dg * xv[i].x;
        acci[i] = dg * xv[i].y;
      }
      if (zinit && ((live >> i) & 1u)) {       // Lanczos: y = A x - b z, the beta term starts the accumulators
        const c128 zv = zinit[base + r - row0];
        accr[i] = fma(-zscale, zv.x, accr[i]);
        acci[i] = fma(-zscale, zv.y, acci[i]);
        if (zinit2) {
          const c128 z2 = zinit2[base + r - row0];
          accr[i] = fma(z2re, z2.x, accr[i]);
          accr[i] = fma(-z2im, z2.y, accr[i]);
          acci[i] = fma(z2re, z2.y, acci[i]);
          acci[i] = fma(z2im, z2.x, acci[i]);
        }
      }
    }
  }

  // bonds inside the high part, lowest first and before the barrier: the partner blocks under the lowest high
  // bonds are resident on this XCD and were requested by their owners a moment ago.  Two bonds at a time:
  // 2 * RPT independent coalesced loads in flight per thread.
  while (hb) {
    // SC_NB bonds at a time: SC_NB * RPT independent coalesced loads in flight per thread
    const c128 *__restrict__ pp[SC_NB];
    double cr[SC_NB], ci[SC_NB];
#pragma unroll
    for (int j = 0; j < SC_NB; ++j) {
      const bool have = hb != 0;
      const int m = have ? __ffsll((long long)hb) - 1 : 0;
      hb &= hb - 1;
      pp[j] = x + (base + rl_i64(delta_v, m));
      cr[j] = have ? rl_f64(c0, m) : 0.0;
      ci[j] = have ? rl_f64(c1, m) : 0.0;
      if (!have) pp[j] = pp[0];
    }
    c128 v[SC_NB][RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = threadIdx.x + i * NT;
#pragma unroll
      for (int j = 0; j < SC_NB; ++j) {
        v[j][i] = make_double2(0.0, 0.0);
        if ((live >> i) & 1u) v[j][i] = pp[j][r];
      }
    }
#pragma unroll
    for (int j = 0; j < SC_NB; ++j) {
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        accr[i] = fma(cr[j], v[j][i].x, accr[i]);
        acci[i] = fma(cr[j], v[j][i].y, acci[i]);
        accr[i] = fma(-ci[j], v[j][i].y, accr[i]);
        acci[i] = fma(ci[j], v[j][i].x, acci[i]);
      }
    }
  }
  __syncthreads();

  // bonds that touch the low part: partner row of the same block from LDS; the bond across the boundary
  // (lo == LB - 1) pairs the top low bit with the lowest high bit and reads its partner from memory
  const uint32_t hbit = (uint32_t)(H & 1ull) << LB;
  const int nlf = cnt[0];
  for (int a = 0; a < nlf; ++a) {
    const int lo = lf_lo[a];
    const double ure = lf_c[a][0], uim = lf_c[a][1], dre = lf_c[a][2], dim_ = lf_c[a][3];
    const bool cross = lo == LB - 1;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = threadIdx.x + i * NT;
      const uint32_t pair = ((lowb[i] | hbit) >> lo) & 3u;
      if (((live >> i) & 1u) && (pair == 1u || pair == 2u)) {
        const bool up = pair == 1u;
        const int ord0 = __popc(lowb[i] & ((1u << lo) - 1u));
        const int d = cl[lo * (LB + 1) + ord0];
        const int rp = up ? r + d : r - d;
        const c128 xp = cross ? x[base + rp] : xs[rp];
        const double cre = up ? ure : dre, cim = up ? uim : dim_;
        accr[i] = fma(cre, xp.x, accr[i]);
        acci[i] = fma(cre, xp.y, acci[i]);
        accr[i] = fma(-cim, xp.y, accr[i]);
        acci[i] = fma(cim, xp.x, acci[i]);
      }
    }
  }

  // everything else: per row, columns by incremental rank (as in sc_matvec_kernel)
  const int ngen = cnt[1];
  for (int a = 0; a < ngen; ++a) {
    const int m = gen_m[a];
    const uint64_t mask = (uint64_t)msc.masks[m];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = threadIdx.x + i * NT;
      if (!((live >> i) & 1u)) continue;
      const uint64_t ket = (H << LB) | lowb[i];
      const uint64_t bra = ket ^ mask;
      int64_t delta = 0;
      if (mask && (mask & (mask + 1)) == 0) {
        // every spin below c flipped: the order among the low patterns is reversed (see sc_matvec_kernel)
        if (__popcll(bra) != kk) continue;
        const int c = 64 - __clzll((long long)mask);
        const int n1 = __popcll(ket & mask);
        int64_t hc = 0;
        uint64_t hb2 = ket >> c;
        int o = n1;
        while (hb2) {
          const int p = c + __ffsll((long long)hb2) - 1;
          ++o;
          hc += gtab[o * ld + p];
          hb2 &= hb2 - 1;
        }
        delta = gtab[n1 * ld + c] - 1 - 2 * (base + r - hc);
      } else if (mask) {
        if (__popcll(bra) != kk) continue;             // leaves the subspace: projection semantics
        const int mlo_ = __ffsll((long long)mask) - 1;
        const int mhi = 63 - __clzll((long long)mask);
        const uint64_t span = (mhi >= 63 ? ~0ull : ((2ull << mhi) - 1)) & ~((1ull << mlo_) - 1);
        const int ord0 = __popcll(ket & ((1ull << mlo_) - 1));
        uint64_t bb = bra & span, kb = ket & span;
        int o = ord0;
        while (bb) {
          const int p = __ffsll((long long)bb) - 1;
          ++o;
          if (o <= p) delta += gtab[o * ld + p];
          bb &= bb - 1;
        }
        o = ord0;
        while (kb) {
          const int p = __ffsll((long long)kb) - 1;
          ++o;
          if (o <= p) delta -= gtab[o * ld + p];
          kb &= kb - 1;
        }
      }
      double cre = 0.0, cim = 0.0;
      for (int64_t t = msc.mask_offsets[m]; t < msc.mask_offsets[m + 1]; ++t) {
        const uint64_t sg = (uint64_t)msc.signs[t];
        const double c = flip_sign(msc.real_coeffs[t], (uint32_t)__popcll(bra & sg) & 1u);
        if (__popcll(mask & sg) & 1) cim += c; else cre += c;   // TERM_REAL
      }
      const c128 xg = x[base + r + delta];
      accr[i] = fma(cre, xg.x, accr[i]);
      acci[i] = fma(cre, xg.y, acci[i]);
      accr[i] = fma(-cim, xg.y, accr[i]);
      acci[i] = fma(cim, xg.x, acci[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = threadIdx.x + i * NT;
    if ((live >> i) & 1u) store_streaming(y + (base + r - row0), accr[i], acci[i]);
  }
  // fused <x, y> and |y|^2 of this block's rows (the block of x is still in LDS); the slot of a workgroup that
  // returned early keeps the zero the host put there
  if (dot_out) {
    double dr = 0.0, di = 0.0, dn = 0.0;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int r = threadIdx.x + i * NT;
      if ((live >> i) & 1u) {
        const c128 xo = xs[r];
        dr = fma(xo.x, accr[i], dr);
        dr = fma(xo.y, acci[i], dr);
        di = fma(xo.x, acci[i], di);
        di = fma(-xo.y, accr[i], di);
        dn = fma(accr[i], accr[i], dn);
        dn = fma(acci[i], acci[i], dn);
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      dr += __shfl_xor(dr, off, 64);
      di += __shfl_xor(di, off, 64);
      dn += __shfl_xor(dn, off, 64);
    }
    __syncthreads();                       // every wavefront is done with xs
    double *red = (double *)xs;            // reuse the tile for the cross-wave sum
    if (lane == 0) {
      red[3 * (threadIdx.x >> 6)] = dr;
      red[3 * (threadIdx.x >> 6) + 1] = di;
      red[3 * (threadIdx.x >> 6) + 2] = dn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double sr = 0.0, si = 0.0, sn = 0.0;
      for (int w = 0; w < NT / 64; ++w) { sr += red[3 * w]; si += red[3 * w + 1]; sn += red[3 * w + 2]; }
      dot_out[3 * (size_t)blockIdx.x] = sr;
      dot_out[3 * (size_t)blockIdx.x + 1] = si;
      dot_out[3 * (size_t)blockIdx.x + 2] = sn;
    }
  }
}

bool sc_block_supported(int lb) { return lb == 10 || lb == 13; }
int sc_block_max_masks() { return SCB_MAXM; }

int64_t sc_block_grid(const ScBlock &blk) {
  const int64_t span = blk.h_last - blk.h_first + 1;
  return blk.perm ? blk.nperm : blk.swizzle ? ((span + 511) & ~(int64_t)511) : span;
}

int launch_sc_block(const DevMsc &msc, const ScMask *scm, const ScBlock &blk, const SubView &sub, int64_t M, int64_t row0,
                    int64_t win_start, int64_t win_len, const double *diag, const void *xw, void *y,
                    hipStream_t st, const void *zinit, double zscale, double *dot_out, const void *zinit2,
                    double z2re, double z2im) {
  DNM_CHECK(msc.nmasks <= SCB_MAXM, "too many masks for the block kernel");
  const int64_t span = blk.h_last - blk.h_first + 1;
  DNM_CHECK(span > 0 && span < (int64_t)1 << 31, "block range out of range");
  const dim3 grid((unsigned)sc_block_grid(blk));
  if (dot_out) DNM_HIP(hipMemsetAsync(dot_out, 0, (size_t)grid.x * 3 * sizeof(double), st));
#define DNM_SCB(LB_, NT_)                                                                                      \
  hipLaunchKernelGGL((sc_block_kernel<LB_, NT_>), grid, dim3(NT_), 0, st, msc, scm, blk, sub, M, row0,           \
                     win_start, win_len, diag, (const c128 *)xw, (c128 *)y, (const c128 *)zinit, zscale, dot_out,      \
                     (const c128 *)zinit2, z2re, z2im)
  switch (blk.lb) {
    case 10: DNM_SCB(10, 64); break;
    case 13: DNM_SCB(13, DNM_SC_NT); break;
    default: DNM_CHECK(false, "unsupported block size %d", blk.lb);
  }
#undef DNM_SCB
  DNM_HIP(hipGetLastError());
  return 0;
}

template <int T>
__global__ void __launch_bounds__(GATHER_NT)
diag_kernel(const DevMsc msc, const SubView sub_g, int64_t M, int64_t row0, double *__restrict__ diag) {
  __shared__ int64_t nck[NCK_LDS_MAX];
  int used = 0;
  const SubView sub = stage_sub<T>(sub_g, nck, used);
  if (used) __syncthreads();
  const int64_t row = (int64_t)blockIdx.x * GATHER_NT + threadIdx.x;
  if (row >= M) return;
  const int64_t st = Sub<T>::i2s(row0 + row, sub);
  double v = 0.0;
  for (int64_t t = 0; t < msc.mask_offsets[1]; ++t)
    v += flip_sign(msc.real_coeffs[t], (uint32_t)__popcll((uint64_t)(st & msc.signs[t])) & 1u);
  diag[row] = v;
}

template <int LT, int RT>
__global__ void __launch_bounds__(GATHER_NT)
norm_kernel(const DevMsc msc, const SubView left_g, const SubView right_g, int64_t M, int64_t row0,
            double *__restrict__ block_max) {
  __shared__ int64_t nck[NCK_LDS_MAX];
  __shared__ double wmax[GATHER_NT / 64];
  int used = 0;
  const SubView left = stage_sub<LT>(left_g, nck, used);
  const SubView right = stage_sub<RT>(right_g, nck, used);
  if (used) __syncthreads();
  double best = 0.0;
  for (int64_t row = (int64_t)blockIdx.x * GATHER_NT + threadIdx.x; row < M;
       row += (int64_t)gridDim.x * GATHER_NT) {
    const int64_t ket = Sub<LT>::i2s(row0 + row, left);
    double sum = 0.0, err = 0.0;   // Kahan, as MatNorm_CPU (bpetsc_template_2.c:964-967)
    for (int m = 0; m < msc.nmasks; ++m) {
      const int64_t mask = msc.masks[m];
      const int64_t bra = ket ^ mask;
      if (!Sub<RT>::contains(bra, right)) continue;       // (membership only: the column index is not needed)
      double cre = 0.0, cim = 0.0;
      bool imag = false;                                   // wave-uniform: does any term of this mask carry an i?
      for (int64_t t = msc.mask_offsets[m]; t < msc.mask_offsets[m + 1]; ++t) {
        const int64_t sg = msc.signs[t];
        const double c = flip_sign(msc.real_coeffs[t], (uint32_t)__popcll((uint64_t)(bra & sg)) & 1u);
        if (__popcll((uint64_t)(mask & sg)) & 1) { cim += c; imag = true; } else cre += c;
      }
      const double comp = (imag ? hypot(cre, cim) : fabs(cre)) - err;
      const double tot = sum + comp;
      err = (tot - sum) - comp;
      sum = tot;
    }
    best = fmax(best, sum);
  }
  // wave max via DPP-free shuffles, then one value per wave through LDS
  for (int off = 32; off > 0; off >>= 1) best = fmax(best, __shfl_xor(best, off, 64));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    double b = wmax[0];
    for (int i = 1; i < GATHER_NT / 64; ++i) b = fmax(b, wmax[i]);
    block_max[blockIdx.x] = b;
  }
}

// CheckConserves (bpetsc_template_2.c:990-1056): does H map the right subspace
// into the left one?  One thread per column; any column whose image leaves the
// left subspace with a non-zero matrix element clears the flag.
template <int LT, int RT>
__global__ void __launch_bounds__(GATHER_NT)
conserves_kernel(const DevMsc msc, const double *__restrict__ coeffs_im, const SubView left_g,
                 const SubView right_g, int64_t N, int *__restrict__ bad, int64_t col0) {
  __shared__ int64_t nck[NCK_LDS_MAX];
  int used = 0;
  const SubView left = stage_sub<LT>(left_g, nck, used);
  const SubView right = stage_sub<RT>(right_g, nck, used);
  if (used) __syncthreads();
  const int64_t col = col0 + (int64_t)blockIdx.x * GATHER_NT + threadIdx.x;
  if (col >= N) return;
  const int64_t bra = Sub<RT>::i2s(col, right);
  for (int m = 0; m < msc.nmasks; ++m) {
    const int64_t ket = bra ^ msc.masks[m];
    if (Sub<LT>::contains(ket, left)) continue;
    // complex sum of the terms of this matrix element (the reference sums msc->coeffs)
    double vr = 0.0, vi = 0.0;
    for (int64_t t = msc.mask_offsets[m]; t < msc.mask_offsets[m + 1]; ++t) {
      const uint32_t p = (uint32_t)__popcll((uint64_t)(bra & msc.signs[t])) & 1u;
      vr += flip_sign(msc.real_coeffs[t], p);
      vi += flip_sign(coeffs_im[t], p);
    }
    if (vr != 0.0 || vi != 0.0) {
      *bad = 1;
      return;
    }
  }
}

// rows / columns per launch of the one-thread-per-row kernels (DNM_LAUNCH_SLICE_LOG2: tests slice small problems)
static int64_t launch_slice() {
  const char *e = knob("DNM_LAUNCH_SLICE_LOG2");
  const int lg = e ? atoi(e) : 30;
  return (int64_t)1 << (lg < 8 ? 8 : (lg > 30 ? 30 : lg));
}

template <int LT>
static int conserves_dispatch_r(const DevMsc &msc, const double *cim, const SubView &l, const SubView &r,
                                int64_t N, int *bad, hipStream_t st) {
  // (one thread per column, fewer than 2^32 threads to a launch: slices of 2^30 columns)
  const int64_t SLICE = launch_slice();
  for (int64_t c0 = 0; c0 < N; c0 += SLICE) {
    const dim3 grid((unsigned)((std::min(SLICE, N - c0) + GATHER_NT - 1) / GATHER_NT)), blk(GATHER_NT);
#define DNM_C(RT)                                                                                      \
  case RT:                                                                                             \
    hipLaunchKernelGGL((conserves_kernel<LT, RT>), grid, blk, 0, st, msc, cim, l, r, N, bad, c0);      \
    break;
    switch (r.type) {
      DNM_C(DNM_FULL) DNM_C(DNM_PARITY) DNM_C(DNM_SPIN_CONSERVE) DNM_C(DNM_EXPLICIT)
      default: set_error("bad right subspace type"); return 1;
    }
#undef DNM_C
    DNM_HIP(hipGetLastError());
  }
  return 0;
}

// msc.real_coeffs must hold the REAL parts here and coeffs_im the imaginary parts
int launch_conserves(const DevMsc &msc, const double *coeffs_im, const SubView &left, const SubView &right,
                     int64_t N, int *bad, hipStream_t st) {
  switch (left.type) {
    case DNM_FULL: return conserves_dispatch_r<DNM_FULL>(msc, coeffs_im, left, right, N, bad, st);
    case DNM_PARITY: return conserves_dispatch_r<DNM_PARITY>(msc, coeffs_im, left, right, N, bad, st);
    case DNM_SPIN_CONSERVE: return conserves_dispatch_r<DNM_SPIN_CONSERVE>(msc, coeffs_im, left, right, N, bad, st);
    case DNM_EXPLICIT: return conserves_dispatch_r<DNM_EXPLICIT>(msc, coeffs_im, left, right, N, bad, st);
  }
  set_error("bad left subspace type");
  return 1;
}

template <int LT>
static int gather_dispatch_r(const DevMsc &msc, const SubView &l, const SubView &r, int64_t M, int64_t row0,
                             int64_t win_start, int xswz, const double *diag, const void *x, void *y,
                             int64_t *colrange, hipStream_t st, ColMark mark) {
  const dim3 grid((unsigned)((M + GATHER_NT - 1) / GATHER_NT)), blk(GATHER_NT);
#define DNM_G(RT)                                                                                         \
  case RT:                                                                                                \
    hipLaunchKernelGGL((gather_matvec_kernel<LT, RT>), grid, blk, 0, st, msc, l, r, M, row0, win_start,  \
                       xswz, diag, (const c128 *)x, (c128 *)y, colrange, mark);                           \
    break;
  switch (r.type) {
    DNM_G(DNM_FULL) DNM_G(DNM_PARITY) DNM_G(DNM_SPIN_CONSERVE) DNM_G(DNM_EXPLICIT)
    default: set_error("bad right subspace type"); return 1;
  }
#undef DNM_G
  DNM_HIP(hipGetLastError());
  return 0;
}

int gather_num_blocks(int64_t M) { return (int)((M + GATHER_NT - 1) / GATHER_NT); }
int gather_rows_per_block() { return GATHER_NT; }

int launch_gather_matvec(const DevMsc &msc, const SubView &left, const SubView &right, int64_t M,
                         const double *diag, const void *x, void *y, hipStream_t st, int64_t row0,
                         int64_t win_start, int xswz, int64_t *colrange, ColMark mark) {
  DNM_CHECK(M > 0 && M + GATHER_NT < (int64_t)1 << 32, "row count out of range (one thread per row, fewer than 2^32 per launch)");
  if (xswz < 0) xswz = right.swz;
  switch (left.type) {
    case DNM_FULL: return gather_dispatch_r<DNM_FULL>(msc, left, right, M, row0, win_start, xswz, diag, x, y, colrange, st, mark);
    case DNM_PARITY: return gather_dispatch_r<DNM_PARITY>(msc, left, right, M, row0, win_start, xswz, diag, x, y, colrange, st, mark);
    case DNM_SPIN_CONSERVE: return gather_dispatch_r<DNM_SPIN_CONSERVE>(msc, left, right, M, row0, win_start, xswz, diag, x, y, colrange, st, mark);
    case DNM_EXPLICIT: return gather_dispatch_r<DNM_EXPLICIT>(msc, left, right, M, row0, win_start, xswz, diag, x, y, colrange, st, mark);
  }
  set_error("bad left subspace type");
  return 1;
}

int launch_diag(const DevMsc &msc, const SubView &sub, int64_t M, int64_t row0, double *diag, hipStream_t st) {
  // One thread per row, and a launch holds fewer than 2^32 threads: rows go in slices of 2^30 (XParity on
  // SpinConserve(36,18) has 4.54 G of them -- its first run, round 5, silently computed with no diagonal at all)
  const int64_t SLICE = launch_slice();
  for (int64_t r0 = 0; r0 < M; r0 += SLICE) {
    const int64_t m = std::min(SLICE, M - r0);
    const dim3 grid((unsigned)((m + GATHER_NT - 1) / GATHER_NT)), blk(GATHER_NT);
    switch (sub.type) {
      case DNM_FULL: hipLaunchKernelGGL((diag_kernel<DNM_FULL>), grid, blk, 0, st, msc, sub, m, row0 + r0, diag + r0); break;
      case DNM_PARITY: hipLaunchKernelGGL((diag_kernel<DNM_PARITY>), grid, blk, 0, st, msc, sub, m, row0 + r0, diag + r0); break;
      case DNM_SPIN_CONSERVE: hipLaunchKernelGGL((diag_kernel<DNM_SPIN_CONSERVE>), grid, blk, 0, st, msc, sub, m, row0 + r0, diag + r0); break;
      case DNM_EXPLICIT: hipLaunchKernelGGL((diag_kernel<DNM_EXPLICIT>), grid, blk, 0, st, msc, sub, m, row0 + r0, diag + r0); break;
      default: set_error("bad subspace type"); return 1;
    }
    DNM_HIP(hipGetLastError());
  }
  return 0;
}

int norm_num_blocks(int64_t M) {
  int64_t nb = (M + GATHER_NT - 1) / GATHER_NT;
  return (int)(nb < 4096 ? nb : 4096);
}

template <int LT>
static int norm_dispatch_r(const DevMsc &msc, const SubView &l, const SubView &r, int64_t M,
                           int64_t row0, double *bm, hipStream_t st) {
  const dim3 grid((unsigned)norm_num_blocks(M)), blk(GATHER_NT);
#define DNM_N(RT)                                                                           \
  case RT:                                                                                  \
    hipLaunchKernelGGL((norm_kernel<LT, RT>), grid, blk, 0, st, msc, l, r, M, row0, bm);    \
    break;
  switch (r.type) {
    DNM_N(DNM_FULL) DNM_N(DNM_PARITY) DNM_N(DNM_SPIN_CONSERVE) DNM_N(DNM_EXPLICIT)
    default: set_error("bad right subspace type"); return 1;
  }
#undef DNM_N
  DNM_HIP(hipGetLastError());
  return 0;
}

int launch_norm(const DevMsc &msc, const SubView &left, const SubView &right, int64_t M,
                int64_t row0, double *block_max, hipStream_t st) {
  switch (left.type) {
    case DNM_FULL: return norm_dispatch_r<DNM_FULL>(msc, left, right, M, row0, block_max, st);
    case DNM_PARITY: return norm_dispatch_r<DNM_PARITY>(msc, left, right, M, row0, block_max, st);
    case DNM_SPIN_CONSERVE: return norm_dispatch_r<DNM_SPIN_CONSERVE>(msc, left, right, M, row0, block_max, st);
    case DNM_EXPLICIT: return norm_dispatch_r<DNM_EXPLICIT>(msc, left, right, M, row0, block_max, st);
  }
  set_error("bad left subspace type");
  return 1;
}

}  // namespace dnm
